import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extra_bench as eb
kzg = eb.load_pkg()
s = kzg.KZGSettings.from_file(eb.SETUP)
rnd = random.Random(5)
nb = 256
blobs = bytearray(rnd.randbytes(nb * 131072))
for i in range(0, len(blobs), 32):
    blobs[i] = 0
blobs = bytes(blobs)
cms = b"".join(kzg.blob_to_kzg_commitment_batch(blobs, nb, s))
for _ in range(3):
    kzg.compute_blob_kzg_proof_batch(blobs, cms, nb, s)
for _ in range(10):
    kzg.compute_blob_kzg_proof(blobs[:131072], cms[:48], s)
s.close()

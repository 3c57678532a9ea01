import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extra_bench as eb
kzg = eb.load_pkg()
s = kzg.KZGSettings.from_file(eb.SETUP)
rnd = random.Random(5)
b = bytearray(rnd.randbytes(131072))
for i in range(0, len(b), 32):
    b[i] = 0
b = bytes(b)
for _ in range(10):
    c = kzg.blob_to_kzg_commitment(b, s)
for _ in range(10):
    kzg.compute_blob_kzg_proof(b, c, s)
s.close()

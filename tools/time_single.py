import os, sys, random, time
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
c = kzg.blob_to_kzg_commitment(b, s)
for name, fn in (("commit", lambda: kzg.blob_to_kzg_commitment(b, s)), ("blob proof", lambda: kzg.compute_blob_kzg_proof(b, c, s))):
    for _ in range(5):
        fn()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    print("%-10s %.3f ms" % (name, (time.perf_counter() - t0) / 50 * 1e3), os.environ.get("KZGAMD_TUNING", "default"))
s.close()
# verification: field work / G1 combinations on the GPU, one pairing check on the host
s = kzg.KZGSettings.from_file(eb.SETUP)
p = kzg.compute_blob_kzg_proof(b, c, s)
assert kzg.verify_blob_kzg_proof(b, c, p, s)
t0 = time.perf_counter()
for _ in range(5):
    kzg.verify_blob_kzg_proof(b, c, p, s)
print("%-10s %.3f ms" % ("verify_blob_kzg_proof", (time.perf_counter() - t0) / 5 * 1e3))
n = 64
assert kzg.verify_blob_kzg_proof_batch([b] * n, [c] * n, [p] * n, s)
t0 = time.perf_counter()
for _ in range(3):
    kzg.verify_blob_kzg_proof_batch([b] * n, [c] * n, [p] * n, s)
print("%-10s %.3f ms per call of %d blobs" % ("verify_blob_kzg_proof_batch", (time.perf_counter() - t0) / 3 * 1e3, n))
s.close()
# the parts of a 64-blob batch verification, buffers joined once
s = kzg.KZGSettings.from_file(eb.SETUP)
blobs, cms, prs = b * n, c * n, p * n
def tm(label, fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    print("  %-44s %.3f ms" % (label, (time.perf_counter() - t0) / reps * 1e3))
    return r
zs, ys = tm("challenges + evaluations (GPU)", lambda: kzg.compute_challenges_and_evaluate_batch(blobs, cms, n, s))
pl, rhs = tm("r-powers + three lincombs (GPU)", lambda: kzg.verify_kzg_proof_batch_g1(cms, b"".join(zs), b"".join(ys), prs, n, s))
g2 = kzg.p2_generator()
tm("pairing check (host)", lambda: kzg.pairings_verify(pl, g2, pl, g2))
s.close()

import os, sys, random, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
kzg = load_package()
s = kzg.KZGSettings.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
rnd = random.Random(5)
blob = bytearray(rnd.randbytes(131072))
for i in range(0, 131072, 32): blob[i] = 0
blob = bytes(blob)
cm = kzg.blob_to_kzg_commitment(blob, s)
cells, cproofs = kzg.compute_cells_and_kzg_proofs(blob, s)
idx = list(range(128))
for _ in range(5):
    t0 = time.perf_counter(); assert kzg.verify_cell_kzg_proof_batch(cm * 128, idx, cells, cproofs, s); print("call %.3f ms" % ((time.perf_counter() - t0) * 1e3))

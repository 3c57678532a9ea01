import os, sys, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_package
kzg = load_package()
s = kzg.KZGSettings.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
L = kzg.lib()
BLOB = 131072
g = torch.Generator(); g.manual_seed(1)
blobs = torch.randint(0, 256, (256, BLOB), dtype=torch.uint8, generator=g); blobs[:, ::32] = 0
for n in (1, 2, 3, 4, 8, 16, 32, 64, 128, 256):
    cm = torch.zeros(48 * n, dtype=torch.uint8); po = torch.zeros(48 * n, dtype=torch.uint8)
    def commit():
        assert L.kzgamd_blob_to_kzg_commitment_batch(C.c_void_p(cm.data_ptr()), C.c_void_p(blobs.data_ptr()), n, C.byref(s.c)) == 0
    def prove():
        assert L.kzgamd_compute_blob_kzg_proof_batch(C.c_void_p(po.data_ptr()), C.c_void_p(blobs.data_ptr()), C.c_void_p(cm.data_ptr()), n, C.byref(s.c)) == 0
    out = []
    for fn in (commit, prove):
        fn(); fn()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        ts.sort(); out.append(ts[4] * 1e3)
    print("n=%3d commit %.3f ms (%.0f/s)  proof %.3f ms (%.0f/s)" % (n, out[0], n / out[0] * 1e3, out[1], n / out[1] * 1e3), flush=True)

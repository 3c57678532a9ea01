#!/usr/bin/env python3
"""Host-buffer commitment / proof batches with the caller's buffers pageable vs page-locked (hipHostMalloc through torch)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_package

kzg = load_package()
s = kzg.KZGSettings.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
BLOB = 131072
L = kzg.lib()
for n in (256, 1024, 4096):
    g = torch.Generator()
    g.manual_seed(n)
    blobs = torch.randint(0, 256, (n, BLOB), dtype=torch.uint8, generator=g)
    blobs[:, ::32] = 0
    pinned = blobs.pin_memory()
    outp = torch.zeros(48 * n, dtype=torch.uint8).pin_memory()
    outq = torch.zeros(48 * n, dtype=torch.uint8)
    res = {}
    for name, b, o in (("pageable", blobs, outq), ("pinned", pinned, outp)):
        def commit():
            rc = L.kzgamd_blob_to_kzg_commitment_batch(C.c_void_p(o.data_ptr()), C.c_void_p(b.data_ptr()), n, C.byref(s.c))
            assert rc == 0
        commit()
        t0 = time.perf_counter()
        for _ in range(3):
            commit()
        res[name + "_commit_per_s"] = round(3 * n / (time.perf_counter() - t0))
        cm = o.clone()
        po = torch.zeros(48 * n, dtype=torch.uint8)
        if name == "pinned":
            cm, po = cm.pin_memory(), po.pin_memory()
        def prove():
            rc = L.kzgamd_compute_blob_kzg_proof_batch(C.c_void_p(po.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(cm.data_ptr()), n, C.byref(s.c))
            assert rc == 0
        prove()
        t0 = time.perf_counter()
        for _ in range(3):
            prove()
        res[name + "_proofs_per_s"] = round(3 * n / (time.perf_counter() - t0))
        res[name + "_digest"] = int(o.sum()) + int(po.sum())
    print(n, res)
s.close()

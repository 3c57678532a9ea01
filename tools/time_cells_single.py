#!/usr/bin/env python3
"""ms per compute_cells_and_kzg_proofs / recover_cells_and_kzg_proofs call on one blob (the direct form: 128 MSMs)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

kzg = load_package()
s = kzg.KZGSettings.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
rnd = random.Random(3)
blob = bytearray(rnd.randbytes(131072))
for i in range(0, 131072, 32):
    blob[i] = 0
blob = bytes(blob)


def med(fn, reps=9):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


cells, proofs = kzg.compute_cells_and_kzg_proofs(blob, s)
print("compute_cells_and_kzg_proofs: %.3f ms" % med(lambda: kzg.compute_cells_and_kzg_proofs(blob, s)))
idx = list(range(0, 128, 2))
part = b"".join(cells[2048 * i: 2048 * (i + 1)] for i in idx)
print("recover_cells_and_kzg_proofs (64 cells): %.3f ms" % med(lambda: kzg.recover_cells_and_kzg_proofs(idx, part, s)))

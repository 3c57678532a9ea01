#!/usr/bin/env python3
"""ms per verify_blob_kzg_proof_batch (64 blobs) and verify_cell_kzg_proof_batch (128 cells of one blob) call, host
buffers; KZGAMD_TUNING=wide_check=0 runs the single-lane membership tests instead of the wave-per-point ones."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import random
from conftest import load_package

kzg = load_package()
s = kzg.KZGSettings.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
rnd = random.Random(5)
n = 64
blobs = bytearray(rnd.randbytes(n * 131072))
for i in range(0, len(blobs), 32):
    blobs[i] = 0
blobs = bytes(blobs)
cms = kzg.blob_to_kzg_commitment_batch(blobs, n, s)
prs = kzg.compute_blob_kzg_proof_batch(blobs, b"".join(cms), n, s)


def med(fn, reps=9):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


bl = [blobs[i * 131072:(i + 1) * 131072] for i in range(n)]
assert kzg.verify_blob_kzg_proof_batch(bl, cms, prs, s)
print("verify_blob_kzg_proof_batch(64): %.3f ms" % med(lambda: kzg.verify_blob_kzg_proof_batch(bl, cms, prs, s)))
print("verify_blob_kzg_proof_batch(8):  %.3f ms" % med(lambda: kzg.verify_blob_kzg_proof_batch(bl[:8], cms[:8], prs[:8], s)))
cells, cproofs = kzg.compute_cells_and_kzg_proofs(bl[0], s)
idx = list(range(128))
assert kzg.verify_cell_kzg_proof_batch(cms[0] * 128, idx, cells, cproofs, s)
print("verify_cell_kzg_proof_batch(128 cells): %.3f ms" % med(lambda: kzg.verify_cell_kzg_proof_batch(cms[0] * 128, idx, cells, cproofs, s)))
print("verify_blob_kzg_proof (single): %.3f ms" % med(lambda: kzg.verify_blob_kzg_proof(bl[0], cms[0], prs[0], s)))

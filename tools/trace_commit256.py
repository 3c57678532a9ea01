#!/usr/bin/env python3
"""A few kzgamd_blob_to_kzg_commitment_batch calls (host buffers) for a kernel + copy trace (tools/trace_batch256.sh)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import random
from conftest import load_package

kzg = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = kzg.KZGSettings.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
rnd = random.Random(3)
blobs = bytearray(rnd.randbytes(n * 131072))
for i in range(0, len(blobs), 32):
    blobs[i] = 0
blobs = bytes(blobs)
for rep in range(6):
    t0 = time.perf_counter()
    kzg.blob_to_kzg_commitment_batch(blobs, n, s)
    print("call %d: %.3f ms" % (rep, (time.perf_counter() - t0) * 1e3), flush=True)

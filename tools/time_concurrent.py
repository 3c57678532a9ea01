#!/usr/bin/env python3
"""Throughput of single-blob host-buffer calls from T threads sharing one CKZGSettings (the reference's rayon pattern,
kzg/src/eip_4844.rs:781-805): calls/s for blob_to_kzg_commitment and compute_blob_kzg_proof, T = 1, 4, 16."""
import json
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package


def measure(kzg, s, blobs, commitments, nthreads, seconds=1.5):
    out = {}
    for name in ("blob_to_kzg_commitment", "compute_blob_kzg_proof"):
        counts = [0] * nthreads
        stop = threading.Event()
        start = threading.Barrier(nthreads + 1)

        def worker(i):
            b, c = blobs[i % len(blobs)], commitments[i % len(blobs)]
            start.wait()
            n = 0
            while not stop.is_set():
                if name == "blob_to_kzg_commitment":
                    kzg.blob_to_kzg_commitment(b, s)
                else:
                    kzg.compute_blob_kzg_proof(b, c, s)
                n += 1
            counts[i] = n

        th = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
        for t in th:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        time.sleep(seconds)
        stop.set()
        for t in th:
            t.join()
        out[name] = sum(counts) / (time.perf_counter() - t0)
    return out


def main():
    kzg = load_package()
    s = kzg.KZGSettings.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
    rnd = random.Random(4)
    blobs = []
    for _ in range(16):
        b = bytearray(rnd.randbytes(131072))
        for i in range(0, 131072, 32):
            b[i] = 0
        blobs.append(bytes(b))
    commitments = [kzg.blob_to_kzg_commitment(b, s) for b in blobs]
    res = {}
    for t in (1, 4, 16):
        measure(kzg, s, blobs, commitments, t, 0.3)  # warm the lanes
        res["threads_%d" % t] = measure(kzg, s, blobs, commitments, t)
    print(json.dumps(res, indent=1))
    s.close()


if __name__ == "__main__":
    main()

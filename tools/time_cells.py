#!/usr/bin/env python3
"""compute_cells_and_kzg_proofs: FK20 (tuning key fk20=1) against the direct form (=0), proofs only, host buffers."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ctypes as C
import extra_bench as eb

kzg = eb.load_pkg()
L = kzg.lib()
rnd = random.Random(3)
BLOB = 131072
nmax = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 16, 64, 256, nmax]
blobs = bytearray(rnd.randbytes(nmax * BLOB))
for i in range(0, nmax * BLOB, 32):
    blobs[i] = 0
blobs = bytes(blobs)
rows = {}
# the key is read when a settings object is created: one object per form, one after the other (each builds its tables)
for mode in ("1", "0"):
    s = kzg.KZGSettings.from_file(eb.SETUP, kzg.make_config(tuning={"fk20": int(mode)}))
    for n in sorted({m for m in sizes if m <= nmax}):
        if mode == "0" and n > 256:
            continue  # the direct form needs 16 MB of quotient vectors per blob
        proofs = C.create_string_buffer(n * 128 * 48)
        def run():
            rc = L.kzgamd_compute_cells_and_kzg_proofs_batch(None, proofs, blobs, n, C.byref(s.c))
            assert rc == 0
        run()
        run()
        reps = 5 if n <= 64 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        dt = (time.perf_counter() - t0) / reps
        rows.setdefault(n, {})["fk20" if mode == "1" else "direct"] = "%.2f ms, %.0f cell proofs/s" % (dt * 1e3, n * 128 / dt)
    s.close()
for n in sorted(rows):
    print(n, rows[n], flush=True)

#!/usr/bin/env python3
"""G1 transforms: the stage forms by grid size (KZGAMD_G1_WIDE_MAX,_QUAD_MAX,_PAIR_MAX; default) against the single-lane ones (0,0,0), for the FK20
cell proofs of 16 ... 256 blobs and for fft_g1 of 2^7 ... 2^15 points.  Every proof / point of the two variants is
compared.  usage: time_g1.py [wide_max,quad_max,pair_max ...]"""
import ctypes as C
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extra_bench as eb

kzg = eb.load_pkg()
L = kzg.lib()
BLOB = 131072
rnd = random.Random(3)
nmax = 256
blobs = bytearray(rnd.randbytes(nmax * BLOB))
for i in range(0, nmax * BLOB, 32):
    blobs[i] = 0
blobs = bytes(blobs)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
ONLY_FFT, ONLY_CELLS = "--only-fft" in sys.argv, "--only-cells" in sys.argv
variants = args or ["0,0,0", "4096,16384,32768"]  # wide_max,quad_max,pair_max
_st = kzg.KZGSettings.from_file(eb.SETUP)
MONO = bytes((kzg.BlstP1 * 4096).from_address(_st.c.g1_values_monomial))
_st.close()
ref = {}
for v in variants:
    parts = v.split(",")
    os.environ["KZGAMD_G1_WIDE_MAX"], os.environ["KZGAMD_G1_QUAD_MAX"], os.environ["KZGAMD_G1_PAIR_MAX"] = parts[:3]
    if len(parts) > 3:
        os.environ["KZGAMD_G1_BF_MAX"] = parts[3]  # four lanes per butterfly, both GLV halves on one chain
    else:
        os.environ.pop("KZGAMD_G1_BF_MAX", None)
    s = kzg.KZGSettings.from_file(eb.SETUP)
    for n in (() if ONLY_FFT else (16, 32, 64, 128, 256)):
        proofs = C.create_string_buffer(n * 128 * 48)

        def run():
            assert L.kzgamd_compute_cells_and_kzg_proofs_batch(None, proofs, blobs, n, C.byref(s.c)) == 0

        run()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            run()
        dt = (time.perf_counter() - t0) / reps
        same = ref.setdefault(("cells", n), proofs.raw) == proofs.raw
        print("wide_max=%s cells n=%d: %.2f ms (%.0f cell proofs/s) %s" % (v, n, dt * 1e3, n * 128 / dt, "same" if same else "DIFFERENT"), flush=True)
    s.close()
    fs = kzg.FFTSettings(15)
    g2 = kzg.p2_generator()
    for logn in (() if ONLY_CELLS else (7, 10, 12, 15)):
        n = 1 << logn
        pts = (kzg.BlstP1 * n)()
        raw = (MONO * ((n * 144 + len(MONO) - 1) // len(MONO)))[: n * 144]  # the setup's monomial points, repeated
        C.memmove(pts, raw, n * 144)
        out = (kzg.BlstP1 * n)()
        for inverse in (0, 1):
            assert L.fft_g1(fs.handle, out, pts, n, inverse) == 0
            t0 = time.perf_counter()
            assert L.fft_g1(fs.handle, out, pts, n, inverse) == 0
            dt = time.perf_counter() - t0
            # 16 outputs compared with the first variant's as group elements: e(a, G2) == e(b, G2)
            sample = [kzg.BlstP1.from_buffer_copy(bytes(out[i])) for i in range(0, n, n // 16)]
            want = ref.setdefault(("fft", logn, inverse), sample)
            same = all(kzg.pairings_verify(a, g2, b, g2) for a, b in zip(sample, want))
            print("wide_max=%s fft_g1 2^%d inverse=%d: %.2f ms %s" % (v, logn, inverse, dt * 1e3, "same" if same else "DIFFERENT"), flush=True)
    fs.close()

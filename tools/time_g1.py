#!/usr/bin/env python3
"""G1 transforms: the stage forms by grid size (tuning keys g1_wide_max, g1_quad_max, g1_pair_max; default) against the single-lane ones (0,0,0), for the FK20
cell proofs of 16 ... 256 blobs and for fft_g1 of 2^7 ... 2^15 points.  Every proof / point of the two variants is
compared.  usage: time_g1.py [wide_max,quad_max,pair_max ...]"""
import ctypes as C
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

torch.cuda.init()  # before the library pulls in libamdhip64 (both resolve the same SONAME; first loaded wins)
import extra_bench as eb

kzg = eb.load_pkg()
L = kzg.lib()
BLOB = 131072
rnd = random.Random(3)
SIZES = tuple(int(x) for x in os.environ.get("TIME_G1_BLOBS", "16,32,64,128,256").split(","))
nmax = max(SIZES)
blobs = bytearray(rnd.randbytes(nmax * BLOB))
for i in range(0, nmax * BLOB, 32):
    blobs[i] = 0
blobs = bytes(blobs)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
ONLY_FFT, ONLY_CELLS = "--only-fft" in sys.argv, "--only-cells" in sys.argv
variants = args or ["0,0,0", "4096,16384,32768"]  # wide_max,quad_max,pair_max
def distinct_points(n):
    """n distinct G1 points as Jacobian blst_p1 (Z = 1) on the host: the library's generator (h_i * G), so that no
    stage of a transform sees equal or cancelling inputs (a periodic test input makes the early stages trivial)"""
    dev = torch.device("cuda", 0)
    aff = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    kzg.generate_points(aff.data_ptr(), n, 11, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    a = aff.cpu().numpy().reshape(n, 96)
    import numpy as np

    one = np.frombuffer(bytes.fromhex("fdff02000000097602000cc40b00f4ebba58c7535798485f455752705358ce776dec56a2971a075c93e480fac35ef615"), dtype=np.uint8)
    out = np.concatenate([a, np.tile(one, (n, 1))], axis=1)
    return out.tobytes()


MONO = distinct_points(1 << 15)
ref = {}
for v in variants:
    parts = v.split(",")
    os.environ["KZGAMD_TUNING"] = "g1_wide_max=%s;g1_quad_max=%s;g1_pair_max=%s" % tuple(parts[:3])
    s = kzg.KZGSettings.from_file(eb.SETUP)
    for n in (() if ONLY_FFT else SIZES):
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
        raw = MONO[: n * 144]
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

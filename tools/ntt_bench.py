#!/usr/bin/env python3
"""NTT timing on one MI355X: forward transforms, n = 4096 x 256 and n = 2^20 (HIP events, min of 10),
plus a byte-compare of a few sizes against a second run (determinism) — parity itself is tests/test_ntt_gpu.py."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_package

kzg = load_package()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fs = kzg.FFTSettings(scale)
res = {}
shapes = ((4096, 256), (4096, 1024), (1 << 20, 1), (8192, 128), (1 << 16, 16), (2048, 512), (1 << 19, 2))
if len(sys.argv) > 2:  # "n x batch" pairs: 4096x1024,1048576x1
    shapes = tuple(tuple(int(v) for v in p.split("x")) for p in sys.argv[2].split(","))
for n, nb in shapes:
    if n > (1 << scale):
        continue
    a = torch.randint(0, 2**31, (nb * n * 8,), dtype=torch.int32, device=dev)
    a[7::8] &= 0x3FFFFFFF
    b = torch.empty_like(a)
    fs.fft_fr_device(b.data_ptr(), a.data_ptr(), n, nb, False, stream)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fs.fft_fr_device(b.data_ptr(), a.data_ptr(), n, nb, False, stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = min(ts)
    res["%d x %d" % (n, nb)] = {"us": ms * 1e3, "alg_GBps": 64 * n * nb / (ms * 1e-3) / 1e9,
                                "G_fr_mul_per_s": nb * (n / 2) * math.log2(n) / (ms * 1e-3) / 1e9}
for n, nb in ((2048, 256), (4096, 256), (1 << 19, 1)):
    if 2 * n > (1 << scale) or len(sys.argv) > 2:
        continue
    a = torch.randint(0, 2**31, (nb * n * 8,), dtype=torch.int32, device=dev)
    a[7::8] &= 0x3FFFFFFF
    b, t = torch.empty_like(a), torch.empty_like(a)
    ts = []
    for _ in range(11):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fs.das_fft_extension_device(b.data_ptr(), a.data_ptr(), t.data_ptr(), n, nb, stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    res["das %d x %d" % (n, nb)] = {"us": min(ts[1:]) * 1e3}
print(json.dumps(res, indent=1))

#!/usr/bin/env python3
"""2^16 .. 2^22 variable-base MSM timings (device-resident, HIP events, min of 15)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_package

kzg = load_package()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
nmax = 1 << 22
pts = torch.empty(nmax * 96, dtype=torch.uint8, device=dev)
kzg.generate_points(pts.data_ptr(), nmax, 2, stream)
torch.cuda.synchronize()
g = torch.Generator(device=dev)
g.manual_seed(2)
sc = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, generator=g, device=dev)
sc[:, 31] &= 0x3F
out = torch.zeros(144, dtype=torch.uint8, device=dev)
res = {}
for logn in (14, 16, 18, 20, 22):
    n = 1 << logn
    h = kzg.DeviceMsm(pts.data_ptr(), n, False)
    kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, 1, False, stream)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, 1, False, stream)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    res["2^%d" % logn] = round(min(ts), 3)
    h.close()
print(res, out[:8].cpu().tolist())

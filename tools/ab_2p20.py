#!/usr/bin/env python3
"""A/B timing of one 2^20 variable-base MSM under two environments, alternating in one process (box-to-box and run-to-run
noise is larger than most of the differences of interest):  python tools/ab_2p20.py flat_digits=1 [logn]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_package

kzg = load_package()
var = sys.argv[1] if len(sys.argv) > 1 else "flat_digits=1"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
k, v = var.split("=")
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
n = 1 << logn
pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
kzg.generate_points(pts.data_ptr(), n, 2, stream)
torch.cuda.synchronize()
g = torch.Generator(device=dev)
g.manual_seed(2)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g, device=dev)
sc[:, 31] &= 0x3F
out = torch.zeros(144, dtype=torch.uint8, device=dev)
# the tuning keys are read when a handle is created: one handle per configuration, both alive, used alternately
handles = {"default": kzg.DeviceMsm(pts.data_ptr(), n, False)}
handles[var] = kzg.DeviceMsm(pts.data_ptr(), n, False, kzg.make_config(tuning={k: int(v)}))


def run(h):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, 1, False, stream)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


res = {"default": [], var: []}
for rnd in range(12):
    for name in ("default", var):
        run(handles[name])
        res[name].append(min(run(handles[name]) for _ in range(3)))
for name, ts in res.items():
    ts.sort()
    print("%-28s min %.3f  median %.3f ms" % (name, ts[0], ts[len(ts) // 2]))

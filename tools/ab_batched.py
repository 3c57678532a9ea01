#!/usr/bin/env python3
"""A/B of batched large MSMs (nbatch MSMs of 2^logn scalars over one set of bases in ONE call) under several tuning
strings, alternating in one process:  python tools/ab_batched.py [logn] [nbatch] [tuning ...]
e.g.  python tools/ab_batched.py 20 4 sub_streams=0 sub_streams=1 sub_streams=2 sub_streams=3
With `trace` as the last argument: four calls of the FIRST configuration only (for rocprofv3 --kernel-trace + timeline.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_package

kzg = load_package()
args = sys.argv[1:]
trace = bool(args) and args[-1] == "trace"
if trace:
    args = args[:-1]
logn = int(args[0]) if args else 20
nbatch = int(args[1]) if len(args) > 1 else 4
variants = args[2:] or ["sub_streams=0", "sub_streams=2"]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
n = 1 << logn
pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
kzg.generate_points(pts.data_ptr(), n, 2, stream)
torch.cuda.synchronize()
g = torch.Generator(device=dev)
g.manual_seed(2)
sc = torch.randint(0, 256, (nbatch * n, 32), dtype=torch.uint8, generator=g, device=dev)
sc[:, 31] &= 0x3F
out = torch.zeros(144 * nbatch, dtype=torch.uint8, device=dev)


def cfg(var):
    t = {}
    for kv in var.split(";"):
        if kv and kv != "default":
            k, v = kv.split("=")
            t[k] = int(v)
    return kzg.make_config(tuning=t) if t else None


handles = {v: kzg.DeviceMsm(pts.data_ptr(), n, False, cfg(v)) for v in variants}


def run(h, nb):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, nb, False, stream)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


if trace:
    h = handles[variants[0]]
    for _ in range(4):
        run(h, nbatch)
    sys.exit(0)
res = {v: [] for v in variants}
one = {v: [] for v in variants}
for rnd in range(8):
    for v in variants:
        run(handles[v], nbatch)
        res[v].append(min(run(handles[v], nbatch) for _ in range(3)))
        one[v].append(min(run(handles[v], 1) for _ in range(3)))
for v in variants:
    ts, t1 = sorted(res[v]), sorted(one[v])
    print("%-28s %d x 2^%d: min %.3f median %.3f ms per call = %.3f ms per MSM   (one MSM alone: %.3f)"
          % (v, nbatch, logn, ts[0], ts[len(ts) // 2], ts[len(ts) // 2] / nbatch, t1[len(t1) // 2]))

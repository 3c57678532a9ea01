#!/usr/bin/env python3
"""One MSM over n = 2^logn resident bases: variable-base handle against a PREPARED handle (fixed-base rows
2^(c w) P_i built once at creation, every window into one bucket set — the reference's BGMW idea,
kzg/src/msm/bgmw.rs, on the bucket engine), alternating in one process; results compared.
python tools/prep_2p20.py [logn]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_package

kzg = load_package()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
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
outs = {}
handles = {}
for name, prep in (("variable", False), ("prepared", True)):
    t0 = time.time()
    handles[name] = kzg.DeviceMsm(pts.data_ptr(), n, prep)
    torch.cuda.synchronize()
    print(name, "handle: %.1f ms to create" % ((time.time() - t0) * 1e3), handles[name].info())
    outs[name] = torch.zeros(144, dtype=torch.uint8, device=dev)


def run(name):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    kzg.msm_prepared_batch_device(handles[name], outs[name].data_ptr(), sc.data_ptr(), n, 1, False, stream)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


res = {k: [] for k in handles}
for rnd in range(10):
    for name in handles:
        run(name)
        res[name].append(min(run(name) for _ in range(3)))
for name, ts in res.items():
    ts.sort()
    print("%-10s min %.3f  median %.3f ms" % (name, ts[0], ts[len(ts) // 2]))
import ctypes as C
sys.path.insert(0, os.path.join(ROOT, "oracle"))
a = bytes(outs["variable"].cpu().numpy().tobytes())
b = bytes(outs["prepared"].cpu().numpy().tobytes())
print("jacobian outputs equal as bytes:", a == b, "(representatives may differ; tests compare compressed points)")

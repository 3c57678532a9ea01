#!/usr/bin/env python3
"""n = 2^20 (and 2^21, 2^22) variable-base MSM against the tuning key tail_pieces (1 = the accumulation and the whole
reduction back to back; 2 .. 4 = the reduction of a piece of the window sets beside the accumulation of the next), and the
batched figure (nbatch MSMs of 2^20 in one call: ms per MSM).  Device-resident, HIP events, min of 15, handles alternated."""
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
out = torch.zeros(144 * 8, dtype=torch.uint8, device=dev)


def t_of(h, n, nbatch=1, reps=15):
    kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, nbatch, False, stream)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, nbatch, False, stream)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


for logn in (20, 21, 22):
    n = 1 << logn
    hs = {p: kzg.DeviceMsm(pts.data_ptr(), n, False, kzg.make_config(tuning={"tail_pieces": p})) for p in (1, 2, 3, 4)}
    row = {}
    for rnd in range(2):
        for p, h in hs.items():
            t = t_of(h, n)
            row[p] = min(row.get(p, 1e9), t)
    print("2^%d" % logn, {("pieces=%d" % p): round(t, 3) for p, t in row.items()}, flush=True)
    for h in hs.values():
        h.close()
n = 1 << 20
h = kzg.DeviceMsm(pts.data_ptr(), n, False)
for nb in (1, 2, 4):
    t = t_of(h, n, nb, reps=9)
    print("2^20 x %d in one call: %.3f ms = %.3f ms per MSM" % (nb, t, t / nb), flush=True)
h.close()

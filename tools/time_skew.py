#!/usr/bin/env python3
"""Variable-base MSM under skewed scalar distributions (n = 2^20): uniform, all equal, 64-bit, two values."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extra_bench as eb
import torch
kzg = eb.load_pkg()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
kzg.generate_points(pts.data_ptr(), n, 2, stream)
g = torch.Generator(device="cpu"); g.manual_seed(2)
base = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g); base[:, 31] &= 0x3F
cases = {"uniform": base.clone()}
eq = base.clone(); eq[:] = base[0]; cases["all equal"] = eq
sm = base.clone(); sm[:, 8:] = 0; cases["64-bit scalars"] = sm
two = base.clone(); two[::2] = base[0]; two[1::2] = base[1]; cases["two values"] = two
out = torch.zeros(144, dtype=torch.uint8, device=dev)
h = kzg.DeviceMsm(pts.data_ptr(), n, False)
for name, sc in cases.items():
    d = sc.to(dev)
    fn = lambda: kzg.msm_prepared_batch_device(h, out.data_ptr(), d.data_ptr(), n, 1, False, stream)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("%-16s %8.3f ms" % (name, min(ts)), os.environ.get("KZGAMD_TUNING", ""), flush=True)

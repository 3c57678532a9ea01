#!/usr/bin/env python3
"""Time the device-resident variable-base MSM at n = 2^logn (default 20) under the current environment."""
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
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g); sc[:, 31] &= 0x3F; sc = sc.to(dev)
out = torch.zeros(144, dtype=torch.uint8, device=dev)
h = kzg.DeviceMsm(pts.data_ptr(), n, False)
fn = lambda: kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, 1, False, stream)
fn(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print("logn", logn, "groups", os.environ.get("KZGAMD_GROUPS", "default"), "ms", round(min(ts), 3), [round(t, 3) for t in ts], out.cpu().numpy()[:8].tolist())

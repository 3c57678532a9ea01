#!/usr/bin/env python3
"""Window sweep for the variable-base engine: for each n, time the MSM for every `window` candidate (KZGAMD_TUNING).
Run on an MI355X:  python tools/sweep_window.py [logn ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extra_bench as eb  # noqa: E402
import torch  # noqa: E402

kzg = eb.load_pkg()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
logns = [int(a) for a in sys.argv[1:]] or [12, 14, 16, 18, 20, 21, 22]
nmax = 1 << max(logns)
pts = torch.empty(nmax * 96, dtype=torch.uint8, device=dev)
kzg.generate_points(pts.data_ptr(), nmax, 2, stream)
g = torch.Generator(device="cpu")
g.manual_seed(2)
sc = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, generator=g)
sc[:, 31] &= 0x3F
sc = sc.to(dev)
out = torch.zeros(144, dtype=torch.uint8, device=dev)
res = {}
for logn in logns:
    n = 1 << logn
    row = {}
    for c in range(max(4, logn - 8), min(22, logn + 1)):
        os.environ["KZGAMD_TUNING"] = "window=%d" % c
        h = kzg.DeviceMsm(pts.data_ptr(), n, False)
        fn = lambda: kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, 1, False, stream)
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        row[c] = round(min(ts), 3)
        h.close()
    res[logn] = row
    print(logn, row, flush=True)
os.environ.pop("KZGAMD_TUNING", None)
print(json.dumps(res))

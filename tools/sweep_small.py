#!/usr/bin/env python3
"""Small variable-base MSMs: window x reduction form.  For n = 2^10 .. 2^18 and every window candidate, the time of one MSM
with the digit-decomposed (tiled) reduction from 1024 buckets up (digit_min_log=10) and with round 5's threshold (14: the
tree below 16384 buckets).  python tools/sweep_small.py [logn ...]"""
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
logns = [int(a) for a in sys.argv[1:]] or [10, 11, 12, 13, 14, 15, 16, 17, 18]
nmax = 1 << max(logns)
pts = torch.empty(nmax * 96, dtype=torch.uint8, device=dev)
kzg.generate_points(pts.data_ptr(), nmax, 2, stream)
g = torch.Generator(device="cpu")
g.manual_seed(2)
sc = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, generator=g)
sc[:, 31] &= 0x3F
sc = sc.to(dev)
out = torch.zeros(144, dtype=torch.uint8, device=dev)
ref = {}
for logn in logns:
    n = 1 << logn
    row = {}
    for c in (9, 10, 11, 12, 13, 14, 16):
        for dml in (10, 14):
            h = kzg.DeviceMsm(pts.data_ptr(), n, False, kzg.make_config(tuning={"window": c, "digit_min_log": dml}))
            fn = lambda: kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, 1, False, stream)
            fn()
            torch.cuda.synchronize()
            res = bytes(out.cpu().numpy().tobytes())
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            row["c%d/%s" % (c, "digits" if dml == 10 else "r5")] = round(min(ts), 3)
            h.close()
            # every configuration computes the same point (projective representations differ: compare through the library)
            ref.setdefault(logn, []).append(res)
    print(logn, row, flush=True)

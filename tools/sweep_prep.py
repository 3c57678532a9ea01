import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extra_bench as eb
import torch
kzg = eb.load_pkg()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
n = 1 << 20
pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
kzg.generate_points(pts.data_ptr(), n, 2, stream)
g = torch.Generator(device="cpu"); g.manual_seed(2)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g); sc[:, 31] &= 0x3F; sc = sc.to(dev)
out = torch.zeros(144, dtype=torch.uint8, device=dev)
for prep in (1, 0):
    for c in ([13, 14, 15, 16, 17, 18, 20] if prep else [13, 15, 16, 17, 18, 19]):
        os.environ["KZGAMD_TUNING"] = ("window_prepared=%d" if prep else "window=%d") % c
        h = kzg.DeviceMsm(pts.data_ptr(), n, bool(prep))
        f = lambda: kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, 1, False, stream)
        f(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        print("prepared" if prep else "variable", "c=%d" % c, "rows", h.info()["rows"], "%.2f ms" % min(ts), flush=True)
        h.close()

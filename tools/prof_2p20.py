"""One MSM of 2^logn points on the bucket engine, four calls on one handle (for rocprofv3): prof_2p20.py [logn] [fixed]
(fixed: a prepared handle — table rows 2^(c j) P, one bucket set; tuning key window_prepared picks c)"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extra_bench as eb
import torch
kzg = eb.load_pkg()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)  # log2 of the MSM size
pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
kzg.generate_points(pts.data_ptr(), n, 2, stream)
torch.cuda.synchronize()  # handles copy the points on their own non-blocking streams
g = torch.Generator(device="cpu"); g.manual_seed(2)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g); sc[:, 31] &= 0x3F; sc = sc.to(dev)
out = torch.zeros(144, dtype=torch.uint8, device=dev)
h = kzg.DeviceMsm(pts.data_ptr(), n, len(sys.argv) > 2 and sys.argv[2] == "fixed")
for _ in range(4):
    kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, 1, False, stream)
torch.cuda.synchronize()

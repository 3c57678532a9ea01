#!/usr/bin/env python3
"""a few n = 2^20 MSMs with tail_pieces = argv[1] (for rocprofv3 --kernel-trace: the timeline of one call)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_package
kzg = load_package()
pieces = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = 1 << 20
stream = torch.cuda.current_stream().cuda_stream
pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
kzg.generate_points(pts.data_ptr(), n, 2, stream)
g = torch.Generator(device="cuda"); g.manual_seed(2)
sc = torch.randint(0, 256, (n * nbatch, 32), dtype=torch.uint8, generator=g, device="cuda"); sc[:, 31] &= 0x3F
out = torch.zeros(144 * nbatch, dtype=torch.uint8, device="cuda")
h = kzg.DeviceMsm(pts.data_ptr(), n, False, kzg.make_config(tuning={"tail_pieces": pieces}))
for _ in range(4):
    kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, nbatch, False, stream)
    torch.cuda.synchronize()
h.close()

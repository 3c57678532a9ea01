#!/usr/bin/env python3
"""Does the accumulation's time depend on the scalars?  1024-blob commitment batches (device-resident, one stream) with the
bench's blobs (top byte of every element zero: 248-bit scalars) and with elements uniform below r (top byte random below
0x73): the proof path multiplies by quotient coefficients of the second kind.  python tools/time_commit_scalar_range.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extra_bench as eb
import torch
kzg = eb.load_pkg()
dev = torch.device("cuda", 0)
B = 1024
s = kzg.KZGSettings.from_file(eb.SETUP)
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(7)
res = {}
for name in ("top byte zero", "uniform below r", "top byte zero", "uniform below r"):
    blobs = torch.randint(0, 256, (B, 4096, 32), dtype=torch.uint8, generator=g, device=dev)
    if name == "top byte zero":
        blobs[:, :, 0] = 0
    else:
        blobs[:, :, 0] = torch.randint(0, 0x73, (B, 4096), dtype=torch.uint8, generator=g, device=dev)
    cm = torch.zeros(B * 48, dtype=torch.uint8, device=dev)
    stat = torch.zeros(B, dtype=torch.int32, device=dev)
    scr = torch.empty(B * 131072, dtype=torch.uint8, device=dev)
    fn = lambda: kzg.blob_to_kzg_commitment_device(cm.data_ptr(), stat.data_ptr(), scr.data_ptr(), blobs.data_ptr(), B, s, st)
    fn(); torch.cuda.synchronize()
    assert int(stat.abs().sum()) == 0
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("%-18s %.3f ms per batch of %d (min of 6; median %.3f)" % (name, min(ts), B, sorted(ts)[3]), flush=True)
s.close()

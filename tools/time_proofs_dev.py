#!/usr/bin/env python3
"""Device-resident proof pipeline under the two forms of the device Fiat-Shamir hash (tuning key sha_lanes = 4 / 1):
ms per lone batch of 1024 blobs on one stream (latency) and proofs/s with batches rotating over four streams
(throughput, bench.py's blob_proofs_device_resident shape).  python tools/time_proofs_dev.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extra_bench as eb
import torch
kzg = eb.load_pkg()
dev = torch.device("cuda", 0)
B, NB, NS = 1024, 8, 4
g = torch.Generator(device=dev); g.manual_seed(7)
blobs = torch.randint(0, 256, (B, 131072), dtype=torch.uint8, generator=g, device=dev)
blobs[:, ::32] = 0
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
for lanes in (4, 1):
    s = kzg.KZGSettings.from_file(eb.SETUP, kzg.make_config(table_budget_gb=100, tuning={"sha_lanes": lanes}))
    st = torch.cuda.current_stream().cuda_stream
    cm = torch.zeros(B * 48, dtype=torch.uint8, device=dev)
    stat = torch.zeros(B, dtype=torch.int32, device=dev)
    scr = torch.empty(B * 131072, dtype=torch.uint8, device=dev)
    kzg.blob_to_kzg_commitment_device(cm.data_ptr(), stat.data_ptr(), scr.data_ptr(), blobs.data_ptr(), B, s, st)
    prs = [torch.zeros(B * 48, dtype=torch.uint8, device=dev) for _ in range(NS)]
    pstat = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NS)]
    pscr = [torch.empty(B * kzg.PROOF_SCRATCH_BYTES, dtype=torch.uint8, device=dev) for _ in range(NS)]
    torch.cuda.synchronize()

    def one(j, stream):
        kzg.compute_blob_kzg_proof_device(prs[j].data_ptr(), pstat[j].data_ptr(), pscr[j].data_ptr(), blobs.data_ptr(), cm.data_ptr(), B, s, stream)

    one(0, st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); one(0, st); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    for _ in range(2):
        for k in range(NB):
            one(k % NS, streams[k % NS].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        for k in range(NB):
            one(k % NS, streams[k % NS].cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    ref = prs[0].cpu().numpy().tobytes()
    assert all(p.cpu().numpy().tobytes() == ref for p in prs) and int(sum(int(x.abs().sum()) for x in pstat)) == 0
    print("sha_lanes=%d  lone batch of %d: %.2f ms   pipelined on %d streams: %.1f k proofs/s" % (lanes, B, min(ts), NS, NB * B / dt / 1e3))
    if lanes == 4:
        ref4 = ref
    else:
        assert ref == ref4, "the two hash forms disagree"
    s.close()

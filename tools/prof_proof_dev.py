"""Device-resident proof pipeline, 1024 blobs per batch, a few batches on one stream (for rocprofv3 --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extra_bench as eb
import torch
kzg = eb.load_pkg()
s = kzg.KZGSettings.from_file(eb.SETUP)
dev = torch.device("cuda", 0)
B = 1024
g = torch.Generator(device=dev); g.manual_seed(7)
blobs = torch.randint(0, 256, (B, 131072), dtype=torch.uint8, generator=g, device=dev)
blobs[:, ::32] = 0
st = torch.cuda.current_stream().cuda_stream
cm = torch.zeros(B * 48, dtype=torch.uint8, device=dev)
stat = torch.zeros(B, dtype=torch.int32, device=dev)
scr = torch.empty(B * 131072, dtype=torch.uint8, device=dev)
kzg.blob_to_kzg_commitment_device(cm.data_ptr(), stat.data_ptr(), scr.data_ptr(), blobs.data_ptr(), B, s, st)
pr = torch.zeros(B * 48, dtype=torch.uint8, device=dev)
pscr = torch.empty(B * kzg.PROOF_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
for _ in range(4):
    kzg.compute_blob_kzg_proof_device(pr.data_ptr(), stat.data_ptr(), pscr.data_ptr(), blobs.data_ptr(), cm.data_ptr(), B, s, st)
torch.cuda.synchronize()
s.close()

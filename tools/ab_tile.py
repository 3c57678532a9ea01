"""A/B in one process: variable-base MSM with k_tile_sums_loop (default) against KZGAMD_TILE_V1=1, alternating handles:
ab_tile.py [logn]"""
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
torch.cuda.synchronize()
g = torch.Generator(device=dev); g.manual_seed(2)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g, device=dev); sc[:, 31] &= 0x3F
o = torch.zeros(144, dtype=torch.uint8, device=dev)
hs = {}
for name, env in (("loop", None), ("v1", "1")):
    if env: os.environ["KZGAMD_TILE_V1"] = env
    else: os.environ.pop("KZGAMD_TILE_V1", None)
    hs[name] = kzg.DeviceMsm(pts.data_ptr(), n, False)
os.environ.pop("KZGAMD_TILE_V1", None)
res = {k: [] for k in hs}
outs = {}
for rep in range(12):
    for name, h in hs.items():
        kzg.msm_prepared_batch_device(h, o.data_ptr(), sc.data_ptr(), n, 1, False, stream); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); kzg.msm_prepared_batch_device(h, o.data_ptr(), sc.data_ptr(), n, 1, False, stream); b.record()
        torch.cuda.synchronize()
        res[name].append(a.elapsed_time(b)); outs[name] = bytes(o.cpu().numpy())
for name, v in res.items():
    v.sort()
    print("2^%d %-5s min %.3f median %.3f ms" % (logn, name, v[0], v[len(v) // 2]))
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
f = lambda s, k: int.from_bytes(s[48 * k:48 * k + 48], "little")
a, b = outs["loop"], outs["v1"]  # Jacobian, Montgomery limbs: the factor cancels in the cross products
same = (f(a, 0) * f(b, 2) ** 2 - f(b, 0) * f(a, 2) ** 2) % P == 0 and (f(a, 1) * f(b, 2) ** 3 - f(b, 1) * f(a, 2) ** 3) % P == 0
print("same point" if same else "DIFFERENT POINTS")

"""Fixed-base (prepared: one bucket set, table rows 2^(c j) P) against variable-base bucket engine on the same points:
time_prepared.py [logn ...]   (windows tried: TIME_PREPARED_WINDOWS=0,16,18,20; 0 = the engine's own choice)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extra_bench as eb
import torch
kzg = eb.load_pkg()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def same_point(a, b):
    """two 144-byte Jacobian blst_p1 (Montgomery limbs): equal as points (the Montgomery factor cancels on both sides)"""
    f = lambda s, k: int.from_bytes(s[48 * k:48 * k + 48], "little")
    x1, y1, z1, x2, y2, z2 = f(a, 0), f(a, 1), f(a, 2), f(b, 0), f(b, 1), f(b, 2)
    if z1 == 0 or z2 == 0:
        return z1 == z2
    return (x1 * z2 * z2 - x2 * z1 * z1) % P == 0 and (y1 * z2 ** 3 - y2 * z1 ** 3) % P == 0


def ev_time(fn, reps=7):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


logns = [int(a) for a in sys.argv[1:]] or [16, 18, 20, 21, 22]
windows = [int(w) for w in os.environ.get("TIME_PREPARED_WINDOWS", "0,16,18,20").split(",")]
nmax = 1 << max(logns)
pts = torch.empty(nmax * 96, dtype=torch.uint8, device=dev)
kzg.generate_points(pts.data_ptr(), nmax, 2, stream)
torch.cuda.synchronize()
g = torch.Generator(device=dev); g.manual_seed(2)
sc = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, generator=g, device=dev); sc[:, 31] &= 0x3F
o = torch.zeros(144, dtype=torch.uint8, device=dev)
for logn in logns:
    n = 1 << logn
    h = kzg.DeviceMsm(pts.data_ptr(), n, False)
    ms = ev_time(lambda: kzg.msm_prepared_batch_device(h, o.data_ptr(), sc.data_ptr(), n, 1, False, stream))
    ref = bytes(o.cpu().numpy()); h.close()
    print("2^%d variable-base  c=%d  %.3f ms" % (logn, 16, ms), flush=True)
    for w in windows:
        if w: os.environ["KZGAMD_TUNING"] = "window_prepared=%d" % w
        else: os.environ.pop("KZGAMD_TUNING", None)
        import time
        t0 = time.perf_counter()
        h = kzg.DeviceMsm(pts.data_ptr(), n, True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        hi = h.info()
        o.zero_()
        ms = ev_time(lambda: kzg.msm_prepared_batch_device(h, o.data_ptr(), sc.data_ptr(), n, 1, False, stream))
        ok = same_point(ref, bytes(o.cpu().numpy()))
        print("2^%d fixed-base     c=%d rows=%d  %.3f ms  (handle %.0f ms)  %s" % (logn, hi["window_bits"], hi["rows"], ms, (t1 - t0) * 1e3,
                                                                          "same point" if ok else "MISMATCH"), flush=True)
        h.close()

#!/usr/bin/env python3
"""Differential fuzz of the MSM engine against the CPU oracle: random sizes, scalar shapes (uniform, short, equal,
few distinct, r - small, GLV boundaries), points with repeats / negations / infinity, host and device entry points.
Not part of the test suite (minutes of GPU + CPU time):  python tools/fuzz_msm.py [seconds] [seed]"""
import ctypes as C
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import extra_bench as eb  # noqa: E402
import oracle_ffi as O  # noqa: E402
import torch  # noqa: E402

kzg = eb.load_pkg()
L = O.lib()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
X2 = 0xd201000000010000 ** 2
stream = torch.cuda.current_stream().cuda_stream

g = O.G1()
L.og1_generator(C.byref(g))
POOL = 512
pool = (O.G1Affine * POOL)()
for i in range(POOL):
    t = O.G1()
    k = O.fr_from_int(rnd.randrange(1, O.R))
    L.og1_mul(C.byref(t), C.byref(g), C.byref(k))
    L.og1_to_affine(C.byref(pool[i]), C.byref(t))


def neg_affine(p):
    q = O.G1Affine()
    C.memmove(C.byref(q), C.byref(p), 96)
    y = O.fp_to_int(p.y)
    q.y = O.fp_from_int((O.P - y) % O.P)
    return q


def scalar(kind):
    if kind == 0:
        return rnd.randrange(O.R)
    if kind == 1:
        return rnd.randrange(1 << rnd.choice([1, 8, 64, 128, 200]))
    if kind == 2:
        return O.R - 1 - rnd.randrange(1 << 16)
    if kind == 3:
        return (rnd.randrange(1 << 127) * X2 + rnd.choice([0, 1, X2 - 1, X2 // 2, X2 // 2 + 1])) % O.R
    return 0


def compressed(p):
    buf = C.create_string_buffer(48)
    L.og1_compress(buf, C.byref(p))
    return buf.raw


t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    n = rnd.choice([1, 2, 3, 7, 8, 9, 63, 64, 65, 100, 255, 256, 257, 1000, 1023, 1025, 4096, 5000, 20000, 40000, 70000])
    pts = (O.G1Affine * n)()
    mode = rnd.randrange(4)
    for i in range(n):
        if mode == 1 and i and rnd.random() < 0.3:
            pts[i] = pts[rnd.randrange(i)]
        elif mode == 2 and i and rnd.random() < 0.3:
            pts[i] = neg_affine(pts[rnd.randrange(i)])
        elif mode == 3 and rnd.random() < 0.1:
            pts[i] = O.G1Affine()
        else:
            pts[i] = pool[rnd.randrange(POOL)]
    smode = rnd.randrange(5)
    if smode == 0:
        vals = [scalar(0) for _ in range(n)]
    elif smode == 1:
        vals = [scalar(rnd.randrange(5)) for _ in range(n)]
    elif smode == 2:
        v = scalar(rnd.randrange(4))
        vals = [v] * n
    elif smode == 3:
        few = [scalar(rnd.randrange(4)) for _ in range(3)]
        vals = [rnd.choice(few) for _ in range(n)]
    else:
        vals = [scalar(1) for _ in range(n)]
    sc = O.fr_array(vals)
    exp = O.G1()
    L.omsm_affine(C.byref(exp), pts, sc, n)
    want = compressed(exp)
    got = O.G1()
    C.memmove(C.byref(got), bytes(kzg.multi_scalar_mult(pts, sc, n)), 144)
    assert compressed(got) == want, ("host variable-base", n, mode, smode, seed, cases)
    if n <= 5000 and rnd.random() < 0.3:
        h = kzg.prepare_multi_scalar_mult(pts, n)
        C.memmove(C.byref(got), bytes(kzg.multi_scalar_mult_prepared(h, sc, n)), 144)
        h.close()
        assert compressed(got) == want, ("prepared", n, mode, smode, seed, cases)
    # device handle, canonical scalars
    d_pts = torch.frombuffer(bytearray(bytes(pts)), dtype=torch.uint8).cuda()
    raw = b"".join(v.to_bytes(32, "little") for v in vals)
    d_sc = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    d_out = torch.zeros(144, dtype=torch.uint8, device="cuda")
    h = kzg.DeviceMsm(d_pts.data_ptr(), n, False)
    kzg.msm_prepared_batch_device(h, d_out.data_ptr(), d_sc.data_ptr(), n, 1, False, stream)
    torch.cuda.synchronize()
    h.close()
    C.memmove(C.byref(got), d_out.cpu().numpy().tobytes(), 144)
    assert compressed(got) == want, ("device variable-base", n, mode, smode, seed, cases)
    cases += 1
print("fuzz ok:", cases, "cases, seed", seed, "library", os.path.basename(kzg.LIB_PATH))

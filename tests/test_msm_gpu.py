"""MSM parity on the GPU: HIP Pippenger (through the C-ABI, B1) vs the CPU oracle.
Mirrors the reference's MSM tests (kzg-bench/src/tests/bls12_381.rs:184-387) and the fuzz
invariant (fuzz/src/lib.rs:81-96: every MSM variant == sequential Pippenger bytes)."""
import ctypes as C
import random

import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def gen_points(L, n, rnd, base=None):
    g = O.G1()
    L.og1_generator(C.byref(g))
    pts = (O.G1Affine * max(n, 1))()
    for i in range(n):
        t = O.G1()
        kf = O.fr_from_int(rnd.randrange(1, O.R))
        L.og1_mul(C.byref(t), C.byref(g), C.byref(kf))
        L.og1_to_affine(C.byref(pts[i]), C.byref(t))
    return pts


def compressed(L, p):
    buf = C.create_string_buffer(48)
    L.og1_compress(buf, C.byref(p))
    return buf.raw


def as_oracle_g1(p1):
    g = O.G1()
    C.memmove(C.byref(g), C.byref(p1), 144)
    return g


def check(L, kzg, pts, sc, n, prepared=None, unprepared=True):
    exp = O.G1()
    L.omsm_affine(C.byref(exp), pts, sc, n)
    want = compressed(L, exp)
    if prepared is not None:
        got = as_oracle_g1(kzg.multi_scalar_mult_prepared(prepared, sc, n))
        assert compressed(L, got) == want
    if unprepared and n > 0:
        got = as_oracle_g1(kzg.multi_scalar_mult(pts, sc, n))
        assert compressed(L, got) == want
    return want


def test_small_sizes_with_and_without_precomputation(oracle, kzg):
    # every n in a range, output buffer semantics (bls12_381.rs:314-387)
    L = oracle.lib()
    rnd = random.Random(11)
    pts = gen_points(L, 130, rnd)
    sc = O.fr_array([rnd.randrange(O.R) for _ in range(130)])
    h = kzg.prepare_multi_scalar_mult(pts, 130)
    for n in list(range(1, 20)) + [31, 32, 33, 64, 100, 128, 130]:
        check(L, kzg, pts, sc, n, prepared=h)
    h.close()


def test_every_size_up_to_128_into_a_garbage_filled_output(oracle, kzg):
    """The reference's exhaustive small sweep (kzg-bench/src/tests/bls12_381.rs:314-387): EVERY n in 0..=128 — without
    precomputation, with ONE handle prepared for all 128 points, and with a handle prepared for exactly n points — each
    time into an output the caller filled with a random point first (`let mut res = TG1::rand()`): the result must not
    depend on what the output held.  Expected values are the oracle's running sums of [s_i]P_i."""
    L = oracle.lib()
    rnd = random.Random(2024)
    N = 128
    pts = gen_points(L, N, rnd)
    sc = O.fr_array([rnd.randrange(O.R) for _ in range(N)])
    # results[i] = sum of the first i terms, as the reference builds them (point by point)
    results = [bytes([0xC0]) + bytes(47)]
    cur = O.G1()
    C.memset(C.byref(cur), 0, C.sizeof(cur))
    for i in range(N):
        pj = O.G1()
        L.og1_from_affine(C.byref(pj), C.byref(pts[i]))
        term = O.G1()
        L.og1_mul(C.byref(term), C.byref(pj), C.byref(sc[i]))
        nxt = O.G1()
        L.og1_add_or_dbl(C.byref(nxt), C.byref(cur), C.byref(term))
        cur = nxt
        results.append(compressed(L, cur))
    garbage = [gen_points(L, 1, rnd)[0] for _ in range(4)]

    def garbage_out(k):
        # a valid random point in blst's Jacobian layout (x, y, z = R mod p): what TG1::rand() leaves in `res`
        g = O.G1()
        L.og1_from_affine(C.byref(g), C.byref(garbage[k % 4]))
        out = kzg.BlstP1()
        C.memmove(C.byref(out), C.byref(g), 144)
        return out

    lib = kzg.lib()
    whole = kzg.prepare_multi_scalar_mult(pts, N)
    small = kzg.make_config(table_budget_gb=0.25)  # 128 handles are built below: a table of c = 10, not the default 160 GB budget's
    for n in range(N + 1):
        out = garbage_out(n)
        err = lib.mult_pippenger(C.byref(out), pts, n, sc)
        assert err.code == 0, (n, err.code)
        assert compressed(L, as_oracle_g1(out)) == results[n], ("unprepared", n)
        out = garbage_out(n + 1)
        err = lib.mult_pippenger_prepared(whole.handle, C.byref(out), n, sc)
        assert err.code == 0, (n, err.code)
        assert compressed(L, as_oracle_g1(out)) == results[n], ("prepared for 128", n)
        if n > 0:  # prepare_msm of no points has no handle to return (the reference's precompute yields None there)
            own = kzg.prepare_multi_scalar_mult(pts, n, small)
            out = garbage_out(n + 2)
            err = lib.mult_pippenger_prepared(own.handle, C.byref(out), n, sc)
            assert err.code == 0, (n, err.code)
            assert compressed(L, as_oracle_g1(out)) == results[n], ("prepared for n", n)
            own.close()
    whole.close()


def test_edge_scalars_and_points(oracle, kzg):
    L = oracle.lib()
    rnd = random.Random(12)
    n = 96
    pts = gen_points(L, n, rnd)
    vals = [rnd.randrange(O.R) for _ in range(n)]
    for i in range(0, n, 7):
        vals[i] = 0                      # ~10% zero scalars (bls12_381.rs:265-311)
    vals[1], vals[2], vals[3] = 1, O.R - 1, O.R - 2
    vals[4] = (1 << 254)                 # top bits
    vals[5] = (1 << 255) % O.R
    for i in range(5, n, 9):
        pts[i] = O.G1Affine()            # infinity points
    pts[10] = pts[11]                    # equal points (doubling inside a bucket when digits agree)
    vals[10] = vals[11]
    pts[20] = pts[21]
    vals[20] = O.R - vals[21]            # P and -P with equal digits -> cancels to infinity in a bucket
    sc = O.fr_array(vals)
    h = kzg.prepare_multi_scalar_mult(pts, n)
    check(L, kzg, pts, sc, n, prepared=h)
    # all-equal scalars, all-zero scalars, all-same point
    sc2 = O.fr_array([vals[7]] * n)
    check(L, kzg, pts, sc2, n, prepared=h)
    sc0 = O.fr_array([0] * n)
    got = check(L, kzg, pts, sc0, n, prepared=h)
    assert got == b"\xc0" + bytes(47)
    h.close()
    same = (O.G1Affine * n)(*[pts[0]] * n)
    hs = kzg.prepare_multi_scalar_mult(same, n)
    check(L, kzg, same, sc, n, prepared=hs)
    check(L, kzg, same, sc2, n, prepared=hs)
    hs.close()


def curve_points_outside_g1(count, seed):
    """(x, y) with y^2 = x^3 + 4 for small x: on the curve, and (the cofactor is ~2^126) outside the r-torsion subgroup"""
    out, x = [], seed
    while len(out) < count:
        x += 1
        rhs = (pow(x, 3, O.P) + 4) % O.P
        y = pow(rhs, (O.P + 1) // 4, O.P)
        if y * y % O.P == rhs:
            out.append((x, y))
    return out


def test_bases_outside_the_subgroup(oracle, kzg):
    """FsG1::from_bytes accepts any curve point (blst/src/types/g1.rs:65-87) and g1_lincomb is a plain sum of k_i P_i for
    them too; the engines' GLV split is an identity of G1 only.  One / several bases outside G1 through the unprepared
    entry point (membership test at creation -> unsplit engine), a device handle, the prepared path (table without the
    split), and a call large enough to skip the test (unsplit engine outright): all equal the oracle's naive sum."""
    import torch

    L = oracle.lib()
    rnd = random.Random(31)
    for n, nbad in ((1, 1), (9, 1), (300, 3), (4096, 1), (40000, 2)):
        pts = gen_points(L, min(n, 300), rnd)
        if n > 300:  # large sets: repeat a few hundred random points (the oracle does not mind)
            big = (O.G1Affine * n)()
            for i in range(n):
                big[i] = pts[i % 300]
            pts = big
        bad = curve_points_outside_g1(nbad, 1000 * n)
        where = rnd.sample(range(n), nbad)
        for (x, y), i in zip(bad, where):
            pts[i].x, pts[i].y = O.fp_from_int(x), O.fp_from_int(y)
        # sanity: the oracle agrees the planted point is on the curve and not in G1
        probe = O.G1()
        L.og1_from_affine(C.byref(probe), C.byref(pts[where[0]]))
        assert L.og1_affine_on_curve(C.byref(pts[where[0]])) == 1 and L.og1_in_subgroup(C.byref(probe)) == 0
        sc = O.fr_array([rnd.randrange(O.R) for _ in range(n)])
        exp = O.G1()
        (L.omsm_naive if n <= 300 else L.omsm_affine)(C.byref(exp), pts, sc, n)
        want = compressed(L, exp)
        assert compressed(L, as_oracle_g1(kzg.multi_scalar_mult(pts, sc, n))) == want, ("unprepared", n)
        if n <= 4096:
            h = kzg.prepare_multi_scalar_mult(pts, n)
            assert compressed(L, as_oracle_g1(kzg.multi_scalar_mult_prepared(h, sc, n))) == want, ("prepared", n)
            h.close()
        d_pts = torch.frombuffer(bytearray(bytes(pts)), dtype=torch.uint8).cuda()
        d_sc = torch.frombuffer(bytearray(bytes(sc)), dtype=torch.uint8).cuda()
        d_out = torch.zeros(144, dtype=torch.uint8, device="cuda")
        h = kzg.DeviceMsm(d_pts.data_ptr(), n, False)
        kzg.msm_prepared_batch_device(h, d_out.data_ptr(), d_sc.data_ptr(), n, 1, True, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = O.G1()
        C.memmove(C.byref(got), d_out.cpu().numpy().tobytes(), 144)
        assert compressed(L, got) == want, ("device handle", n)
        h.close()


def test_glv_split_boundaries(oracle, kzg):
    """The variable-base engine splits k = k1 + k2*x^2 (x the BLS parameter): scalars on the quotient /
    remainder boundaries, including the largest quotient below r, through the host and the device entry points."""
    L = oracle.lib()
    rnd = random.Random(13)
    X2 = 0xd201000000010000 ** 2
    qmax = (O.R - 1) // X2
    vals = []
    for q in (0, 1, 2, (1 << 127) - 1, 1 << 127, qmax - 1, qmax):
        for d in (0, 1, 2, X2 - 2, X2 - 1, X2 // 2 - 1, X2 // 2, X2 // 2 + 1, (1 << 127) - 1, 1 << 127, 1 << 64, (1 << 64) - 1):
            k = q * X2 + d
            if k < O.R:
                vals.append(k)
    vals += [X2 << s for s in range(0, 127, 9) if (X2 << s) < O.R]
    vals += [O.R - 1 - X2, O.R - X2, (O.R - 1) // 2, (O.R + 1) // 2, (O.R - 1) // 2 - 1, (O.R + 1) // 2 + 1]
    vals += [O.R - v for v in vals[:20] if v]
    n = len(vals)
    pts = gen_points(L, n, rnd)
    sc = O.fr_array(vals)
    check(L, kzg, pts, sc, n)
    # one scalar at a time on a single point: isolates each split
    for i in range(0, n, 5):
        one = (O.G1Affine * 1)(pts[i])
        check(L, kzg, one, O.fr_array([vals[i]]), 1)


def test_index_range_partials_combine_with_g1_sum(oracle, kzg):
    """The multi-GPU split of one MSM (sharding.msm_sharded), run here as three slices on one GPU: the
    partials added by the library's host helper equal the whole MSM."""
    L = oracle.lib()
    rnd = random.Random(14)
    n = 301
    pts = gen_points(L, n, rnd)
    sc = O.fr_array([rnd.randrange(O.R) for _ in range(n)])
    parts = []
    for lo, hi in ((0, 100), (100, 101), (101, n)):
        sub_p = C.cast(C.byref(pts, lo * 96), C.POINTER(O.G1Affine * (hi - lo))).contents
        sub_s = C.cast(C.byref(sc, lo * 32), C.POINTER(O.Fr * (hi - lo))).contents
        parts.append(bytes(kzg.multi_scalar_mult(sub_p, sub_s, hi - lo)))
    parts.append(bytes(144))  # a rank with an empty slice contributes the point at infinity
    total = O.G1()
    C.memmove(C.byref(total), kzg.g1_sum(parts), 144)
    exp = O.G1()
    L.omsm_affine(C.byref(exp), pts, sc, n)
    assert compressed(L, total) == compressed(L, exp)


def test_sum_of_multiples_of_generator(oracle, kzg):
    # sum (i+1)*G with scalars (i+1), n = 255 (bls12_381.rs:184-219)
    L = oracle.lib()
    n = 255
    g = O.G1()
    L.og1_generator(C.byref(g))
    pts = (O.G1Affine * n)()
    acc = O.G1()
    L.og1_generator(C.byref(acc))
    for i in range(n):
        L.og1_to_affine(C.byref(pts[i]), C.byref(acc))
        L.og1_add_or_dbl(C.byref(acc), C.byref(acc), C.byref(g))
    sc = O.fr_array([i + 1 for i in range(n)])
    h = kzg.prepare_multi_scalar_mult(pts, n)
    got = check(L, kzg, pts, sc, n, prepared=h)
    tot = O.fr_from_int(sum((i + 1) ** 2 for i in range(n)))
    e = O.G1()
    L.og1_mul(C.byref(e), C.byref(g), C.byref(tot))
    assert compressed(L, e) == got
    h.close()


def test_trusted_setup_4096_random_scalars_and_batch(oracle, oracle_settings, kzg):
    # BASELINE configs[1]: n=4096 random Fr x trusted-setup Lagrange points
    L = oracle.lib()
    rnd = random.Random(1)
    n = 4096
    pts = oracle_settings.g1_lagrange_brp
    h = kzg.prepare_multi_scalar_mult(pts, n)
    info = h.info()
    c = info["window_bits"]
    assert info["npoints"] == n and info["rows"] == ((127 + c) // c if info["wide_glv"] else 255 // c + 1)
    nb = 3
    vals = [rnd.randrange(O.R) for _ in range(nb * n)]
    sc = O.fr_array(vals)
    outs = kzg.multi_scalar_mult_prepared_batch(h, sc, n, nb)
    for b in range(nb):
        scb = (O.Fr * n).from_buffer(sc, b * n * 32)
        exp = O.G1()
        L.omsm_affine(C.byref(exp), pts, scb, n)
        assert compressed(L, as_oracle_g1(outs[b])) == compressed(L, exp)
    # unprepared path on the same inputs, and a shorter MSM on the prepared handle
    scb = (O.Fr * n).from_buffer(sc, 0)
    check(L, kzg, pts, scb, n, prepared=h)
    check(L, kzg, pts, scb, 1000, prepared=h, unprepared=False)
    h.close()


def test_linearity_property_large(oracle, kzg):
    # size-independent property at n = 2^16: MSM(s) + MSM(t) == MSM(s + t); points = 16 distinct
    # oracle-checked points tiled (every bucket sees repeated points -> doubling path exercised)
    L = oracle.lib()
    rnd = random.Random(5)
    n = 1 << 16
    base = gen_points(L, 16, rnd)
    pts = (O.G1Affine * n)()
    for i in range(n):
        pts[i] = base[i % 16]
    s = [rnd.randrange(O.R) for _ in range(n)]
    t = [rnd.randrange(O.R) for _ in range(n)]
    import numpy as np

    def fr_bulk(vals):
        # Montgomery limbs via python ints (R = 2^256)
        arr = (O.Fr * len(vals))()
        raw = b"".join(((v << 256) % O.R).to_bytes(32, "little") for v in vals)
        C.memmove(arr, raw, len(raw))
        return arr

    a = as_oracle_g1(kzg.multi_scalar_mult(pts, fr_bulk(s), n))
    b = as_oracle_g1(kzg.multi_scalar_mult(pts, fr_bulk(t), n))
    c = as_oracle_g1(kzg.multi_scalar_mult(pts, fr_bulk([(x + y) % O.R for x, y in zip(s, t)]), n))
    ab = O.G1()
    L.og1_add_or_dbl(C.byref(ab), C.byref(a), C.byref(b))
    assert compressed(L, ab) == compressed(L, c)
    # closed form: sum_i s_i * base[i%16] = sum_j (sum_{i%16==j} s_i) * base[j]
    folded = O.fr_array([sum(s[j::16]) % O.R for j in range(16)])
    e = O.G1()
    L.omsm_affine(C.byref(e), base, folded, 16)
    assert compressed(L, e) == compressed(L, a)


def _splitmix_scalar(seed, i):
    """h_i of kzgamd_generate_points: four splitmix64 outputs, top byte cleared (little-endian limbs)"""
    M = (1 << 64) - 1
    st = seed ^ ((0xD1B54A32D192ED03 * (i + 1)) & M)
    out = []
    for _ in range(4):
        st = (st + 0x9E3779B97F4A7C15) & M
        z = st
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        out.append(z ^ (z >> 31))
    out[3] &= 0x00FFFFFFFFFFFFFF
    return sum(v << (64 * k) for k, v in enumerate(out))


def test_empty_msm_and_generated_points(oracle, kzg):
    import torch

    L = oracle.lib()
    # npoints == 0: identity, output overwritten
    out = kzg.multi_scalar_mult((O.G1Affine * 1)(), (O.Fr * 1)(), 0)
    assert bytes(out) == bytes(144)
    # device point generator == h_i * G (oracle) on a subsample
    n = 4096
    d = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    kzg.generate_points(d.data_ptr(), n, 2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host = d.cpu().numpy().tobytes()
    g = O.G1()
    L.og1_generator(C.byref(g))
    for i in (0, 1, 77, 4095):
        t, a = O.G1(), O.G1Affine()
        kf = O.fr_from_int(_splitmix_scalar(2, i))
        L.og1_mul(C.byref(t), C.byref(g), C.byref(kf))
        L.og1_to_affine(C.byref(a), C.byref(t))
        assert bytes(a) == host[96 * i:96 * i + 96], i


def test_msm_2p20_matches_oracle(oracle, kzg):
    # BASELINE configs[2] size: n = 2^20 device-generated points, random scalars, GPU (variable-base engine,
    # device-resident inputs) vs the CPU oracle's Booth Pippenger on the same data
    import numpy as np
    import torch

    L = oracle.lib()
    n = 1 << 20
    stream = torch.cuda.current_stream().cuda_stream
    d_pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    kzg.generate_points(d_pts.data_ptr(), n, 7, stream)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(11)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=gen)
    sc[:, 31] &= 0x3F  # canonical little-endian scalars < 2^254
    sc[::10] = 0       # 10 % zero scalars
    d_sc = sc.cuda()
    pts = (O.G1Affine * n).from_buffer_copy(d_pts.cpu().numpy().tobytes())
    exp = O.G1()
    L.omsm_tiling_pippenger(C.byref(exp), pts, sc.numpy().tobytes(), n)
    # the accumulation in one piece (the default), and in two, three, four pieces whose reductions run beside the
    # accumulation of the next piece (tuning key tail_pieces): the same sum
    for pieces in (0, 2, 3, 4):
        d_out = torch.zeros(144, dtype=torch.uint8, device="cuda")
        h = kzg.DeviceMsm(d_pts.data_ptr(), n, False, kzg.make_config(tuning={"tail_pieces": pieces}))
        for _ in range(2):  # the second call reuses the streams and events of the first
            kzg.msm_prepared_batch_device(h, d_out.data_ptr(), d_sc.data_ptr(), n, 1, False, stream)
            torch.cuda.synchronize()
            got = O.G1()
            C.memmove(C.byref(got), d_out.cpu().numpy().tobytes(), 144)
            assert compressed(L, got) == compressed(L, exp), pieces
        h.close()
    # round 6's forms of the reduction: 16-row tiles (k_tile_sums_loop<16>) for a lone MSM, and a batch of two with the side
    # streams forced on at this size (sub_large: accumulations on the caller's stream, sort and reduction on side streams)
    d_sc2 = torch.cat([d_sc, d_sc])
    # ... and the tiles' upper tree levels with one lane per addition (tile_quad = 0; the default runs them four lanes each)
    for tuning, nb in (({"tile_rows": 16}, 1), ({"sub_large": 1, "sub_streams": 2, "tile_rows": 16}, 2), ({"sub_large": 1, "sub_prio": 0}, 2),
                       ({"tile_quad": 0}, 1), ({"tile_quad": 0, "tile_rows": 16}, 1)):
        d_out = torch.zeros(144 * nb, dtype=torch.uint8, device="cuda")
        h = kzg.DeviceMsm(d_pts.data_ptr(), n, False, kzg.make_config(tuning=tuning))
        kzg.msm_prepared_batch_device(h, d_out.data_ptr(), d_sc2.data_ptr(), n, nb, False, stream)
        torch.cuda.synchronize()
        raw = d_out.cpu().numpy().tobytes()
        for b in range(nb):
            got = O.G1()
            C.memmove(C.byref(got), raw[144 * b:144 * b + 144], 144)
            assert compressed(L, got) == compressed(L, exp), (tuning, b)
        h.close()


@pytest.mark.parametrize("logn", [16, 18, 19, 21, 22, 23, 24])
def test_msm_2p22_split_property(oracle, kzg, logn):
    # BASELINE configs[2] (n = 2^16 … 2^22: every size bench.py times, 2^20 has its own oracle test above) and beyond: a
    # size-independent property instead of a CPU recomputation:
    # MSM over all points == MSM(first half) + MSM(second half), plus one oracle-checked small prefix.
    # 2^23 is the largest size of the two-level sort (24-bit point indices for P_i and [x^2]P_i), 2^24 runs the
    # one-level sort again.
    import torch

    L = oracle.lib()
    n = 1 << logn
    stream = torch.cuda.current_stream().cuda_stream
    d_pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    kzg.generate_points(d_pts.data_ptr(), n, 22, stream)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(22)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=gen)
    sc[:, 31] &= 0x3F
    d_sc = sc.cuda()
    outs = []
    for lo, cnt in ((0, n), (0, n // 2), (n // 2, n // 2), (0, 512)):
        h = kzg.DeviceMsm(d_pts.data_ptr() + lo * 96, cnt, False)
        d_out = torch.zeros(144, dtype=torch.uint8, device="cuda")
        kzg.msm_prepared_batch_device(h, d_out.data_ptr(), d_sc.data_ptr() + lo * 32, cnt, 1, False, stream)
        torch.cuda.synchronize()
        g = O.G1()
        C.memmove(C.byref(g), d_out.cpu().numpy().tobytes(), 144)
        outs.append(g)
        h.close()
    s = O.G1()
    L.og1_add_or_dbl(C.byref(s), C.byref(outs[1]), C.byref(outs[2]))
    assert compressed(L, s) == compressed(L, outs[0])
    pts = (O.G1Affine * 512).from_buffer_copy(d_pts[: 512 * 96].cpu().numpy().tobytes())
    exp = O.G1()
    L.omsm_tiling_pippenger(C.byref(exp), pts, sc[:512].numpy().tobytes(), 512)
    assert compressed(L, exp) == compressed(L, outs[3])


def test_prepared_bucket_path_without_wide_table(oracle, kzg):
    """A prepared handle whose wide table does not fit (here: disabled) runs the fixed-base-rows bucket engine:
    one bucket set per MSM, single call and batch, skewed and edge scalars."""
    L = oracle.lib()
    rnd = random.Random(15)
    n = 700
    pts = gen_points(L, n, rnd)
    pts[13] = O.G1Affine()
    h = kzg.prepare_multi_scalar_mult(pts, n, kzg.make_config(no_tables=True))
    assert not h.info()["wide_table"] and h.info()["rows"] > 1
    vals = [rnd.randrange(O.R) for _ in range(n)]
    vals[0], vals[1], vals[2] = 0, O.R - 1, 1
    batches = [vals, [rnd.randrange(1 << 200) for _ in range(n)], [vals[5]] * n, [0] * n, [rnd.randrange(O.R) for _ in range(n)]]
    for v in batches[:3]:
        check(L, kzg, pts, O.fr_array(v), n, prepared=h, unprepared=False)
    flat = O.fr_array([x for v in batches for x in v])
    got = kzg.multi_scalar_mult_prepared_batch(h, flat, n, len(batches))
    for b, v in enumerate(batches):
        exp = O.G1()
        L.omsm_affine(C.byref(exp), pts, O.fr_array(v), n)
        assert compressed(L, as_oracle_g1(got[b])) == compressed(L, exp), b
    h.close()


@pytest.mark.parametrize("tuning", [{}, {"quad_accum_max": 0}, {"no_wide_tree": 1}, {"quad_accum_max": 0, "no_wide_tree": 1},
                                    {"quad_accum_max": 8, "wide_fold_max": 8}, {"spl1_max": 1, "wide_fold_max": 1, "quad_accum_max": 1}],
                         ids=lambda t: ";".join("%s=%d" % kv for kv in t.items()) or "default")
def test_few_commitments_every_accumulation_and_fold_form(oracle, oracle_settings, kzg, tuning):
    """One to eight MSMs per call over the 4096-point setup (the call shape of every c-kzg consumer: a handful of
    blobs): a lane per (scalar, half) with one lane or FOUR lanes per chain (k_fbw_accum_quad, tuning key quad_accum_max),
    folded by the one-launch tree of limb-parallel additions (k_wide_tree), by the two launches of k_wide_fold64
    (no_wide_tree) or by k_blocksum_hybrid, in every combination the keys select — against the oracle.  Scalars include
    zeros (skipped windows, chains that stay at infinity), r - 1, small values and a batch member that is all zero."""
    L = oracle.lib()
    pts = oracle_settings.g1_lagrange_brp
    n = 4096
    h = kzg.prepare_multi_scalar_mult(pts, n, kzg.make_config(table_budget_gb=8.0, tuning=tuning))  # c = 11: 6.4 GB
    info = h.info()
    assert info["wide_table"] and info["wide_glv"]
    rnd = random.Random(4096)
    sets = []
    for k in range(8):
        vals = [rnd.randrange(O.R) for _ in range(n)]
        if k == 1:
            vals = [0] * n
        if k == 2:
            vals[:2048] = [0] * 2048
            vals[4000:] = [O.R - 1] * 96
        if k == 3:
            vals = [rnd.randrange(1 << 16) for _ in range(n)]
        sets.append(vals)
    want = []
    for vals in sets:
        exp = O.G1()
        L.omsm_affine(C.byref(exp), pts, O.fr_array(vals), n)
        want.append(compressed(L, exp))
    for nbatch in (1, 2, 3, 8):
        flat = O.fr_array([x for vals in sets[:nbatch] for x in vals])
        for _ in range(2):  # the second call finds the counters where the first left them
            got = kzg.multi_scalar_mult_prepared_batch(h, flat, n, nbatch)
            assert [compressed(L, as_oracle_g1(got[b])) for b in range(nbatch)] == want[:nbatch], nbatch
    # single calls on every set (k = 1: all-zero scalars -> the point at infinity)
    for k in (1, 2, 3):
        assert compressed(L, as_oracle_g1(kzg.multi_scalar_mult_prepared(h, O.fr_array(sets[k]), n))) == want[k], k
    assert want[1] == b"\xc0" + bytes(47)
    h.close()


@pytest.mark.parametrize("outside", [False, True])
def test_large_prepared_handle_takes_the_variable_base_shape(oracle, kzg, outside):
    """A prepared handle too large for a wide table (here: the threshold lowered to 2^9 points, the table disabled) runs the
    GLV-split engine on the plain bases instead of table rows — unless a base fails the subgroup test, then it keeps
    the rows.  Same results either way, single call and batch."""
    L = oracle.lib()
    rnd = random.Random(16)
    n = 700
    pts = gen_points(L, n, rnd)
    pts[13] = O.G1Affine()
    if outside:
        x, y = curve_points_outside_g1(1, 5)[0]
        pts[7].x, pts[7].y = O.fp_from_int(x), O.fp_from_int(y)
    h = kzg.prepare_multi_scalar_mult(pts, n, kzg.make_config(no_tables=True, tuning={"fixed_as_variable_min": 9}))
    info = h.info()
    assert not info["wide_table"]
    assert (info["rows"] > 1) == outside, info
    vals = [rnd.randrange(O.R) for _ in range(n)]
    vals[0], vals[1], vals[2] = 0, O.R - 1, 1
    batches = [vals, [rnd.randrange(1 << 200) for _ in range(n)], [vals[5]] * n, [0] * n]
    for v in batches[:2]:
        check(L, kzg, pts, O.fr_array(v), n, prepared=h, unprepared=False)
    flat = O.fr_array([x for v in batches for x in v])
    got = kzg.multi_scalar_mult_prepared_batch(h, flat, n, len(batches))
    for b, v in enumerate(batches):
        exp = O.G1()
        L.omsm_affine(C.byref(exp), pts, O.fr_array(v), n)
        assert compressed(L, as_oracle_g1(got[b])) == compressed(L, exp), b
    h.close()


@pytest.mark.parametrize("budget_gb,glv", [(2.0, True), (0.6, True), (0.3, True), (0.1, True), (2.0, False), (0.3, False)])
def test_wide_table_every_window_shape(oracle, kzg, budget_gb, glv):
    """The wide table under shrinking budgets: each budget picks another (window, rows, GLV / plain) shape — among them
    window counts that are not a multiple of four (the selector rows are padded) — and every shape gives the oracle's
    result, single call and batch."""
    L = oracle.lib()
    rnd = random.Random(151)
    n = 96
    pts = gen_points(L, n, rnd)
    pts[5] = O.G1Affine()
    h = kzg.prepare_multi_scalar_mult(pts, n, kzg.make_config(table_budget_gb=budget_gb, tuning=None if glv else {"fbw_glv": 0}))
    info = h.info()
    assert info["wide_table"], info
    # n = 96: 2.0 GB -> GLV c=15 (9 windows), 0.6 -> GLV c=13 (10), 0.3 -> GLV c=12 (11), 0.1 -> GLV c=10 (13);
    # without the split 2.0 -> c=14 (19 windows), 0.3 -> c=10 (26)
    want = {(2.0, True): 9, (0.6, True): 10, (0.3, True): 11, (0.1, True): 13, (2.0, False): 19, (0.3, False): 26}
    assert info["rows"] == want[(budget_gb, glv)], info
    assert info["wide_glv"] == glv
    vals = [rnd.randrange(O.R) for _ in range(n)]
    vals[0], vals[1], vals[2] = 0, O.R - 1, 1
    batches = [vals, [rnd.randrange(1 << 130) for _ in range(n)], [O.R - 1 - rnd.randrange(1 << 20) for _ in range(n)],
               [0] * n, [rnd.randrange(O.R) for _ in range(n)]]
    check(L, kzg, pts, O.fr_array(vals), n, prepared=h, unprepared=False)
    flat = O.fr_array([x for v in batches for x in v])
    got = kzg.multi_scalar_mult_prepared_batch(h, flat, n, len(batches))
    for b, v in enumerate(batches):
        exp = O.G1()
        L.omsm_affine(C.byref(exp), pts, O.fr_array(v), n)
        assert compressed(L, as_oracle_g1(got[b])) == compressed(L, exp), (b, info)
    h.close()


@pytest.mark.parametrize("nbatch", [3, 70])
def test_variable_base_device_handle_batched(oracle, kzg, nbatch):
    """Several MSMs over one variable-base device handle in one launch: 3 (top-of-tree sums + limb-parallel
    Horner, one wave per MSM) and 70 (more than 64 bucket sets: plain tree + single-lane Horner)."""
    import torch

    L = oracle.lib()
    rnd = random.Random(16 + nbatch)
    n = 600
    pts = gen_points(L, n, rnd)
    stream = torch.cuda.current_stream().cuda_stream
    d_pts = torch.frombuffer(bytearray(bytes(pts)), dtype=torch.uint8).cuda()
    h = kzg.DeviceMsm(d_pts.data_ptr(), n, False)
    scal = [[rnd.randrange(O.R) for _ in range(n)] for _ in range(nbatch)]
    scal[1] = [scal[1][0]] * n          # equal scalars: one bucket per window takes everything
    scal[2][::3] = [0] * len(scal[2][::3])
    raw = b"".join(v.to_bytes(32, "little") for row in scal for v in row)  # canonical little-endian
    d_sc = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    d_out = torch.zeros(144 * nbatch, dtype=torch.uint8, device="cuda")
    kzg.msm_prepared_batch_device(h, d_out.data_ptr(), d_sc.data_ptr(), n, nbatch, False, stream)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy().tobytes()
    for b in range(nbatch):
        got = O.G1()
        C.memmove(C.byref(got), out[144 * b:144 * b + 144], 144)
        exp = O.G1()
        L.omsm_affine(C.byref(exp), pts, O.fr_array(scal[b]), n)
        assert compressed(L, got) == compressed(L, exp), b
    h.close()


@pytest.mark.parametrize("tuning", [{}, {"groups": 2}, {"one_level_sort": 1}, {"groups": 3, "one_level_sort": 1}, {"no_wide_tail": 1},
                                    {"tree_tail": 1}, {"tree_tail": 1, "groups": 2}, {"fine_bits": 9}, {"fine_bits": 10, "lgc": 3},
                                    {"lgc": 7}, {"flat_digits": 1}, {"flat_digits": 1, "no_wide_tail": 1}, {"direct_scatter": 1},
                                    {"scatter_atomics": 1}, {"no_wide_tail": 1, "fine_bits": 8}, {"lgc": 3}, {"lgc": 4}])
def test_variable_base_engine_variants(oracle, kzg, tuning):
    """Every selectable shape of the variable-base engine (two-level / one-level sort, window groups on their own
    streams, limb-parallel / single-lane tails, digit-decomposed / tree bucket reduction) on the same 40 000-point MSM
    with a skewed scalar distribution."""
    import torch

    L = oracle.lib()
    n = 40000
    stream = torch.cuda.current_stream().cuda_stream
    d_pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    kzg.generate_points(d_pts.data_ptr(), n, 21, stream)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(23)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=gen)
    sc[:, 31] &= 0x3F
    sc[::7] = sc[3]        # one scalar repeated 5 700 times: heavy buckets in every window
    sc[1::50, 8:] = 0      # short scalars
    d_sc = sc.cuda()
    d_out = torch.zeros(144, dtype=torch.uint8, device="cuda")
    h = kzg.DeviceMsm(d_pts.data_ptr(), n, False, kzg.make_config(tuning=tuning))
    kzg.msm_prepared_batch_device(h, d_out.data_ptr(), d_sc.data_ptr(), n, 1, False, stream)
    torch.cuda.synchronize()
    got = O.G1()
    C.memmove(C.byref(got), d_out.cpu().numpy().tobytes(), 144)
    pts = (O.G1Affine * n).from_buffer_copy(d_pts.cpu().numpy().tobytes())
    exp = O.G1()
    L.omsm_tiling_pippenger(C.byref(exp), pts, sc.numpy().tobytes(), n)
    assert compressed(L, got) == compressed(L, exp)
    h.close()


@pytest.mark.parametrize("sign", [1, -1])
def test_horner_exceptional_additions(oracle, kzg, sign):
    """Window sums that collide in the Horner recombination: Q = +-2^16 * P with scalars (2^16, 1) makes the last
    addition of the limb-parallel chain a doubling (+) or a cancellation to infinity (-); c = 16 is forced so that
    the digits fall on window boundaries whatever the size-dependent default."""
    import torch

    L = oracle.lib()
    rnd = random.Random(31)
    n = 40000  # large enough for the default 16-bit windows of the variable-base engine
    few = gen_points(L, 64, rnd)
    pts = (O.G1Affine * n)(*[few[i % 64] for i in range(n)])  # all but the first two carry a zero scalar
    p = O.G1()
    L.og1_from_affine(C.byref(p), C.byref(pts[0]))
    q = O.G1()
    k = O.fr_from_int((sign * (1 << 16)) % O.R)
    L.og1_mul(C.byref(q), C.byref(p), C.byref(k))
    L.og1_to_affine(C.byref(pts[1]), C.byref(q))
    vals = [0] * n
    vals[0], vals[1] = 1 << 16, 1
    sc = O.fr_array(vals)
    check(L, kzg, pts, sc, n)  # host entry point (host-side Horner)
    stream = torch.cuda.current_stream().cuda_stream
    d_pts = torch.frombuffer(bytearray(bytes(pts)), dtype=torch.uint8).cuda()
    raw = b"".join(v.to_bytes(32, "little") for v in vals)
    d_sc = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    d_out = torch.ones(144, dtype=torch.uint8, device="cuda")
    h = kzg.DeviceMsm(d_pts.data_ptr(), n, False)
    assert h.info()["window_bits"] == 16
    kzg.msm_prepared_batch_device(h, d_out.data_ptr(), d_sc.data_ptr(), n, 1, False, stream)
    torch.cuda.synchronize()
    h.close()
    got = O.G1()
    C.memmove(C.byref(got), d_out.cpu().numpy().tobytes(), 144)
    exp = O.G1()
    L.omsm_affine(C.byref(exp), pts, sc, n)
    assert compressed(L, got) == compressed(L, exp)
    if sign < 0:
        assert compressed(L, got) == b"\xc0" + bytes(47)


@pytest.mark.parametrize("tuning", [{}, {"tile_quad": 0}, {"tile_rows": 16}])
def test_tile_tree_exceptional_additions(oracle, kzg, tuning):
    """The tiled bucket reduction on ONE repeated point: every bucket sum is a small multiple m * P (m = 0 included: P and
    -P meet through the signed digits), so the row and column trees of k_tile_sums_loop add equal operands (the doubling
    branch), opposite ones (infinity) and infinities at every level — the four-lane levels (grp::dadd_body<4> /
    dbl_body<4>) and the single-lane ones alike.  Expected value without any MSM: (sum of the scalars) * P."""
    import torch

    L = oracle.lib()
    rnd = random.Random(77)
    n = 1 << 15  # 16-bit windows, 2^15 buckets per set: the tiled digit reduction
    one = gen_points(L, 1, rnd)
    pts = (O.G1Affine * n)(*[one[0]] * n)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(78)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=gen)
    sc[:, 31] &= 0x3F
    raw = sc.numpy().tobytes()
    total = sum(int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(n)) % O.R
    p = O.G1()
    L.og1_from_affine(C.byref(p), C.byref(pts[0]))
    exp = O.G1()
    k = O.fr_from_int(total)
    L.og1_mul(C.byref(exp), C.byref(p), C.byref(k))
    stream = torch.cuda.current_stream().cuda_stream
    d_pts = torch.frombuffer(bytearray(bytes(pts)), dtype=torch.uint8).cuda()
    d_sc = sc.cuda()
    d_out = torch.ones(144, dtype=torch.uint8, device="cuda")
    h = kzg.DeviceMsm(d_pts.data_ptr(), n, False, kzg.make_config(tuning=tuning))
    assert h.info()["window_bits"] == 16
    kzg.msm_prepared_batch_device(h, d_out.data_ptr(), d_sc.data_ptr(), n, 1, False, stream)
    torch.cuda.synchronize()
    h.close()
    got = O.G1()
    C.memmove(C.byref(got), d_out.cpu().numpy().tobytes(), 144)
    assert compressed(L, got) == compressed(L, exp), tuning


@pytest.mark.parametrize("nbatch,sub_streams", [(2, 0), (5, 0), (2, 3), (5, 3), (7, 2), (9, 1), (5, 6)])
def test_several_large_msms_in_one_call(oracle, kzg, nbatch, sub_streams):
    """Batches of MSMs over a 40 000-point variable-base handle.  With sub_streams = 0 (everything on the caller's stream,
    round 5's form): nbatch = 2 — 16 bucket sets of 32 768 buckets go through the tiled digit reduction and the
    limb-parallel cell sums together (set indexing of every stage with more than one MSM); nbatch = 5 — more coarse bins
    than the two-level sort holds in one launch: sub-batches of 2 + 2 + 1 (output and scalar offsets of every sub-batch).
    With side streams (the default, 3, for MSMs below 2^18 points): the batch is always cut (1 + 1; 2 + 2 + 1; 2 + 2 + 2 + 1
    over two side streams; 2 + 2 + 2 + 2 + 1 over one; 2 + 2 + 1 with six streams on offer), the accumulations chained on
    the caller's stream, sorts and reductions on the high-priority side streams with their own workspaces; two calls back to
    back reuse streams, events and workspaces."""
    import torch

    L = oracle.lib()
    n = 40000
    stream = torch.cuda.current_stream().cuda_stream
    d_pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    kzg.generate_points(d_pts.data_ptr(), n, 31, stream)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(32)
    sc = torch.randint(0, 256, (nbatch * n, 32), dtype=torch.uint8, generator=gen)
    sc[:, 31] &= 0x3F
    sc[n + 5:n + 3000, 4:] = 0  # the second MSM has a run of short scalars
    d_sc = sc.cuda()
    d_out = torch.zeros(144 * nbatch, dtype=torch.uint8, device="cuda")
    h = kzg.DeviceMsm(d_pts.data_ptr(), n, False, kzg.make_config(tuning={"sub_streams": sub_streams}))
    d_out2 = torch.zeros(144 * nbatch, dtype=torch.uint8, device="cuda")
    kzg.msm_prepared_batch_device(h, d_out.data_ptr(), d_sc.data_ptr(), n, nbatch, False, stream)
    kzg.msm_prepared_batch_device(h, d_out2.data_ptr(), d_sc.data_ptr(), n, nbatch, False, stream)  # no sync in between
    torch.cuda.synchronize()
    out, out2 = d_out.cpu().numpy().tobytes(), d_out2.cpu().numpy().tobytes()
    pts = (O.G1Affine * n).from_buffer_copy(d_pts.cpu().numpy().tobytes())
    for b in range(nbatch):
        exp = O.G1()
        L.omsm_tiling_pippenger(C.byref(exp), pts, sc[b * n:(b + 1) * n].numpy().tobytes(), n)
        for o in (out, out2):
            got = O.G1()
            C.memmove(C.byref(got), o[144 * b:144 * b + 144], 144)
            assert compressed(L, got) == compressed(L, exp), b
    h.close()


def test_batch_of_large_msms_sorts_ahead(oracle, kzg):
    """Five MSMs of 2^18 points in one call: sub-batches of 2 + 2 + 1 whose sorts run on two alternating side streams beside
    the previous sub-batch's reduction while accumulations and reductions stay on the caller's stream (tuning key
    sort_ahead = 1), two calls back to back without a synchronisation (the third sub-batch and
    the second call reuse streams, events and workspaces), against the same call in the default form (sub-batches one after the other on one stream) and
    against the oracle."""
    import torch

    L = oracle.lib()
    n, nbatch = 1 << 18, 5
    stream = torch.cuda.current_stream().cuda_stream
    d_pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    kzg.generate_points(d_pts.data_ptr(), n, 41, stream)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(42)
    sc = torch.randint(0, 256, (nbatch * n, 32), dtype=torch.uint8, generator=gen)
    sc[:, 31] &= 0x3F
    sc[3 * n + 7:3 * n + 5000, 6:] = 0  # the fourth MSM has a run of short scalars
    d_sc = sc.cuda()
    pts = (O.G1Affine * n).from_buffer_copy(d_pts.cpu().numpy().tobytes())
    want = []
    for b in range(nbatch):
        exp = O.G1()
        L.omsm_tiling_pippenger(C.byref(exp), pts, sc[b * n:(b + 1) * n].numpy().tobytes(), n)
        want.append(compressed(L, exp))
    for tuning in ({"sort_ahead": 1}, {}, {"sort_ahead": 1, "sub_streams": 2, "sub_prio": 0}):
        h = kzg.DeviceMsm(d_pts.data_ptr(), n, False, kzg.make_config(tuning=tuning))
        d_out = torch.zeros(144 * nbatch, dtype=torch.uint8, device="cuda")
        d_out2 = torch.ones(144 * nbatch, dtype=torch.uint8, device="cuda")
        kzg.msm_prepared_batch_device(h, d_out.data_ptr(), d_sc.data_ptr(), n, nbatch, False, stream)
        kzg.msm_prepared_batch_device(h, d_out2.data_ptr(), d_sc.data_ptr(), n, nbatch, False, stream)
        torch.cuda.synchronize()
        for o in (d_out.cpu().numpy().tobytes(), d_out2.cpu().numpy().tobytes()):
            for b in range(nbatch):
                got = O.G1()
                C.memmove(C.byref(got), o[144 * b:144 * b + 144], 144)
                assert compressed(L, got) == want[b], (tuning, b)
        h.close()

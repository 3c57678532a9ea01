"""Concurrent callers on ONE B1 / B2 handle (calls of the same kind and length are combined into batched launches below
the seam: msm.hip msm_run_host_combined, ntt.hip run_host_combined).  The reference shares its handles between rayon workers: SpparkPrecomputation
is Send + Sync and lives in an Arc (kzg/src/msm/sppark.rs:24-44), verify_blob_kzg_proof_batch calls the MSM from
par_chunks (kzg/src/eip_4844.rs:781-805), FFT settings are shared by reference.  Sixteen threads per handle, every
result against the oracle: mult_pippenger_prepared (whose concurrent calls are combined into one launch, msm.hip:
msm_run_host_combined), ntt_fr / das_fft_extension, fft_g1."""
import ctypes as C
import random
import threading

import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu
THREADS = 16


def compressed(L, p):
    buf = C.create_string_buffer(48)
    g = O.G1()
    C.memmove(C.byref(g), C.byref(p), 144)
    L.og1_compress(buf, C.byref(g))
    return buf.raw


def fr_bulk(vals):
    arr = (O.Fr * len(vals))()
    raw = b"".join(((v << 256) % O.R).to_bytes(32, "little") for v in vals)
    C.memmove(arr, raw, len(raw))
    return arr


def run_threads(work):
    """work(t) on THREADS threads released together; returns the list of failures the threads recorded"""
    failures = []
    gate = threading.Barrier(THREADS)

    def body(t):
        try:
            gate.wait()
            work(t)
        except Exception as e:  # noqa: BLE001
            failures.append((t, repr(e)))

    ts = [threading.Thread(target=body, args=(t,)) for t in range(THREADS)]
    for th in ts:
        th.start()
    for th in ts:
        th.join()
    return failures


def test_sixteen_threads_share_one_prepared_msm_handle(kzg, oracle, oracle_settings):
    L = oracle.lib()
    n = 4096
    pts = oracle_settings.g1_lagrange_brp
    h = kzg.prepare_multi_scalar_mult(pts, n)
    lengths = [4096, 4096, 4096, 1000, 4096, 8, 4096, 77]  # equal lengths are combined, the others run between them
    calls = len(lengths)
    inputs = {}
    for t in range(THREADS):
        rnd = random.Random(9000 + t)
        for k in range(calls):
            m = lengths[(k + t) % calls]
            vals = [rnd.randrange(O.R) for _ in range(m)]
            if k == 3:
                vals[0] = 0
                vals[-1] = O.R - 1
            inputs[(t, k)] = (m, fr_bulk(vals))
    got = {}

    def work(t):
        for k in range(calls):
            m, sc = inputs[(t, k)]
            got[(t, k)] = compressed(L, kzg.multi_scalar_mult_prepared(h, sc, m))

    assert run_threads(work) == []
    for (t, k), (m, sc) in inputs.items():
        exp = O.G1()
        L.omsm_affine(C.byref(exp), pts, sc, m)
        assert got[(t, k)] == compressed(L, exp), (t, k, m)
    # a failing call among good ones fails alone: more scalars than the handle holds
    bad = fr_bulk([1] * (n + 1))

    def work2(t):
        if t == 5:
            with pytest.raises(kzg.KzgAmdError):
                kzg.multi_scalar_mult_prepared(h, bad, n + 1)
        else:
            m, sc = inputs[(t, 0)]
            assert compressed(L, kzg.multi_scalar_mult_prepared(h, sc, m)) == got[(t, 0)]

    assert run_threads(work2) == []
    h.close()


def test_sixteen_threads_share_one_ntt_handle(kzg, oracle):
    L = oracle.lib()
    scale = 13
    fs = kzg.FFTSettings(scale)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), scale) == 0
    sizes = [4096, 8192, 64, 4096, 1, 1024]
    inputs = {}
    for t in range(THREADS):
        rnd = random.Random(7000 + t)
        for k, n in enumerate(sizes):
            inputs[(t, k)] = (n, (t + k) % 2 == 1, fr_bulk([rnd.randrange(O.R) for _ in range(n)]))
    got = {}

    def work(t):
        for k in range(len(sizes)):
            n, inv, data = inputs[(t, k)]
            got[(t, k)] = bytes(fs.fft_fr(data, n, inverse=inv))[: 32 * n]
            if 2 <= n <= 4096:
                got[(t, k, "das")] = bytes(fs.das_fft_extension(data, n))[: 32 * n]

    assert run_threads(work) == []
    for (t, k), (n, inv, data) in inputs.items():
        exp = (O.Fr * n)()
        assert L.offt_fr(C.byref(ofs), exp, data, n, 1 if inv else 0) == 0
        assert got[(t, k)] == bytes(exp), (t, k, n, inv)
        if 2 <= n <= 4096:
            odds = (O.Fr * n)()
            assert L.odas_fft_extension(C.byref(ofs), odds, data, n) == 0
            assert got[(t, k, "das")] == bytes(odds), (t, k, n)
    # the same calls with the combining off (tuning key combine=0: every call on the handle's mutex): the same bytes
    fs2 = kzg.FFTSettings(scale, kzg.make_config(tuning={"combine": 0}))
    got2 = {}

    def work_plain(t):
        for k in range(len(sizes)):
            n, inv, data = inputs[(t, k)]
            got2[(t, k)] = bytes(fs2.fft_fr(data, n, inverse=inv))[: 32 * n]

    assert run_threads(work_plain) == []
    assert all(got2[key] == got[key] for key in got2)
    fs2.close()
    # errors keep their codes through the combined path, and a call longer than the combining limit takes the plain one
    with pytest.raises(kzg.KzgAmdError, match="power-of-two"):
        fs.fft_fr(inputs[(0, 0)][2], 12)
    big = kzg.FFTSettings(15)
    n = 1 << 15
    rnd = random.Random(5)
    data = fr_bulk([rnd.randrange(O.R) for _ in range(n)])
    ofs15 = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs15), 15) == 0
    exp = (O.Fr * n)()
    assert L.offt_fr(C.byref(ofs15), exp, data, n, 0) == 0
    assert bytes(big.fft_fr(data, n))[: 32 * n] == bytes(exp)
    big.close()
    L.offt_settings_free(C.byref(ofs15))
    fs.close()
    L.offt_settings_free(C.byref(ofs))


def test_sixteen_threads_share_one_fft_g1_handle(kzg, oracle):
    L = oracle.lib()
    scale = 8
    fs = kzg.FFTSettings(scale)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), scale) == 0
    g = O.G1()
    L.og1_generator(C.byref(g))
    rnd = random.Random(81)
    pool = []
    for _ in range(40):
        p = O.G1()
        k = O.fr_from_int(rnd.randrange(1, O.R))
        L.og1_mul(C.byref(p), C.byref(g), C.byref(k))
        pool.append(p)
    sizes = [64, 16, 256, 64]
    inputs = {}
    for t in range(THREADS):
        for k, n in enumerate(sizes):
            data = (O.G1 * n)()
            for i in range(n):
                data[i] = O.G1() if (i + t) % 11 == 0 else pool[rnd.randrange(len(pool))]
            inputs[(t, k)] = (n, (t + k) % 2 == 0, data)
    got = {}

    def work(t):
        for k in range(len(sizes)):
            n, inv, data = inputs[(t, k)]
            out = fs.fft_g1(data, n, inverse=inv)
            got[(t, k)] = [compressed(L, out[i]) for i in range(n)]

    assert run_threads(work) == []
    for (t, k), (n, inv, data) in inputs.items():
        exp = (O.G1 * n)()
        assert L.offt_g1(C.byref(ofs), exp, data, n, 1 if inv else 0) == 0
        assert got[(t, k)] == [compressed(L, exp[i]) for i in range(n)], (t, k, n, inv)
    fs.close()
    L.offt_settings_free(C.byref(ofs))


def test_calls_at_and_beyond_the_combining_limit_share_one_handle(kzg, oracle):
    """Calls of up to 2^16 scalars are combined (a page-locked slot holds exactly that many), longer ones take the
    handle's own stream under its mutex (msm.hip: COMBINE_NMAX): both kinds at once on one handle, each against the
    oracle's Pippenger on the same points."""
    import torch

    L = oracle.lib()
    lim = 1 << 16
    n = lim + 64
    d_pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    kzg.generate_points(d_pts.data_ptr(), n, 21, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    pts = (O.G1Affine * n).from_buffer_copy(d_pts.cpu().numpy().tobytes())
    h = kzg.prepare_multi_scalar_mult(pts, n, kzg.make_config(table_budget_gb=8))
    rnd = random.Random(33)
    want, scalars = {}, {}
    for m in (lim, n, lim - 1):
        vals = [rnd.randrange(O.R) for _ in range(m)]
        vals[0], vals[-1] = 0, O.R - 1
        exp = O.G1()
        L.omsm_tiling_pippenger(C.byref(exp), pts, b"".join(v.to_bytes(32, "little") for v in vals), m)
        want[m] = compressed(L, exp)
        scalars[m] = fr_bulk(vals)  # Montgomery form, what mult_pippenger_prepared takes
    order = [lim, n, lim - 1, n]
    got = {}

    def work(t):
        for k in range(3):
            m = order[(t + k) % len(order)]
            got[(t, k)] = (m, compressed(L, kzg.multi_scalar_mult_prepared(h, scalars[m], m)))

    assert run_threads(work) == []
    assert len(got) == 3 * THREADS
    for key, (m, c) in got.items():
        assert c == want[m], (key, m)
    h.close()

"""NTT parity on the GPU (B2) vs the CPU oracle; mirrors kzg-bench/src/tests/fft_fr.rs and das.rs."""
import ctypes as C
import hashlib
import random

import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def fr_bulk(vals):
    arr = (O.Fr * len(vals))()
    raw = b"".join(((v << 256) % O.R).to_bytes(32, "little") for v in vals)
    C.memmove(arr, raw, len(raw))
    return arr


def limbs(fr):
    return [int(x) for x in fr.l]


def test_inverse_fft_kat_and_das_kat(kzg, oracle, kats):
    L = oracle.lib()
    fs = kzg.FFTSettings(4)
    data = fr_bulk(list(range(16)))
    out = fs.fft_fr(data, 16, inverse=True)
    for i, row in enumerate(kats["inverse_fft"]["expected"]):
        got = (C.c_uint64 * 4)()
        f = O.Fr()
        C.memmove(C.byref(f), C.byref(out[i]), 32)
        L.ofr_to_u64_arr(got, C.byref(f))
        assert list(got) == row, i
    odds = fs.das_fft_extension(fr_bulk(list(range(8))), 8)
    for i, row in enumerate(kats["das_extension_known"]["expected"]):
        got = (C.c_uint64 * 4)()
        f = O.Fr()
        C.memmove(C.byref(f), C.byref(odds[i]), 32)
        L.ofr_to_u64_arr(got, C.byref(f))
        assert list(got) == row, i
    fs.close()


@pytest.mark.parametrize("logn", list(range(0, 17)))
def test_fft_matches_oracle(kzg, oracle, logn):
    L = oracle.lib()
    scale = max(logn, 1) + 1
    fs = kzg.FFTSettings(scale)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), scale) == 0
    n = 1 << logn
    rnd = random.Random(100 + logn)
    data = fr_bulk([rnd.randrange(O.R) for _ in range(n)])
    for inv in (False, True):
        exp = (O.Fr * n)()
        assert L.offt_fr(C.byref(ofs), exp, data, n, 1 if inv else 0) == 0
        got = fs.fft_fr(data, n, inverse=inv)
        assert bytes(got)[: 32 * n] == bytes(exp), (logn, inv)
    if n >= 1:
        exp = (O.Fr * n)()
        rc = L.odas_fft_extension(C.byref(ofs), exp, data, n)
        assert rc == 0
        got = fs.das_fft_extension(data, n)
        assert bytes(got)[: 32 * n] == bytes(exp)
    L.offt_settings_free(C.byref(ofs))
    fs.close()


def test_every_length_on_one_long_lived_handle(kzg, oracle):
    """Stride invariance (kzg-bench/src/tests/fft_fr.rs:87-106: `stride_fft`, a 2^9-point transform on settings of scale
    9 and of scale 12 gives the same values): ONE handle of scale 15 serves every power-of-two length up to its
    max_width — forward, inverse and the DAS extension (whose lists are at most half the width) — each against the
    oracle on settings of the SAME scale (the reference's stride = max_width / n walk over one table of roots) and against
    the oracle on settings made for exactly that length."""
    L = oracle.lib()
    SCALE = 15
    fs = kzg.FFTSettings(SCALE)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), SCALE) == 0
    rnd = random.Random(1509)
    for logn in range(0, SCALE + 1):
        n = 1 << logn
        data = fr_bulk([rnd.randrange(O.R) for _ in range(n)])
        ofs_n = O.FFTSettings()
        assert L.offt_settings_new(C.byref(ofs_n), max(logn, 1)) == 0
        for inv in (False, True):
            exp = (O.Fr * n)()
            assert L.offt_fr(C.byref(ofs), exp, data, n, 1 if inv else 0) == 0
            exp_n = (O.Fr * n)()
            assert L.offt_fr(C.byref(ofs_n), exp_n, data, n, 1 if inv else 0) == 0
            assert bytes(exp) == bytes(exp_n), (logn, inv)  # the oracle itself is stride-invariant
            got = fs.fft_fr(data, n, inverse=inv)
            assert bytes(got)[: 32 * n] == bytes(exp), (logn, inv)
        if logn < SCALE:  # das_fft_extension of n evens needs 2n <= max_width
            exp = (O.Fr * n)()
            assert L.odas_fft_extension(C.byref(ofs), exp, data, n) == 0
            got = fs.das_fft_extension(data, n)
            assert bytes(got)[: 32 * n] == bytes(exp), logn
        L.offt_settings_free(C.byref(ofs_n))
    # and the reference's own case: 2^9 points of 0, 1, 2, ... on scale 9 and scale 12 handles
    data = fr_bulk(list(range(512)))
    a, b = kzg.FFTSettings(9), kzg.FFTSettings(12)
    assert bytes(a.fft_fr(data, 512)) == bytes(b.fft_fr(data, 512)) == bytes(fs.fft_fr(data, 512))
    a.close()
    b.close()
    L.offt_settings_free(C.byref(ofs))
    fs.close()


@pytest.mark.parametrize("logn,nbatch", [(0, 5), (1, 3), (3, 1000), (7, 3), (7, 64), (8, 17), (10, 5), (11, 3), (12, 3), (13, 3), (14, 2)])
def test_device_batches_match_oracle(kzg, oracle, logn, nbatch):
    # kzgamd_ntt_fr_device: nbatch contiguous transforms in one call, partial tiles included (4096 does not divide
    # n * nbatch), every transform against the oracle
    import torch

    L = oracle.lib()
    n = 1 << logn
    fs = kzg.FFTSettings(max(logn, 1))
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), max(logn, 1)) == 0
    rnd = random.Random(1000 * logn + nbatch)
    vals = [rnd.randrange(O.R) for _ in range(n * nbatch)]
    raw = b"".join(v.to_bytes(32, "little") for v in vals)
    d_in = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    d_out = torch.zeros_like(d_in)
    stream = torch.cuda.current_stream().cuda_stream
    for inv in (False, True):
        fs.fft_fr_device(d_out.data_ptr(), d_in.data_ptr(), n, nbatch, inv, stream)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().tobytes()
        for b in range(nbatch):
            one = (O.Fr * n).from_buffer_copy(raw[32 * n * b: 32 * n * (b + 1)])
            exp = (O.Fr * n)()
            assert L.offt_fr(C.byref(ofs), exp, one, n, 1 if inv else 0) == 0
            assert got[32 * n * b: 32 * n * (b + 1)] == bytes(exp), (logn, nbatch, inv, b)
    L.offt_settings_free(C.byref(ofs))
    fs.close()


@pytest.mark.parametrize("logh,nbatch", [(0, 3), (3, 9), (6, 64), (10, 3), (11, 5), (12, 2), (14, 1)])
def test_das_extension_device_batches(kzg, oracle, logh, nbatch):
    # kzgamd_das_fft_extension_device: batches of half-size lists, each against the oracle's restatement of the
    # reference's fused network (data_availability_sampling.rs:14-100)
    import torch

    L = oracle.lib()
    n = 1 << logh
    fs = kzg.FFTSettings(logh + 1)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), logh + 1) == 0
    rnd = random.Random(77 * logh + nbatch)
    raw = b"".join(rnd.randrange(O.R).to_bytes(32, "little") for _ in range(n * nbatch))
    d_in = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    d_out, d_tmp = torch.zeros_like(d_in), torch.zeros_like(d_in)
    fs.das_fft_extension_device(d_out.data_ptr(), d_in.data_ptr(), d_tmp.data_ptr(), n, nbatch, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().tobytes()
    for b in range(nbatch):
        one = (O.Fr * n).from_buffer_copy(raw[32 * n * b: 32 * n * (b + 1)])
        exp = (O.Fr * n)()
        assert L.odas_fft_extension(C.byref(ofs), exp, one, n) == 0
        assert got[32 * n * b: 32 * n * (b + 1)] == bytes(exp), (logh, nbatch, b)
    with pytest.raises(kzg.KzgAmdError):
        fs.das_fft_extension_device(d_in.data_ptr(), d_in.data_ptr(), d_tmp.data_ptr(), n, nbatch, 0)
    L.offt_settings_free(C.byref(ofs))
    fs.close()


def test_das_extension_half_of_2p20_digest(kzg, oracle):
    # BASELINE configs[3]: the DAS extension of 2^19 evens (the odd half of a 2^20 domain) against the oracle, by digest
    L = oracle.lib()
    logh = 19
    n = 1 << logh
    fs = kzg.FFTSettings(logh + 1)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), logh + 1) == 0
    data = fr_bulk(list(range(n)))
    got = fs.das_fft_extension(data, n)
    exp = (O.Fr * n)()
    assert L.odas_fft_extension(C.byref(ofs), exp, data, n) == 0
    assert hashlib.sha256(bytes(got)[: 32 * n]).digest() == hashlib.sha256(bytes(exp)).digest()
    L.offt_settings_free(C.byref(ofs))
    fs.close()


def test_roots_and_errors(kzg, oracle):
    L = oracle.lib()
    fs = kzg.FFTSettings(8)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), 8) == 0
    r, rr, br = fs.roots()
    assert bytes(r) == bytes((O.Fr * 257).from_address(C.addressof(ofs.roots_of_unity.contents)))
    assert bytes(rr) == bytes((O.Fr * 257).from_address(C.addressof(ofs.reverse_roots_of_unity.contents)))
    assert bytes(br) == bytes((O.Fr * 256).from_address(C.addressof(ofs.brp_roots_of_unity.contents)))
    buf = (O.Fr * 512)()
    with pytest.raises(kzg.KzgAmdError, match="longer than the available max width"):
        fs.fft_fr(buf, 512)
    with pytest.raises(kzg.KzgAmdError, match="power-of-two"):
        fs.fft_fr(buf, 24)
    with pytest.raises(kzg.KzgAmdError, match="non-zero list"):
        fs.das_fft_extension(buf, 0)
    with pytest.raises(kzg.KzgAmdError, match="power-of-two"):
        fs.das_fft_extension(buf, 24)
    with pytest.raises(kzg.KzgAmdError, match="longer than the available max width"):
        fs.das_fft_extension(buf, 256)
    with pytest.raises(kzg.KzgAmdError):
        kzg.FFTSettings(32)
    L.offt_settings_free(C.byref(ofs))
    fs.close()


def test_large_2p20_roundtrip_and_digest(kzg, oracle):
    # BASELINE configs[3]: n = 2^20.  fft then ifft == identity (size-independent property) and the
    # forward transform equals the oracle's (compared by digest; the oracle takes ~1 s at this size)
    L = oracle.lib()
    logn = 20
    n = 1 << logn
    fs = kzg.FFTSettings(logn)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), logn) == 0
    data = fr_bulk(list(range(n)))  # data[i] = Fr(i), as in kzg-bench/src/tests/fft_fr.rs:35-37
    fwd = fs.fft_fr(data, n)
    back = fs.fft_fr(fwd, n, inverse=True)
    assert bytes(back) == bytes(data)
    exp = (O.Fr * n)()
    assert L.offt_fr(C.byref(ofs), exp, data, n, 0) == 0
    assert hashlib.sha256(bytes(fwd)).digest() == hashlib.sha256(bytes(exp)).digest()
    L.offt_settings_free(C.byref(ofs))
    fs.close()


def test_three_pass_2p25_matches_oracle(kzg, oracle):
    # n = 2^25 needs the third pass (stages 24..); raw random residues (< 2^254) as input
    if kzg.LIB_PATH.endswith("_exact.so"):
        pytest.skip("the Fr transforms contain no exact-zero test: both builds are the same kernels; the oracle's 2^25 "
                    "transform (25 s on the host) runs once")
    import numpy as np

    L = oracle.lib()
    logn = 25
    n = 1 << logn
    fs = kzg.FFTSettings(logn)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), logn) == 0
    rng = np.random.default_rng(25)
    raw = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    raw[:, 31] &= 0x3F
    data = (O.Fr * n).from_buffer(raw)
    fwd = fs.fft_fr(data, n)
    exp = (O.Fr * n)()
    assert L.offt_fr(C.byref(ofs), exp, data, n, 0) == 0
    assert hashlib.sha256(bytes(fwd)).digest() == hashlib.sha256(bytes(exp)).digest()
    back = fs.fft_fr(fwd, n, inverse=True)
    assert bytes(back) == raw.tobytes()
    L.offt_settings_free(C.byref(ofs))
    fs.close()

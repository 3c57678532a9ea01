"""fft_g1 on the GPU (B2 widening row) vs the CPU oracle; mirrors kzg-bench/src/tests/fft_g1.rs
(compare_ft_fft, roundtrip_fft, stride_fft) plus the exceptional cases of the group law."""
import ctypes as C
import random

import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu

# The four forms of a stage (fftg1.hip): a wave / four lanes / two lanes / one lane per half-butterfly.  They are chosen
# by grid size; the tuning keys (read when the handle is created, KzgAmdConfig.tuning) force one of them for every size
# so that each is checked against the oracle on the same cases.
FORMS = {"by_size": {}, "wave": {"g1_wide_max": 1000000},
         "four_lanes": {"g1_wide_max": 0, "g1_quad_max": 1000000},
         "two_lanes": {"g1_wide_max": 0, "g1_quad_max": 0, "g1_pair_max": 1000000},
         "one_lane": {"g1_wide_max": 0, "g1_quad_max": 0, "g1_pair_max": 0}}


@pytest.fixture(params=list(FORMS))
def form(request, kzg):
    """the KzgAmdConfig that forces the form (None for the by-size default)"""
    t = FORMS[request.param]
    return kzg.make_config(tuning=t) if t else None


def make_data(L, n):
    """kzg-bench/src/tests/fft_g1.rs make_data: G, 2G, 3G, ... (Jacobian, Z != 1 after the first)"""
    g, acc = O.G1(), O.G1()
    L.og1_generator(C.byref(g))
    L.og1_generator(C.byref(acc))
    data = (O.G1 * n)()
    for i in range(n):
        data[i] = acc
        L.og1_add_or_dbl(C.byref(acc), C.byref(acc), C.byref(g))
    return data


def random_points(L, n, seed):
    rnd = random.Random(seed)
    g = O.G1()
    L.og1_generator(C.byref(g))
    data = (O.G1 * n)()
    for i in range(n):
        k = O.fr_from_int(rnd.randrange(O.R))
        L.og1_mul(C.byref(data[i]), C.byref(g), C.byref(k))
    return data


def compressed(L, arr, n):
    out = []
    for i in range(n):
        buf = C.create_string_buffer(48)
        p = O.G1()
        C.memmove(C.byref(p), C.byref(arr[i]), 144)
        L.og1_compress(buf, C.byref(p))
        out.append(buf.raw)
    return out


@pytest.mark.parametrize("logn", [0, 1, 3, 6])
def test_fft_g1_matches_oracle(kzg, oracle, logn, form):
    L = oracle.lib()
    scale = max(logn, 1)
    n = 1 << logn
    fs = kzg.FFTSettings(scale, form)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), scale) == 0
    data = make_data(L, n)
    if n >= 8:
        data[3] = O.G1()  # infinity (Z == 0)
        data[5] = data[4]  # equal neighbours
    for inv in (False, True):
        exp = (O.G1 * n)()
        assert L.offt_g1(C.byref(ofs), exp, data, n, 1 if inv else 0) == 0
        got = fs.fft_g1(data, n, inverse=inv)
        assert compressed(L, got, n) == compressed(L, exp, n), (logn, inv)
    fs.close()
    L.offt_settings_free(C.byref(ofs))


def test_fft_g1_exceptional_inputs(kzg, oracle, form):
    """all points equal (every first-stage butterfly doubles / cancels), all infinity, P and -P pairs"""
    L = oracle.lib()
    n = 16
    fs = kzg.FFTSettings(4, form)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), 4) == 0
    g = O.G1()
    L.og1_generator(C.byref(g))
    neg = O.G1()
    minus_one = O.fr_from_int(O.R - 1)
    L.og1_mul(C.byref(neg), C.byref(g), C.byref(minus_one))
    cases = []
    same = (O.G1 * n)()
    for i in range(n):
        same[i] = g
    cases.append(same)
    cases.append((O.G1 * n)())
    alt = (O.G1 * n)()
    for i in range(n):
        alt[i] = g if i % 2 == 0 else neg
    cases.append(alt)
    for data in cases:
        for inv in (False, True):
            exp = (O.G1 * n)()
            assert L.offt_g1(C.byref(ofs), exp, data, n, 1 if inv else 0) == 0
            got = fs.fft_g1(data, n, inverse=inv)
            assert compressed(L, got, n) == compressed(L, exp, n)
    fs.close()
    L.offt_settings_free(C.byref(ofs))


def test_fft_g1_roundtrip_scale_10(kzg, oracle):
    # roundtrip_fft: forward then inverse returns the data (scale 10)
    L = oracle.lib()
    n = 1 << 10
    fs = kzg.FFTSettings(10)
    data = make_data(L, n)
    coeffs = fs.fft_g1(data, n)
    back = fs.fft_g1(coeffs, n, inverse=True)
    assert compressed(L, back, n) == compressed(L, data, n)
    # spot-check the forward transform against the definition at two output positions: sum_j w^(ij) * P_j
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), 10) == 0
    for i in (1, 777):
        acc = O.G1()
        for j in range(n):
            v = O.G1()
            L.og1_mul(C.byref(v), C.byref(data[j]), C.byref(ofs.roots_of_unity[(i * j) % n]))
            L.og1_add_or_dbl(C.byref(acc), C.byref(acc), C.byref(v))
        assert compressed(L, coeffs, n)[i] == compressed(L, (O.G1 * 1)(acc), 1)[0]
    fs.close()
    L.offt_settings_free(C.byref(ofs))


def test_fft_g1_stride_and_batch(kzg, oracle, form):
    # stride_fft: the same data through settings of scale 9 and 12 gives the same result
    L = oracle.lib()
    n = 1 << 9
    data = random_points(L, n, 42)
    fs1, fs2 = kzg.FFTSettings(9, form), kzg.FFTSettings(12, form)
    a = fs1.fft_g1(data, n)
    b = fs2.fft_g1(data, n)
    assert compressed(L, a, n) == compressed(L, b, n)
    # batch of 4 x 128 equals four single transforms
    m = 128
    batch = fs2.fft_g1(data, m, nbatch=4)
    cb = compressed(L, batch, n)
    for k in range(4):
        part = (O.G1 * m)()
        C.memmove(part, C.byref(data, k * m * 144), m * 144)
        single = fs2.fft_g1(part, m)
        assert compressed(L, single, m) == cb[k * m:(k + 1) * m]
    fs1.close()
    fs2.close()


def test_fft_g1_errors(kzg):
    fs = kzg.FFTSettings(4)
    buf = (kzg.BlstP1 * 32)()
    with pytest.raises(kzg.KzgAmdError, match="longer than the available max width"):
        fs.fft_g1(buf, 32)
    with pytest.raises(kzg.KzgAmdError, match="power-of-two"):
        fs.fft_g1(buf, 12)
    fs.close()

"""The configuration of a handle (KzgAmdConfig: device, table budget, tuning keys — include/kzg_mi355x.h) and the matrix
handle behind G1LinComb::g1_lincomb_batch (kzg/src/lib.rs:156-181 -> BgmwTable::multiply_batch, kzg/src/msm/bgmw.rs:306-380;
its one user is FK20's 128 MSMs of 64 points over x_ext_fft_columns, kzg/src/das.rs:682-686)."""
import ctypes as C
import os
import random
import time

import pytest

import oracle_ffi as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
SETUP = os.path.join(GOLDEN, "trusted_setup.txt")


def compressed(L, p):
    buf = C.create_string_buffer(48)
    g = O.G1()
    C.memmove(C.byref(g), C.byref(p), 144)
    L.og1_compress(buf, C.byref(g))
    return buf.raw


def test_matrix_handle_fk20_shape_against_the_oracle_and_separate_calls(kzg, oracle):
    """128 rows x 64 columns = the settings' x_ext_fft_columns: one kzgamd_mult_pippenger_matrix call gives the 128 sums
    the oracle gives row by row, and what 128 mult_pippenger calls on the rows give (the form the trait's default
    g1_lincomb_batch takes without a table, kzg/src/lib.rs:170-178) — timed side by side."""
    L = oracle.lib()
    s = kzg.KZGSettings.from_file(SETUP, kzg.make_config(table_budget_gb=8))
    rows, cols = 128, 64
    aff = (O.G1Affine * (rows * cols))()
    colptrs = (C.c_void_p * rows).from_address(s.c.x_ext_fft_columns)
    for r in range(rows):
        col = (kzg.BlstP1 * cols).from_address(colptrs[r])
        for c in range(cols):
            p = O.G1()
            C.memmove(C.byref(p), C.byref(col[c]), 144)
            L.og1_to_affine(C.byref(aff[r * cols + c]), C.byref(p))
    rnd = random.Random(128064)
    nmat = 3
    vals = [rnd.randrange(O.R) for _ in range(nmat * rows * cols)]
    vals[0], vals[1], vals[2] = 0, O.R - 1, 1
    vals[7 * cols:8 * cols] = [0] * cols            # row 7 of matrix 0: the point at infinity
    vals[9 * cols:10 * cols] = [vals[9 * cols]] * cols
    sc = O.fr_array(vals)
    h = kzg.MatrixMsm(aff, rows, cols, kzg.make_config(table_budget_gb=16))
    assert h.info()["wide_table"] and h.info()["npoints"] == rows * cols
    out = h.multiply_batch(sc, nmat)
    want = []
    for k in range(nmat * rows):
        exp = O.G1()
        base = (O.G1Affine * cols).from_buffer(aff, (k % rows) * cols * 96)
        sub = (O.Fr * cols).from_buffer(sc, k * cols * 32)
        L.omsm_affine(C.byref(exp), base, sub, cols)
        want.append(compressed(L, exp))
    assert [compressed(L, out[k]) for k in range(nmat * rows)] == want
    assert want[7] == b"\xc0" + bytes(47)
    # one scalar matrix (what g1_lincomb_batch passes): matrix call against 128 separate unprepared calls
    one = h.multiply_batch(sc, 1)
    assert [compressed(L, one[k]) for k in range(rows)] == want[:rows]
    t0 = time.perf_counter()
    for _ in range(5):
        h.multiply_batch(sc, 1)
    t_matrix = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    sep = []
    for r in range(rows):
        base = (O.G1Affine * cols).from_buffer(aff, r * cols * 96)
        sub = (O.Fr * cols).from_buffer(sc, r * cols * 32)
        sep.append(compressed(L, kzg.multi_scalar_mult(base, sub, cols)))
    t_sep = time.perf_counter() - t0
    assert sep == want[:rows]
    print("matrix 128 x 64: %.3f ms per call; 128 separate mult_pippenger calls: %.1f ms" % (t_matrix * 1e3, t_sep * 1e3))
    assert t_matrix < t_sep
    with pytest.raises(kzg.KzgAmdError):   # a matrix call on a plain prepared handle
        kzg._check(kzg.lib().kzgamd_mult_pippenger_matrix(s.msm_handle(), out, sc, 1), "matrix")
    h.close()
    # the matrix as a part of the Lagrange-points handle (one PrecomputationTable = points + matrix in the reference):
    # the same handle then answers both g1_lincomb and g1_lincomb_batch
    both = kzg.prepare_multi_scalar_mult(s.g1_lagrange_affine(), 4096, kzg.make_config(table_budget_gb=8))
    both.attach_matrix(aff, rows, cols, kzg.make_config(table_budget_gb=16))
    with pytest.raises(kzg.KzgAmdError):  # one matrix per handle, once: a re-attach would free a table matrix calls may be using
        both.attach_matrix(aff, rows, cols, kzg.make_config(table_budget_gb=16))
    got = both.multiply_batch(sc, 1)
    assert [compressed(L, got[k]) for k in range(rows)] == want[:rows]
    lag = s.g1_lagrange_affine()
    sc4096 = O.fr_array([rnd.randrange(O.R) for _ in range(4096)])
    exp = O.G1()
    L.omsm_affine(C.byref(exp), C.cast(lag, C.POINTER(O.G1Affine)), sc4096, 4096)
    assert compressed(L, kzg.multi_scalar_mult_prepared(both, sc4096, 4096)) == compressed(L, exp)
    both.close()
    s.close()


def test_no_table_fits_the_budget_means_no_matrix_handle(kzg, oracle_settings):
    pts = oracle_settings.g1_lagrange_brp
    with pytest.raises(kzg.KzgAmdError):
        kzg.MatrixMsm(pts, 64, 64, kzg.make_config(no_tables=True))
    with pytest.raises(kzg.KzgAmdError):
        kzg.MatrixMsm(pts, 64, 64, kzg.make_config(table_budget_gb=0.001))


def test_configuration_is_validated(kzg, oracle_settings):
    pts = oracle_settings.g1_lagrange_brp
    keys = kzg.tuning_keys()
    assert keys["fk20"][:3] == (-1, -1, 1) and keys["g1_pair_max"][0] == 32768 and len(keys) >= 35
    for bad in ("no_such_key=1", "spl=99", "spl", "spl=x", "fk20=2"):
        with pytest.raises(kzg.KzgAmdError):
            kzg.prepare_multi_scalar_mult(pts, 64, kzg.make_config(tuning=bad))
        with pytest.raises(kzg.KzgAmdError):
            kzg.FFTSettings(4, kzg.make_config(tuning=bad))
        with pytest.raises(kzg.KzgAmdError):
            kzg.KZGSettings.from_file(SETUP, kzg.make_config(tuning=bad))
    cfg = kzg.make_config()
    cfg.struct_size = 4
    with pytest.raises(kzg.KzgAmdError):
        kzg.FFTSettings(4, cfg)
    with pytest.raises(kzg.KzgAmdError):   # a device that does not exist
        kzg.FFTSettings(4, kzg.make_config(device=kzg.device_count() + 3))
    # separators and spaces are free-form; the caller's device is left alone
    before = kzg.get_device()
    h = kzg.prepare_multi_scalar_mult(pts, 64, kzg.make_config(device=0, table_budget_gb=0.5, tuning=" spl=1, combine=0 ;fbw_glv=0"))
    assert h.info()["wide_table"] and not h.info()["wide_glv"] and kzg.get_device() == before
    h.close()


def test_environment_tuning_string_and_explicit_config_precedence(kzg, oracle_settings, monkeypatch):
    """KZGAMD_TUNING is the one environment form of the tuning keys (measurement tools); an explicit KzgAmdConfig wins over
    it; a string that does not parse fails handle creation instead of being ignored."""
    pts = oracle_settings.g1_lagrange_brp
    monkeypatch.setenv("KZGAMD_TUNING", "fbw_glv=0")
    monkeypatch.setenv("KZGAMD_FBW_MAX_GB", "0.5")
    h = kzg.prepare_multi_scalar_mult(pts, 64)
    assert h.info()["wide_table"] and not h.info()["wide_glv"]
    h.close()
    h = kzg.prepare_multi_scalar_mult(pts, 64, kzg.make_config(tuning={"fbw_glv": 1}, table_budget_gb=1.0))
    assert h.info()["wide_glv"]
    h.close()
    monkeypatch.setenv("KZGAMD_TUNING", "fbw_glv=0;typo=1")
    with pytest.raises(kzg.KzgAmdError):
        kzg.prepare_multi_scalar_mult(pts, 64)

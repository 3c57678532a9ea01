"""Pins the CPU oracle against everything the reference's own tests hold for the hot path
(SURVEY.md §8c): c-kzg-4844 mainnet vectors + hard-coded known-answer constants.  CPU only."""
import ctypes as C
import hashlib
import os
import random

import pytest

import oracle_ffi as O

BLOB = 131072


def hx(b):
    return "0x" + bytes(b).hex()


def test_field_ops_vs_python_ints(oracle):
    L = oracle.lib()
    rnd = random.Random(1)
    for _ in range(200):
        a, b = rnd.randrange(O.P), rnd.randrange(O.P)
        fa, fb, fc = O.fp_from_int(a), O.fp_from_int(b), O.Fp()
        L.ofp_mul(C.byref(fc), C.byref(fa), C.byref(fb))
        assert O.fp_to_int(fc) == a * b % O.P
        L.ofp_add(C.byref(fc), C.byref(fa), C.byref(fb))
        assert O.fp_to_int(fc) == (a + b) % O.P
        L.ofp_sub(C.byref(fc), C.byref(fa), C.byref(fb))
        assert O.fp_to_int(fc) == (a - b) % O.P
    a = rnd.randrange(1, O.P)
    fa, fc = O.fp_from_int(a), O.Fp()
    L.ofp_inv(C.byref(fc), C.byref(fa))
    assert O.fp_to_int(fc) == pow(a, -1, O.P)
    for _ in range(200):
        a, b = rnd.randrange(O.R), rnd.randrange(O.R)
        fa, fb, fc = O.fr_from_int(a), O.fr_from_int(b), O.Fr()
        L.ofr_mul(C.byref(fc), C.byref(fa), C.byref(fb))
        assert O.fr_to_int(fc) == a * b % O.R
        L.ofr_sub(C.byref(fc), C.byref(fa), C.byref(fb))
        assert O.fr_to_int(fc) == (a - b) % O.R
    # edge values
    for a in (0, 1, O.R - 1):
        for b in (0, 1, O.R - 1):
            fa, fb, fc = O.fr_from_int(a), O.fr_from_int(b), O.Fr()
            L.ofr_mul(C.byref(fc), C.byref(fa), C.byref(fb))
            assert O.fr_to_int(fc) == a * b % O.R
            L.ofr_add(C.byref(fc), C.byref(fa), C.byref(fb))
            assert O.fr_to_int(fc) == (a + b) % O.R


def test_generator_limbs_match_reference_constant(oracle, kats):
    # blst/src/consts.rs:52-84 (Montgomery limbs of G, Z = R mod p)
    g = O.G1()
    oracle.lib().og1_generator(C.byref(g))
    got = list(g.x.l) + list(g.y.l) + list(g.z.l)
    assert got == kats["g1_generator_mont_limbs"]["xyz"]
    assert oracle.lib().og1_in_subgroup(C.byref(g)) == 1


def test_scale2_roots_match_reference_table(oracle, kats):
    for k, row in enumerate(kats["scale2_root_of_unity"]["limbs"]):
        out = (C.c_uint64 * 4)()
        oracle.lib().oscale2_root_of_unity(out, k)
        assert list(out) == row, k


def test_expected_powers(oracle, kats):
    # compute_powers_test, kzg-bench/src/tests/eip_4844.rs:72-82
    L = oracle.lib()
    base = O.fr_from_int(kats["expected_powers"]["base"])
    acc = O.fr_from_int(1)
    for row in kats["expected_powers"]["limbs"]:
        out = (C.c_uint64 * 4)()
        L.ofr_to_u64_arr(out, C.byref(acc))
        assert list(out) == row
        L.ofr_mul(C.byref(acc), C.byref(acc), C.byref(base))


def test_bytes_to_bls_field(oracle):
    # bytes_to_bls_field_test, eip_4844.rs:62-70; and the >= r rejection (fr.rs:64-86)
    L = oracle.lib()
    f = O.Fr()
    b = (329).to_bytes(32, "big")
    assert L.ofr_from_be32(C.byref(f), b) == 1
    out = C.create_string_buffer(32)
    L.ofr_to_be32(out, C.byref(f))
    assert out.raw == b
    assert L.ofr_from_be32(C.byref(f), O.R.to_bytes(32, "big")) == 0
    assert L.ofr_from_be32(C.byref(f), (O.R - 1).to_bytes(32, "big")) == 1
    assert L.ofr_from_be32(C.byref(f), b"\xff" * 32) == 0


def test_inverse_fft_kat(oracle, kats):
    L = oracle.lib()
    fs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(fs), 4) == 0
    data = O.fr_array(list(range(16)))
    out = (O.Fr * 16)()
    assert L.offt_fr(C.byref(fs), out, data, 16, 1) == 0
    for i, row in enumerate(kats["inverse_fft"]["expected"]):
        limbs = (C.c_uint64 * 4)()
        L.ofr_to_u64_arr(limbs, C.byref(out[i]))
        assert list(limbs) == row, i
    L.offt_settings_free(C.byref(fs))


def test_das_extension_kat(oracle, kats):
    L = oracle.lib()
    fs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(fs), 4) == 0
    evens = O.fr_array(list(range(8)))
    odds = (O.Fr * 8)()
    assert L.odas_fft_extension(C.byref(fs), odds, evens, 8) == 0
    for i, row in enumerate(kats["das_extension_known"]["expected"]):
        limbs = (C.c_uint64 * 4)()
        L.ofr_to_u64_arr(limbs, C.byref(odds[i]))
        assert list(limbs) == row, i
    L.offt_settings_free(C.byref(fs))


def test_fft_properties(oracle):
    # compare_sft_fft / roundtrip_fft / stride_fft, kzg-bench/src/tests/fft_fr.rs:5-106 (smaller n for the slow DFT)
    L = oracle.lib()
    fs, fs2 = O.FFTSettings(), O.FFTSettings()
    assert L.offt_settings_new(C.byref(fs), 8) == 0
    assert L.offt_settings_new(C.byref(fs2), 12) == 0
    n = 256
    rnd = random.Random(3)
    vals = [rnd.randrange(O.R) for _ in range(n)]
    data = O.fr_array(vals)
    a, b, c = (O.Fr * n)(), (O.Fr * n)(), (O.Fr * n)()
    assert L.offt_fr(C.byref(fs), a, data, n, 0) == 0
    L.offt_fr_slow(C.byref(fs), b, data, n)
    assert bytes(a) == bytes(b)
    assert L.offt_fr(C.byref(fs2), c, data, n, 0) == 0  # stride invariance
    assert bytes(a) == bytes(c)
    assert L.offt_fr(C.byref(fs), b, a, n, 1) == 0
    assert bytes(b) == bytes(data)
    # python check of one output: sum v_j w^(ij)
    w = pow(7, (O.R - 1) >> 8, O.R)
    i = 5
    assert O.fr_to_int(a[i]) == sum(v * pow(w, i * j, O.R) for j, v in enumerate(vals)) % O.R
    # error codes (fft_fr.rs:118-132)
    big = (O.Fr * 512)()
    assert L.offt_fr(C.byref(fs), big, big, 512, 0) == 1
    assert L.offt_fr(C.byref(fs), big, big, 24, 0) == 2
    # DAS: odd half of the inverse FFT of the interleaved vector is zero (das.rs:35-68)
    evens = O.fr_array(vals[:64])
    odds = (O.Fr * 64)()
    assert L.odas_fft_extension(C.byref(fs), odds, evens, 64) == 0
    inter = (O.Fr * 128)()
    for k in range(64):
        inter[2 * k] = evens[k]
        inter[2 * k + 1] = odds[k]
    co = (O.Fr * 128)()
    assert L.offt_fr(C.byref(fs), co, inter, 128, 1) == 0
    assert all(O.fr_to_int(co[k]) == 0 for k in range(64, 128))
    assert L.odas_fft_extension(C.byref(fs), odds, evens, 0) == 1
    assert L.odas_fft_extension(C.byref(fs), odds, evens, 24) == 2
    assert L.odas_fft_extension(C.byref(fs), big, big, 256) == 3
    L.offt_settings_free(C.byref(fs))
    L.offt_settings_free(C.byref(fs2))


def test_sha256(oracle):
    for msg in (b"", b"abc", b"a" * 55, b"a" * 56, b"a" * 64, bytes(range(256)) * 513):
        out = C.create_string_buffer(32)
        oracle.lib().osha256(out, msg, len(msg))
        assert out.raw == hashlib.sha256(msg).digest()


def test_booth_window_sizes(oracle):
    # pippenger_window_size, kzg/src/msm/pippenger_utils.rs:300-317 (values quoted in SURVEY §8 a3)
    f = oracle.lib().opippenger_window_size
    assert [f(1 << k) for k in (12, 16, 20, 21, 22)] == [10, 13, 17, 18, 19]


def test_msm_small_vs_naive(oracle):
    # MSM correctness in the style of kzg-bench/src/tests/bls12_381.rs:184-387
    L = oracle.lib()
    rnd = random.Random(7)
    g = O.G1()
    L.og1_generator(C.byref(g))
    for n in (0, 1, 7, 8, 9, 33, 255):
        pts = (O.G1Affine * max(n, 1))()
        sc = (O.Fr * max(n, 1))()
        ks = []
        for i in range(n):
            k = rnd.randrange(1, O.R)
            ks.append(k)
            t = O.G1()
            kf = O.fr_from_int(k)
            L.og1_mul(C.byref(t), C.byref(g), C.byref(kf))
            L.og1_to_affine(C.byref(pts[i]), C.byref(t))
            s = 0 if rnd.random() < 0.1 else rnd.randrange(O.R)
            sc[i] = O.fr_from_int(s)
            ks[-1] = (k, s)
        if n > 3:
            pts[2] = O.G1Affine()  # infinity point
            ks[2] = (0, ks[2][1])
        a, b, e = O.G1(), O.G1(), O.G1()
        L.omsm_affine(C.byref(a), pts, sc, n)
        L.omsm_naive(C.byref(b), pts, sc, n)
        assert L.og1_equal(C.byref(a), C.byref(b)) == 1, n
        tot = O.fr_from_int(sum(k * s for k, s in ks) % O.R)
        L.og1_mul(C.byref(e), C.byref(g), C.byref(tot))
        assert L.og1_equal(C.byref(a), C.byref(e)) == 1, n
        if n >= 64:
            L.omsm_affine_mt(C.byref(b), pts, sc, n, 3)
            assert L.og1_equal(C.byref(a), C.byref(b)) == 1


def test_trusted_setup_parser_rejects(oracle, trusted_setup_text):
    # shapes from kzg-bench/src/tests/c_bindings.rs:344-430
    t = trusted_setup_text
    assert oracle.load_settings(t.replace(b"4096", b"4097", 1))[0] != 0
    assert oracle.load_settings(t.replace(b"65", b"64", 1))[0] != 0
    assert oracle.load_settings(b"")[0] != 0
    assert oracle.load_settings(t[: len(t) // 2])[0] != 0
    assert oracle.load_settings(t.replace(b"a", b"g", 1))[0] != 0


def test_kat_commitment_and_proof(oracle, oracle_settings, kats):
    L = oracle.lib()
    k = kats["blob_to_kzg_commitment_test"]
    blob = bytes.fromhex(k["field_element"][2:]) + bytes(BLOB - 32)
    out = C.create_string_buffer(48)
    assert L.oblob_to_kzg_commitment(out, blob, C.byref(oracle_settings)) == 0
    assert hx(out.raw) == k["commitment"]
    k = kats["compute_kzg_proof_test"]
    blob = bytes.fromhex(k["field_element"][2:]) + bytes(BLOB - 32)
    y = C.create_string_buffer(32)
    assert L.ocompute_kzg_proof(out, y, blob, bytes.fromhex(k["z"][2:]), C.byref(oracle_settings)) == 0
    assert hx(out.raw) == k["proof"]


def test_vectors_blob_to_kzg_commitment(oracle, oracle_settings, golden, blob_loader):
    L = oracle.lib()
    nvalid = 0
    for case in golden["blob_to_kzg_commitment"]:
        blob = blob_loader(case["blob"])
        out = C.create_string_buffer(48)
        rc = 1 if len(blob) != BLOB else L.oblob_to_kzg_commitment(out, blob, C.byref(oracle_settings))
        if case["output"] is None:
            assert rc != 0, case["name"]
        else:
            assert rc == 0 and hx(out.raw) == case["output"], case["name"]
            nvalid += 1
    assert nvalid == 7


def test_vectors_blob_to_kzg_commitment_bgmw(oracle, oracle_settings, golden, blob_loader):
    """The BGMW fixed-base path (kzg/src/msm/bgmw.rs — the reference's default with feature `bgmw`, and the algorithm
    bench.py's cpu_baseline times) on the same vectors: window 13, 20 table rows for the 4096-point setup."""
    L = oracle.lib()
    assert L.obgmw_window_size(4096) == 13 and L.obgmw_window_size(1 << 20) == 20 and L.obgmw_window_size(8) == 6
    nvalid = 0
    for case in golden["blob_to_kzg_commitment"]:
        blob = blob_loader(case["blob"])
        out = C.create_string_buffer(48)
        rc = 1 if len(blob) != BLOB else L.oblob_to_kzg_commitment_bgmw(out, blob, C.byref(oracle_settings))
        if case["output"] is None:
            assert rc != 0, case["name"]
        else:
            assert rc == 0 and hx(out.raw) == case["output"], case["name"]
            nvalid += 1
    assert nvalid == 7


def test_vectors_compute_challenge(oracle, golden, blob_loader):
    L = oracle.lib()
    n = 0
    for case in golden["compute_challenge"]:
        blob = blob_loader(case["blob"])
        poly = (O.Fr * 4096)()
        assert L.oblob_to_fr(poly, blob) == 0
        z = O.Fr()
        L.ocompute_challenge(C.byref(z), poly, bytes.fromhex(case["commitment"][2:]))
        out = C.create_string_buffer(32)
        L.ofr_to_be32(out, C.byref(z))
        assert hx(out.raw) == case["output"], case["name"]
        n += 1
    assert n == 9


def test_vectors_compute_kzg_proof(oracle, oracle_settings, golden, blob_loader):
    L = oracle.lib()
    nvalid = 0
    for case in golden["compute_kzg_proof"]:
        blob = blob_loader(case["blob"])
        z = bytes.fromhex(case["z"][2:])
        pr, y = C.create_string_buffer(48), C.create_string_buffer(32)
        rc = 1 if (len(blob) != BLOB or len(z) != 32) else L.ocompute_kzg_proof(pr, y, blob, z, C.byref(oracle_settings))
        if case["output"] is None:
            assert rc != 0, case["name"]
        else:
            assert rc == 0, case["name"]
            assert [hx(pr.raw), hx(y.raw)] == case["output"], case["name"]
            nvalid += 1
    assert nvalid == 42


def test_vectors_compute_blob_kzg_proof(oracle, oracle_settings, golden, blob_loader):
    L = oracle.lib()
    nvalid = 0
    for case in golden["compute_blob_kzg_proof"]:
        blob = blob_loader(case["blob"])
        cm = bytes.fromhex(case["commitment"][2:])
        pr = C.create_string_buffer(48)
        rc = 1 if (len(blob) != BLOB or len(cm) != 48) else L.ocompute_blob_kzg_proof(pr, blob, cm, C.byref(oracle_settings))
        if case["output"] is None:
            assert rc != 0, case["name"]
        else:
            assert rc == 0 and hx(pr.raw) == case["output"], case["name"]
            nvalid += 1
    assert nvalid == 7


def test_vectors_compute_cells_pin_ntt(oracle, oracle_settings, golden, blob_loader):
    L = oracle.lib()
    nvalid = 0
    for case in golden["compute_cells"]:
        blob = blob_loader(case["blob"])
        out = C.create_string_buffer(8192 * 32)
        rc = 1 if len(blob) != BLOB else L.ocompute_cells(out, blob, C.byref(oracle_settings))
        if case["output"] is None:
            assert rc != 0, case["name"]
        else:
            assert rc == 0, case["name"]
            assert hx(out.raw[:2048]) == case["output"]["cell0"], case["name"]
            assert hx(out.raw[-2048:]) == case["output"]["cell127"], case["name"]
            assert hashlib.sha256(out.raw).hexdigest() == case["output"]["sha256"], case["name"]
            nvalid += 1
    assert nvalid == 7


def test_vectors_cell_proofs_pin_definition(oracle, oracle_settings, golden, blob_loader):
    # the cell proofs of compute_cells_and_kzg_proofs (FK20 in the reference) equal the
    # per-cell quotient commitments; three cells of two vectors (one 4096-point CPU MSM each)
    L = oracle.lib()
    done = 0
    for case in golden["compute_cells_and_kzg_proofs"]:
        if case["output"] is None or done >= 2:
            continue
        blob = blob_loader(case["blob"])
        for k, key in ((0, "proof0"), (1, "proof1"), (127, "proof127")):
            out = C.create_string_buffer(48)
            assert L.ocompute_cell_proof(out, blob, k, C.byref(oracle_settings)) == 0
            assert hx(out.raw) == case["output"][key], (case["name"], k)
        done += 1
    assert done == 2


def test_fft_g1_properties(oracle):
    # compare_ft_fft / roundtrip_fft / stride_fft of kzg-bench/src/tests/fft_g1.rs, at CPU-friendly sizes
    L = oracle.lib()
    fs, fs2 = O.FFTSettings(), O.FFTSettings()
    assert L.offt_settings_new(C.byref(fs), 4) == 0
    assert L.offt_settings_new(C.byref(fs2), 6) == 0
    n = 16
    g = O.G1()
    L.og1_generator(C.byref(g))
    data = (O.G1 * n)()
    acc = O.G1()
    L.og1_generator(C.byref(acc))
    for i in range(n):  # make_data: ascending multiples of the generator
        data[i] = acc
        L.og1_add_or_dbl(C.byref(acc), C.byref(acc), C.byref(g))
    data[5] = O.G1()  # a point at infinity
    fast, slow, other, back = (O.G1 * n)(), (O.G1 * n)(), (O.G1 * n)(), (O.G1 * n)()
    assert L.offt_g1(C.byref(fs), fast, data, n, 0) == 0
    L.offt_g1_slow(C.byref(fs), slow, data, n)
    assert L.offt_g1(C.byref(fs2), other, data, n, 0) == 0
    assert L.offt_g1(C.byref(fs), back, fast, n, 1) == 0
    for i in range(n):
        assert L.og1_equal(C.byref(fast[i]), C.byref(slow[i])) == 1
        assert L.og1_equal(C.byref(fast[i]), C.byref(other[i])) == 1
        assert L.og1_equal(C.byref(back[i]), C.byref(data[i])) == 1
    big = (O.G1 * 32)()
    assert L.offt_g1(C.byref(fs), big, big, 32, 0) == 1
    assert L.offt_g1(C.byref(fs), big, big, 12, 0) == 2
    L.offt_settings_free(C.byref(fs))
    L.offt_settings_free(C.byref(fs2))


def test_recover_model_pinned_on_reference_vectors(kats):
    """tests/recover_model.py (the Python restatement of recover_cells, das.rs:566-657) against the reference's three
    half-missing recover_cells_and_kzg_proofs vectors: every recovered cell."""
    import gzip
    import json

    import recover_model as M
    from conftest import GOLDEN

    with open(os.path.join(GOLDEN, "kzg_mainnet_7594.json")) as f:
        v = json.load(f)
    with gzip.open(os.path.join(GOLDEN, v["cells_file"]), "rb") as f:
        raw = f.read()
    cells = [raw[i: i + 2048] for i in range(0, len(raw), 2048)]
    limbs = kats["scale2_root_of_unity"]["limbs"][13]
    root = sum(int(x) << (64 * i) for i, x in enumerate(limbs))
    assert pow(root, 8192, M.R) == 1 and pow(root, 4096, M.R) != 1
    n = 0
    for case in v["recover_cells_and_kzg_proofs"]:
        if case["output"] is None or len(case["cell_indices"]) == 128:
            continue
        provided = {i: M.cell_to_ints(cells[ref]) for i, ref in zip(case["cell_indices"], case["cells"])}
        got = M.recover_cells(provided, root)
        want = [cells[ref] for ref in case["output"]["cells"]]
        assert [M.ints_to_cell(c) for c in got] == want, case["name"]
        n += 1
    assert n == 3

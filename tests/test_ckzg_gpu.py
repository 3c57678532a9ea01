"""c-kzg-4844 surface (B3) on the GPU against the reference's golden vectors and the oracle."""
import ctypes as C
import os
import random

import pytest

import oracle_ffi as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
BLOB = 131072


def hx(b):
    return "0x" + bytes(b).hex()


@pytest.fixture(scope="module")
def settings(kzg):
    s = kzg.KZGSettings.from_file(os.path.join(GOLDEN, "trusted_setup.txt"))
    yield s
    s.close()


def test_settings_arrays_match_oracle(kzg, settings, oracle, oracle_settings):
    L = oracle.lib()
    pts = settings.g1_lagrange_brp()
    for i in (0, 1, 2, 77, 4095):
        a = O.G1()
        C.memmove(C.byref(a), C.byref(pts[i]), 144)
        b = O.G1()
        L.og1_from_affine(C.byref(b), C.byref(oracle_settings.g1_lagrange_brp[i]))
        assert L.og1_equal(C.byref(a), C.byref(b)) == 1
    roots = (O.Fr * 8193).from_address(settings.c.roots_of_unity)
    brp = (O.Fr * 8192).from_address(settings.c.brp_roots_of_unity)
    rev = (O.Fr * 8193).from_address(settings.c.reverse_roots_of_unity)
    fs = oracle_settings.fs
    for i in (0, 1, 2, 4096, 8191, 8192):
        assert bytes(roots[i]) == bytes(fs.roots_of_unity[i])
        assert bytes(rev[i]) == bytes(fs.reverse_roots_of_unity[i])
    for i in (0, 1, 2, 4096, 8191):
        assert bytes(brp[i]) == bytes(fs.brp_roots_of_unity[i])


def test_vectors_blob_to_kzg_commitment(kzg, settings, golden, blob_loader):
    nvalid = 0
    for case in golden["blob_to_kzg_commitment"]:
        blob = blob_loader(case["blob"])
        if case["output"] is None:
            with pytest.raises(kzg.KzgAmdError):
                kzg.blob_to_kzg_commitment(blob, settings)
        else:
            assert hx(kzg.blob_to_kzg_commitment(blob, settings)) == case["output"], case["name"]
            nvalid += 1
    assert nvalid == 7


def test_kat_commitment(kzg, settings, kats):
    k = kats["blob_to_kzg_commitment_test"]
    blob = bytes.fromhex(k["field_element"][2:]) + bytes(BLOB - 32)
    assert hx(kzg.blob_to_kzg_commitment(blob, settings)) == k["commitment"]


def test_batch_commit_matches_oracle(kzg, settings, oracle, oracle_settings):
    L = oracle.lib()
    rnd = random.Random(4844)
    n = 5
    blobs = bytearray(rnd.randbytes(n * BLOB))
    for i in range(0, n * BLOB, 32):
        blobs[i] = 0  # generate_random_blob_bytes, kzg-bench/src/tests/eip_4844.rs:28-37
    got = kzg.blob_to_kzg_commitment_batch(bytes(blobs), n, settings)
    for b in range(n):
        out = C.create_string_buffer(48)
        assert L.oblob_to_kzg_commitment(out, bytes(blobs[b * BLOB:(b + 1) * BLOB]), C.byref(oracle_settings)) == 0
        assert got[b] == out.raw
    # element == r  => BadArgs for the whole call (kzg-bench/src/tests/c_bindings.rs:65-97)
    blobs[BLOB + 64:BLOB + 96] = O.R.to_bytes(32, "big")
    with pytest.raises(kzg.KzgAmdError):
        kzg.blob_to_kzg_commitment_batch(bytes(blobs), n, settings)


def test_load_rejects_bad_setups(kzg, trusted_setup_text):
    import tempfile

    def try_load(text):
        with tempfile.NamedTemporaryFile("wb", suffix=".txt", delete=False) as f:
            f.write(text)
            path = f.name
        try:
            s = kzg.KZGSettings.from_file(path)
            s.close()
            return True
        except kzg.KzgAmdError:
            return False
        finally:
            os.unlink(path)

    t = trusted_setup_text
    assert try_load(t)
    assert not try_load(t.replace(b"4096", b"4097", 1))
    assert not try_load(t[: len(t) // 2])
    assert not try_load(b"")
    # "old" monomial-form file: put the monomial points where the Lagrange ones belong
    lines = t.split(b"\n")
    swapped = lines[:2] + lines[2 + 4096 + 65:2 + 4096 + 65 + 4096] + lines[2 + 4096:2 + 4096 + 65] + lines[2:2 + 4096]
    assert not try_load(b"\n".join(swapped) + b"\n")
    # free is idempotent and null-safe (c_bindings.rs:490-544)
    s = kzg.KZGSettings.from_file(os.path.join(GOLDEN, "trusted_setup.txt"))
    kzg.lib().free_trusted_setup(C.byref(s.c))
    assert s.c.g1_values_lagrange_brp is None and s.c.roots_of_unity is None
    kzg.lib().free_trusted_setup(C.byref(s.c))
    kzg.lib().free_trusted_setup(None)
    s.loaded = False


def test_vectors_compute_kzg_proof(kzg, settings, golden, blob_loader):
    nvalid = 0
    for case in golden["compute_kzg_proof"]:
        blob = blob_loader(case["blob"])
        z = bytes.fromhex(case["z"][2:])
        if case["output"] is None:
            with pytest.raises(kzg.KzgAmdError):
                kzg.compute_kzg_proof(blob, z, settings)
        else:
            proof, y = kzg.compute_kzg_proof(blob, z, settings)
            assert [hx(proof), hx(y)] == case["output"], case["name"]
            nvalid += 1
    assert nvalid == 42


def test_vectors_compute_blob_kzg_proof(kzg, settings, golden, blob_loader):
    nvalid = 0
    for case in golden["compute_blob_kzg_proof"]:
        blob = blob_loader(case["blob"])
        cm = bytes.fromhex(case["commitment"][2:])
        if case["output"] is None:
            with pytest.raises(kzg.KzgAmdError):
                kzg.compute_blob_kzg_proof(blob, cm, settings)
        else:
            assert hx(kzg.compute_blob_kzg_proof(blob, cm, settings)) == case["output"], case["name"]
            nvalid += 1
    assert nvalid == 7


def test_vectors_compute_challenge(kzg, golden, blob_loader):
    n = 0
    for case in golden["compute_challenge"]:
        blob = blob_loader(case["blob"])
        p1 = kzg.bytes_to_kzg_commitment(bytes.fromhex(case["commitment"][2:]))
        z = kzg.compute_challenge(blob, p1)
        assert hx(kzg.bytes_from_bls_field(z)) == case["output"], case["name"]
        n += 1
    assert n == 9


def test_kat_proof_and_domain_points(kzg, settings, kats, oracle, oracle_settings):
    k = kats["compute_kzg_proof_test"]
    blob = bytes.fromhex(k["field_element"][2:]) + bytes(BLOB - 32)
    proof, y = kzg.compute_kzg_proof(blob, bytes.fromhex(k["z"][2:]), settings)
    assert hx(proof) == k["proof"]
    # z inside the evaluation domain (compute_and_verify_kzg_proof_within_domain_test,
    # kzg-bench/src/tests/eip_4844.rs:236-288), checked against the oracle
    L = oracle.lib()
    rnd = random.Random(9)
    blob = bytearray(rnd.randbytes(BLOB))
    for i in range(0, BLOB, 32):
        blob[i] = 0
    blob = bytes(blob)
    brp = (O.Fr * 8192).from_address(settings.c.brp_roots_of_unity)
    for idx in (0, 1, 5, 4095):
        zb = C.create_string_buffer(32)
        f = O.Fr()
        C.memmove(C.byref(f), C.byref(brp[idx]), 32)
        L.ofr_to_be32(zb, C.byref(f))
        ep, ey = C.create_string_buffer(48), C.create_string_buffer(32)
        assert L.ocompute_kzg_proof(ep, ey, blob, zb.raw, C.byref(oracle_settings)) == 0
        proof, y = kzg.compute_kzg_proof(blob, zb.raw, settings)
        assert (proof, y) == (ep.raw, ey.raw), idx
        assert y == blob[32 * idx:32 * idx + 32]


def test_blob_proof_batch_equals_singles_and_oracle(kzg, settings, oracle, oracle_settings):
    L = oracle.lib()
    rnd = random.Random(77)
    n = 4
    blobs = bytearray(rnd.randbytes(n * BLOB))
    for i in range(0, n * BLOB, 32):
        blobs[i] = 0
    blobs = bytes(blobs)
    cms = kzg.blob_to_kzg_commitment_batch(blobs, n, settings)
    got = kzg.compute_blob_kzg_proof_batch(blobs, b"".join(cms), n, settings)
    for b in range(n):
        bl = blobs[b * BLOB:(b + 1) * BLOB]
        assert got[b] == kzg.compute_blob_kzg_proof(bl, cms[b], settings)
        ep = C.create_string_buffer(48)
        assert L.ocompute_blob_kzg_proof(ep, bl, cms[b], C.byref(oracle_settings)) == 0
        assert got[b] == ep.raw
    # commitment = infinity is accepted (kzg-bench/src/tests/c_bindings.rs:584-616)
    inf = b"\xc0" + bytes(47)
    ep = C.create_string_buffer(48)
    assert L.ocompute_blob_kzg_proof(ep, blobs[:BLOB], inf, C.byref(oracle_settings)) == 0
    assert kzg.compute_blob_kzg_proof(blobs[:BLOB], inf, settings) == ep.raw


@pytest.mark.parametrize("n", [3, 7, 16, 40])
def test_batches_with_zero_and_constant_blobs(kzg, settings, oracle, oracle_settings, n):
    """Points at infinity inside a batch (the all-zero blob commits to infinity and so does its proof; a constant blob's
    proof is infinity too): the host-side batch compression of small batches shares one inversion over the finite
    points only, the device-side one (above 16 blobs) likewise; every result against the oracle."""
    L = oracle.lib()
    rnd = random.Random(300 + n)
    blobs = []
    for i in range(n):
        if i % 3 == 0:
            blobs.append(bytes(BLOB))
        elif i % 3 == 1:
            blobs.append(rnd.randrange(O.R).to_bytes(32, "big") * 4096)
        else:
            b = bytearray(rnd.randbytes(BLOB))
            for k in range(0, BLOB, 32):
                b[k] = 0
            blobs.append(bytes(b))
    flat = b"".join(blobs)
    cms = kzg.blob_to_kzg_commitment_batch(flat, n, settings)
    proofs = kzg.compute_blob_kzg_proof_batch(flat, b"".join(cms), n, settings)
    for i in range(n):
        ec, ep = C.create_string_buffer(48), C.create_string_buffer(48)
        assert L.oblob_to_kzg_commitment(ec, blobs[i], C.byref(oracle_settings)) == 0
        assert cms[i] == ec.raw, (n, i)
        assert L.ocompute_blob_kzg_proof(ep, blobs[i], ec.raw, C.byref(oracle_settings)) == 0
        assert proofs[i] == ep.raw, (n, i)
        if i % 3 == 0:
            assert cms[i] == b"\xc0" + bytes(47)
        if i % 3 != 2:
            assert proofs[i] == b"\xc0" + bytes(47)


def test_compute_cells_vectors_pin_gpu_ntt(kzg, golden, blob_loader, oracle):
    # polynomial half of compute_cells (kzg/src/das.rs:258-279) through the GPU NTT:
    # ifft(brp(blob)) -> zero-extend -> fft 8192 -> brp, against the c-kzg vectors
    import hashlib

    fs = kzg.FFTSettings(13)

    def brp(vals):
        n = len(vals)
        bits = n.bit_length() - 1
        return [vals[int(format(i, "0%db" % bits)[::-1], 2)] for i in range(n)]

    nvalid = 0
    for case in golden["compute_cells"]:
        if case["output"] is None:
            continue
        blob = blob_loader(case["blob"])
        elems = [int.from_bytes(blob[32 * i:32 * i + 32], "big") for i in range(4096)]
        raw = b"".join(((v << 256) % O.R).to_bytes(32, "little") for v in brp(elems))
        poly = (kzg.BlstFr * 4096)()
        C.memmove(poly, raw, len(raw))
        mono = fs.fft_fr(poly, 4096, inverse=True)
        ext = (kzg.BlstFr * 8192)()
        C.memmove(ext, mono, 4096 * 32)
        ev = fs.fft_fr(ext, 8192)
        ints = []
        for i in range(8192):
            f = O.Fr()
            C.memmove(C.byref(f), C.byref(ev[i]), 32)
            ints.append(O.fr_to_int(f))
        cells = b"".join(v.to_bytes(32, "big") for v in brp(ints))
        assert hx(cells[:2048]) == case["output"]["cell0"], case["name"]
        assert hashlib.sha256(cells).hexdigest() == case["output"]["sha256"], case["name"]
        nvalid += 1
    assert nvalid == 7
    fs.close()


@pytest.mark.parametrize("n", [6, 70])
def test_batch_rejects_invalid_commitments_host_and_device_checks(kzg, settings, golden, blob_loader, n):
    # batches of up to 64 blobs validate their commitments on the host's cores while the GPU proves, larger ones on the
    # device (k_check_commitments, one lane each on a low-priority stream); every invalid-commitment vector of
    # compute_blob_kzg_proof (bad flags, x >= p, not on curve, not in the subgroup) must fail the batch either way
    rnd = random.Random(31)
    blobs = bytearray(rnd.randbytes(n * BLOB))
    for i in range(0, n * BLOB, 32):
        blobs[i] = 0
    blobs = bytes(blobs)
    cms = kzg.blob_to_kzg_commitment_batch(blobs, n, settings)
    assert len(kzg.compute_blob_kzg_proof_batch(blobs, b"".join(cms), n, settings)) == n
    bad = [c for c in golden["compute_blob_kzg_proof"] if "invalid_commitment" in c["name"]]
    assert len(bad) == 4
    for c in bad:
        cm = bytes.fromhex(c["commitment"][2:])
        if len(cm) != 48:
            continue
        for pos in (0, n - 1):
            mixed = list(cms)
            mixed[pos] = cm
            with pytest.raises(kzg.KzgAmdError):
                kzg.compute_blob_kzg_proof_batch(blobs, b"".join(mixed), n, settings)


def test_vectors_compute_cells_and_kzg_proofs(kzg, settings, golden, blob_loader):
    # SURVEY §8(f) item 1; c-kzg vectors: 7 valid (128 cells + 128 proofs each), 4 invalid blobs
    import hashlib

    nvalid = 0
    for case in golden["compute_cells_and_kzg_proofs"]:
        blob = blob_loader(case["blob"])
        if case["output"] is None:
            with pytest.raises(kzg.KzgAmdError):
                kzg.compute_cells_and_kzg_proofs(blob, settings)
            continue
        cells, proofs = kzg.compute_cells_and_kzg_proofs(blob, settings)
        exp = case["output"]
        assert hx(proofs[:48]) == exp["proof0"] and hx(proofs[-48:]) == exp["proof127"], case["name"]
        assert hashlib.sha256(cells).hexdigest() == exp["cells_sha256"], case["name"]
        assert hashlib.sha256(proofs).hexdigest() == exp["proofs_sha256"], case["name"]
        nvalid += 1
    assert nvalid == 7
    # cells-only and proofs-only calls, and the batch form
    blob = blob_loader(golden["compute_cells_and_kzg_proofs"][-1]["blob"])
    c_only, none = kzg.compute_cells_and_kzg_proofs(blob, settings, want_proofs=False)
    assert none is None and c_only == cells
    none, p_only = kzg.compute_cells_and_kzg_proofs(blob, settings, want_cells=False)
    assert none is None and p_only == proofs
    blob2 = blob_loader(golden["compute_cells_and_kzg_proofs"][-2]["blob"])
    bc, bp = kzg.compute_cells_and_kzg_proofs_batch(blob2 + blob, 2, settings)
    assert bc[128 * 2048:] == cells and bp[128 * 48:] == proofs
    assert hashlib.sha256(bp[:128 * 48]).hexdigest() == golden["compute_cells_and_kzg_proofs"][-2]["output"]["proofs_sha256"]


def test_batch_256_blobs_commit_and_prove(kzg, settings, oracle, oracle_settings, monkeypatch):
    # BASELINE configs[4] size: 256 blobs in one batched call; sampled blobs checked against the oracle,
    # all of them against the single-blob path through a digest of digests
    import hashlib

    L = oracle.lib()
    rnd = random.Random(256)
    n = 256
    blobs = bytearray(rnd.randbytes(n * BLOB))
    for i in range(0, n * BLOB, 32):
        blobs[i] = 0
    blobs[7 * BLOB:8 * BLOB] = bytes(BLOB)                       # an all-zero blob -> infinity commitment
    blobs[9 * BLOB:10 * BLOB] = (b"\x00" * 31 + b"\x05") * 4096  # all elements equal
    blobs = bytes(blobs)
    cms = kzg.blob_to_kzg_commitment_batch(blobs, n, settings)
    proofs = kzg.compute_blob_kzg_proof_batch(blobs, b"".join(cms), n, settings)
    assert cms[7] == b"\xc0" + bytes(47)
    # every one of the 256 commitments and proofs against the oracle (its C code runs without the GIL: one worker per core)
    from concurrent.futures import ThreadPoolExecutor

    def check(b):
        bl = blobs[b * BLOB:(b + 1) * BLOB]
        ec, ep = C.create_string_buffer(48), C.create_string_buffer(48)
        assert L.oblob_to_kzg_commitment(ec, bl, C.byref(oracle_settings)) == 0
        assert L.ocompute_blob_kzg_proof(ep, bl, ec.raw, C.byref(oracle_settings)) == 0
        return (cms[b], proofs[b]) == (ec.raw, ep.raw)

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        assert all(ex.map(check, range(n)))
    singles = hashlib.sha256()
    for b in range(0, n, 16):
        bl = blobs[b * BLOB:(b + 1) * BLOB]
        singles.update(kzg.blob_to_kzg_commitment(bl, settings) + kzg.compute_blob_kzg_proof(bl, cms[b], settings))
    batch = hashlib.sha256()
    for b in range(0, n, 16):
        batch.update(cms[b] + proofs[b])
    assert singles.digest() == batch.digest()
    # the pipelined large-batch path rejects a bad blob (element >= r) and a bad commitment wherever they sit
    for pos in (3, 200):
        bad = bytearray(blobs)
        bad[pos * BLOB + 64:pos * BLOB + 96] = b"\xff" * 32
        with pytest.raises(kzg.KzgAmdError):
            kzg.compute_blob_kzg_proof_batch(bytes(bad), b"".join(cms), n, settings)
        badc = list(cms)
        badc[pos] = b"\x9f" + b"\xff" * 47  # compressed flag set, x >= p: not a field element
        with pytest.raises(kzg.KzgAmdError):
            kzg.compute_blob_kzg_proof_batch(blobs, b"".join(badc), n, settings)
    assert kzg.compute_blob_kzg_proof_batch(blobs, b"".join(cms), n, settings) == proofs  # and recovers afterwards


def test_concurrent_callers_share_one_settings_handle(kzg, settings, oracle, oracle_settings):
    """The reference shares the settings / MSM handle through Arc and calls it from rayon workers
    (kzg/src/eip_4844.rs:781-805): eight threads commit and prove at once on one CKZGSettings and on one prepared
    MSM handle; every result equals the sequential one."""
    import threading

    L = oracle.lib()
    rnd = random.Random(91)
    nblobs = 8
    blobs = []
    for _ in range(nblobs):
        b = bytearray(rnd.randbytes(BLOB))
        for i in range(0, BLOB, 32):
            b[i] = 0
        blobs.append(bytes(b))
    want_c = [kzg.blob_to_kzg_commitment(b, settings) for b in blobs]
    want_p = [kzg.compute_blob_kzg_proof(b, c, settings) for b, c in zip(blobs, want_c)]
    ec = C.create_string_buffer(48)
    assert L.oblob_to_kzg_commitment(ec, blobs[0], C.byref(oracle_settings)) == 0
    assert want_c[0] == ec.raw
    n = 64
    pts = (O.G1Affine * n)()
    g = O.G1()
    L.og1_generator(C.byref(g))
    for i in range(n):
        t = O.G1()
        k = O.fr_from_int(rnd.randrange(1, O.R))
        L.og1_mul(C.byref(t), C.byref(g), C.byref(k))
        L.og1_to_affine(C.byref(pts[i]), C.byref(t))
    scs = [O.fr_array([rnd.randrange(O.R) for _ in range(n)]) for _ in range(nblobs)]
    h = kzg.prepare_multi_scalar_mult(pts, n)
    want_m = [bytes(kzg.multi_scalar_mult_prepared(h, sc, n)) for sc in scs]
    errors = []

    def worker(i):
        try:
            for _ in range(3):
                assert kzg.blob_to_kzg_commitment(blobs[i], settings) == want_c[i]
                assert kzg.compute_blob_kzg_proof(blobs[i], want_c[i], settings) == want_p[i]
                assert bytes(kzg.multi_scalar_mult_prepared(h, scs[i], n)) == want_m[i]
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(nblobs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    h.close()
    assert not errors, errors


def test_sixteen_concurrent_callers_every_result_against_the_oracle(kzg, settings, oracle, oracle_settings, golden):
    """16 threads on ONE CKZGSettings, as the reference's rayon workers use it (kzg/src/eip_4844.rs:781-805): single
    commitments, single proofs, proofs at explicit points and small batches interleaved (the calls overlap on the
    lanes of the settings object, see ckzg_shared.h and ckzg.hip), a large batch on the parent's own pipeline in the middle; every
    result is compared with the oracle's, and an invalid blob in one thread fails only that call."""
    import threading

    L = oracle.lib()
    rnd = random.Random(16)
    nthreads = 16
    blobs = []
    for _ in range(nthreads):
        b = bytearray(rnd.randbytes(BLOB))
        for i in range(0, BLOB, 32):
            b[i] = 0
        blobs.append(bytes(b))
    want_c, want_p, want_zp = [], [], []
    zs = [rnd.randrange(O.R).to_bytes(32, "big") for _ in range(nthreads)]
    for b, z in zip(blobs, zs):
        c = C.create_string_buffer(48)
        assert L.oblob_to_kzg_commitment(c, b, C.byref(oracle_settings)) == 0
        pr = C.create_string_buffer(48)
        assert L.ocompute_blob_kzg_proof(pr, b, c.raw, C.byref(oracle_settings)) == 0
        pz, y = C.create_string_buffer(48), C.create_string_buffer(32)
        assert L.ocompute_kzg_proof(pz, y, b, z, C.byref(oracle_settings)) == 0
        want_c.append(c.raw)
        want_p.append(pr.raw)
        want_zp.append((pz.raw, y.raw))
    bad = bytearray(blobs[0])
    bad[:32] = O.R.to_bytes(32, "big")  # element == r
    # an element of the blob's evaluation domain
    brp = (O.Fr * 8192).from_address(settings.c.brp_roots_of_unity)
    zd, f = C.create_string_buffer(32), O.Fr()
    C.memmove(C.byref(f), C.byref(brp[5]), 32)
    L.ofr_to_be32(zd, C.byref(f))
    z_dom = zd.raw
    # the reference's invalid-commitment vectors: bad flags, x >= p, not on the curve, not in the subgroup
    bad_cms = [bytes.fromhex(c["commitment"][2:]) for c in golden["compute_blob_kzg_proof"] if "invalid_commitment" in c["name"]]
    bad_cms = [c for c in bad_cms if len(c) == 48]
    assert len(bad_cms) >= 2
    pz, y = C.create_string_buffer(48), C.create_string_buffer(32)
    assert L.ocompute_kzg_proof(pz, y, blobs[9], z_dom, C.byref(oracle_settings)) == 0
    want_dom = (pz.raw, y.raw)
    errors = []
    start = threading.Barrier(nthreads)

    def worker(i):
        try:
            start.wait()
            for rep in range(4):
                assert kzg.blob_to_kzg_commitment(blobs[i], settings) == want_c[i]
                assert kzg.compute_blob_kzg_proof(blobs[i], want_c[i], settings) == want_p[i]
                assert kzg.compute_kzg_proof(blobs[i], zs[i], settings) == want_zp[i]
                j = (i + 1) % nthreads
                assert kzg.blob_to_kzg_commitment_batch(blobs[i] + blobs[j], 2, settings) == [want_c[i], want_c[j]]
                assert kzg.compute_blob_kzg_proof_batch(blobs[i] + blobs[j], want_c[i] + want_c[j], 2, settings) == \
                    [want_p[i], want_p[j]]
                if i == 3 and rep == 1:  # beyond the lane size: the parent's pipeline, while the others keep calling
                    big = kzg.blob_to_kzg_commitment_batch(b"".join(blobs) * 2, 2 * nthreads, settings)
                    assert big == want_c * 2
                if i == 5:
                    with pytest.raises(kzg.KzgAmdError):
                        kzg.blob_to_kzg_commitment(bytes(bad), settings)
                    with pytest.raises(kzg.KzgAmdError):
                        kzg.compute_blob_kzg_proof(bytes(bad), want_c[0], settings)
                    with pytest.raises(kzg.KzgAmdError):
                        kzg.compute_kzg_proof(bytes(bad), zs[i], settings)
                if i == 7:
                    # a commitment that is not a point of G1, an evaluation point that is not canonical: each fails
                    # alone, inside whatever batch the call was merged into
                    for cm in bad_cms:
                        with pytest.raises(kzg.KzgAmdError):
                            kzg.compute_blob_kzg_proof(blobs[i], cm, settings)
                    with pytest.raises(kzg.KzgAmdError):
                        kzg.compute_kzg_proof(blobs[i], O.R.to_bytes(32, "big"), settings)
                if i == 9:
                    # an evaluation point inside the domain (the quotient's special column), merged with the others
                    assert kzg.compute_kzg_proof(blobs[i], z_dom, settings) == want_dom
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]


def test_more_concurrent_callers_than_staging_slots(kzg, settings, oracle, oracle_settings):
    """64 threads on one CKZGSettings: more callers than page-locked staging slots (48) — the ones that find no slot
    leave the copy of their blob to the batch leader; every result against the oracle."""
    import threading

    L = oracle.lib()
    rnd = random.Random(64)
    nthreads, ndistinct = 64, 8
    blobs, want_c, want_p = [], [], []
    for _ in range(ndistinct):
        b = bytearray(rnd.randbytes(BLOB))
        for i in range(0, BLOB, 32):
            b[i] = 0
        blobs.append(bytes(b))
        c, pr = C.create_string_buffer(48), C.create_string_buffer(48)
        assert L.oblob_to_kzg_commitment(c, blobs[-1], C.byref(oracle_settings)) == 0
        assert L.ocompute_blob_kzg_proof(pr, blobs[-1], c.raw, C.byref(oracle_settings)) == 0
        want_c.append(c.raw)
        want_p.append(pr.raw)
    errors = []
    start = threading.Barrier(nthreads)

    def worker(i):
        try:
            k = i % ndistinct
            start.wait()
            for rep in range(3):
                assert kzg.blob_to_kzg_commitment(blobs[k], settings) == want_c[k]
                assert kzg.compute_blob_kzg_proof(blobs[k], want_c[k], settings) == want_p[k]
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]


def test_challenges_and_evaluations_for_batched_verification(kzg, settings, oracle, oracle_settings, golden, blob_loader):
    """compute_challenges_and_evaluate_polynomial (kzg/src/eip_4844.rs:690-719), the field half of
    verify_blob_kzg_proof_batch: z from the oracle's Fiat-Shamir restatement, y from the oracle's evaluation; the
    blobs of the compute_blob_kzg_proof vectors plus random ones; batch of 1, of 3 (host-side commitment check)
    and of 9 (device-side check)."""
    L = oracle.lib()
    rnd = random.Random(57)
    blobs, cms = [], []
    for case in golden["compute_blob_kzg_proof"]:
        if case["output"] is None:
            continue
        blobs.append(blob_loader(case["blob"]))
        cms.append(bytes.fromhex(case["commitment"][2:]))
    while len(blobs) < 9:
        b = bytearray(rnd.randbytes(BLOB))
        for i in range(0, BLOB, 32):
            b[i] = 0
        blobs.append(bytes(b))
        cms.append(kzg.blob_to_kzg_commitment(blobs[-1], settings))
    for n in (1, 3, 9):
        zs, ys = kzg.compute_challenges_and_evaluate_batch(b"".join(blobs[:n]), b"".join(cms[:n]), n, settings)
        for i in range(n):
            poly = (O.Fr * 4096)()
            assert L.oblob_to_fr(poly, blobs[i]) == 0
            z = O.Fr()
            L.ocompute_challenge(C.byref(z), poly, cms[i])
            zb = C.create_string_buffer(32)
            L.ofr_to_be32(zb, C.byref(z))
            assert zs[i] == zb.raw, (n, i)
            ep, ey = C.create_string_buffer(48), C.create_string_buffer(32)
            assert L.ocompute_kzg_proof(ep, ey, blobs[i], zb.raw, C.byref(oracle_settings)) == 0
            assert ys[i] == ey.raw, (n, i)
    # an invalid commitment is rejected, as validate_batched_input does
    bad = bytearray(cms[0])
    bad[5] ^= 1
    with pytest.raises(kzg.KzgAmdError):
        kzg.compute_challenges_and_evaluate_batch(blobs[0] + blobs[1], bytes(bad) + cms[1], 2, settings)


def test_device_entry_point_on_two_streams(kzg, settings):
    """Independent batches on two streams share the settings' MSM handle (one workspace per stream): results equal
    the host-buffer path whatever the interleaving."""
    import torch

    rnd = random.Random(101)
    nb = 24
    dev = torch.device("cuda", 0)
    host = []
    for _ in range(2):
        b = bytearray(rnd.randbytes(nb * BLOB))
        for i in range(0, nb * BLOB, 32):
            b[i] = 0
        host.append(bytes(b))
    want = [kzg.blob_to_kzg_commitment_batch(h, nb, settings) for h in host]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    d_blobs = [torch.frombuffer(bytearray(h), dtype=torch.uint8).to(dev) for h in host]
    outs = [torch.zeros(nb * 48, dtype=torch.uint8, device=dev) for _ in range(2)]
    stats = [torch.zeros(nb, dtype=torch.int32, device=dev) for _ in range(2)]
    scratch = [torch.empty(nb * BLOB, dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(6):  # interleaved launches, no synchronisation in between
        k = rep % 2
        kzg.blob_to_kzg_commitment_device(outs[k].data_ptr(), stats[k].data_ptr(), scratch[k].data_ptr(),
                                          d_blobs[k].data_ptr(), nb, settings, streams[k].cuda_stream)
    torch.cuda.synchronize()
    for k in range(2):
        assert int(stats[k].sum().item()) == 0
        got = outs[k].cpu().numpy().tobytes()
        assert [got[48 * i:48 * i + 48] for i in range(nb)] == want[k]


@pytest.mark.parametrize("sha_lanes", [0, 4, 1])
def test_device_resident_blob_proofs_match_the_host_buffer_path(kzg, settings, golden, blob_loader, sha_lanes):
    """kzgamd_compute_blob_kzg_proof_device (SHA-256 challenge on the GPU) against the reference vectors and against
    the host-buffer entry point on random blobs; bad blobs / commitments only flag their own slot.  Both forms of the
    device hash (tuning key sha_lanes): four lanes per blob — the message schedules of four blocks side by side, the
    what a batch of this size takes by default (sha_lanes = 0) — and one lane per blob; 72 blobs = four full waves of the four-lane form and half of a fifth."""
    import torch

    module_settings = settings
    if sha_lanes != 0:
        settings = kzg.KZGSettings.from_file(os.path.join(GOLDEN, "trusted_setup.txt"),
                                             kzg.make_config(table_budget_gb=8, tuning={"sha_lanes": sha_lanes}))

    dev = torch.device("cuda", 0)
    cases = [c for c in golden["compute_blob_kzg_proof"] if c["output"] is not None]
    rnd = random.Random(21)
    blobs, cms, want = [], [], []
    for c in cases:
        blobs.append(blob_loader(c["blob"]))
        cms.append(bytes.fromhex(c["commitment"][2:]))
        want.append(bytes.fromhex(c["output"][2:]))
    for _ in range(70 - len(cases)):
        b = bytearray(rnd.randbytes(BLOB))
        for i in range(0, BLOB, 32):
            b[i] = 0
        blobs.append(bytes(b))
    extra = blobs[len(cases):]
    ecm = kzg.blob_to_kzg_commitment_batch(b"".join(extra), len(extra), module_settings)
    cms += ecm
    want += kzg.compute_blob_kzg_proof_batch(b"".join(extra), b"".join(ecm), len(extra), module_settings)  # host SHA-256
    n = len(blobs)
    # slot n: blob with an element == r; slot n + 1: commitment that is no G1 element
    bad_blob = bytearray(blobs[-1])
    bad_blob[32:64] = O.R.to_bytes(32, "big")
    blobs.append(bytes(bad_blob))
    cms.append(cms[-1])
    blobs.append(blobs[0])
    cms.append(bytes.fromhex("8123456789abcdef0123456789abcdef0123456789abcdef0123456789abcdef0123456789abcdef0123456789abcdef"))
    m = len(blobs)
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev)
    d_cms = torch.frombuffer(bytearray(b"".join(cms)), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(m * 48, dtype=torch.uint8, device=dev)
    d_stat = torch.zeros(m, dtype=torch.int32, device=dev)
    d_scr = torch.empty(m * kzg.PROOF_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(device=dev)
    kzg.compute_blob_kzg_proof_device(d_out.data_ptr(), d_stat.data_ptr(), d_scr.data_ptr(), d_blobs.data_ptr(),
                                      d_cms.data_ptr(), m, settings, st.cuda_stream)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy().tobytes()
    stat = d_stat.cpu().tolist()
    assert stat[:n] == [0] * n and stat[n] != 0 and stat[n + 1] != 0
    for i in range(n):
        assert out[48 * i:48 * i + 48] == want[i], i
    if settings is not module_settings:
        settings.close()


def test_settings_fk20_columns_match_oracle(kzg, settings, oracle, oracle_settings):
    """x_ext_fft_columns of the settings struct (FsKZGSettings::new, blst/src/types/kzg_settings.rs:84-101): the
    size-128 G1 transform of the strided monomial points, for three offsets, against the oracle's offt_g1."""
    L = oracle.lib()
    rows = (C.POINTER(kzg.BlstP1) * 128).from_address(settings.c.x_ext_fft_columns)
    for offset in (0, 1, 63):
        start = 4096 - 64 - 1 - offset
        x = (O.G1 * 128)()
        for i in range(63):
            L.og1_from_affine(C.byref(x[i]), C.byref(oracle_settings.g1_monomial[start - 64 * i]))
        want = (O.G1 * 128)()
        assert L.offt_g1(C.byref(oracle_settings.fs), want, x, 128, 0) == 0
        for row in (0, 1, 2, 64, 127):
            got = O.G1()
            C.memmove(C.byref(got), C.byref(rows[row][offset]), 144)
            assert L.og1_equal(C.byref(got), C.byref(want[row])) == 1, (offset, row)


def test_hybrid_fold_exact_zero_test_regression(kzg, settings, oracle, oracle_settings):
    """A sparse blob whose quotient polynomial sends the limb-parallel addition of the small-batch fold
    (k_blocksum_hybrid: four waves, each on its own chain) into its rare exact zero test.  Until round 4 that test
    ran workgroup barriers the other waves did not, and batches of 2 .. 8 such blobs got a wrong proof (about one
    small batch in 10^5; found by tools/fuzz_ckzg.py).  Every batch size against the single call and the oracle."""
    import json

    with open(os.path.join(GOLDEN, "sparse_blob_hybrid_fold.json")) as f:
        fx = json.load(f)
    blob = bytearray(BLOB)
    for i, h in fx["elements"]:
        blob[32 * i: 32 * i + 32] = bytes.fromhex(h)
    blob = bytes(blob)
    cm = bytes.fromhex(fx["commitment"])
    L = oracle.lib()
    want = C.create_string_buffer(48)
    assert L.ocompute_blob_kzg_proof(want, blob, cm, C.byref(oracle_settings)) == 0
    assert kzg.blob_to_kzg_commitment(blob, settings) == cm
    assert kzg.compute_blob_kzg_proof(blob, cm, settings) == want.raw
    for n in (1, 2, 3, 5, 8, 9, 16, 17):
        assert kzg.compute_blob_kzg_proof_batch(blob * n, cm * n, n, settings) == [want.raw] * n, n
        assert kzg.blob_to_kzg_commitment_batch(blob * n, n, settings) == [cm] * n, n

"""Verification half of the c-kzg-4844 surface on the GPU box: the reference's verify_* vectors through the C-ABI, the
G1 half of batched verification against the oracle, and the trusted-setup text fixtures of the binding test-suite."""
import ctypes as C
import gzip
import os
import random

import pytest

import oracle_ffi as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
BLOB = 131072


def unhex(s):
    return bytes.fromhex(s[2:])


@pytest.fixture(scope="module")
def settings(kzg):
    s = kzg.KZGSettings.from_file(os.path.join(GOLDEN, "trusted_setup.txt"))
    yield s
    s.close()


def test_settings_carry_the_g2_points(kzg, settings, trusted_setup_text):
    toks = trusted_setup_text.split()
    g2 = (kzg.BlstP2 * 65).from_address(settings.c.g2_values_monomial)
    for i in (0, 1, 2, 64):
        assert kzg.p2_compress(g2[i]) == bytes.fromhex(toks[2 + 4096 + i].decode())


def test_vectors_verify_kzg_proof(kzg, settings, golden):
    seen = {True: 0, False: 0, None: 0}
    for case in golden["verify_kzg_proof"]:
        args = [unhex(case[k]) for k in ("commitment", "z", "y", "proof")]
        if case["output"] is None:
            with pytest.raises(kzg.KzgAmdError):
                kzg.verify_kzg_proof(*args, settings)
        else:
            assert kzg.verify_kzg_proof(*args, settings) == case["output"], case["name"]
        seen[case["output"]] += 1
    assert seen[True] >= 30 and seen[False] >= 30 and seen[None] >= 10


def test_vectors_verify_blob_kzg_proof(kzg, settings, golden, blob_loader):
    seen = {True: 0, False: 0, None: 0}
    for case in golden["verify_blob_kzg_proof"]:
        blob = blob_loader(case["blob"])
        c, p = unhex(case["commitment"]), unhex(case["proof"])
        if case["output"] is None:
            with pytest.raises(kzg.KzgAmdError):
                kzg.verify_blob_kzg_proof(blob, c, p, settings)
        else:
            assert kzg.verify_blob_kzg_proof(blob, c, p, settings) == case["output"], case["name"]
        seen[case["output"]] += 1
    assert seen[True] >= 5 and seen[False] >= 2 and seen[None] >= 5


def test_vectors_verify_blob_kzg_proof_batch(kzg, settings, golden, blob_loader):
    seen = {True: 0, False: 0, None: 0}
    for case in golden["verify_blob_kzg_proof_batch"]:
        blobs = [blob_loader(b) for b in case["blobs"]]
        cs, ps = [unhex(c) for c in case["commitments"]], [unhex(p) for p in case["proofs"]]
        if case["output"] is None:
            with pytest.raises(kzg.KzgAmdError):
                kzg.verify_blob_kzg_proof_batch(blobs, cs, ps, settings)
        else:
            assert kzg.verify_blob_kzg_proof_batch(blobs, cs, ps, settings) == case["output"], case["name"]
        seen[case["output"]] += 1
    assert seen[True] >= 5 and seen[False] >= 2 and seen[None] >= 8


def _valid_triples(kzg, settings, golden, blob_loader):
    out = []
    for case in golden["compute_blob_kzg_proof"]:
        if case["output"] is not None:
            out.append((blob_loader(case["blob"]), unhex(case["commitment"]), unhex(case["output"])))
    return out


def test_batch_g1_half_matches_oracle(kzg, settings, golden, blob_loader, oracle, oracle_settings):
    """kzgamd_verify_(blob_)kzg_proof_batch_g1 == the oracle's restatement of verify_kzg_proof_batch up to the pairing
    (kzg/src/eip_4844.rs:328-435), and the pair it returns satisfies the pairing equation for honest proofs."""
    L = oracle.lib()
    triples = _valid_triples(kzg, settings, golden, blob_loader)
    assert len(triples) == 7
    rnd = random.Random(8)
    for n in (1, 2, 7, 20, 70):
        pick = [triples[rnd.randrange(len(triples))] for _ in range(n)]
        blobs, cs, ps = b"".join(t[0] for t in pick), b"".join(t[1] for t in pick), b"".join(t[2] for t in pick)
        zs, ys = kzg.compute_challenges_and_evaluate_batch(blobs, cs, n, settings)
        a, b = kzg.verify_kzg_proof_batch_g1(cs, b"".join(zs), b"".join(ys), ps, n, settings)
        a2, b2 = kzg.verify_blob_kzg_proof_batch_g1(blobs, cs, ps, n, settings)
        oa, ob = O.G1(), O.G1()
        assert L.overify_kzg_proof_batch_g1(C.byref(oa), C.byref(ob), cs, b"".join(zs), b"".join(ys), ps, n) == 0
        for got, want in ((a, oa), (b, ob), (a2, oa), (b2, ob)):
            g = O.G1()
            C.memmove(C.byref(g), C.byref(got), 144)
            assert L.og1_equal(C.byref(g), C.byref(want)) == 1, n
        g2 = (kzg.BlstP2 * 65).from_address(settings.c.g2_values_monomial)
        assert kzg.pairings_verify(a, g2[1], b, kzg.p2_generator())
    # a proof that does not belong: the G1 half still computes, the pairing says no
    t0 = triples[0]
    t1 = next(t for t in triples if t[2] != t0[2] and t[2][0] != 0xC0 and t0[2] != t[2])
    blobs = t0[0] + t1[0]
    cs, ps = t0[1] + t1[1], t1[2] + t0[2]
    a, b = kzg.verify_blob_kzg_proof_batch_g1(blobs, cs, ps, 2, settings)
    g2 = (kzg.BlstP2 * 65).from_address(settings.c.g2_values_monomial)
    assert not kzg.pairings_verify(a, g2[1], b, kzg.p2_generator())


def test_batch_g1_half_rejects_bad_input(kzg, settings, golden, blob_loader):
    blob, c, p = _valid_triples(kzg, settings, golden, blob_loader)[0]
    z, y = bytes(32), bytes(32)
    r_be = O.R.to_bytes(32, "big")
    # the reference's "invalid commitment" vectors: well-formed 48-byte strings that are no valid G1 element
    not_in_g1 = next(unhex(k["commitment"]) for k in golden["verify_kzg_proof"]
                     if "invalid_commitment" in k["name"] and len(unhex(k["commitment"])) == 48
                     and k["commitment"].startswith("0x8123"))
    for cs, zs, ys, ps in ((c, r_be, y, p), (c, z, r_be, p), (not_in_g1, z, y, p), (c, z, y, not_in_g1),
                           (bytes(48), z, y, p)):
        with pytest.raises(kzg.KzgAmdError):
            kzg.verify_kzg_proof_batch_g1(cs * 2, zs * 2, ys * 2, ps * 2, 2, settings)
    # n == 0: two points at infinity
    a, b = kzg.verify_kzg_proof_batch_g1(b"", b"", b"", b"", 0, settings)
    assert bytes(a) == bytes(144) and bytes(b) == bytes(144)


def test_setup_text_fixtures(kzg, golden, tmp_path):
    """kzg-bench/src/tests/c_bindings.rs:344-489: every fixture file through load_trusted_setup_file."""
    files = golden["setup_fixtures"]["files"]
    assert len(files) == 11
    for name, meta in sorted(files.items()):
        path = tmp_path / (name + ".txt")
        with gzip.open(os.path.join(GOLDEN, "setup_fixtures", name + ".txt.gz"), "rb") as f:
            path.write_bytes(f.read())
        if meta["expect"] == "ok":
            s = kzg.KZGSettings.from_file(str(path))
            blob = bytes(BLOB)
            assert kzg.blob_to_kzg_commitment(blob, s)[0] == 0xC0
            s.close()
        else:
            with pytest.raises(kzg.KzgAmdError):
                kzg.KZGSettings.from_file(str(path))


def test_load_trusted_setup_rejects_bad_g2(kzg, trusted_setup_text):
    toks = trusted_setup_text.split()
    g1l = b"".join(bytes.fromhex(t.decode()) for t in toks[2:2 + 4096])
    g2 = [bytes.fromhex(t.decode()) for t in toks[2 + 4096:2 + 4096 + 65]]
    g1m = b"".join(bytes.fromhex(t.decode()) for t in toks[2 + 4096 + 65:2 + 4096 + 65 + 4096])
    bad = bytearray(g2[3])
    ok = False
    for k in range(1, 9):   # find an x with no point on the twist
        bad[95] = g2[3][95] ^ k
        try:
            kzg.p2_uncompress(bytes(bad))
        except kzg.KzgAmdError:
            ok = True
            break
    assert ok
    with pytest.raises(kzg.KzgAmdError):
        kzg.KZGSettings.from_bytes(g1m, g1l, b"".join(g2[:3] + [bytes(bad)] + g2[4:]))
    # swapped sections (monomial points in the Lagrange slot): the reference's pairing test rejects it
    with pytest.raises(kzg.KzgAmdError):
        kzg.KZGSettings.from_bytes(g1m, g1m, b"".join(g2))

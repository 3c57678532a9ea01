"""Host-side G2 / pairing code of the product (rust-kzg_amd/csrc/host_pairing.h) on the CPU, no GPU involved:
bilinearity, the G2 wire format on the 65 setup points, the reference's Lagrange-form test on the real setup, and the
122 verify_kzg_proof vectors of the reference (check_proof_single rebuilt from the library's own primitives, with the
oracle doing the G1 side)."""
import ctypes as C
import os

import pytest

import oracle_ffi as O
from conftest import GOLDEN


def unhex(s):
    return bytes.fromhex(s[2:])


def to_p1(kzg, og1):
    p = kzg.BlstP1()
    C.memmove(C.byref(p), C.byref(og1), 144)
    return p


def g1_mul_gen(L, k):
    g, out = O.G1(), O.G1()
    L.og1_generator(C.byref(g))
    f = O.fr_from_int(k % O.R)
    L.og1_mul(C.byref(out), C.byref(g), C.byref(f))
    return out


def fr_mont(kzg, k):
    f = O.fr_from_int(k % O.R)
    out = kzg.BlstFr()
    C.memmove(C.byref(out), C.byref(f), 32)
    return out


@pytest.fixture(scope="module")
def g2_setup(kzg, trusted_setup_text):
    toks = trusted_setup_text.split()
    assert toks[0] == b"4096" and toks[1] == b"65"
    return [kzg.p2_uncompress(bytes.fromhex(t.decode())) for t in toks[2 + 4096:2 + 4096 + 65]], \
           [bytes.fromhex(t.decode()) for t in toks[2 + 4096:2 + 4096 + 65]]


def test_g2_wire_format_round_trip(kzg, g2_setup):
    pts, raw = g2_setup
    for p, b in zip(pts, raw):
        assert kzg.p2_compress(p) == b
    assert kzg.p2_compress(kzg.p2_generator()) == raw[0]         # [tau^0]G2 is the generator
    inf = bytes([0xC0]) + bytes(95)
    assert kzg.p2_compress(kzg.p2_uncompress(inf)) == inf
    for bad in (bytes(96),                                        # compression flag missing
                bytes([0xE0]) + bytes(95),                        # infinity with the sort flag
                bytes([0xC0]) + bytes(94) + b"\x01",              # infinity with a non-zero x
                bytes([0x9F]) + b"\xff" * 95):                    # x >= p
        with pytest.raises(kzg.KzgAmdError):
            kzg.p2_uncompress(bad)
    # an x whose x^3 + 4(1+u) is not a square: flip low bits of a valid point until decoding fails
    b = bytearray(raw[1])
    failures = 0
    for k in range(1, 9):
        b[95] = raw[1][95] ^ k
        try:
            kzg.p2_uncompress(bytes(b))
        except kzg.KzgAmdError:
            failures += 1
    assert failures >= 1


def test_bilinearity(kzg):
    L = O.lib()
    g2 = kzg.p2_generator()
    a, b = 0x1234567890ABCDEF1122334455667788, 0x0FEDCBA987654321DEADBEEF
    pa, pab, p1 = to_p1(kzg, g1_mul_gen(L, a)), to_p1(kzg, g1_mul_gen(L, a * b)), to_p1(kzg, g1_mul_gen(L, 1))
    qb = kzg.p2_mult(g2, fr_mont(kzg, b))
    assert kzg.pairings_verify(pa, qb, pab, g2)          # e(aG, bH) == e(abG, H)
    assert kzg.pairings_verify(pab, g2, p1, kzg.p2_mult(g2, fr_mont(kzg, a * b)))
    assert not kzg.pairings_verify(pa, qb, p1, g2)       # non-degenerate
    assert not kzg.pairings_verify(pa, g2, pab, g2)
    # additivity in G2 and infinity on either side
    q = kzg.p2_add(kzg.p2_mult(g2, fr_mont(kzg, 5)), kzg.p2_mult(g2, fr_mont(kzg, 7)))
    assert kzg.pairings_verify(p1, q, to_p1(kzg, g1_mul_gen(L, 12)), g2)
    inf1, inf2 = kzg.BlstP1(), kzg.BlstP2()
    assert kzg.pairings_verify(inf1, g2, p1, inf2)
    assert not kzg.pairings_verify(inf1, g2, p1, g2)


def test_real_setup_pairing_relations(kzg, g2_setup, oracle_settings):
    """e([tau]G1, G2) == e(G1, [tau]G2) on the mainnet setup; and the reference's Lagrange-form test
    (is_trusted_setup_in_lagrange_form, kzg/src/eip_4844.rs:1005-1020) accepts the file's Lagrange section and
    would reject its monomial section."""
    L = O.lib()
    pts, _ = g2_setup

    def aff(a):
        j = O.G1()
        L.og1_from_affine(C.byref(j), C.byref(a))
        return to_p1(kzg, j)

    m0, m1 = aff(oracle_settings.g1_monomial[0]), aff(oracle_settings.g1_monomial[1])
    assert kzg.pairings_verify(m1, pts[0], m0, pts[1])           # "is monomial form" holds for the monomial points
    l0, l1 = aff(oracle_settings.g1_lagrange_brp[0]), aff(oracle_settings.g1_lagrange_brp[2048])  # file order 0, 1
    assert not kzg.pairings_verify(l1, pts[0], l0, pts[1])


def test_reference_verify_kzg_proof_vectors(kzg, golden, g2_setup):
    """check_proof_single (blst/src/types/kzg_settings.rs:178-196): e(C - [y]G, G2) == e(proof, [tau]G2 - [z]G2)."""
    L = O.lib()
    pts, _ = g2_setup
    g2 = kzg.p2_generator()
    counts = {True: 0, False: 0, None: 0}
    for case in golden["verify_kzg_proof"]:
        c, z, y, pr = (unhex(case[k]) for k in ("commitment", "z", "y", "proof"))
        ok = None
        ca, pa = O.G1Affine(), O.G1Affine()
        zf, yf = O.Fr(), O.Fr()
        valid = len(c) == 48 and len(pr) == 48 and len(z) == 32 and len(y) == 32
        valid = valid and L.og1_uncompress(C.byref(ca), c) == 1 and L.og1_uncompress(C.byref(pa), pr) == 1
        valid = valid and L.ofr_from_be32(C.byref(zf), z) == 1 and L.ofr_from_be32(C.byref(yf), y) == 1
        if valid:
            cj, pj = O.G1(), O.G1()
            L.og1_from_affine(C.byref(cj), C.byref(ca))
            L.og1_from_affine(C.byref(pj), C.byref(pa))
            valid = (L.og1_is_inf(C.byref(cj)) or L.og1_in_subgroup(C.byref(cj))) and \
                    (L.og1_is_inf(C.byref(pj)) or L.og1_in_subgroup(C.byref(pj)))
        if valid:
            g, yg, cmy = O.G1(), O.G1(), O.G1()
            L.og1_generator(C.byref(g))
            L.og1_mul(C.byref(yg), C.byref(g), C.byref(yf))
            L.og1_neg(C.byref(yg), C.byref(yg))
            L.og1_add_or_dbl(C.byref(cmy), C.byref(cj), C.byref(yg))
            zneg = O.Fr()
            zero = O.fr_from_int(0)
            L.ofr_sub(C.byref(zneg), C.byref(zero), C.byref(zf))
            zm = kzg.BlstFr()
            C.memmove(C.byref(zm), C.byref(zneg), 32)
            s_minus_z = kzg.p2_add(pts[1], kzg.p2_mult(g2, zm))
            ok = kzg.pairings_verify(to_p1(kzg, cmy), g2, to_p1(kzg, pj), s_minus_z)
        assert ok == case["output"], case["name"]
        counts[ok] += 1
    assert counts[True] >= 30 and counts[False] >= 30 and counts[None] >= 10, counts

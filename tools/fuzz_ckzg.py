#!/usr/bin/env python3
"""Differential fuzz of the c-kzg surface and the NTT entry points against the CPU oracle: random blob shapes
(uniform, sparse, equal elements, r - 1, zero), evaluation points inside and outside the domain, batch sizes on both
sides of every internal switch (1-4 host-side paths, 5+ device paths, 128+ chunk pipeline), NTT sizes 1 .. 2^14.
Not part of the test suite:  python tools/fuzz_ckzg.py [seconds] [seed]"""
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

kzg = eb.load_pkg()
L = O.lib()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
BLOB = 131072
s = kzg.KZGSettings.from_file(eb.SETUP)
with open(eb.SETUP, "rb") as f:
    rc, os_ = O.load_settings(f.read())
assert rc == 0
roots = [O.fr_to_int(os_.fs.roots_of_unity[i]) for i in (0, 1, 2, 4095, 2048, 77)]  # 8192-th roots; even ones are in the blob domain


def make_blob():
    kind = rnd.randrange(6)
    if kind == 0:
        vals = [rnd.randrange(O.R) for _ in range(4096)]
    elif kind == 1:
        vals = [0] * 4096
        for _ in range(rnd.randrange(1, 20)):
            vals[rnd.randrange(4096)] = rnd.randrange(O.R)
    elif kind == 2:
        vals = [rnd.randrange(O.R)] * 4096
    elif kind == 3:
        vals = [O.R - 1 - rnd.randrange(3) for _ in range(4096)]
    elif kind == 4:
        vals = [0] * 4096
    else:
        vals = [rnd.randrange(1 << rnd.choice([8, 64, 200])) for _ in range(4096)]
    return b"".join(v.to_bytes(32, "big") for v in vals)


t_end = time.time() + budget
cases = 0
fss = {}
while time.time() < t_end:
    n = rnd.choice([1, 1, 2, 3, 4, 5, 8, 17, 130])
    blobs = [make_blob() for _ in range(min(n, 6))]
    blobs = [blobs[i % len(blobs)] for i in range(n)]
    flat = b"".join(blobs)
    cms = kzg.blob_to_kzg_commitment_batch(flat, n, s)
    proofs = kzg.compute_blob_kzg_proof_batch(flat, b"".join(cms), n, s)
    zs, ys = kzg.compute_challenges_and_evaluate_batch(flat, b"".join(cms), n, s)
    for b in sorted(set([0, n - 1, rnd.randrange(n)])):
        ec, ep = C.create_string_buffer(48), C.create_string_buffer(48)
        assert L.oblob_to_kzg_commitment(ec, blobs[b], C.byref(os_)) == 0
        assert cms[b] == ec.raw, ("commit", n, b, seed, cases)
        assert L.ocompute_blob_kzg_proof(ep, blobs[b], ec.raw, C.byref(os_)) == 0
        if proofs[b] != ep.raw:
            again = kzg.compute_blob_kzg_proof_batch(flat, b"".join(cms), n, s)
            single = kzg.compute_blob_kzg_proof(blobs[b], cms[b], s)
            print("blob proof mismatch: n", n, "b", b, "seed", seed, "case", cases, "second batch call right:", again[b] == ep.raw,
                  "single call right:", single == ep.raw, "got", proofs[b].hex()[:20], "want", ep.raw.hex()[:20],
                  "blob head", blobs[b][:64].hex(), "distinct blobs", len(set(blobs)), flush=True)
            raise AssertionError(("blob proof", n, b, seed, cases))
        ey = C.create_string_buffer(32)
        assert L.ocompute_kzg_proof(ep, ey, blobs[b], zs[b], C.byref(os_)) == 0
        assert ys[b] == ey.raw, ("evaluate", n, b, seed, cases)
    # the verification entry points on what was just produced (decode + membership test + two-row MSM on the GPU, pairing on
    # the host): the batch verifies, and does not with two proofs exchanged (unless they are equal)
    assert kzg.verify_blob_kzg_proof_batch(blobs, cms, proofs, s), ("verify batch", n, seed, cases)
    if n >= 2 and proofs[0] != proofs[1]:
        swapped = [proofs[1], proofs[0]] + list(proofs[2:])
        assert not kzg.verify_blob_kzg_proof_batch(blobs, cms, swapped, s), ("verify swapped", n, seed, cases)
    assert kzg.verify_kzg_proof(cms[0], zs[0], ys[0], proofs[0], s), ("verify_kzg_proof", seed, cases)
    # compute_kzg_proof at chosen points: in the domain (even powers of the 8192-th root) and outside
    z = rnd.choice([pow(roots[1], 2 * rnd.randrange(4096), O.R), rnd.randrange(O.R), 0, 1, O.R - 1])
    zb = z.to_bytes(32, "big")
    p, y = kzg.compute_kzg_proof(blobs[0], zb, s)
    ep, ey = C.create_string_buffer(48), C.create_string_buffer(32)
    assert L.ocompute_kzg_proof(ep, ey, blobs[0], zb, C.byref(os_)) == 0
    assert (p, y) == (ep.raw, ey.raw), ("kzg proof", hex(z), seed, cases)
    # NTT
    logn = rnd.randrange(0, 15)
    nn = 1 << logn
    scale = max(logn, 1) + rnd.randrange(0, 2)
    if scale not in fss:
        ofs = O.FFTSettings()
        assert L.offt_settings_new(C.byref(ofs), scale) == 0
        fss[scale] = (kzg.FFTSettings(scale), ofs)
    fs, ofs = fss[scale]
    data = O.fr_array([rnd.randrange(O.R) for _ in range(nn)])
    inv = rnd.random() < 0.5
    exp = (O.Fr * nn)()
    assert L.offt_fr(C.byref(ofs), exp, data, nn, 1 if inv else 0) == 0
    assert bytes(fs.fft_fr(data, nn, inverse=inv))[: 32 * nn] == bytes(exp), ("ntt", logn, inv, seed, cases)
    if 1 <= logn < scale:
        odds = (O.Fr * nn)()
        assert L.odas_fft_extension(C.byref(ofs), odds, data, nn) == 0
        assert bytes(fs.das_fft_extension(data, nn))[: 32 * nn] == bytes(odds), ("das", logn, seed, cases)
    cases += 1
print("fuzz ok:", cases, "cases, seed", seed, "library", os.path.basename(kzg.LIB_PATH))

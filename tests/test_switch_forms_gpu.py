"""Forms of the c-kzg paths that a settings object chooses by tuning key.  The keys are read ONCE, when the
object is created (KzgAmdConfig.tuning), so every form gets its own object here — with 8 GB tables (KzgAmdConfig.
table_budget_bytes), and in a module of its own so that the 137 + 43 + 77 GB of another module's default object are
released before these are built."""
import hashlib
import os
import random

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
BLOB = 131072
SETUP = os.path.join(GOLDEN, "trusted_setup.txt")


@pytest.fixture(scope="module")
def forms(kzg):
    """settings objects: default form / FK20 forced / direct cell proofs forced / device SHA-256; each with 8 GB per
    table through its KzgAmdConfig — four objects side by side on one GPU, no environment variable involved"""
    made = {}
    try:
        for name, tuning in (("default", None), ("fk20", {"fk20": 1}), ("direct", {"fk20": 0}), ("device_sha", {"device_sha": 1})):
            made[name] = kzg.KZGSettings.from_file(SETUP, kzg.make_config(table_budget_gb=8, tuning=tuning))
        yield made
    finally:
        for s in made.values():
            s.close()


def test_cell_proofs_fk20_matches_vectors_and_the_direct_form(kzg, forms, golden, blob_loader):
    """The batch algorithm for cell proofs (FK20, kzg/src/das.rs:630-696: Toeplitz vectors, 64 transforms of 128,
    128 MSMs of 64 points over x_ext_fft_columns, two G1 transforms) against the reference's vectors and against the
    direct form (one fixed-base MSM per cell), which is what single blobs use."""
    cases = [c for c in golden["compute_cells_and_kzg_proofs"] if c["output"] is not None]
    blobs = b"".join(blob_loader(c["blob"]) for c in cases)
    cells, proofs = kzg.compute_cells_and_kzg_proofs_batch(blobs, len(cases), forms["fk20"])
    for k, c in enumerate(cases):
        assert hashlib.sha256(proofs[k * 6144:(k + 1) * 6144]).hexdigest() == c["output"]["proofs_sha256"], c["name"]
        assert hashlib.sha256(cells[k * 262144:(k + 1) * 262144]).hexdigest() == c["output"]["cells_sha256"], c["name"]
    assert kzg.compute_cells_and_kzg_proofs_batch(blobs, len(cases), forms["direct"]) == (cells, proofs)
    # a single blob through FK20 as well
    _, p1 = kzg.compute_cells_and_kzg_proofs(blob_loader(cases[3]["blob"]), forms["fk20"], want_cells=False)
    assert p1 == proofs[3 * 6144:4 * 6144]
    # random blobs, batch of 70 (above the size from which FK20 is the default): default == FK20 == direct
    rnd = random.Random(720)
    n = 70
    rb = bytearray(rnd.randbytes(n * BLOB))
    for i in range(0, n * BLOB, 32):
        rb[i] = 0
    rb[5 * BLOB:6 * BLOB] = bytes(BLOB)  # the zero polynomial: every proof is the point at infinity
    rb = bytes(rb)
    _, p_default = kzg.compute_cells_and_kzg_proofs_batch(rb, n, forms["default"])
    _, p_fk20 = kzg.compute_cells_and_kzg_proofs_batch(rb, n, forms["fk20"])
    _, p_direct = kzg.compute_cells_and_kzg_proofs_batch(rb, n, forms["direct"])
    assert p_default == p_direct and p_fk20 == p_direct
    assert p_direct[5 * 6144:5 * 6144 + 48] == b"\xc0" + bytes(47)


def test_proof_batch_with_device_sha(kzg, forms):
    """Tuning key device_sha=1 (the Fiat-Shamir hashes of a host-buffer batch on the GPU, no host threads): the same
    proofs, challenges and evaluations as the default object, and the same rejections."""
    rnd = random.Random(257)
    n = 256
    blobs = bytearray(rnd.randbytes(n * BLOB))
    for i in range(0, n * BLOB, 32):
        blobs[i] = 0
    blobs[7 * BLOB:8 * BLOB] = bytes(BLOB)
    blobs = bytes(blobs)
    s1, s2 = forms["default"], forms["device_sha"]
    cms = kzg.blob_to_kzg_commitment_batch(blobs, n, s1)
    assert kzg.blob_to_kzg_commitment_batch(blobs, n, s2) == cms
    proofs = kzg.compute_blob_kzg_proof_batch(blobs, b"".join(cms), n, s1)
    assert kzg.compute_blob_kzg_proof_batch(blobs, b"".join(cms), n, s2) == proofs
    assert kzg.compute_challenges_and_evaluate_batch(blobs, b"".join(cms), n, s2) == \
        kzg.compute_challenges_and_evaluate_batch(blobs, b"".join(cms), n, s1)
    assert kzg.verify_blob_kzg_proof_batch([blobs[i * BLOB:(i + 1) * BLOB] for i in range(0, n, 37)], cms[::37], proofs[::37], s2)
    bad = bytearray(blobs)
    bad[200 * BLOB + 64:200 * BLOB + 96] = b"\xff" * 32
    with pytest.raises(kzg.KzgAmdError):
        kzg.compute_blob_kzg_proof_batch(bytes(bad), b"".join(cms), n, s2)
    badc = list(cms)
    badc[130] = b"\x9f" + b"\xff" * 47
    with pytest.raises(kzg.KzgAmdError):
        kzg.compute_blob_kzg_proof_batch(blobs, b"".join(badc), n, s2)
    assert kzg.compute_blob_kzg_proof_batch(blobs, b"".join(cms), n, s2) == proofs  # and recovers afterwards
    # a mid-sized batch (one workgroup per blob in k_quotient, a single chunk)
    m = 40
    assert kzg.compute_blob_kzg_proof_batch(blobs[:m * BLOB], b"".join(cms[:m]), m, s2) == proofs[:m]


def test_one_lane_stage_at_production_scale(kzg, forms):
    """The one-lane chain (k_g1_stage_chain<1>) is the form FK20 uses above 256 blobs and fft_g1 above 2^15 points:
    1024 blobs are 131 072 half-butterflies per stage, 1.4 million chains of ~170 point operations in one call — enough
    for a 1e-5-per-chain event to show several times, where the suite only ever ran this form at 64 points.  (It is a
    test of scale, not of round 4's g1::dbl pad bug: measured on the GPU, a library with the old 8p pad passes it — the Y
    of a stage input is the output of a two-product reduction R*(Q - X3) + (8p - S)*PPP, whose (8p - S)*PPP / 2^392
    term keeps it above 2^364, so tiny values reach the doubling only through raw inputs.  The directed host test
    tests/test_host_cpu.py::test_dbl_of_a_negated_point_with_a_tiny_y is what pins that bug.)
    Every proof against the same blobs in batches of 64 (the four-lane form: other kernels, other formulas' schedule),
    which the vectors and the direct form pin."""
    import ctypes as C

    import numpy as np

    n = 1024
    rng = np.random.default_rng(1024)
    arr = rng.integers(0, 256, size=(n * 4096, 32), dtype=np.uint8)
    arr[:, 0] = 0
    blobs = arr.tobytes()
    L = kzg.lib()
    s = forms["default"]
    big = C.create_string_buffer(n * 6144)
    assert L.kzgamd_compute_cells_and_kzg_proofs_batch(None, big, blobs, n, C.byref(s.c)) == 0
    small = C.create_string_buffer(64 * 6144)
    bad = []
    for k in range(0, n, 64):
        assert L.kzgamd_compute_cells_and_kzg_proofs_batch(None, small, blobs[k * BLOB:(k + 64) * BLOB], 64, C.byref(s.c)) == 0
        if small.raw != big.raw[k * 6144:(k + 64) * 6144]:
            bad += [k + j for j in range(64) if small.raw[j * 6144:(j + 1) * 6144] != big.raw[(k + j) * 6144:(k + j + 1) * 6144]]
    assert not bad, bad
    # and two of those batches through the direct form (one fixed-base MSM per cell; what the reference's vectors pin)
    d = forms["direct"]
    for k in (0, 960):
        assert L.kzgamd_compute_cells_and_kzg_proofs_batch(None, small, blobs[k * BLOB:(k + 64) * BLOB], 64, C.byref(d.c)) == 0
        assert small.raw == big.raw[k * 6144:(k + 64) * 6144], k

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

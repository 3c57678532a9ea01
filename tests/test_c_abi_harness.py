"""A native C consumer of the boundary (tests/c_abi_harness.c): compiled with `gcc -std=c99 -pedantic -Werror` against
include/kzg_mi355x.h and libkzg_mi355x.so, it replays the reference's vectors (tests/golden) and oracle-made cases for
the plug-in symbols through the header's own prototypes and struct layouts — the counterpart of the reference linking
its staticlib into the c-kzg-4844 bindings (run-c-kzg-4844-tests.sh:36-57).  The header, not the ctypes mirror, is
what these vectors pin."""
import ctypes as C
import gzip
import json
import os
import random
import struct
import subprocess

import pytest

import oracle_ffi as O
from conftest import GOLDEN, ROOT, load_blob

BLOB = 131072
CELL = 2048
(OP_COMMIT, OP_PROOF, OP_BLOB_PROOF, OP_VERIFY, OP_VERIFY_BLOB, OP_VERIFY_BATCH, OP_CELLS, OP_RECOVER, OP_VERIFY_CELLS,
 OP_CELL_CHALLENGE, OP_NTT, OP_DAS, OP_MSM, OP_FFT_G1, OP_CHALLENGE, OP_LOAD_BYTES, OP_COMMIT_BATCH, OP_PROOF_BATCH,
 OP_G1_SUM, OP_MATRIX) = range(1, 21)


def unhex(s):
    return bytes.fromhex(s[2:])


def build_harness(tmp_path, kzg):
    exe = str(tmp_path / "c_abi_harness")
    libdir = os.path.dirname(kzg.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_harness.c"), "-L" + libdir, "-l:" + os.path.basename(kzg.LIB_PATH),
                           "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_header_is_c99_and_layout_matches_the_reference(tmp_path, kzg):
    """No GPU needed: the header compiles as strict C99 inside a real consumer, the consumer links against every
    symbol it uses, and the struct layouts are the reference's (kzg/src/eth/c_bindings.rs:16-113, 429-474)."""
    kzg.lib()  # the library must be built
    exe = build_harness(tmp_path, kzg)
    p = subprocess.run([exe, "layout"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert p.returncode == 0, p.stdout.decode()
    assert b"layout ok" in p.stdout


class Records:
    def __init__(self, path):
        self.f = open(path, "wb")
        self.counts = {}

    def add(self, op, *fields):
        self.f.write(struct.pack("<II", op, len(fields)))
        for fld in fields:
            fld = bytes(fld)
            self.f.write(struct.pack("<Q", len(fld)))
            self.f.write(fld)
        self.counts[op] = self.counts.get(op, 0) + 1

    def close(self):
        self.f.close()


def u64s(vals):
    return struct.pack("<%dQ" % len(vals), *vals)


def write_records(path, oracle_settings):
    """Every case the fixed-size C signatures can express; cases whose byte strings have the wrong length are rejected by
    the reference's bindings before the call and are skipped here."""
    L = O.lib()
    with open(os.path.join(GOLDEN, "kzg_mainnet.json")) as f:
        g = json.load(f)
    with open(os.path.join(GOLDEN, "kzg_mainnet_7594.json")) as f:
        g7 = json.load(f)
    with gzip.open(os.path.join(GOLDEN, g7["cells_file"]), "rb") as f:
        raw = f.read()
    cellv = [raw[i: i + CELL] for i in range(0, len(raw), CELL)]
    r = Records(path)

    def sized(*pairs):
        return all(len(b) == n for b, n in pairs)

    # OP_LOAD_BYTES first: the second settings object the *_multi calls use
    with open(os.path.join(GOLDEN, "trusted_setup.txt"), "rb") as f:
        toks = f.read().split()
    n1, n2 = int(toks[0]), int(toks[1])
    lag = b"".join(bytes.fromhex(t.decode()) for t in toks[2: 2 + n1])
    g2m = b"".join(bytes.fromhex(t.decode()) for t in toks[2 + n1: 2 + n1 + n2])
    mono = b"".join(bytes.fromhex(t.decode()) for t in toks[2 + n1 + n2: 2 + 2 * n1 + n2])
    first = next(c for c in g["blob_to_kzg_commitment"] if c["output"] is not None)
    r.add(OP_LOAD_BYTES, mono, lag, g2m, load_blob(first["blob"]), unhex(first["output"]))

    for c in g["blob_to_kzg_commitment"]:
        blob = load_blob(c["blob"])
        if sized((blob, BLOB)):
            r.add(OP_COMMIT, blob, unhex(c["output"]) if c["output"] else b"")
    for c in g["compute_kzg_proof"]:
        blob, z = load_blob(c["blob"]), unhex(c["z"])
        if sized((blob, BLOB), (z, 32)):
            r.add(OP_PROOF, blob, z, unhex(c["output"][0]) + unhex(c["output"][1]) if c["output"] else b"")
    for c in g["compute_blob_kzg_proof"]:
        blob, cm = load_blob(c["blob"]), unhex(c["commitment"])
        if sized((blob, BLOB), (cm, 48)):
            r.add(OP_BLOB_PROOF, blob, cm, unhex(c["output"]) if c["output"] else b"")
    for c in g["compute_challenge"]:
        r.add(OP_CHALLENGE, load_blob(c["blob"]), unhex(c["commitment"]), unhex(c["output"]))
    verdict = lambda o: b"" if o is None else bytes([1 if o else 0])  # noqa: E731
    for c in g["verify_kzg_proof"]:
        a = [unhex(c[k]) for k in ("commitment", "z", "y", "proof")]
        if sized((a[0], 48), (a[1], 32), (a[2], 32), (a[3], 48)):
            r.add(OP_VERIFY, *a, verdict(c["output"]))
    for c in g["verify_blob_kzg_proof"]:
        blob, cm, pf = load_blob(c["blob"]), unhex(c["commitment"]), unhex(c["proof"])
        if sized((blob, BLOB), (cm, 48), (pf, 48)):
            r.add(OP_VERIFY_BLOB, blob, cm, pf, verdict(c["output"]))
    for c in g["verify_blob_kzg_proof_batch"]:
        blobs = [load_blob(b) for b in c["blobs"]]
        cs, ps = [unhex(x) for x in c["commitments"]], [unhex(x) for x in c["proofs"]]
        if len(cs) == len(blobs) == len(ps) and all(len(b) == BLOB for b in blobs) and all(len(x) == 48 for x in cs + ps):
            r.add(OP_VERIFY_BATCH, b"".join(blobs), b"".join(cs), b"".join(ps), verdict(c["output"]))
    for c in g["compute_cells_and_kzg_proofs"]:
        blob = load_blob(c["blob"])
        if not sized((blob, BLOB)):
            continue
        e = c["output"]
        r.add(OP_CELLS, blob, b"" if e is None else bytes.fromhex(e["cells_sha256"]) + bytes.fromhex(e["proofs_sha256"]) +
              unhex(e["proof0"]) + unhex(e["proof127"]))

    def cells_of(refs):
        out, ok = [], True
        for ref in refs:
            if isinstance(ref, dict):
                out.append(unhex(ref["hex"]))
                ok = ok and len(out[-1]) == CELL
            else:
                out.append(cellv[ref])
        return b"".join(out), ok

    for c in g7["recover_cells_and_kzg_proofs"]:
        cells, ok = cells_of(c["cells"])
        if not ok or len(c["cell_indices"]) != len(c["cells"]):
            continue
        e = c["output"]
        r.add(OP_RECOVER, u64s(c["cell_indices"]), cells,
              b"" if e is None else cells_of(e["cells"])[0] + bytes.fromhex(e["proofs_sha256"]))
    for c in g7["verify_cell_kzg_proof_batch"]:
        cells, ok = cells_of(c["cells"])
        cs, ps = [unhex(x) for x in c["commitments"]], [unhex(x) for x in c["proofs"]]
        n = len(c["cell_indices"])
        if not (ok and len(cs) == n and len(ps) == n and len(c["cells"]) == n and all(len(x) == 48 for x in cs + ps)):
            continue
        r.add(OP_VERIFY_CELLS, b"".join(cs), u64s(c["cell_indices"]), cells, b"".join(ps), verdict(c["output"]))
    for c in g7["compute_verify_cell_kzg_proof_batch_challenge"]:
        cells, ok = cells_of(c["cells"])
        assert ok
        r.add(OP_CELL_CHALLENGE, b"".join(unhex(x) for x in c["commitments"]), u64s(c["commitment_indices"]),
              u64s(c["cell_indices"]), cells, b"".join(unhex(x) for x in c["proofs"]), unhex(c["output"]))

    # plug-in symbols: expectations from the oracle
    rnd = random.Random(2024)
    fs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(fs), 16) == 0
    for n in (1, 2, 8, 128, 4096, 8192):
        for inverse in (0, 1):
            a = O.fr_array([rnd.randrange(O.R) for _ in range(n)])
            out = (O.Fr * n)()
            assert L.offt_fr(C.byref(fs), out, a, n, inverse) == 0
            r.add(OP_NTT, bytes([inverse]), bytes(a), bytes(out))
    r.add(OP_NTT, b"\0", bytes(O.fr_array([1, 2, 3])), b"")  # not a power of two: the reference's error
    for n in (1, 4, 64, 2048):
        a = O.fr_array([rnd.randrange(O.R) for _ in range(n)])
        out = (O.Fr * n)()
        assert L.odas_fft_extension(C.byref(fs), out, a, n) == 0
        r.add(OP_DAS, bytes(a), bytes(out))
    gen = O.G1()
    L.og1_generator(C.byref(gen))

    def points(n, seed):
        rr = random.Random(seed)
        jac = (O.G1 * n)()
        aff = (O.G1Affine * n)()
        acc = O.G1()
        k0 = O.fr_from_int(rr.randrange(1, O.R))
        L.og1_mul(C.byref(acc), C.byref(gen), C.byref(k0))
        for i in range(n):
            C.memmove(C.byref(jac[i]), C.byref(acc), 144)
            L.og1_to_affine(C.byref(aff[i]), C.byref(acc))
            L.og1_add_or_dbl(C.byref(acc), C.byref(acc), C.byref(gen))
        return jac, aff

    def compressed(p):
        b = C.create_string_buffer(48)
        L.og1_compress(b, C.byref(p))
        return b.raw

    for n in (1, 7, 8, 300, 5000):
        jac, aff = points(n, n)
        sc = O.fr_array([rnd.randrange(O.R) for _ in range(n)])
        want = O.G1()
        L.omsm_affine(C.byref(want), aff, sc, n)
        r.add(OP_MSM, bytes(aff), bytes(sc), compressed(want))
    for n in (4, 16):
        jac, _ = points(n, 100 + n)
        for inverse in (0, 1):
            out = (O.G1 * n)()
            assert L.offt_g1(C.byref(fs), out, jac, n, inverse) == 0
            r.add(OP_FFT_G1, bytes([inverse]), bytes(jac), b"".join(compressed(out[i]) for i in range(n)))
    jac, _ = points(9, 77)
    total = O.G1()
    C.memmove(C.byref(total), C.byref(jac[0]), 144)
    for i in range(1, 9):
        L.og1_add_or_dbl(C.byref(total), C.byref(total), C.byref(jac[i]))
    r.add(OP_G1_SUM, bytes(jac), compressed(total))

    # matrix handle: 64 rows of 64 of the setup's Lagrange points, two scalar matrices (short, zero and full-size scalars)
    rows, cols, nmat = 64, 64, 2
    mpts = oracle_settings.g1_lagrange_brp
    vals = [rnd.randrange(O.R) for _ in range(nmat * rows * cols)]
    vals[0], vals[1], vals[cols] = 0, O.R - 1, 1
    vals[5 * cols:6 * cols] = [0] * cols  # a row that sums to infinity
    msc = O.fr_array(vals)
    sums = b""
    for m in range(nmat):
        for rr_ in range(rows):
            want = O.G1()
            base = (O.G1Affine * cols).from_address(C.addressof(mpts.contents) + rr_ * cols * 96)
            sub = (O.Fr * cols).from_buffer(msc, (m * rows + rr_) * cols * 32)
            L.omsm_affine(C.byref(want), base, sub, cols)
            sums += compressed(want)
    r.add(OP_MATRIX, C.string_at(mpts, rows * cols * 96), u64s([rows, cols, nmat]), bytes(msc), sums)

    # batch forms (single settings object, and the in-library multi-GPU form over two objects)
    blobs = []
    for _ in range(21):
        b = bytearray(rnd.randbytes(BLOB))
        for i in range(0, BLOB, 32):
            b[i] = 0
        blobs.append(bytes(b))
    cms, prs = [], []
    for b in blobs:
        o = C.create_string_buffer(48)
        assert L.oblob_to_kzg_commitment(o, b, C.byref(oracle_settings)) == 0
        cms.append(o.raw)
        p = C.create_string_buffer(48)
        assert L.ocompute_blob_kzg_proof(p, b, o.raw, C.byref(oracle_settings)) == 0
        prs.append(p.raw)
    for n in (21, 3):
        r.add(OP_COMMIT_BATCH, b"".join(blobs[:n]), b"".join(cms[:n]))
        r.add(OP_PROOF_BATCH, b"".join(blobs[:n]), b"".join(cms[:n]), b"".join(prs[:n]))
    r.close()
    return r.counts


@pytest.mark.gpu
def test_c_harness_replays_the_vectors(tmp_path, kzg, oracle_settings):
    exe = build_harness(tmp_path, kzg)
    rec = str(tmp_path / "records.bin")
    counts = write_records(rec, oracle_settings)
    # what the fixed-size signatures can express of the reference's vectors (the rest is rejected by its bindings)
    assert (counts[OP_COMMIT], counts[OP_PROOF], counts[OP_BLOB_PROOF], counts[OP_CHALLENGE]) == (9, 48, 11, 9)
    assert (counts[OP_VERIFY], counts[OP_VERIFY_BLOB], counts[OP_VERIFY_BATCH]) == (114, 23, 15)
    assert (counts[OP_CELLS], counts[OP_RECOVER], counts[OP_VERIFY_CELLS], counts[OP_CELL_CHALLENGE]) == (9, 14, 22, 10)
    # (its two settings objects live side by side on the one GPU: the harness gives each 40 GB per table through KzgAmdConfig)
    p = subprocess.run([exe, "run", os.path.join(GOLDEN, "trusted_setup.txt"), rec], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=1500)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-4000:]
    assert "0 failures" in out
    for op in counts:
        assert "op %d: " % op in out and ("op %d: 0 ok" % op) not in out, out[-3000:]

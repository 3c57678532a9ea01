#!/usr/bin/env python3
"""Differential fuzz of the G1 transforms (fftg1.hip): fft_g1 / kzgamd_fft_g1_batch against the CPU oracle on random
sizes, batches, directions and inputs (random points, infinities, repeats, negations), under every stage form
(tuning keys g1_{wide,quad,pair}_max forced per handle through KzgAmdConfig); and the FK20 cell proofs of batches of every form against the
single-blob entry point (the direct form: one fixed-base MSM per cell).
Not part of the test suite:  python tools/fuzz_g1.py [seconds] [seed]"""
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
FORMS = {"by_size": {}, "wave": {"g1_wide_max": 100000000},
         "four": {"g1_wide_max": 0, "g1_quad_max": 100000000},
         "two": {"g1_wide_max": 0, "g1_quad_max": 0, "g1_pair_max": 100000000},
         "one": {"g1_wide_max": 0, "g1_quad_max": 0, "g1_pair_max": 0}}


def with_form(name):
    """the KzgAmdConfig that forces the form (None: by size)"""
    return kzg.make_config(tuning=FORMS[name]) if FORMS[name] else None


g = O.G1()
L.og1_generator(C.byref(g))
pool = []  # random multiples of the generator
for _ in range(64):
    p = O.G1()
    k = O.fr_from_int(rnd.randrange(1, O.R))
    L.og1_mul(C.byref(p), C.byref(g), C.byref(k))
    pool.append(p)
minus_one = O.fr_from_int(O.R - 1)


def compressed(arr, n):
    out = []
    for i in range(n):
        buf = C.create_string_buffer(48)
        q = O.G1()
        C.memmove(C.byref(q), C.byref(arr[i]), 144)
        L.og1_compress(buf, C.byref(q))
        out.append(buf.raw)
    return out


def random_input(n):
    data = (O.G1 * n)()
    shape = rnd.choice(["random", "few", "inf_some", "all_inf", "pairs", "equal"])
    for i in range(n):
        if shape == "random":
            data[i] = rnd.choice(pool)
        elif shape == "few":
            data[i] = pool[rnd.randrange(3)]
        elif shape == "inf_some":
            data[i] = O.G1() if rnd.random() < 0.4 else rnd.choice(pool)
        elif shape == "all_inf":
            data[i] = O.G1()
        elif shape == "pairs":
            p = pool[(i // 2) % len(pool)]
            if i % 2:
                q = O.G1()
                L.og1_mul(C.byref(q), C.byref(p), C.byref(minus_one))
                data[i] = q
            else:
                data[i] = p
        else:
            data[i] = pool[7]
    return data, shape


cases = bad = 0
t_end = time.time() + budget * 0.6
while time.time() < t_end:
    form = rnd.choice(list(FORMS))
    cfg = with_form(form)
    logn = rnd.choice([0, 1, 2, 3, 4, 5, 6, 7, 8])
    n = 1 << logn
    nbatch = rnd.choice([1, 1, 2, 3, 5])
    scale = rnd.choice([max(logn, 1), max(logn, 1) + 2])
    fs = kzg.FFTSettings(scale, cfg)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), scale) == 0
    inverse = rnd.random() < 0.5
    data, shape = random_input(n * nbatch)
    got = fs.fft_g1(data, n, inverse=inverse, nbatch=nbatch)
    cg = compressed(got, n * nbatch)
    for b in range(nbatch):
        part = (O.G1 * n)()
        C.memmove(part, C.byref(data, b * n * 144), n * 144)
        exp = (O.G1 * n)()
        assert L.offt_g1(C.byref(ofs), exp, part, n, 1 if inverse else 0) == 0
        if compressed(exp, n) != cg[b * n:(b + 1) * n]:
            bad += 1
            print("MISMATCH fft_g1", form, n, nbatch, inverse, shape, flush=True)
    cases += 1
    fs.close()
    L.offt_settings_free(C.byref(ofs))
print("fft_g1: %d cases, %d mismatches" % (cases, bad), flush=True)

# FK20 batches under every form against the single-blob entry point
BLOB = 131072
fcases = fbad = 0
t_end = time.time() + budget * 0.4
first = True
while time.time() < t_end or first:
    first = False
    form = rnd.choice(list(FORMS))
    s = kzg.KZGSettings.from_file(eb.SETUP, with_form(form))
    n = rnd.choice([16, 17, 33, 64, 70, 130, 200])
    blobs = bytearray(rnd.randbytes(n * BLOB))
    for i in range(0, n * BLOB, 32):
        blobs[i] = 0
    kind = rnd.choice(["random", "zero_some", "sparse"])
    if kind == "zero_some":
        for b in rnd.sample(range(n), 3):
            blobs[b * BLOB:(b + 1) * BLOB] = bytes(BLOB)
    elif kind == "sparse":
        for b in rnd.sample(range(n), 3):
            blobs[b * BLOB:(b + 1) * BLOB] = bytes(BLOB - 32) + (5).to_bytes(32, "big")
    blobs = bytes(blobs)
    cells, proofs = kzg.compute_cells_and_kzg_proofs_batch(blobs, n, s)
    for b in rnd.sample(range(n), 4):
        c1, p1 = kzg.compute_cells_and_kzg_proofs(blobs[b * BLOB:(b + 1) * BLOB], s)
        if cells[b * 262144:(b + 1) * 262144] != c1 or proofs[b * 6144:(b + 1) * 6144] != p1:
            fbad += 1
            print("MISMATCH fk20", form, n, kind, b, flush=True)
    fcases += 1
    s.close()
print("fk20: %d batches, %d mismatches" % (fcases, fbad), flush=True)
if bad or fbad:
    sys.exit(1)
print("fuzz ok:", cases + fcases, "cases, seed", seed, "library", os.path.basename(kzg.LIB_PATH))

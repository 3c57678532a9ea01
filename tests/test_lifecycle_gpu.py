"""Lifecycle and soak of the settings object (B3) next to B1 / B2 handles.

A drop-in that owns up to 257 GB of HBM (three fixed-base tables, 15 lanes, page-locked staging) has to give all of it
back: the reference's contract is load_trusted_setup -> use -> free_trusted_setup, as often as the caller likes
(kzg-bench/src/tests/c_bindings.rs:490-544 loads and frees a settings object per test; blst/src/eip_4844.rs:111-161).

* `test_load_use_free_returns_hbm_and_host_memory`: hipMemGetInfo and the process's resident / locked host memory are
  recorded, then 20 x (load -> commitment, proof, cells of a batch and of one blob = all three tables -> a 16-thread
  burst of single-blob calls = lanes and coalescing queues -> free); after EVERY cycle the free HBM is back within
  a few MB of the baseline and the host memory does not grow.  Two of the cycles take the default table budget (the
  full 137 + 43 + 77 GB), the others 6 GB per table so that the test stays inside its minute.
* `test_soak_mixed_callers_with_a_second_object_coming_and_going`: threads mixing B1 (mult_pippenger_prepared), B2
  (ntt_fr, das_fft_extension) and c-kzg calls (commitment, proof, verification, cells) on shared handles, every result
  against values computed up front (oracle-checked), while one more thread loads, uses and frees a SECOND settings
  object the whole time.  8 s by default; KZGAMD_SOAK_SECONDS (or KZGAMD_FUZZ_SCALE >= 2: three minutes, 32 threads) for
  the long form (`profiles/r06_soak.log`)."""
import ctypes as C
import os
import random
import threading
import time

import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup.txt")
MB = 1 << 20


def _host_kb():
    """resident, locked and pinned host memory of this process in KiB (/proc/self/status)"""
    out = {}
    with open("/proc/self/status") as f:
        for line in f:
            k, _, v = line.partition(":")
            if k in ("VmRSS", "VmLck", "VmPin"):
                out[k] = int(v.split()[0])
    return out


def _blobs(seed, n):
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        b = bytearray(rnd.randbytes(131072))
        for i in range(0, 131072, 32):
            b[i] = 0
        out.append(bytes(b))
    return out


def _fr_bulk(vals):
    arr = (O.Fr * len(vals))()
    raw = b"".join(((v << 256) % O.R).to_bytes(32, "little") for v in vals)
    C.memmove(arr, raw, len(raw))
    return arr


def _compressed(L, p):
    buf = C.create_string_buffer(48)
    g = O.G1()
    C.memmove(C.byref(g), C.byref(p), 144)
    L.og1_compress(buf, C.byref(g))
    return buf.raw


def test_load_use_free_returns_hbm_and_host_memory(kzg, oracle, oracle_settings):
    import torch

    L = oracle.lib()
    blobs = _blobs(61, 20)
    want_c, want_p = [], []
    for b in blobs[:4]:
        o = C.create_string_buffer(48)
        assert L.oblob_to_kzg_commitment(o, b, C.byref(oracle_settings)) == 0
        p = C.create_string_buffer(48)
        assert L.ocompute_blob_kzg_proof(p, b, o.raw, C.byref(oracle_settings)) == 0
        want_c.append(o.raw)
        want_p.append(p.raw)

    def cycle(config, check):
        s = kzg.KZGSettings.from_file(SETUP, config)
        try:
            # table 1 (Lagrange setup): commitments and proofs, single calls and a batch
            c0 = kzg.blob_to_kzg_commitment(blobs[0], s)
            p0 = kzg.compute_blob_kzg_proof(blobs[0], c0, s)
            cs = kzg.blob_to_kzg_commitment_batch(b"".join(blobs), len(blobs), s)
            ps = kzg.compute_blob_kzg_proof_batch(b"".join(blobs[:4]), b"".join(cs[:4]), 4, s)
            # table 2 (FK20, batches of >= 3 blobs) and table 3 (monomial setup, the direct form of one blob)
            cells, proofs = kzg.compute_cells_and_kzg_proofs_batch(b"".join(blobs[:4]), 4, s)
            c1, p1 = kzg.compute_cells_and_kzg_proofs(blobs[0], s)
            assert cells[:262144] == c1 and proofs[:6144] == p1
            if check:
                assert [c0] + cs[1:4] == want_c and cs[0] == c0
                assert ps == want_p and p0 == want_p[0]
            # lanes, leaders and the coalescing queues: 16 threads of single-blob calls on the one object
            errs = []

            def work(t):
                try:
                    assert kzg.blob_to_kzg_commitment(blobs[t], s) == cs[t]
                    if t < 4:
                        assert kzg.compute_blob_kzg_proof(blobs[t], cs[t], s) == ps[t]
                except Exception as e:  # noqa: BLE001
                    errs.append((t, repr(e)))

            ts = [threading.Thread(target=work, args=(t,)) for t in range(16)]
            for th in ts:
                th.start()
            for th in ts:
                th.join()
            assert errs == []
            return kzg.lib().kzgamd_settings_table_info is not None
        finally:
            s.close()

    torch.cuda.synchronize()
    small = kzg.make_config(table_budget_gb=6)
    cycle(small, True)  # the runtime's own one-off allocations (code objects, queues, signal pools) happen here
    torch.cuda.synchronize()
    base_free, total = torch.cuda.mem_get_info(0)
    base_host = _host_kb()
    log = []
    t0 = time.time()
    for k in range(20):
        big = k in (3, 11)  # the default budget: every table at its full size (137 + 43 + 77 GB when the HBM is free)
        cycle(None if big else small, k < 2 or big)
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info(0)
        host = _host_kb()
        log.append((k, big, (base_free - free) / MB, {key: host[key] - base_host.get(key, 0) for key in host}))
        # HBM: everything a settings object took is back (the tolerance is the allocator's granularity, not a leak rate:
        # it does not grow with k)
        assert base_free - free <= 8 * MB, log
        # host: page-locked staging (VmLck / VmPin where the driver accounts it there) and resident memory
        assert host.get("VmLck", 0) - base_host.get("VmLck", 0) <= 4096, log
        assert host.get("VmPin", 0) - base_host.get("VmPin", 0) <= 4096, log
        assert host["VmRSS"] - base_host["VmRSS"] <= 96 * 1024, log  # KiB; no growth per cycle (checked below)
    # no per-cycle growth: the second half of the run sits where the first half did
    first = max(e[3]["VmRSS"] for e in log[:10])
    last = max(e[3]["VmRSS"] for e in log[10:])
    assert last - first <= 16 * 1024, log
    print("lifecycle: 20 cycles in %.1f s; HBM delta MB per cycle: %s; RSS delta KiB: %s"
          % (time.time() - t0, ["%.1f" % e[2] for e in log], [e[3]["VmRSS"] for e in log]))
    # free is idempotent and a freed object is refused, not dereferenced
    s = kzg.KZGSettings.from_file(SETUP, small)
    s.close()
    s.close()
    with pytest.raises(kzg.KzgAmdError):
        kzg.blob_to_kzg_commitment(blobs[0], s)


def _soak_seconds():
    if os.environ.get("KZGAMD_SOAK_SECONDS"):
        return float(os.environ["KZGAMD_SOAK_SECONDS"])
    try:
        if float(os.environ.get("KZGAMD_FUZZ_SCALE", "1")) >= 2:
            return 180.0
    except ValueError:
        pass
    return 8.0


def test_soak_mixed_callers_with_a_second_object_coming_and_going(kzg, oracle, oracle_settings):
    L = oracle.lib()
    seconds = _soak_seconds()
    nthreads = 32 if seconds >= 60 else 12
    # shared handles: one settings object (40 GB per table so that a second object fits beside it), one prepared MSM
    # handle over the setup, one NTT handle
    s = kzg.KZGSettings.from_file(SETUP, kzg.make_config(table_budget_gb=40))
    pts = oracle_settings.g1_lagrange_brp
    h = kzg.prepare_multi_scalar_mult(pts, 4096, kzg.make_config(table_budget_gb=8))
    fs = kzg.FFTSettings(13)
    ofs = O.FFTSettings()
    assert L.offt_settings_new(C.byref(ofs), 13) == 0
    # expected values, computed once (oracle)
    blobs = _blobs(97, 6)
    want_c, want_p = [], []
    for b in blobs:
        o = C.create_string_buffer(48)
        assert L.oblob_to_kzg_commitment(o, b, C.byref(oracle_settings)) == 0
        p = C.create_string_buffer(48)
        assert L.ocompute_blob_kzg_proof(p, b, o.raw, C.byref(oracle_settings)) == 0
        want_c.append(o.raw)
        want_p.append(p.raw)
    cells0, cproofs0 = kzg.compute_cells_and_kzg_proofs(blobs[0], s)  # pinned on the reference's vectors elsewhere
    rnd = random.Random(5)
    msm_in, ntt_in = [], []
    for m in (4096, 4096, 1000, 33):
        sc = _fr_bulk([rnd.randrange(O.R) for _ in range(m)])
        exp = O.G1()
        L.omsm_affine(C.byref(exp), pts, sc, m)
        msm_in.append((m, sc, _compressed(L, exp)))
    for n, inv in ((4096, False), (4096, True), (8192, False), (256, True)):
        data = _fr_bulk([rnd.randrange(O.R) for _ in range(n)])
        exp = (O.Fr * n)()
        assert L.offt_fr(C.byref(ofs), exp, data, n, 1 if inv else 0) == 0
        das = None
        if n <= 4096:
            d = (O.Fr * n)()
            assert L.odas_fft_extension(C.byref(ofs), d, data, n) == 0
            das = bytes(d)
        ntt_in.append((n, inv, data, bytes(exp), das))
    stop = time.time() + seconds
    failures, counts = [], [0] * (nthreads + 1)

    def worker(t):
        r = random.Random(1000 + t)
        try:
            while time.time() < stop:
                kind = r.randrange(9)
                if kind <= 1:  # B1
                    m, sc, want = msm_in[r.randrange(len(msm_in))]
                    assert _compressed(L, kzg.multi_scalar_mult_prepared(h, sc, m)) == want
                elif kind <= 3:  # B2
                    n, inv, data, want, das = ntt_in[r.randrange(len(ntt_in))]
                    assert bytes(fs.fft_fr(data, n, inverse=inv))[: 32 * n] == want
                    if das is not None and r.random() < 0.5:
                        assert bytes(fs.das_fft_extension(data, n))[: 32 * n] == das
                elif kind == 4:
                    i = r.randrange(len(blobs))
                    assert kzg.blob_to_kzg_commitment(blobs[i], s) == want_c[i]
                elif kind == 5:
                    i = r.randrange(len(blobs))
                    assert kzg.compute_blob_kzg_proof(blobs[i], want_c[i], s) == want_p[i]
                elif kind == 6:
                    i = r.randrange(len(blobs))
                    good = r.random() < 0.7
                    assert kzg.verify_blob_kzg_proof(blobs[i], want_c[i], want_p[i if good else (i + 1) % len(blobs)], s) is good
                elif kind == 7:
                    k = r.randrange(2, 5)
                    assert kzg.blob_to_kzg_commitment_batch(b"".join(blobs[:k]), k, s) == want_c[:k]
                    assert kzg.compute_blob_kzg_proof_batch(b"".join(blobs[:k]), b"".join(want_c[:k]), k, s) == want_p[:k]
                elif t % 4 == 0:  # cells are milliseconds of GPU: a quarter of the threads
                    c1, p1 = kzg.compute_cells_and_kzg_proofs(blobs[0], s)
                    assert c1 == cells0 and p1 == cproofs0
                counts[t] += 1
        except Exception as e:  # noqa: BLE001
            failures.append((t, repr(e)))

    def churn():
        # a second settings object is created, used and freed the whole time (6 GB tables: the load is what is exercised)
        try:
            while time.time() < stop:
                s2 = kzg.KZGSettings.from_file(SETUP, kzg.make_config(table_budget_gb=6))
                try:
                    assert kzg.blob_to_kzg_commitment(blobs[1], s2) == want_c[1]
                    assert kzg.compute_blob_kzg_proof(blobs[1], want_c[1], s2) == want_p[1]
                    h2 = kzg.prepare_multi_scalar_mult(pts, 512, kzg.make_config(table_budget_gb=1))
                    kzg.multi_scalar_mult_prepared(h2, msm_in[3][1], 33)
                    h2.close()
                finally:
                    s2.close()
                counts[nthreads] += 1
        except Exception as e:  # noqa: BLE001
            failures.append(("churn", repr(e)))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)] + [threading.Thread(target=churn)]
    for th in ts:
        th.start()
    for th in ts:
        th.join()
    print("soak: %.0f s, %d threads, %d calls, %d load/free cycles of the second object" % (seconds, nthreads, sum(counts[:-1]), counts[-1]))
    assert failures == []
    assert counts[-1] >= 1 and min(counts[:-1]) >= 1
    # the shared handles still answer after the soak
    assert kzg.blob_to_kzg_commitment(blobs[2], s) == want_c[2]
    L.offt_settings_free(C.byref(ofs))
    fs.close()
    h.close()
    s.close()

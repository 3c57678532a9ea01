#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs[2..4] (not the headline bench line):
  - G1 MSM scaling sweep n = 2^16 .. 2^22 (variable-base engine, device-resident inputs)
  - Fr NTT n = 4096 (batched) and n = 2^20, forward
  - compute_blob_kzg_proof_batch over 256 blobs (host buffers in/out, Fiat-Shamir hash on host)
Prints one JSON object; run on an MI355X:  python tools/extra_bench.py > profiles/r01_extra.json
"""
import ctypes as C
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup.txt")


def load_pkg():
    path = os.path.join(ROOT, "rust-kzg_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("rust_kzg_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rust_kzg_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def window_for(n):  # reference pippenger_window_size (kzg/src/msm/pippenger_utils.rs:300-317)
    b = n.bit_length()
    return b - 4 if b > 13 else (b - 3 if b > 5 else 2)


def main():
    import torch

    kzg = load_pkg()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    res = {}

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return min(ts)

    # ---- MSM sweep ----
    sweep = []
    nmax = 1 << 22
    pts = torch.empty(nmax * 96, dtype=torch.uint8, device=dev)
    kzg.generate_points(pts.data_ptr(), nmax, 2, stream)
    torch.cuda.synchronize()  # handles copy the points on their own non-blocking streams
    g = torch.Generator(device="cpu")
    g.manual_seed(2)
    sc = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, generator=g)
    sc[:, 31] &= 0x3F
    sc = sc.to(dev)
    out = torch.zeros(144, dtype=torch.uint8, device=dev)
    for logn in (12, 14, 16, 18, 20, 21, 22):
        n = 1 << logn
        h = kzg.DeviceMsm(pts.data_ptr(), n, False)
        info = h.info()
        ms = timed(lambda: kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, 1, False, stream))
        c = window_for(n)
        w = -(-255 // c)
        adds = n * w + (1 << c) * w  # algorithmic adds with the reference's window
        sweep.append({"n": n, "ms": ms, "pairs_per_s": n / (ms * 1e-3), "g1_adds_per_s": adds / (ms * 1e-3),
                      "algorithmic_GBps": 128 * n / (ms * 1e-3) / 1e9, "kernel_window_bits": info["window_bits"]})
        h.close()
    res["msm_sweep_variable_base"] = sweep
    # prepared handle at n = 2^20 (fixed-base rows, one bucket set, no Horner) and the host-buffer B1 calls
    n = 1 << 20
    t0 = time.perf_counter()
    h = kzg.DeviceMsm(pts.data_ptr(), n, True)
    torch.cuda.synchronize()
    prep_s = time.perf_counter() - t0
    info = h.info()
    ms = timed(lambda: kzg.msm_prepared_batch_device(h, out.data_ptr(), sc.data_ptr(), n, 1, False, stream))
    res["msm_2p20_prepared"] = {"ms": ms, "prepare_s": prep_s, "window_bits": info["window_bits"], "rows": info["rows"],
                                "wide_table": info["wide_table"]}
    h.close()
    hp = pts[: 4096 * 96].cpu().numpy().tobytes()
    hs = bytes(4096 * 32)
    import ctypes as C2
    sc_host = sc[:4096].cpu().numpy().copy()
    sc_host[:, 31] = 0
    hs = sc_host.tobytes()  # any value < r is a valid Montgomery residue
    kzg.multi_scalar_mult(hp, hs, 4096)
    t0 = time.perf_counter()
    for _ in range(10):
        kzg.multi_scalar_mult(hp, hs, 4096)
    res["mult_pippenger_n4096_host_call_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    del pts, sc

    # ---- NTT ----
    ntt = {}
    fs = kzg.FFTSettings(20)
    for n, nb in ((4096, 256), (1 << 20, 1)):
        a = torch.randint(0, 2**31, (nb * n * 8,), dtype=torch.int32, device=dev)
        a[7::8] &= 0x3FFFFFFF  # any 256-bit pattern below r is a valid Montgomery residue
        b = torch.empty_like(a)
        ms = timed(lambda: fs.fft_fr_device(b.data_ptr(), a.data_ptr(), n, nb, False, stream), reps=5)
        import math

        ntt["n=%d x %d" % (n, nb)] = {"ms": ms, "transforms_per_s": nb / (ms * 1e-3),
                                       "algorithmic_GBps": 64 * n * nb / (ms * 1e-3) / 1e9,
                                       "fr_mul_per_s": nb * (n / 2) * math.log2(n) / (ms * 1e-3)}
    res["ntt_forward"] = ntt
    # fft_g1 (host buffers in/out): the reference's bench_fft_g1 shape is scale 15 (kzg-bench/src/benches/fft.rs:13-21)
    import numpy as np
    g1 = {}
    for logn, nb in ((7, 64), (12, 1), (15, 1)):
        n = 1 << logn
        tot = n * nb
        aff = torch.empty(tot * 96, dtype=torch.uint8, device=dev)
        kzg.generate_points(aff.data_ptr(), tot, 9, stream)
        torch.cuda.synchronize()
        a = aff.cpu().numpy().reshape(tot, 96)
        one = np.array([0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745,
                        0x5c071a97a256ec6d, 0x15f65ec3fa80e493], dtype="<u8").tobytes()  # 2^384 mod p (Z = 1)
        jac = np.concatenate([a, np.tile(np.frombuffer(one, dtype=np.uint8), (tot, 1))], axis=1).copy()
        buf = (kzg.BlstP1 * tot).from_buffer(jac)
        fs.fft_g1(buf, n, nbatch=nb)
        t0 = time.perf_counter()
        fs.fft_g1(buf, n, nbatch=nb)
        dt = time.perf_counter() - t0
        g1["n=%d x %d" % (n, nb)] = {"ms": dt * 1e3, "scalar_muls_per_s": nb * (n / 2) * logn / dt}
        del aff
    res["fft_g1_forward_host_buffers"] = g1
    fs.close()

    # ---- blob proofs, batch of 256 through the host-buffer entry point ----
    s = kzg.KZGSettings.from_file(SETUP)
    import random

    rnd = random.Random(5)
    nb = 256
    blobs = bytearray(rnd.randbytes(nb * 131072))
    for i in range(0, len(blobs), 32):
        blobs[i] = 0
    blobs = bytes(blobs)
    cms = b"".join(kzg.blob_to_kzg_commitment_batch(blobs, nb, s))
    kzg.compute_blob_kzg_proof_batch(blobs, cms, nb, s)
    t0 = time.perf_counter()
    for _ in range(3):
        kzg.compute_blob_kzg_proof_batch(blobs, cms, nb, s)
    dt = (time.perf_counter() - t0) / 3
    res["compute_blob_kzg_proof_batch_256"] = {"ms": dt * 1e3, "proofs_per_s": nb / dt,
                                               "note": "host buffers in/out, SHA-256 challenges on host threads"}
    t0 = time.perf_counter()
    for _ in range(3):
        kzg.blob_to_kzg_commitment_batch(blobs, nb, s)
    dt = (time.perf_counter() - t0) / 3
    res["blob_to_kzg_commitment_batch_256_host_buffers"] = {"ms": dt * 1e3, "commitments_per_s": nb / dt}
    one = blobs[:131072]
    t0 = time.perf_counter()
    for _ in range(20):
        kzg.blob_to_kzg_commitment(one, s)
    res["blob_to_kzg_commitment_single_call_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.perf_counter()
    for _ in range(20):
        kzg.compute_blob_kzg_proof(one, cms[:48], s)
    res["compute_blob_kzg_proof_single_call_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    # EIP-7594: cells + 128 cell proofs per blob (128 fixed-base MSMs over the monomial setup)
    nb2 = 8
    kzg.compute_cells_and_kzg_proofs_batch(blobs[:nb2 * 131072], nb2, s)
    t0 = time.perf_counter()
    for _ in range(3):
        kzg.compute_cells_and_kzg_proofs_batch(blobs[:nb2 * 131072], nb2, s)
    dt = (time.perf_counter() - t0) / 3
    res["compute_cells_and_kzg_proofs_batch_8"] = {"ms": dt * 1e3, "blobs_per_s": nb2 / dt, "cell_proofs_per_s": 128 * nb2 / dt}
    t0 = time.perf_counter()
    for _ in range(5):
        kzg.compute_cells_and_kzg_proofs(one, s)
    res["compute_cells_and_kzg_proofs_single_call_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    s.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

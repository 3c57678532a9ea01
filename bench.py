#!/usr/bin/env python3
"""bench.py — blob_to_kzg_commitment throughput on MI355X (BASELINE.json metric).

One step = one pass of the hot path over one batch of synthetic blobs per GPU:
  B blobs (4096 x 32-byte field elements each, already resident in HBM)
    -> canonical scalars -> fixed-base Pippenger MSM over the mainnet trusted setup
    -> 48-byte compressed commitments (in HBM).
Workload = BASELINE.json configs[1] (n = 4096 G1 MSM over the trusted-setup points, random Fr
scalars), batched B per GPU; N > 1 shards whole blobs across ranks (weak scaling, no data-path
collective).  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup.txt")
BLOB = 131072
N = 4096
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALG_BYTES_PER_COMMIT = 128 * N             # 96 B point + 32 B scalar per pair
ALG_ADDS_PER_COMMIT = 20 * N + 8192        # BGMW count for the fixed-base 4096 case
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")


def load_pkg():
    path = os.path.join(ROOT, "rust-kzg_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("rust_kzg_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rust_kzg_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def make_blobs(torch, nblobs, seed, device):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    b = torch.randint(0, 256, (nblobs, N, 32), dtype=torch.uint8, generator=g)
    b[:, :, 0] = 0  # generate_random_blob_bytes: every element < r (kzg-bench/src/tests/eip_4844.rs:28-37)
    return b.reshape(nblobs, BLOB).to(device)


def host_cores():
    """usable host cores: affinity mask capped by the cgroup CPU quota (cpu.max)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(blobs_host, budget_s=12.0):
    """Times the CPU oracle (portable C restatement of the reference's Pippenger path — NOT blst asm)
    on this box's host cores: one thread per core, each committing to its own blobs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O

    L = O.lib()
    with open(SETUP, "rb") as f:
        rc, s = O.load_settings(f.read())
    assert rc == 0
    out = C.create_string_buffer(48)
    t0 = time.perf_counter()
    assert L.oblob_to_kzg_commitment(out, blobs_host[0], C.byref(s)) == 0
    t1 = time.perf_counter() - t0
    cores = host_cores()
    per_thread = max(1, min(256, int(budget_s / max(t1, 1e-3))))
    done = [0] * cores

    def work(t):
        o = C.create_string_buffer(48)
        for k in range(per_thread):
            L.oblob_to_kzg_commitment(o, blobs_host[(t + k) % len(blobs_host)], C.byref(s))
            done[t] += 1

    th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    return {"value": sum(done) / dt, "unit": "commitments/s", "cores": cores, "kind": "port",
            "single_thread_ms": t1 * 1e3,
            "sample": "%d commitments (%d threads x %d, seeded random blobs, mainnet setup) in %.1f s; "
                      "portable-C oracle (oracle/msm.c tiling Pippenger), not blst asm" % (sum(done), cores, per_thread, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="blobs per GPU per step")
    ap.add_argument("--streams", type=int, default=4,
                    help="1: batches back to back on one stream; 2-4: consecutive batches rotate over that many streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large", action="store_true", help="skip the 2^20-point MSM latency line")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group("nccl", device_id=dev)

    kzg = load_pkg()
    settings = kzg.KZGSettings.from_file(SETUP)
    handle = settings.msm_handle()
    info = kzg.PreparedMsm.info(type("H", (), {"handle": handle})())

    B = args.batch
    blobs = make_blobs(torch, B, 4844 + rank, dev)
    # A few streams, in rotation: every step is one whole batch (bytes in HBM -> commitments in HBM) on one stream;
    # consecutive batches are independent, so the low-occupancy tail of step i (block sums, compression) runs under
    # the accumulation kernel of step i + 1.  Each stream has its own outputs and scratch.
    NS = max(1, min(args.streams, 4))
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    outs = [torch.zeros(B * 48, dtype=torch.uint8, device=dev) for _ in range(NS)]
    stats = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NS)]
    scratch = [torch.empty(B * BLOB, dtype=torch.uint8, device=dev) for _ in range(NS)]
    out, status = outs[0], stats[0]
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()

    def step(i):
        k = i % NS
        kzg.blob_to_kzg_commitment_device(outs[k].data_ptr(), stats[k].data_ptr(), scratch[k].data_ptr(), blobs.data_ptr(), B,
                                          settings, streams[k].cuda_stream)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # set-up, like building the table: one pass over every stream so that each has its workspace allocated (a first
    # use calls hipMalloc, which synchronises the device) — then the W warm-up steps and the K timed steps
    for i in range(NS):
        step(i)
    sync_all()
    for i in range(args.warmup):
        step(i)
    sync_all()
    kzg.msm_set_profile(handle, True)  # HIP events around the dominant kernel, on its launch stream
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all()
    wall = time.perf_counter() - t0
    prof = kzg.msm_get_profile(handle)  # over the timed region (with several streams: while sharing the GPU)
    # the same kernel on a few launches that run alone, after the timed region: its own duration, which is what the
    # committed rocprofv3 summary measures and what the VALU utilisation is quoted on
    kzg.msm_set_profile(handle, True)
    for _ in range(3):
        step(0)
    torch.cuda.synchronize()
    prof_alone = kzg.msm_get_profile(handle)
    kzg.msm_set_profile(handle, False)
    assert all(int(st.sum().item()) == 0 for st in stats)
    if NS >= 2 and args.steps >= 2:
        assert all(torch.equal(outs[0], o) for o in outs[1:min(NS, args.steps)])  # the same blobs on every stream
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall_max = float(tmax.item())

    total_commits = B * args.steps * world
    value = total_commits / wall_max
    res = {
        "metric": "blob_to_kzg_commitment_per_s",
        "value": value,
        "unit": "commitments/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 (14x28-bit limbs, v_mad_u64_u32)",
        "data": "synthetic: seeded random blobs (byte 0 of each element zeroed), Ethereum mainnet trusted setup",
        "config": {"workload": "G1 Pippenger MSM n=4096 (EIP-4844 blob) x trusted-setup Lagrange points, "
                               "blob bytes -> 48-byte commitment, batched",
                   "blobs_per_gpu_per_step": B, "msm_window_bits": info["window_bits"], "table_rows": info["rows"],
                   "parallelism": "blobs sharded across %d GPU(s), table replicated, no collective" % world},
        "g1_adds_per_s": value * ALG_ADDS_PER_COMMIT,
        "streams": NS,
    }
    if prof is not None:
        accum_ms, total_ms, cnt = prof
        alg_bytes = ALG_BYTES_PER_COMMIT * B
        # with several streams the launches of the timed region share the GPU, so an individual launch lasts ~NS times
        # its own duration; the roofline is quoted on the kernel's own duration (launches that run alone, measured
        # with the same HIP events right after the timed region; `bench.py --streams 1` times them inside it)
        own_ms = prof_alone[0] if (prof_alone and NS > 1) else accum_ms
        ach = alg_bytes / (own_ms * 1e-3) / 1e9
        traffic = None
        try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (same kernel, same window), scaled to B
            pm = json.load(open(PMC_SUMMARY))
            if info.get("wide_table") and pm.get("window_bits") == info["window_bits"]:
                traffic = pm["k_fbw_accum"]["hbm_bytes_per_launch"] / pm["batch"] * B
        except Exception:
            pass
        res["roofline"] = {"bound": "hbm", "kernel": "k_fbw_accum" if info.get("wide_table") else "k_accum", "achieved": ach,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                           "kernel_ms": own_ms, "kernel_ms_in_timed_region_sharing_the_gpu": accum_ms,
                           "pipeline_ms": total_ms, "launches_averaged": cnt,
                           "algorithmic_bytes_per_launch": alg_bytes,
                           "note": "MSM is integer-VALU bound, not HBM bound (SURVEY §8d); see `valu`"}
        # The kernel is bound by VALU instruction issue (one wave-instruction per SIMD per 4 cycles), not by HBM.
        # Instructions per launch and the busy fraction come from the committed rocprofv3 PMC passes of this same
        # kernel (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE), the duration is the live one.
        try:
            pm = json.load(open(PMC_SUMMARY))
            pk = pm["k_fbw_accum"]
            if info.get("wide_table") and pm.get("window_bits") == info["window_bits"]:
                wave_instr = pk["SQ_INSTS_VALU"] / pm["batch"] * B
                peak = 1024 * 2.4e9 / 4  # 1024 SIMDs, nominal 2.4 GHz, 4 cycles per wave64 VALU instruction
                alone_ms = prof_alone[0] if prof_alone else accum_ms
                res["valu"] = {"bound": "VALU issue", "achieved": wave_instr / (alone_ms * 1e-3), "peak": peak,
                               "unit": "VALU wave-instructions/s", "frac": wave_instr / (alone_ms * 1e-3) / peak,
                               "kernel_ms_alone": alone_ms,
                               "busy_frac_at_sustained_clock": pk.get("valu_busy_frac"),
                               "sustained_clock_ghz": pk.get("effective_clock_ghz"),
                               "note": "frac is against the nominal 2.4 GHz; under this all-VALU load the chip sustains "
                                       "~2.0 GHz, where the VALUs are busy busy_frac of the kernel's cycles "
                                       "(SQ_ACTIVE_INST_VALU x 4 / SIMD cycles from GRBM_GUI_ACTIVE)"}
        except Exception:
            pass

    # second half of the metric: 2^20-point G1 MSM latency (variable-base engine, device-resident inputs)
    if not args.no_large and rank == 0:
        n = 1 << 20
        pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
        kzg.generate_points(pts.data_ptr(), n, 2, stream)
        g = torch.Generator(device="cpu")
        g.manual_seed(2)
        sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
        sc[:, 31] &= 0x3f  # little-endian canonical scalars < 2^254 < r
        sc = sc.to(dev)
        big = kzg.DeviceMsm(pts.data_ptr(), n, False)
        o = torch.zeros(144, dtype=torch.uint8, device=dev)
        kzg.msm_prepared_batch_device(big, o.data_ptr(), sc.data_ptr(), n, 1, False, stream)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            kzg.msm_prepared_batch_device(big, o.data_ptr(), sc.data_ptr(), n, 1, False, stream)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res["msm_2p20_ms"] = min(ts)
        res["msm_2p20_pairs_per_s"] = n / (min(ts) * 1e-3)
        big.close()

    if rank == 0 and not args.no_large:
        # host buffers in / host buffers out through the c-kzg batch entry point (PCIe both ways) — never `value`
        nb = min(B, 256)
        hb = blobs[:nb].cpu().numpy().tobytes()
        kzg.blob_to_kzg_commitment_batch(hb, nb, settings)
        t0 = time.perf_counter()
        for _ in range(3):
            kzg.blob_to_kzg_commitment_batch(hb, nb, settings)
        res["pcie_inclusive_commitments_per_s"] = 3 * nb / (time.perf_counter() - t0)
        # blob proofs through the batched host-buffer entry point (Fiat-Shamir hashes on host threads)
        cm = b"".join(kzg.blob_to_kzg_commitment_batch(hb, nb, settings))
        kzg.compute_blob_kzg_proof_batch(hb, cm, nb, settings)
        t0 = time.perf_counter()
        for _ in range(3):
            kzg.compute_blob_kzg_proof_batch(hb, cm, nb, settings)
        res["blob_proofs_per_s_host_buffers"] = 3 * nb / (time.perf_counter() - t0)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        host = blobs[: min(B, 64)].cpu().numpy()
        res["cpu_baseline"] = cpu_baseline([host[i].tobytes() for i in range(host.shape[0])])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res))
    settings.close()


if __name__ == "__main__":
    main()

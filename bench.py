#!/usr/bin/env python3
"""bench.py — blob_to_kzg_commitment throughput on MI355X (BASELINE.json metric), plus one driver-timed figure
with its own roofline object for every other BASELINE.json config.

One step = one pass of the hot path over `--batches-per-step` batches of synthetic blobs per GPU:
  B blobs (4096 x 32-byte field elements each, already resident in HBM)
    -> canonical scalars -> fixed-base MSM over the mainnet trusted setup -> 48-byte compressed commitments (in HBM).
Workload = BASELINE.json configs[1] (n = 4096 G1 MSM over the trusted-setup points, random Fr scalars), batched;
N > 1 shards whole blobs across ranks (weak scaling, table replicated, no data-path collective).

  python bench.py [--gpus N] [--steps K] [--warmup W]
`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself
(python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...); under an external
launcher the world size must equal --gpus.  Rank 0 prints ONE JSON line.
"""
import argparse
import csv
import ctypes as C
import glob
import hashlib
import importlib.util
import json
import math
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup.txt")
BLOB = 131072
N = 4096
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK = 1024 * 2.4e9 / 4               # 1024 SIMDs x one wave64 VALU instruction per 4 cycles at the nominal 2.4 GHz
VALU_PEAK_MODEL = ("1 wave64 instruction / SIMD / 4 cycles at 2.4 GHz: the issue rate tools/ffbench.hip measures for "
                   "v_mad_u64_u32-class instructions (profiles/r03_ffbench.log: 4.2-5.7 wall cycles per instruction for every "
                   "three-operand or 64-bit opcode; 2.9-3.0 for two-operand 32-bit ones)")
# MI355X_MICROARCH.md: SIMD-32, a wave64 VALU instruction issues over 2 cycles (the FP32 vector rate, 157.3 TFLOP/s)
VALU_PEAK_GUIDE_NOMINAL = 1024 * 2.4e9 / 2
ISA_HISTOGRAM = os.path.join(ROOT, "profiles", "r06_isa_histogram.json")  # tools/isa_histogram.py


def opcode_weighted_peak(kernel_substr, path=None):
    """VALU wave-instructions/s the chip can issue for THIS kernel's opcode mix: the static class counts of the kernel
    (tools/isa_histogram.py) weighted with the issue cycles per class tools/ffbench.hip measured; None without the file"""
    try:
        with open(path or ISA_HISTOGRAM) as f:
            h = json.load(f)
    except (OSError, ValueError):
        return None
    for name, k in h["kernels"].items():
        if kernel_substr in name and k.get("mean_issue_cycles_per_valu_instruction"):
            mean = k["mean_issue_cycles_per_valu_instruction"]
            return {"peak": 1024 * 2.4e9 / mean, "mean_issue_cycles_per_instruction": mean, "classes": k["classes"],
                    "class_issue_cycles": k["class_issue_cycles"], "kernel_symbol": name,
                    "source": os.path.relpath(path or ISA_HISTOGRAM, ROOT) + " (static opcode classes of the kernel) x profiles/r03_ffbench.log "
                              "(measured wall cycles per wave-instruction per class at the nominal 2.4 GHz)"}
    return None

ALG_BYTES_PER_COMMIT = 128 * N             # 96 B point + 32 B scalar per pair
ALG_ADDS_PER_COMMIT = 20 * N + 8192        # BGMW count for the fixed-base 4096 case
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06_pmc_summary.json")
PMC_FALLBACK = os.path.join(ROOT, "profiles", "r05_pmc_summary.json")
CSRC = os.path.join(ROOT, "rust-kzg_amd", "csrc")
# the sources a kernel family is compiled from: counters collected from another text of these files are not printed
KERNEL_SOURCES = {
    "msm": ["msm.hip", "msm_internal.h", "fp28.hip.h", "ff28.hip.h", "g1_28.hip.h", "g1_io.hip.h", "ff.hip.h", "glv.hip.h", "fpw.hip.h", "g1w.hip.h"],
    "ntt": ["ntt.hip", "ntt_internal.h", "ntt_plan.h", "fr29.hip.h", "ff.hip.h"],
}


def source_hash(family):
    """sha256 over the sources of a kernel family (what the kernel binary is a function of, with the pinned toolchain)"""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[family]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def load_pkg():
    path = os.path.join(ROOT, "rust-kzg_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("rust_kzg_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rust_kzg_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def make_blobs(torch, nblobs, seed, device):
    """generate_random_blob_bytes (kzg-bench/src/tests/eip_4844.rs:28-37): random bytes, byte 0 of every element
    zero so that it is < r.  Generated on the device (seeded), nothing to copy."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    b = torch.randint(0, 256, (nblobs, N, 32), dtype=torch.uint8, generator=g, device=device)
    b[:, :, 0] = 0
    return b.reshape(nblobs, BLOB)


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


def cpu_baseline(blobs_host, gpu_commitments, budget_s=10.0):
    """Times the CPU oracle (portable C restatement of the reference — NOT blst asm) on this box's host cores, one
    thread per core, each committing to its own blobs: the reference's default fixed-base algorithm (BGMW,
    kzg/src/msm/bgmw.rs) when the oracle has it, and the tiling Pippenger the round-1 line used.  Also the checker of
    this run: the oracle's commitments of the sampled blobs must equal the GPU's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O

    L = O.lib()
    with open(SETUP, "rb") as f:
        rc, s = O.load_settings(f.read())
    assert rc == 0
    cores = host_cores()
    out = {}
    algos = [("pippenger", L.oblob_to_kzg_commitment)]
    if hasattr(L, "oblob_to_kzg_commitment_bgmw"):
        algos.insert(0, ("bgmw", L.oblob_to_kzg_commitment_bgmw))
    checked = 0
    for name, fn in algos:
        o = C.create_string_buffer(48)
        assert fn(o, blobs_host[0], C.byref(s)) == 0  # untimed: the BGMW table is built by the first call
        t0 = time.perf_counter()
        assert fn(o, blobs_host[0], C.byref(s)) == 0
        t1 = time.perf_counter() - t0
        assert o.raw == gpu_commitments[0], "GPU commitment differs from the oracle (%s)" % name
        checked += 1
        per_thread = max(1, min(256, int(budget_s / len(algos) / max(t1, 1e-3))))
        done = [0] * cores
        bad = [0]

        def work(t):
            ob = C.create_string_buffer(48)
            for k in range(per_thread):
                i = (t + k) % len(blobs_host)
                fn(ob, blobs_host[i], C.byref(s))
                if ob.raw != gpu_commitments[i]:
                    bad[0] += 1
                done[t] += 1

        th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        assert bad[0] == 0, "GPU commitments differ from the oracle's"
        out[name] = {"commitments_per_s": sum(done) / dt, "single_thread_ms": t1 * 1e3, "commitments": sum(done), "seconds": dt}
    best = max(out, key=lambda k: out[k]["commitments_per_s"])
    return {"value": out[best]["commitments_per_s"], "unit": "commitments/s", "cores": cores, "kind": "port",
            "algorithm": best, "single_thread_ms": out[best]["single_thread_ms"], "by_algorithm": out,
            "gpu_commitments_checked_against_oracle": min(len(blobs_host), cores + 255),
            "sample": "%d commitments (%d threads, seeded random blobs, mainnet setup) in %.1f s with %s; portable-C oracle "
                      "(oracle/msm.c), not blst asm; every one compared with the GPU's commitment of the same blob"
                      % (out[best]["commitments"], cores, out[best]["seconds"], best)}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def reference_window(n):  # pippenger_window_size (kzg/src/msm/pippenger_utils.rs:300-317)
    b = n.bit_length()
    return b - 4 if b > 13 else (b - 3 if b > 5 else 2)


def pmc_summary():
    """The committed rocprofv3 PMC summary (tools/collect_round_profiles.sh + tools/summarize_profiles.py), used for the
    counter fields this run does not collect itself (collect_counters() covers the headline kernel; the NTT and 2^20 MSM
    counters come from here).  A family's counters are only used when the summary records the hash of the sources the
    profiled kernels were built from and it equals the hash of the sources in this tree: a summary of other kernels
    yields null fields, not stale numbers."""
    for p in (PMC_SUMMARY, PMC_FALLBACK):
        try:
            pm = json.load(open(p))
        except Exception:
            continue
        src = os.path.relpath(p, ROOT)
        if pm.get("collected_at_commit"):
            src += " (collected at commit %s)" % pm["collected_at_commit"]
        valid = {fam: (pm.get("source_sha256", {}).get(fam) == source_hash(fam)) for fam in KERNEL_SOURCES}
        return pm, src, valid
    return None, None, {fam: False for fam in KERNEL_SOURCES}


def collect_counters(batch, timeout_s=90):
    """Hardware counters of the headline kernel, collected by THIS run: rocprofv3 wraps a short serialised run of this
    script (--streams 1: a launch runs alone), one counter group per pass (the HBM guide's recipe: separate --pmc
    passes, --kernel-trace only), outside the timed region and after this process has freed its tables.  Returns
    (per-launch dict, note); the dict is None when rocprofv3 is missing or a pass fails."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found: counter fields come from the committed profile if its kernels are these"
    passes = ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES", "GRBM_GUI_ACTIVE"]
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--batches-per-step", "2",
             "--batch", str(batch), "--streams", "1", "--no-cpu-baseline", "--no-extras", "--no-counters"]
    out = {}
    tmp = tempfile.mkdtemp(prefix="kzg_pmc_", dir="/tmp")
    # the child is a plain one-GPU run, whatever launched this process (a torchrun rank passes its rendezvous on)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS")}
    env["TMPDIR"] = "/tmp"
    try:
        for i, p in enumerate(passes):
            d = os.path.join(tmp, "pass%d" % i)
            r = subprocess.run([exe, "--pmc"] + p.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--"] + child,
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 pass %r failed (rc %d)" % (p, r.returncode)
            agg = {}
            for row in csv.DictReader(open(files[0])):
                if "fbw_accum" in row["Kernel_Name"]:
                    agg.setdefault((row["Kernel_Name"], row["Grid_Size"]), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            if not agg:
                return None, "no k_fbw_accum dispatch in the counter pass"
            # the batch-sized launches (the largest grid)
            key = max(agg, key=lambda k: int(k[1]))
            for cn, v in agg[key].items():
                out[cn] = sum(v) / len(v)
            out["kernel"] = key[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if "GRBM_GUI_ACTIVE" in p:
                durs = []
                for tf in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                    for row in csv.DictReader(open(tf)):
                        if row["Kernel_Name"] == key[0] and (row.get("Grid_Size_X") or row.get("Grid_Size")) in (key[1], str(int(key[1]))):
                            durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
                if durs:
                    out["kernel_s_in_grbm_pass"] = sum(durs) / len(durs) * 1e-9
    except Exception as e:  # noqa: BLE001
        return None, "counter collection failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {"hbm_bytes_per_launch": out["FETCH_SIZE"] * 1024 * 2 + out["WRITE_SIZE"] * 1024,  # gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md)
           "SQ_INSTS_VALU": out["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": out["SQ_ACTIVE_INST_VALU"], "kernel": out["kernel"], "batch": batch}
    if "kernel_s_in_grbm_pass" in out:
        xcd_cycles = out["GRBM_GUI_ACTIVE"] / 8
        res["effective_clock_ghz"] = xcd_cycles / out["kernel_s_in_grbm_pass"] / 1e9
        res["valu_busy_frac"] = out["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * xcd_cycles)
        res["kernel_ms_in_grbm_pass"] = out["kernel_s_in_grbm_pass"] * 1e3
    return res, "collected by this run: rocprofv3 --pmc <one group per pass> --kernel-trace -- python bench.py --streams 1 (4 passes)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="blobs per batch (one MSM launch)")
    ap.add_argument("--batches-per-step", type=int, default=8, help="batches per GPU per step, each with its own blobs")
    ap.add_argument("--streams", type=int, default=4,
                    help="1: batches back to back on one stream; 2-4: consecutive batches rotate over that many streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip the NTT / MSM sweep / proof / host-buffer legs")
    ap.add_argument("--no-large", action="store_true", help="alias of --no-extras")
    ap.add_argument("--no-counters", action="store_true", help="skip the rocprofv3 counter passes over the headline kernel")
    ap.add_argument("--no-multi", action="store_true", help="skip the in-process multi-GPU leg (kzgamd_*_batch_multi)")
    args = ap.parse_args()
    if args.no_large:
        args.no_extras = True

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # self-launch: one rank per GPU over RCCL
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or let --gpus N launch them)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback\n")
        sys.stderr.flush()
        if world > 1:  # the launcher ends the other ranks as soon as one exits: let every rank say why first
            time.sleep(3)
        raise SystemExit(2)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group("nccl", device_id=dev)

    kzg = load_pkg()
    kzg.set_device(local_rank)
    settings = kzg.KZGSettings.from_file(SETUP)
    assert settings.device() == local_rank
    handle = settings.msm_handle()
    info = kzg.PreparedMsm.info(type("H", (), {"handle": handle})())

    B, NB = args.batch, max(1, args.batches_per_step)
    blobs = make_blobs(torch, B * NB, 4844 + 1000 * rank, dev)          # every batch of a step has its own blobs
    batch_ptr = [blobs[k * B:(k + 1) * B].data_ptr() for k in range(NB)]
    # A few streams, in rotation: every batch (bytes in HBM -> commitments in HBM) runs on one stream; consecutive
    # batches are independent, so the low-occupancy tail of one (block sums, compression) runs under the
    # accumulation kernel of the next.  Each (stream, batch) has its own outputs; each stream its own scratch.
    NS = max(1, min(args.streams, 4))
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    outs = [torch.zeros(B * 48, dtype=torch.uint8, device=dev) for _ in range(NB)]
    stats = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NB)]
    scratch = [torch.empty(B * BLOB, dtype=torch.uint8, device=dev) for _ in range(NS)]
    stream = torch.cuda.current_stream().cuda_stream
    for s_ in streams:
        settings.reserve(B, s_.cuda_stream)   # workspaces allocated here, not inside the timed region
    torch.cuda.synchronize()
    seq = [0]

    def step():
        for k in range(NB):
            j = seq[0] % NS
            seq[0] += 1
            kzg.blob_to_kzg_commitment_device(outs[k].data_ptr(), stats[k].data_ptr(), scratch[j].data_ptr(), batch_ptr[k], B,
                                              settings, streams[j].cuda_stream)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    kzg.msm_set_profile(handle, True)  # HIP events around the dominant kernel, on its launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    wall = time.perf_counter() - t0
    prof = kzg.msm_get_profile(handle)  # over the timed region (with several streams: launches share the GPU)
    # the same kernel on launches that run alone (one stream, back to back), after the timed region: its own duration,
    # which is what rocprofv3 --streams 1 measures and what the VALU utilisation is quoted on
    kzg.msm_set_profile(handle, True)
    for k in range(min(NB, 4)):
        kzg.blob_to_kzg_commitment_device(outs[k].data_ptr(), stats[k].data_ptr(), scratch[0].data_ptr(), batch_ptr[k], B,
                                          settings, streams[0].cuda_stream)
    torch.cuda.synchronize()
    prof_alone = kzg.msm_get_profile(handle)
    kzg.msm_set_profile(handle, False)
    assert all(int(st.sum().item()) == 0 for st in stats)
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall_max = float(tmax.item())
    devices = [{"rank": rank, "device": kzg.get_device(), "name": torch.cuda.get_device_name(local_rank),
                "settings_device": settings.device()}]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])
        devices = gathered

    gather_ms = None
    if dist is not None:
        # the one collective a sharded batch needs: all ranks' 48-byte results of one batch, gathered as
        # device tensors (ncclAllGather over RCCL); outside the timed region, timed on its own
        import importlib.util as _u

        _sp = _u.spec_from_file_location("kzg_sharding", os.path.join(ROOT, "rust-kzg_amd", "sharding.py"))
        _sh = _u.module_from_spec(_sp)
        _sp.loader.exec_module(_sh)
        _sh.gather_results(outs[0], B * world, 48, dist)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        allc = _sh.gather_results(outs[0], B * world, 48, dist)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t0) * 1e3
        assert allc.numel() == B * world * 48 and torch.equal(allc[rank * B * 48:(rank + 1) * B * 48], outs[0])

    total_commits = B * NB * args.steps * world
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
                   "blobs_per_batch": B, "batches_per_step": NB, "blobs_per_gpu_per_step": B * NB,
                   "msm_window_bits": info["window_bits"], "table_rows": info["rows"], "glv_split": info["wide_glv"],
                   "mixed_adds_per_scalar": info["adds_per_scalar"],
                   "parallelism": "blobs sharded across %d GPU(s), table replicated, no collective" % world},
        "timed_region_s": wall_max,
        "g1_adds_per_s": value * ALG_ADDS_PER_COMMIT,
        "streams": NS,
        "devices": devices,
    }
    # what a scaling run can check by itself: the ranks that took part in the barrier / max-over-ranks / all-gather, as
    # torch.distributed sees them under the nccl (= RCCL) backend; 1 and "none" for a single process
    res["rccl_ranks"] = dist.get_world_size() if dist is not None else 1
    res["dist_backend"] = dist.get_backend() if dist is not None else "none"
    res["distinct_devices"] = len({(d.get("device"), d.get("name")) for d in devices}) if dist is None else len({d.get("device") for d in devices})
    if gather_ms is not None:
        res["result_allgather_ms"] = gather_ms
    pm, pm_src, pm_valid = pmc_summary()
    kern = "k_fbw_accum" if info.get("wide_table") else "k_accum"
    own_ms = None
    if prof is not None:
        accum_ms, total_ms, cnt = prof
        alg_bytes = ALG_BYTES_PER_COMMIT * B
        own_ms = prof_alone[0] if (prof_alone and NS > 1) else accum_ms
        ach = alg_bytes / (own_ms * 1e-3) / 1e9
        res["roofline"] = {"bound": "hbm", "kernel": kern, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                           "kernel_ms": own_ms, "kernel_ms_in_timed_region_sharing_the_gpu": accum_ms,
                           # what a launch costs the timed configuration: co-resident launches on the other streams fill
                           # the issue slots a lone launch leaves idle, so this is below kernel_ms x launches
                           "effective_ms_per_launch_in_timed_region": wall_max / args.steps * 1e3 / NB,
                           "achieved_in_timed_region": alg_bytes / (wall_max / args.steps / NB) / 1e9,
                           "launches_per_step": NB, "pipeline_ms_per_launch": total_ms, "launches_averaged": cnt,
                           "algorithmic_bytes_per_launch": alg_bytes,
                           "note": "MSM is integer-VALU bound, not HBM bound (SURVEY §8d); see `valu`. kernel_ms is one "
                                   "launch (one batch) running alone; a step is launches_per_step of them"}

    def fill_counters(pk, pk_batch, source):
        """roofline.traffic and the `valu` object from per-launch counters of the headline kernel (pk, collected on
        launches of pk_batch blobs), durations from this run's HIP events"""
        if own_ms is None or pk is None:
            return
        res["roofline"]["traffic"] = pk["hbm_bytes_per_launch"] / pk_batch * B
        res["roofline"]["traffic_source"] = source
        wave_instr = pk["SQ_INSTS_VALU"] / pk_batch * B
        res["valu"] = {"bound": "VALU issue", "kernel": kern, "achieved": wave_instr / (own_ms * 1e-3), "peak": VALU_PEAK,
                       "unit": "VALU wave-instructions/s", "frac": wave_instr / (own_ms * 1e-3) / VALU_PEAK,
                       "peak_model": VALU_PEAK_MODEL,
                       "peak_guide_nominal": VALU_PEAK_GUIDE_NOMINAL,
                       "frac_vs_guide_nominal": wave_instr / (own_ms * 1e-3) / VALU_PEAK_GUIDE_NOMINAL,
                       "opcode_weighted": (lambda ow: None if ow is None else dict(ow, frac=wave_instr / (own_ms * 1e-3) / ow["peak"]))(
                           opcode_weighted_peak("k_fbw_accumILi8ELb%d" % (1 if info["wide_glv"] else 0))),  # every SPL >= 2 has the same mix
                       "kernel_ms_alone": own_ms, "valu_instructions_per_mixed_add": wave_instr * 64 / (B * N * info["adds_per_scalar"]),
                       "busy_frac_at_sustained_clock": pk.get("valu_busy_frac"),
                       "sustained_clock_ghz": pk.get("effective_clock_ghz"),
                       "kernel_ms_in_counter_pass": pk.get("kernel_ms_in_grbm_pass"),
                       "source": "instruction count, busy fraction and clock: %s; duration: live HIP events" % source,
                       "note": "frac is against the nominal 2.4 GHz; under this all-VALU load the chip sustains "
                               "sustained_clock_ghz (GRBM_GUI_ACTIVE / 8 XCDs / kernel time), where the VALUs are busy "
                               "busy_frac of the kernel's cycles"}

    if pm and pm_valid["msm"] and info.get("wide_table") and pm.get("window_bits") == info["window_bits"] and kern in pm:
        fill_counters(pm[kern], pm["batch"], pm_src + " (same kernel sources as this tree)")

    def ev_time(fn, reps=5):
        """min over reps of the HIP-event time of fn() on torch's current stream (the library launches there)"""
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

    def ev_time_back_to_back(fn, k=16, reps=3):
        """ms per call of k calls enqueued back to back between two events (min over reps): a single-call event pair also
        times the launch gap in front of the kernel (~5 us), which a caller that keeps the stream busy never sees"""
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(k):
                fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / k)
        return min(ts)

    # ---- blob proofs, device-resident pipeline (every rank; weak scaling like the headline) ----------------------
    # blobs and commitments in HBM -> proofs in HBM: SHA-256 challenge, barycentric evaluation + quotient, MSM, all
    # on the GPU; batches rotate over the streams so that the serial hash of one batch runs under the MSM of another
    if not args.no_extras:
        pouts = [torch.zeros(B * 48, dtype=torch.uint8, device=dev) for _ in range(NB)]
        pstat = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NB)]
        pscr = [torch.empty(B * kzg.PROOF_SCRATCH_BYTES, dtype=torch.uint8, device=dev) for _ in range(NS)]

        def prove_pass():
            for k in range(NB):
                j = k % NS
                kzg.compute_blob_kzg_proof_device(pouts[k].data_ptr(), pstat[k].data_ptr(), pscr[j].data_ptr(), batch_ptr[k],
                                                  outs[k].data_ptr(), B, settings, streams[j].cuda_stream)

        prove_pass()
        sync_all()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            prove_pass()
        sync_all()
        tp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        assert all(int(st.sum().item()) == 0 for st in pstat)
        # spot check of the device pipeline against the host-buffer entry point (host SHA-256) on two blobs
        hb2 = blobs[:2].cpu().numpy().tobytes()
        cm2 = outs[0][:96].cpu().numpy().tobytes()
        assert b"".join(kzg.compute_blob_kzg_proof_batch(hb2, cm2, 2, settings)) == pouts[0][:96].cpu().numpy().tobytes()
        res["blob_proofs_device_resident"] = {
            "proofs_per_s": reps * NB * B * world / float(tp.item()), "blobs_per_batch": B, "batches": reps * NB,
            "path": "kzgamd_compute_blob_kzg_proof_device: blobs + commitments in HBM -> proofs in HBM, SHA-256 challenge on "
                    "the GPU, batches rotating over %d streams" % NS,
            "algorithmic_bytes_per_proof": ALG_BYTES_PER_COMMIT + 131152}
        del pscr

    # ---- configs[4]: blob proofs, 256 blobs sharded over the ranks (host buffers in/out) -------------------------
    if not args.no_extras:
        nshard = max(1, 256 // world)
        hb = blobs[:nshard].cpu().numpy().tobytes()
        cm = b"".join(kzg.blob_to_kzg_commitment_batch(hb, nshard, settings))
        kzg.compute_blob_kzg_proof_batch(hb, cm, nshard, settings)
        sync_all()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            kzg.compute_blob_kzg_proof_batch(hb, cm, nshard, settings)
        sync_all()
        tp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        res["blob_proof_batch_256"] = {"proofs_per_s": reps * nshard * world / float(tp.item()), "blobs_per_rank": nshard,
                                       "ms_per_batch": float(tp.item()) / reps * 1e3,
                                       "path": "kzgamd_compute_blob_kzg_proof_batch: host buffers in and out (PCIe both ways), "
                                               "Fiat-Shamir SHA-256 on host threads",
                                       "algorithmic_bytes_per_proof": ALG_BYTES_PER_COMMIT + 131120}

    if rank == 0 and world == 1 and not args.no_extras:
        # ---- host-buffer calls of 1 .. 256 blobs (the reference-shaped entry points; median of 7 calls each) ----------
        table = []
        for nb_ in (1, 4, 16, 64, 256):
            hbn = blobs[:nb_].cpu().numpy().tobytes()
            cmn = b"".join(kzg.blob_to_kzg_commitment_batch(hbn, nb_, settings))
            row = {"blobs": nb_}
            for key, fn in (("commit_ms", lambda: kzg.blob_to_kzg_commitment_batch(hbn, nb_, settings)),
                            ("proof_ms", lambda: kzg.compute_blob_kzg_proof_batch(hbn, cmn, nb_, settings))):
                fn()
                ts = []
                for _ in range(7):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
                ts.sort()
                row[key] = ts[3] * 1e3
            table.append(row)
        res["host_buffer_batch_calls"] = {"rows": table, "path": "kzgamd_blob_to_kzg_commitment_batch / "
                                          "kzgamd_compute_blob_kzg_proof_batch through the ctypes mirror, host buffers in and out"}

    if rank == 0 and not args.no_extras:
        # ---- configs[3]: Fr NTT n = 4096 (batched) and n = 2^20, forward, inverse and the DAS extension of half -------
        fs = kzg.FFTSettings(20)
        ntt = {}
        # per transform call: HBM bytes and VALU instructions from the committed PMC passes — only if they are of these kernels
        npm = (pm or {}).get("ntt", {}) if pm_valid["ntt"] else {}
        for n, nb in ((4096, 256), (1 << 20, 1)):
            a = torch.randint(0, 2**31, (nb * n * 8,), dtype=torch.int32, device=dev)
            a[7::8] &= 0x3FFFFFFF  # any 256-bit pattern below r is a valid Montgomery residue
            b = torch.empty_like(a)
            ms = ev_time(lambda: fs.fft_fr_device(b.data_ptr(), a.data_ptr(), n, nb, False, stream), reps=9)
            ms_inv = ev_time(lambda: fs.fft_fr_device(b.data_ptr(), a.data_ptr(), n, nb, True, stream), reps=9)
            ms_b2b = ev_time_back_to_back(lambda: fs.fft_fr_device(b.data_ptr(), a.data_ptr(), n, nb, False, stream))
            # DAS extension of the half-size lists (BASELINE configs[3]: "fft_fr + DAS extension"): two half-size transforms
            h = n // 2
            t = torch.empty(nb * h * 8, dtype=torch.int32, device=dev)
            ms_das = ev_time(lambda: fs.das_fft_extension_device(b.data_ptr(), a.data_ptr(), t.data_ptr(), h, nb, stream), reps=9)
            alg = 64 * n * nb
            muls = nb * (n // 2) * int(math.log2(n))
            key = "n=%d x %d" % (n, nb)
            pk = npm.get(key) or {}
            winstr = pk.get("SQ_INSTS_VALU")
            ntt[key] = {
                "ms": ms, "ms_inverse": ms_inv, "ms_back_to_back": ms_b2b, "transforms_per_s": nb / (ms * 1e-3), "fr_mul_per_s": muls / (ms * 1e-3),
                "das_extension": {"half_n": h, "lists": nb, "ms": ms_das, "algorithmic_bytes": 2 * 64 * h * nb,
                                  "achieved_GBps": 2 * 64 * h * nb / (ms_das * 1e-3) / 1e9,
                                  "path": "kzgamd_das_fft_extension_device: inverse + forward transform of half_n points, the "
                                          "twist by the 2n-th roots and n^-1 in their last-pass multiplications"},
                "roofline": {"bound": "hbm", "kernel": "k_ntt_pass" + (" x %d passes" % len(pk.get("passes", [0, 0])) if n > 4096 else ""),
                             "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes": alg,
                             "traffic": pk.get("hbm_bytes_per_call"),
                             "traffic_source": (pm_src + " (FETCH_SIZE x2 per the gfx950 note + WRITE_SIZE; n = 2^20 crosses HBM "
                                                "twice: two passes)") if pk.get("hbm_bytes_per_call") else None,
                             "note": "64*n algorithmic bytes (SURVEY §8d); the transform is VALU-issue bound, see `valu`"},
                "valu": None if not winstr else {
                    "bound": "VALU issue", "achieved": winstr / (ms * 1e-3), "peak": VALU_PEAK, "unit": "VALU wave-instructions/s",
                    "frac": winstr / (ms * 1e-3) / VALU_PEAK, "wave_instructions_per_call": winstr,
                    "peak_model": VALU_PEAK_MODEL, "frac_vs_guide_nominal": winstr / (ms * 1e-3) / VALU_PEAK_GUIDE_NOMINAL,
                    "opcode_weighted": (lambda ow: None if ow is None else dict(ow, frac=winstr / (ms * 1e-3) / ow["peak"]))(
                        opcode_weighted_peak("k_ntt_passILi%d" % (0 if n <= 4096 else 2), os.path.join(ROOT, "profiles", "r06_isa_histogram_ntt.json"))),
                    "instructions_per_butterfly": winstr * 64 / muls,
                    "floor_us_at_nominal_clock": winstr / VALU_PEAK * 1e6,
                    "busy_frac_at_sustained_clock": pk.get("valu_busy_frac"), "scratch_bytes_per_lane": pk.get("scratch", 0),
                    "source": "instruction count and busy fraction: %s; duration: live HIP events" % pm_src}}
            del a, b, t
        res["ntt"] = ntt
        fs.close()

        # ---- configs[2]: G1 MSM sweep n = 2^16 .. 2^22 (variable-base engine, device-resident inputs) --------------
        nmax = 1 << 22
        pts = torch.empty(nmax * 96, dtype=torch.uint8, device=dev)
        kzg.generate_points(pts.data_ptr(), nmax, 2, stream)
        torch.cuda.synchronize()  # the handles below copy the points on their own (non-blocking) streams
        g = torch.Generator(device=dev)
        g.manual_seed(2)
        sc = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, generator=g, device=dev)
        sc[:, 31] &= 0x3F  # little-endian canonical scalars < 2^254 < r
        o = torch.zeros(144, dtype=torch.uint8, device=dev)
        sweep = []
        sweep_pm = (pm or {}).get("msm_sweep", {}) if pm_valid["msm"] else {}
        for logn in (16, 18, 20, 21, 22):
            n = 1 << logn
            h = kzg.DeviceMsm(pts.data_ptr(), n, False)
            hi = h.info()
            ms = ev_time(lambda: kzg.msm_prepared_batch_device(h, o.data_ptr(), sc.data_ptr(), n, 1, False, stream), reps=9)
            c = reference_window(n)
            w = -(-255 // c)
            adds = n * w + (1 << c) * w  # algorithmic adds with the reference's window
            gbs = 128 * n / (ms * 1e-3) / 1e9
            sweep.append({"n": n, "ms": ms, "pairs_per_s": n / (ms * 1e-3), "g1_adds_per_s": adds / (ms * 1e-3),
                          "kernel_window_bits": hi["window_bits"],
                          "roofline": {"bound": "hbm", "kernel": "k_accum (+ sort and reduction kernels)", "achieved": gbs,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                       "algorithmic_bytes": 128 * n,
                                       "traffic": (sweep_pm.get(str(n)) or {}).get("hbm_bytes_per_call"),
                                       "traffic_source": pm_src if sweep_pm.get(str(n)) else None}})
            if logn == 20:
                res["msm_2p20_ms"] = ms
                res["msm_2p20_pairs_per_s"] = n / (ms * 1e-3)
                # the batched figure: four MSMs of 2^20 scalars over the same bases in ONE call (scalars 4 x 2^20 = the whole
                # synthetic array), ms per MSM
                o4 = torch.zeros(4 * 144, dtype=torch.uint8, device=dev)
                ms4 = ev_time(lambda: kzg.msm_prepared_batch_device(h, o4.data_ptr(), sc.data_ptr(), n, 4, False, stream), reps=5)
                res["msm_2p20_batched"] = {"msms_per_call": 4, "ms_per_call": ms4, "ms_per_msm": ms4 / 4}
            h.close()
        res["msm_sweep"] = sweep
        del pts, sc

        # ---- host buffers in / host buffers out through the c-kzg batch entry point (PCIe both ways) — never `value`
        nb = min(4096, B * NB)
        hb = blobs[:nb].cpu().numpy().tobytes()
        cmh = kzg.blob_to_kzg_commitment_batch(hb, nb, settings)
        nc = nb  # one large call: chunks of a quarter of the call pipelined over four streams (1024 per call: 84 k/s)
        hc = hb[:nc * BLOB]
        kzg.blob_to_kzg_commitment_batch(hc, nc, settings)
        t0 = time.perf_counter()
        for _ in range(3):
            kzg.blob_to_kzg_commitment_batch(hc, nc, settings)
        res["pcie_inclusive_commitments_per_s"] = 3 * nc / (time.perf_counter() - t0)
        res["pcie_inclusive_blobs_per_call"] = nc
        res["pcie_inclusive_proof_blobs_per_call"] = nb
        # the same for proofs: one large host-buffer call (chunks pipelined over the streams, SHA-256 on the host pool)
        cmb = b"".join(cmh)
        kzg.compute_blob_kzg_proof_batch(hb, cmb, nb, settings)
        t0 = time.perf_counter()
        for _ in range(2):
            kzg.compute_blob_kzg_proof_batch(hb, cmb, nb, settings)
        res["pcie_inclusive_proofs_per_s"] = 2 * nb / (time.perf_counter() - t0)

    if rank == 0 and world == 1 and not args.no_extras:
        # ---- the reference's concurrency pattern (kzg/src/eip_4844.rs:781-805): native threads sharing one settings
        # object, single-blob calls (tools/concurrent_bench, built by __graft_entry__.build())
        cb = os.path.join(ROOT, "tools", "concurrent_bench")
        if os.path.exists(cb):
            try:
                # its table next to this process's 137 GB one: capped so that both fit
                env = dict(os.environ, KZGAMD_FBW_MAX_GB="100", LD_LIBRARY_PATH=os.path.join(ROOT, "rust-kzg_amd", "csrc") + ":" + os.environ.get("LD_LIBRARY_PATH", "") + ":/opt/rocm/lib")
                o16 = json.loads(subprocess.run([cb, SETUP, "0.8", "16"], stdout=subprocess.PIPE, env=env, timeout=120).stdout.decode().strip().splitlines()[-1])
                o1 = json.loads(subprocess.run([cb, SETUP, "0.5", "1"], stdout=subprocess.PIPE, env=env, timeout=120).stdout.decode().strip().splitlines()[-1])
                # B1 seam: the same threads on ONE prepared handle, mult_pippenger_prepared (what a Rust caller through
                # the kzg:: traits reaches); with and without the combining of concurrent calls into one launch
                env_nc = dict(env, KZGAMD_TUNING="combine=0")
                b16nc = json.loads(subprocess.run([cb, SETUP, "0.8", "16", "2"], stdout=subprocess.PIPE, env=env_nc, timeout=120).stdout.decode().strip().splitlines()[-1])
                # B2 seam: the same for ONE NTT handle, ntt_fr of 4096 elements
                n16 = json.loads(subprocess.run([cb, SETUP, "0.6", "16", "3"], stdout=subprocess.PIPE, env=env, timeout=120).stdout.decode().strip().splitlines()[-1])
                n1 = json.loads(subprocess.run([cb, SETUP, "0.4", "1", "3"], stdout=subprocess.PIPE, env=env, timeout=120).stdout.decode().strip().splitlines()[-1])
                n16nc = json.loads(subprocess.run([cb, SETUP, "0.6", "16", "3"], stdout=subprocess.PIPE, env=env_nc, timeout=120).stdout.decode().strip().splitlines()[-1])
                res["concurrent_callers"] = {
                    "threads_16": {"blob_to_kzg_commitment_per_s": o16.get("commit_threads_16"), "compute_blob_kzg_proof_per_s": o16.get("proof_threads_16")},
                    "threads_1": {"blob_to_kzg_commitment_per_s": o1.get("commit_threads_1"), "compute_blob_kzg_proof_per_s": o1.get("proof_threads_1")},
                    "b1_prepared_threads_1": o1.get("b1_prepared_threads_1"),
                    "b1_prepared_threads_16": o16.get("b1_prepared_threads_16"),
                    "b1_prepared_threads_16_over_1": (o16.get("b1_prepared_threads_16") or 0) / max(o1.get("b1_prepared_threads_1") or 1, 1),
                    "b1_prepared_threads_16_calls_not_combined": b16nc.get("b1_prepared_threads_16"),
                    "b2_ntt_fr_4096_threads_1": n1.get("b2_ntt_fr_4096_threads_1"),
                    "b2_ntt_fr_4096_threads_16": n16.get("b2_ntt_fr_4096_threads_16"),
                    "b2_ntt_fr_4096_threads_16_calls_not_combined": n16nc.get("b2_ntt_fr_4096_threads_16"),
                    "b2_unit": "ntt_fr calls/s (4096 elements, host buffers) on one shared NTT handle",
                    "b1_unit": "mult_pippenger_prepared calls/s (4096 scalars each) on one shared prepared handle (24 GB table budget: c = 13, 20 additions per scalar)",
                    "failed_or_different_from_the_serial_results": (o16.get("failed_or_different_from_the_serial_results", 0) or 0)
                                                                   + (o1.get("failed_or_different_from_the_serial_results", 0) or 0)
                                                                   + (b16nc.get("failed_or_different_from_the_serial_results", 0) or 0)
                                                                   + sum((x.get("failed_or_different_from_the_serial_results", 0) or 0) for x in (n16, n1, n16nc)),
                    "path": "native threads, one CKZGSettings, host buffers; calls are merged into batches on up to three lanes "
                            "(own process: its settings object is loaded next to this one); every result is compared with "
                            "the one a serial call gave"}
            except Exception as e:  # noqa: BLE001
                res["concurrent_callers"] = {"error": repr(e)}


    if rank == 0 and world == 1 and not args.no_extras:
        # ---- EIP-7594 cell proofs, 256 blobs per call (FK20 on the GPU), host buffers in and out
        ncell = min(256, B * NB)
        hbc = blobs[:ncell].cpu().numpy().tobytes()
        import ctypes as C_

        cbuf, pbuf = C_.create_string_buffer(ncell * 128 * 2048), C_.create_string_buffer(ncell * 128 * 48)

        def cells_call():
            rc = kzg.lib().kzgamd_compute_cells_and_kzg_proofs_batch(cbuf, pbuf, hbc, ncell, C_.byref(settings.c))
            if rc != 0:
                raise RuntimeError("kzgamd_compute_cells_and_kzg_proofs_batch: %d" % rc)

        cells_call()
        t0 = time.perf_counter()
        cells_call()
        cells_call()
        dtc = (time.perf_counter() - t0) / 2
        def cells_n(nn):
            rc = kzg.lib().kzgamd_compute_cells_and_kzg_proofs_batch(cbuf, pbuf, hbc, nn, C_.byref(settings.c))
            if rc != 0:
                raise RuntimeError("kzgamd_compute_cells_and_kzg_proofs_batch: %d" % rc)

        small = {}
        for nn in (1, 4, 8, 16, 64):  # 1 and 2 blobs: the direct form; from 3 blobs FK20
            cells_n(nn)
            t0 = time.perf_counter()
            cells_n(nn)
            cells_n(nn)
            small["ms_per_call_%d_blobs" % nn] = (time.perf_counter() - t0) / 2 * 1e3
        res["cells_and_proofs_small_batches"] = small
        res["cells_and_proofs_256"] = {"ms_per_call": dtc * 1e3, "cell_proofs_per_s": ncell * 128 / dtc, "blobs_per_s": ncell / dtc,
                                       "path": "kzgamd_compute_cells_and_kzg_proofs_batch: 128 cells + 128 cell proofs per blob, "
                                               "FK20 (64 NTTs of 128, 128 MSMs of 64 points, two G1 transforms of 128)"}
        del hbc

    if rank == 0 and world == 1 and not args.no_extras:
        # ---- EIP-7594 recovery and cell verification, one blob, host buffers -----------------------
        blob1 = blobs[:1].cpu().numpy().tobytes()
        cells1, proofs1 = kzg.compute_cells_and_kzg_proofs(blob1, settings)
        cm1 = kzg.blob_to_kzg_commitment(blob1, settings)
        half = list(range(0, 128, 2))
        part = b"".join(cells1[2048 * i:2048 * (i + 1)] for i in half)
        rc_, rp_ = kzg.recover_cells_and_kzg_proofs(half, part, settings)
        assert rc_ == cells1 and rp_ == proofs1
        t0 = time.perf_counter()
        for _ in range(3):
            kzg.recover_cells_and_kzg_proofs(half, part, settings)
        t_rec = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        for _ in range(3):
            kzg.recover_cells_and_kzg_proofs(half, part, settings, want_proofs=False)
        t_rec_cells = (time.perf_counter() - t0) / 3
        allidx = list(range(128))
        assert kzg.verify_cell_kzg_proof_batch(cm1 * 128, allidx, cells1, proofs1, settings)
        t0 = time.perf_counter()
        for _ in range(3):
            kzg.verify_cell_kzg_proof_batch(cm1 * 128, allidx, cells1, proofs1, settings)
        t_ver = (time.perf_counter() - t0) / 3
        res["eip7594"] = {"recover_cells_and_kzg_proofs_ms": t_rec * 1e3, "recover_cells_only_ms": t_rec_cells * 1e3,
                          "verify_cell_kzg_proof_batch_128_cells_ms": t_ver * 1e3,
                          "path": "64 of 128 cells -> all cells (five 8192-point transforms on the GPU) + 128 proofs (direct "
                                  "form); verification: interpolation polynomial, decode, subgroup checks and one two-row MSM on the GPU, pairing on the host"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ns = min(B, 64)
        host = blobs[:ns].cpu().numpy()
        gpu = outs[0][:ns * 48].cpu().numpy().tobytes()
        res["cpu_baseline"] = cpu_baseline([host[i].tobytes() for i in range(ns)], [gpu[48 * i:48 * i + 48] for i in range(ns)])
    # ---- everything below runs after this rank has freed its tables ------------------------------------------------
    host_blobs = None
    if rank == 0 and not args.no_multi and not args.no_extras:
        host_blobs = blobs[:min(B * NB, 1024 * max(1, world))].cpu().numpy().tobytes() if world == 1 else \
            (blobs[:min(B * NB, 1024)].cpu().numpy().tobytes() * world)
    settings.close()
    del blobs, outs, scratch
    torch.cuda.empty_cache()
    if dist is not None:
        dist.barrier()

    if rank == 0 and not args.no_counters:
        # ---- hardware counters of the headline kernel, collected by this run (separate rocprofv3 passes) ------------
        pk, note = collect_counters(B) if info.get("wide_table") else (None, "bucket engine: no k_fbw_accum")
        res["counters"] = {"collected": pk is not None, "note": note, "kernel_source_sha256": {f: source_hash(f) for f in KERNEL_SOURCES}}
        if pk is not None:
            fill_counters(pk, pk["batch"], note)

    if rank == 0 and not args.no_multi and not args.no_extras:
        # ---- multi-GPU from ONE process, inside the library (csrc/multi.hip): one settings object per GPU, contiguous
        # slabs of the batch, one host thread per device; host buffers in and out (PCIe both ways — never `value`).
        # Under torchrun the other ranks have freed their tables and wait at the barrier below.
        try:
            ndev = min(world, kzg.device_count())
            ms = kzg.MultiKZGSettings(SETUP, list(range(ndev)))
            nmb = len(host_blobs) // BLOB
            cms = ms.commit_batch(host_blobs, nmb)
            t0 = time.perf_counter()
            for _ in range(3):
                cms = ms.commit_batch(host_blobs, nmb)
            t_c = (time.perf_counter() - t0) / 3
            cmb = b"".join(cms)
            ms.proof_batch(host_blobs, cmb, nmb)
            t0 = time.perf_counter()
            for _ in range(2):
                prs = ms.proof_batch(host_blobs, cmb, nmb)
            t_p = (time.perf_counter() - t0) / 2
            ok = ms.verify_blob_batch(host_blobs[:64 * BLOB], cmb[:64 * 48], b"".join(prs[:64]), min(64, nmb))
            res["in_process_multi"] = {
                "devices": ms.settings_devices(), "objects": ndev, "blobs_per_call": nmb, "commitments_per_s": nmb / t_c, "proofs_per_s": nmb / t_p,
                "host_read_GBps": nmb * BLOB / t_c / 1e9,
                "first_64_proofs_verify": bool(ok),
                "path": "kzgamd_blob_to_kzg_commitment_batch_multi / kzgamd_compute_blob_kzg_proof_batch_multi: one process, "
                        "%d settings object(s), slabs of the batch per device on one host thread each, host buffers in and out "
                        "(PCIe both ways); no torch.distributed, no collective" % ndev}
            ms.close()
        except Exception as e:  # noqa: BLE001
            res["in_process_multi"] = {"error": repr(e)}
        if world == 1:
            # ---- the N = 8 host side on the hardware there is: EIGHT settings objects on this one GPU (8 GB per table), fed by
            # the library's eight host threads from pageable memory and from page-locked memory (kzgamd_pin_host_buffer).  The
            # GPU is shared eight ways, so commitments/s says nothing about eight GPUs; what the leg shows is what ONE host
            # process sustains reading caller buffers through eight slab threads — the first wall an 8-GPU node would meet
            # (8 x ~87 k blobs/s = ~90 GB/s of host reads).
            try:
                import ctypes as C

                ms8 = kzg.MultiKZGSettings(SETUP, [0] * 8, kzg.make_config(table_budget_gb=8))
                nmb = len(host_blobs) // BLOB
                L = kzg.lib()
                rows = {}
                # pageable: the bytes object as it is; pinned: a page-locked copy of it (and of the output)
                pin_in = C.create_string_buffer(host_blobs, len(host_blobs))
                pin_out = C.create_string_buffer(nmb * 48)
                for mode in ("pageable", "pinned"):
                    if mode == "pinned":
                        assert L.kzgamd_pin_host_buffer(pin_in, len(pin_in)) == 0 and L.kzgamd_pin_host_buffer(pin_out, len(pin_out)) == 0
                    src = host_blobs if mode == "pageable" else pin_in
                    out = C.create_string_buffer(nmb * 48) if mode == "pageable" else pin_out
                    assert L.kzgamd_blob_to_kzg_commitment_batch_multi(out, src, nmb, ms8.ptrs, 8) == 0
                    t0 = time.perf_counter()
                    for _ in range(3):
                        assert L.kzgamd_blob_to_kzg_commitment_batch_multi(out, src, nmb, ms8.ptrs, 8) == 0
                    t_c8 = (time.perf_counter() - t0) / 3
                    cm8 = out.raw
                    pout = C.create_string_buffer(nmb * 48)
                    assert L.kzgamd_compute_blob_kzg_proof_batch_multi(pout, src, cm8, nmb, ms8.ptrs, 8) == 0
                    t0 = time.perf_counter()
                    for _ in range(2):
                        assert L.kzgamd_compute_blob_kzg_proof_batch_multi(pout, src, cm8, nmb, ms8.ptrs, 8) == 0
                    t_p8 = (time.perf_counter() - t0) / 2
                    rows[mode] = {"commitments_per_s": nmb / t_c8, "proofs_per_s": nmb / t_p8,
                                  "host_read_GBps_commit": nmb * BLOB / t_c8 / 1e9, "host_read_GBps_prove": nmb * BLOB / t_p8 / 1e9,
                                  "equals_one_object": cm8 == b"".join(cms)}
                L.kzgamd_unpin_host_buffer(pin_in)
                L.kzgamd_unpin_host_buffer(pin_out)
                # 256 blobs, 32 per object: BASELINE configs[4]'s shape
                t0 = time.perf_counter()
                for _ in range(3):
                    ms8.proof_batch(host_blobs[:256 * BLOB], cm8[:256 * 48], 256)
                rows["configs4_256_blobs_ms"] = (time.perf_counter() - t0) / 3 * 1e3
                res["in_process_multi"]["eight_objects_one_gpu"] = dict(rows, objects=8, blobs_per_call=nmb,
                    path="8 settings objects on GPU 0 (table_budget_bytes = 8 GB each), kzgamd_*_batch_multi: one host thread per "
                         "object; the GPU is shared eight ways — the figure of interest is the host-side feed rate")
                ms8.close()
            except Exception as e:  # noqa: BLE001
                res["in_process_multi"]["eight_objects_one_gpu"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_extras:
        # ---- what a smaller commitment table costs (KzgAmdConfig.table_budget_bytes): device-resident commitments/s per HBM budget ----
        rows = []
        dblobs = make_blobs(torch, 1024, 99, dev)
        dout = torch.zeros(1024 * 48, dtype=torch.uint8, device=dev)
        dstat = torch.zeros(1024, dtype=torch.int32, device=dev)
        dscr = [torch.empty(1024 * BLOB, dtype=torch.uint8, device=dev) for _ in range(2)]
        sts = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for gb in (10, 20, 40, 80, 160):
            try:
                sb = kzg.KZGSettings.from_file(SETUP, kzg.make_config(table_budget_gb=gb))
                hi = kzg.PreparedMsm.info(type("H", (), {"handle": sb.msm_handle()})())
                for st_ in sts:
                    sb.reserve(1024, st_.cuda_stream)

                def run(reps):
                    for i in range(reps):
                        kzg.blob_to_kzg_commitment_device(dout.data_ptr(), dstat.data_ptr(), dscr[i % 2].data_ptr(), dblobs.data_ptr(), 1024,
                                                          sb, sts[i % 2].cuda_stream)
                    torch.cuda.synchronize()

                run(2)
                t0 = time.perf_counter()
                run(6)
                dt = time.perf_counter() - t0
                rows.append({"budget_gb": gb, "window_bits": hi["window_bits"], "rows": hi["rows"], "glv": hi["wide_glv"],
                             "wide_table": hi["wide_table"], "mixed_adds_per_scalar": hi["adds_per_scalar"],
                             "table_gb": (hi["rows"] * N * (1 << (hi["window_bits"] - 1)) * 128 / 1e9) if hi["wide_table"] else 0.0,
                             "commitments_per_s": 6 * 1024 / dt})
                sb.close()
            except Exception as e:  # noqa: BLE001
                rows.append({"budget_gb": gb, "error": repr(e)})
        res["throughput_vs_table_budget"] = {"rows": rows, "path": "device-resident commitments, batches of 1024 on two streams, a "
                                             "settings object per KzgAmdConfig.table_budget_bytes value (the default budget is 160 GB)"}

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()

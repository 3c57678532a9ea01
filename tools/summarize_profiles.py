#!/usr/bin/env python3
"""Turn the raw rocprofv3 output tools/collect_round_profiles.sh left under gpurun_out/ into the committed
summaries under profiles/ (round tag TAG, default r04): bench line, kernel statistics (headline 4-stream run, 1-stream run, NTT,
2^20 MSM timeline), the PMC counter files and r02_pmc_summary.json (what bench.py reads for roofline.traffic and
the VALU figures, labelled with this file as their source)."""
import collections
import csv
import glob
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get("TAG", "r04")
G = os.path.join(R, "gpurun_out")
P = os.path.join(R, "profiles")


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def by_grid(trace_csv, out_csv):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(trace_csv)):
        g = r.get("Grid_Size_X") or r.get("Grid_Size") or ""
        agg[(short(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out_csv, "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Grid_Size_X", "Calls", "AverageNs", "MinNs", "MaxNs"])
        for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, g, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
    return agg


def counters(pattern, match):
    out = {}
    for d in sorted(glob.glob(os.path.join(G, pattern))):
        for f in glob.glob(os.path.join(d, TAG + "_counter_collection.csv")):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for row in csv.DictReader(open(f)):
                if match in row["Kernel_Name"]:
                    agg[(short(row["Kernel_Name"]), row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k, v in agg.items():
                for cn, x in v.items():
                    out.setdefault("%s grid=%s" % k, {})[cn] = {"launches": len(x), "avg": sum(x) / len(x)}
            tag = os.path.basename(d)[4:]
            shutil.copy(f, os.path.join(P, "%s_pmc_%s_counter_collection.csv" % (TAG, tag)))
    return out


def durations(dirname, match):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(G, dirname, TAG + "_kernel_trace.csv")):
        for row in csv.DictReader(open(f)):
            if match in row["Kernel_Name"]:
                out["%s grid=%s" % (short(row["Kernel_Name"]), row.get("Grid_Size_X") or row.get("Grid_Size"))].append(
                    int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return out


def compact_counter_files(limit=200 * 1024):
    """Raw per-dispatch counter files above `limit` bytes are replaced by their aggregate per (kernel, grid, counter):
    launches, average, minimum, maximum — what every figure in DESIGN.md and the summary is computed from."""
    for f in glob.glob(os.path.join(P, TAG + "_pmc_*_counter_collection.csv")):
        if os.path.getsize(f) <= limit:
            continue
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(short(row["Kernel_Name"]), row["Grid_Size"], row.get("Workgroup_Size", ""), row["Counter_Name"])].append(float(row["Counter_Value"]))
        out = f.replace("_counter_collection.csv", "_counter_summary.csv")
        with open(out, "w") as o:
            w = csv.writer(o)
            w.writerow(["Kernel", "Grid_Size", "Workgroup_Size", "Counter_Name", "Launches", "Average", "Min", "Max"])
            for k, v in sorted(agg.items()):
                w.writerow(list(k) + [len(v), sum(v) / len(v), min(v), max(v)])
        os.remove(f)


def main():
    bench = json.load(open(os.path.join(G, "bench_final.json")))
    shutil.copy(os.path.join(G, "bench_final.json"), os.path.join(P, TAG + "_bench.json"))
    B = bench["config"]["blobs_per_batch"]
    cfg = bench["config"]
    for name in ("headline", "streams1", "ntt", "2p20", "cells", "proofs", "conc", "verify", "fftg1"):
        d = os.path.join(G, "prof_" + name)
        if os.path.exists(os.path.join(d, TAG + "_kernel_stats.csv")):
            shutil.copy(os.path.join(d, TAG + "_kernel_stats.csv"), os.path.join(P, "%s_%s_kernel_stats.csv" % (TAG, name)))
            by_grid(os.path.join(d, TAG + "_kernel_trace.csv"), os.path.join(P, "%s_%s_kernel_stats_by_grid.csv" % (TAG, name)))
    # ---- headline kernel ----
    fbw = {k: v for k, v in counters("pmc_[FWSG]*", "fbw_accum").items()}
    name = max(fbw, key=lambda k: fbw[k].get("SQ_INSTS_VALU", {}).get("avg", 0))
    ka = fbw[name]
    fetch_kb, write_kb = ka["FETCH_SIZE"]["avg"], ka["WRITE_SIZE"]["avg"]
    adds = cfg["mixed_adds_per_scalar"]
    summary = {
        "run": "rocprofv3 --pmc <one counter group per pass> --kernel-trace -- python bench.py --steps 3 --warmup 1 "
               "--batches-per-step 4 --streams 1 --no-cpu-baseline --no-extras",
        "batch": B, "window_bits": cfg["msm_window_bits"], "table_rows": cfg["table_rows"], "glv_split": cfg["glv_split"],
        "mixed_adds_per_scalar": adds, "kernel": name,
        "k_fbw_accum": {
            "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
            "fetch_bytes_corrected_x2": fetch_kb * 1024 * 2, "write_bytes": write_kb * 1024,
            "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
            "analytic_gather_bytes": B * adds * 4096 * 128 + B * 131072,
            "note": "gfx950 FETCH_SIZE tallies 128-B requests as 64 B (MI355X_MICROARCH.md HBM section), hence x2; the "
                    "corrected figure matches the analytic gather count (one 128-B table slot per mixed addition + the scalars)",
            **{cn: ka[cn]["avg"] for cn in ka if cn not in ("FETCH_SIZE", "WRITE_SIZE")},
        },
    }
    k = summary["k_fbw_accum"]
    k["valu_instructions_per_mixed_add"] = k["SQ_INSTS_VALU"] * 64 / (B * 4096 * adds)
    durs = durations("pmc_GRBM_GUI_ACTIVE", "fbw_accum").get(name, [])
    if durs and "GRBM_GUI_ACTIVE" in k:
        dur_s = sum(durs) / len(durs) * 1e-9
        xcd_cycles = k["GRBM_GUI_ACTIVE"] / 8
        k["kernel_s_in_grbm_pass"] = dur_s
        k["effective_clock_ghz"] = xcd_cycles / dur_s / 1e9
        k["valu_busy_frac"] = k["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * xcd_cycles)
        k["valu_note"] = ("valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): share of the "
                          "kernel's cycles in which a SIMD's VALU is executing, at the clock the chip sustained under this load")
    # ---- NTT: per bench shape, summed over the passes of one transform call (tools/ntt_bench.py runs 11 calls) ----
    def per_call(pattern, match, calls):
        """counter totals over every dispatch of `match` kernels in the run, divided by the number of calls"""
        tot = collections.defaultdict(float)
        launches = collections.Counter()
        meta = {}
        for d in sorted(glob.glob(os.path.join(G, pattern))):
            for f in glob.glob(os.path.join(d, TAG + "_counter_collection.csv")):
                seen = set()
                for row in csv.DictReader(open(f)):
                    if match in row["Kernel_Name"]:
                        tot[row["Counter_Name"]] += float(row["Counter_Value"])
                        seen.add(row["Dispatch_Id"])
                        # registers / scratch PER KERNEL NAME (a call is several kernels: the last row seen is not "the" kernel)
                        # (the profiler's VGPR_Count column is not the code object's .vgpr_count — 52 against 104-114 for
                        # k_ntt_pass — and is left out; DESIGN.md quotes the code-object notes)
                        meta["scratch"] = max(meta.get("scratch", 0), int(float(row.get("Scratch_Size") or 0)))
                for cn in set(r_["Counter_Name"] for r_ in csv.DictReader(open(f)) if match in r_["Kernel_Name"]):
                    launches[cn] = len(seen)
                shutil.copy(f, os.path.join(P, "%s_%s_counter_collection.csv" % (TAG, os.path.basename(d))))
        e = {cn: v / calls for cn, v in tot.items()}
        e["launches_per_call"] = (max(launches.values()) / calls) if launches else 0
        e.update(meta)
        return e

    def per_call_duration(dirname, match, calls):
        ds = []
        for f in glob.glob(os.path.join(G, dirname, TAG + "_kernel_trace.csv")):
            for row in csv.DictReader(open(f)):
                if match in row["Kernel_Name"]:
                    ds.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        return sum(ds) / calls * 1e-9 if ds else None

    for shape, key in (("4096x256", "n=4096 x 256"), ("1048576x1", "n=1048576 x 1")):
        e = per_call("pmc_ntt_%s_*" % shape, "k_ntt_pass", 11)
        if not e.get("SQ_INSTS_VALU"):
            continue
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_call"] = e["FETCH_SIZE"] * 1024 * 2 + e["WRITE_SIZE"] * 1024
        dur = per_call_duration("pmc_ntt_%s_GRBM_GUI_ACTIVE" % shape, "k_ntt_pass", 11)
        if dur and "GRBM_GUI_ACTIVE" in e:
            e["kernel_s_per_call_in_grbm_pass"] = dur
            e["effective_clock_ghz"] = e["GRBM_GUI_ACTIVE"] / 8 / dur / 1e9
            e["valu_busy_frac"] = e["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * e["GRBM_GUI_ACTIVE"] / 8)
        summary.setdefault("ntt", {})[key] = e
    # ---- one 2^20 variable-base MSM (4 calls per run): every kernel of the call together ----
    e = per_call("pmc_2p20_*", "k_", 4)
    # k_gen_points (the one-off point generator of the tool) is not part of the MSM
    # nor are the kernels of the handle's creation (bases in, subgroup test)
    for setup in ("k_gen_points", "k_points_in", "k_bases_in_g1", "k_copy_affpt"):
        g = per_call("pmc_2p20_*", setup, 4)
        for cn in list(e):
            if isinstance(e[cn], float) and cn != "launches_per_call":
                e[cn] -= g.get(cn, 0.0)
    if e.get("SQ_INSTS_VALU"):
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_call"] = e["FETCH_SIZE"] * 1024 * 2 + e["WRITE_SIZE"] * 1024
        e["note"] = "all kernels of one n = 2^20 MSM call (sort, accumulation, reduction); algorithmic bytes 128 * n = 134 MB"
        summary.setdefault("msm_sweep", {})[str(1 << 20)] = e
    # the other sizes of the sweep: HBM traffic only
    for logn in (16, 18, 21, 22):
        e2 = per_call("pmc_msm%d_*" % logn, "k_", 4)
        for setup in ("k_gen_points", "k_points_in", "k_bases_in_g1", "k_copy_affpt"):
            g = per_call("pmc_msm%d_*" % logn, setup, 4)
            for cn in list(e2):
                if isinstance(e2[cn], float) and cn != "launches_per_call":
                    e2[cn] -= g.get(cn, 0.0)
        if "FETCH_SIZE" in e2 and "WRITE_SIZE" in e2:
            e2["hbm_bytes_per_call"] = e2["FETCH_SIZE"] * 1024 * 2 + e2["WRITE_SIZE"] * 1024
            e2["note"] = "all kernels of one n = 2^%d MSM call; algorithmic bytes 128 * n" % logn
            summary.setdefault("msm_sweep", {})[str(1 << logn)] = e2
    # the sources the profiled kernels were built from: bench.py prints these counters only for the same sources
    import importlib.util
    spec = importlib.util.spec_from_file_location("kzg_bench", os.path.join(R, "bench.py"))
    bench_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench_mod)
    summary["source_sha256"] = {fam: bench_mod.source_hash(fam) for fam in bench_mod.KERNEL_SOURCES}
    # ---- the latency-bound kernels: counters per launch, by kernel and grid ----
    lat = {}
    for tag_dir, match in (("pmc_cells_*", "k_g1_st"), ("pmc_cells_*", "k_fbw_accum"), ("pmc_proofs_*", "k_quotient"),
                           ("pmc_proofs_*", "k_challenge_sha256"), ("pmc_proofs_*", "k_check_commitments")):
        for name, cs in counters(tag_dir, match).items():
            lat[name] = {cn: v["avg"] for cn, v in cs.items()}
            lat[name]["launches"] = max(v["launches"] for v in cs.values())
            if "FETCH_SIZE" in lat[name] and "WRITE_SIZE" in lat[name]:
                lat[name]["hbm_bytes_per_launch"] = lat[name]["FETCH_SIZE"] * 1024 * 2 + lat[name]["WRITE_SIZE"] * 1024
    if lat:
        summary["latency_bound_kernels"] = lat
    try:
        import subprocess
        summary["collected_at_commit"] = subprocess.check_output(["git", "-C", R, "rev-parse", "--short", "HEAD"]).decode().strip()
    except Exception:
        pass
    json.dump(summary, open(os.path.join(P, TAG + "_pmc_summary.json"), "w"), indent=1)
    compact_counter_files()
    print("k_fbw_accum: hbm bytes/launch %.3e  VALU/add %.0f  busy %.3f  clock %.2f GHz" % (
        k["hbm_bytes_per_launch"], k["valu_instructions_per_mixed_add"], k.get("valu_busy_frac", 0), k.get("effective_clock_ghz", 0)))
    for key, e in summary.get("ntt", {}).items():
        print(key, {x: (round(y, 3) if y < 100 else int(y)) for x, y in e.items() if x in ("valu_busy_frac", "hbm_bytes_per_call", "kernel_s_per_call_in_grbm_pass", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU")})


if __name__ == "__main__":
    main()

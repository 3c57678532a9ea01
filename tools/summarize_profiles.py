#!/usr/bin/env python3
"""Turn the raw rocprofv3 output that tools/collect_round_profiles.sh left under gpurun_out/ into the
committed summaries under profiles/ (round tag r01): bench line, kernel stats (overall and per launch
shape), PMC counter files and r01_pmc_summary.json (what bench.py reads for roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r01"


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    bench = json.load(open(os.path.join(R, "gpurun_out", "bench_final.json")))
    B = bench["config"]["blobs_per_gpu_per_step"]
    c, rows = bench["config"]["msm_window_bits"], bench["config"]["table_rows"]
    out = {}
    for d in sorted(glob.glob(os.path.join(R, "gpurun_out", "pmc_*/"))):
        for f in glob.glob(d + "*counter_collection.csv"):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for row in csv.DictReader(open(f)):
                agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k, v in agg.items():
                for cn, x in v.items():
                    xs = sorted(x)
                    full = [y for y in xs if y > 0.5 * xs[-1]]  # full-batch launches only
                    out.setdefault(k, {})[cn] = {"launches": len(full), "avg_full_batch": sum(full) / len(full)}
            tag = os.path.basename(d[:-1])[4:]
            shutil.copy(f, os.path.join(R, "profiles", "%s_pmc_%s_counter_collection.csv" % (TAG, tag)))
    # the instantiation that ran the full-batch launches (k_fbw_accum<4> at 1024 blobs): the one with the most work
    name = max((k for k in out if k.startswith("k_fbw_accum")), key=lambda k: out[k].get("FETCH_SIZE", {}).get("avg_full_batch", 0))
    ka = out[name]
    fetch_kb, write_kb = ka["FETCH_SIZE"]["avg_full_batch"], ka["WRITE_SIZE"]["avg_full_batch"]
    summary = {
        "run": "rocprofv3 --pmc <one counter group per pass> --kernel-trace -- python bench.py --steps 4 --warmup 1 "
               "--streams 1 --no-cpu-baseline --no-large",
        "batch": B, "window_bits": c, "table_rows": rows,
        "k_fbw_accum": {
            "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
            "fetch_bytes_corrected_x2": fetch_kb * 1024 * 2, "write_bytes": write_kb * 1024,
            "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
            "analytic_gather_bytes": B * rows * 4096 * 128 + B * 131072, "analytic_write_bytes": B * 4096 * 224,
            "note": "gfx950 FETCH_SIZE tallies 128-B requests as 64 B (MI355X_MICROARCH.md HBM section), hence x2; "
                    "the corrected figure matches the analytic gather count (one 128-B table slot per (window, scalar) "
                    "+ the scalars)",
            **{cn: ka[cn]["avg_full_batch"] for cn in ka if cn not in ("FETCH_SIZE", "WRITE_SIZE")},
        },
        "all": out,
    }
    # VALU utilisation from the counters alone (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count quad-cycles; effective
    # clock = GRBM_GUI_ACTIVE / kernel wall time, the counter being summed over the 8 XCDs): duration of the same
    # launches inside the GRBM pass
    durs = []
    for f in glob.glob(os.path.join(R, "gpurun_out", "pmc_GRBM_GUI_ACTIVE", "*kernel_trace.csv")):
        for row in csv.DictReader(open(f)):
            if short(row["Kernel_Name"]) == name:
                durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    if durs and "GRBM_GUI_ACTIVE" in ka and "SQ_ACTIVE_INST_VALU" in ka:
        full = [d for d in durs if d > 0.5 * max(durs)]
        dur_s = sum(full) / len(full) * 1e-9
        xcd_cycles = ka["GRBM_GUI_ACTIVE"]["avg_full_batch"] / 8
        summary["k_fbw_accum"]["kernel_s_in_grbm_pass"] = dur_s
        summary["k_fbw_accum"]["effective_clock_ghz"] = xcd_cycles / dur_s / 1e9
        summary["k_fbw_accum"]["valu_busy_frac"] = ka["SQ_ACTIVE_INST_VALU"]["avg_full_batch"] * 4 / (1024 * xcd_cycles)
        summary["k_fbw_accum"]["valu_note"] = ("valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): "
                                               "share of the kernel's cycles in which a SIMD's VALU is executing, at the clock the chip "
                                               "sustained under this load")
    json.dump(summary, open(os.path.join(R, "profiles", TAG + "_pmc_summary.json"), "w"), indent=1)
    shutil.copy(os.path.join(R, "gpurun_out", "prof_final", "r01_kernel_stats.csv"),
                os.path.join(R, "profiles", TAG + "_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(R, "gpurun_out", "bench_final.json"), os.path.join(R, "profiles", TAG + "_bench.json"))
    rowsx = list(csv.DictReader(open(os.path.join(R, "gpurun_out", "prof_final", "r01_kernel_trace.csv"))))
    agg = collections.defaultdict(list)
    for r in rowsx:
        g = r.get("Grid_Size_X") or r.get("Grid_Size") or ""
        agg[(short(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(R, "profiles", TAG + "_bench_kernel_stats_by_grid.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Grid_Size_X", "Calls", "AverageNs", "MinNs", "MaxNs"])
        for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, g, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
    k = summary["k_fbw_accum"]
    print("hbm bytes/launch %.3e  SQ_INSTS_VALU %.3e" % (k["hbm_bytes_per_launch"], k["SQ_INSTS_VALU"]))
    for (kk, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(kk[:30].ljust(30), str(g).rjust(9), len(v), round(sum(v) / len(v) / 1e3, 1), "us")


if __name__ == "__main__":
    main()

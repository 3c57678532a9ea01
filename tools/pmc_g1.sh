# counters of the G1 stage kernels at low and high occupancy (same kernel form): bash tools/pmc_g1.sh [variant]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-0,0,65536}
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE" "SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$tag -o x -- python $R/tools/time_g1.py --only-cells $V > /tmp/pmc_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_g1_stage" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("::")[-1].split("(")[0], int(r["Grid_Size"]))
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("/tmp/pmc_GRBM_GUI_ACTIVE/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_g1_stage" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("::")[-1].split("(")[0], int(r.get("Grid_Size_X") or r.get("Grid_Size")))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(agg):
    c = {cn: sum(v) / len(v) for cn, v in agg[k].items()}
    d = sum(dur[k]) / max(1, len(dur[k])) * 1e-9
    line = "%-22s grid %8d  avg %7.1f us" % (k[0], k[1], d * 1e6)
    if "GRBM_GUI_ACTIVE" in c and d:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        line += "  clock %.2f GHz" % (cyc / d / 1e9)
        if "SQ_ACTIVE_INST_VALU" in c:
            line += "  valu_busy(all SIMDs) %.3f  per-wave-SIMD %.3f" % (c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), c["SQ_ACTIVE_INST_VALU"] * 4 / (min(1024, c.get("SQ_WAVES", 1)) * cyc))
    for cn in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_IFETCH", "SQ_INSTS_VMEM_RD", "FETCH_SIZE", "WRITE_SIZE"):
        if cn in c: line += "  %s %.3g" % (cn.replace("SQ_", ""), c[cn])
    print(line)
PY

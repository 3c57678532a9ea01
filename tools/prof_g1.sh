# kernel statistics of the G1 transform stages per stage form (run on the GPU box): bash tools/prof_g1.sh [--only-cells]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:---only-fft}
for v in ${VARIANTS:-0,0,0 0,0,32768 0,32768,65536}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o x -- python $R/tools/time_g1.py $MODE $v > /tmp/log_$v.txt 2>&1
  echo "== $v"; grep "2\^15\|n=256\|n=128\|n=64" /tmp/log_$v.txt
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/prof_$v/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_g1" in r["Name"] or "fbw_accum" in r["Name"]:
        print("  %-40s calls %5s avg %10.1f us  min %10.1f  max %10.1f" % (r["Name"].split("(")[-2 if r["Name"].startswith("void") else 1][-40:] if False else r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done

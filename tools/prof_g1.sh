# per-dispatch durations of the G1 transform stages of ONE fft_g1 of 2^15 points per stage form (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:---only-fft}
for v in ${VARIANTS:-0,0,0 0,0,65536 0,65536,65536}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o x -- python $R/tools/time_g1.py $MODE $v > /tmp/log_$v.txt 2>&1
  echo "== $v"; grep "2\^15\|n=256\|n=128\|n=64" /tmp/log_$v.txt
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/prof_$v/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_g1_st" in r["Kernel_Name"] or "k_g1_scale" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = max(int(r.get("Grid_Size_X") or r.get("Grid_Size")) for r in rows)
seq = [(r["Kernel_Name"].split("::")[-1].split("(")[0], int(r.get("Grid_Size_X") or r.get("Grid_Size")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
# the last 2 x 15 stage launches are the timed forward / inverse transforms of 2^15 points
tail = [s for s in seq if s[1] >= big // 4][-34:] if "$MODE" == "--only-fft" else seq[-28:]
print("  " + " ".join("%s:%d" % (n[5:14], d) for n, g, d in tail))
PY
done

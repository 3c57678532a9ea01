cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spl in 0 4; do
KZGAMD_TUNING="spl=$spl" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$spl -o x -- python $R/tools/time_g1.py --only-cells 4096,16384,32768 > /tmp/log_$spl.txt 2>&1
echo "== SPL=$spl"; grep "n=256" /tmp/log_$spl.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/prof_$spl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last cells call (n = 256): kernels after the last k_blob_to_fr_brp
idx = max(i for i, r in enumerate(rows) if "k_blob_to_fr_brp" in r["Kernel_Name"])
agg = collections.OrderedDict()
for r in rows[idx:]:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
t0, t1 = int(rows[idx]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows[idx:])
print("  span %.2f ms" % ((t1 - t0) / 1e6))
for k, (c, d) in agg.items(): print("  %-34s x%-3d %9.1f us" % (k, c, d))
PY
done

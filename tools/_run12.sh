cd /tmp && export TMPDIR=/tmp
for c in 15 14; do
KZGAMD_TUNING="window=$c" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r12/prof_c$c -o t -- python /root/repo/tools/prof_2p20.py 16 > /root/repo/gpurun_out/r12/prof_c$c.log 2>&1
echo "== c=$c"; python - <<PY
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r12/prof_c$c/t_kernel_stats.csv')))
for r in rows[:20]:
    n=r['Name']
    if 'gen_points' in n or 'bases_in' in n or 'at::' in n: continue
    print("%-70s calls=%s avg=%8.1f us" % (n.replace('(anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_2p20
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_2p20 -o r01 -- python $R/tools/prof_2p20.py > $R/gpurun_out/prof_2p20.log 2>&1
python3 - <<'PY'
import csv,os
R=os.environ['GRAFT_REPO_ROOT']
rows=list(csv.DictReader(open(R+'/gpurun_out/prof_2p20/r01_kernel_stats.csv')))
for r in rows[:14]:
    print(r['Name'].replace('(anonymous namespace)::','')[:50].ljust(50), r['Calls'], round(float(r['AverageNs'])/1e3,1),'us avg', r['Percentage'])
PY

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_proof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_proof -o r01 -- python $R/tools/prof_proof.py > $R/gpurun_out/prof_proof.log 2>&1
python3 - <<'PY'
import csv,os,collections
R=os.environ['GRAFT_REPO_ROOT']
rows=list(csv.DictReader(open(R+'/gpurun_out/prof_proof/r01_kernel_trace.csv')))
agg=collections.defaultdict(list)
for r in rows:
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0]
    agg[(k, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for (k,g),v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    if 'fbw_chain' in k or 'fbw_affine' in k: continue
    print(k[:28].ljust(28), 'grid', str(g).rjust(9), 'calls', len(v), 'avg us', round(sum(v)/len(v),1))
PY

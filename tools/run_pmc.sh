mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_$tag
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o r01 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-large > $R/gpurun_out/pmc_$tag.log 2>&1
  echo "== $pass"; tail -1 $R/gpurun_out/pmc_$tag.log | cut -c1-200
  ls $R/gpurun_out/pmc_$tag | head
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc_*/')):
    for f in glob.glob(d+'*counter_collection.csv'):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'][:50]
            agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in agg.items():
            if 'fbw_accum' in k or 'blocksum' in k or 'k_final' in k or 'blob_to' in k:
                print(os.path.basename(d[:-1]), k, {c:(len(x), sum(x)/len(x)) for c,x in v.items()})
PY

mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_r01_c.json 2> gpurun_out/bench_r01_c.err; tail -2 gpurun_out/bench_r01_c.err; cat gpurun_out/bench_r01_c.json
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_c
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c -o r01 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_c.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_c.log | cut -c1-300

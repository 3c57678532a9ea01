mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-large > $GRAFT_REPO_ROOT/gpurun_out/prof_b.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_b.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_b -type f | head
cat $(find $GRAFT_REPO_ROOT/gpurun_out/prof_b -name "*kernel_stats.csv" | head -1)

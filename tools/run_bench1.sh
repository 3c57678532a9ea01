set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r01_a.json 2> gpurun_out/bench_r01_a.err; tail -5 gpurun_out/bench_r01_a.err; cat gpurun_out/bench_r01_a.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_a -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_a.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_a.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_a -name "*stats*" | head

# full GPU pass (run on the MI355X box through gpurun): parity tests, smoke, bench, rocprofv3 kernel stats, PMC passes
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -2 gpurun_out/bench_final.err; cat gpurun_out/bench_final.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_final $R/gpurun_out/pmc_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o r01 -- python $R/bench.py --steps 5 --warmup 1 --streams 1 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o r01 -- python $R/bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --no-large > $R/gpurun_out/pmc_$tag.log 2>&1
done
ls $R/gpurun_out

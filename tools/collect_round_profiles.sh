# Round-2 profile collection (run on the MI355X box through gpurun).  rocprofv3 passes, each in its own run:
#   1. kernel trace + stats of the HEADLINE configuration (bench.py defaults: 4 streams) and of --streams 1
#   2. PMC passes over the headline kernel (--streams 1, so that a launch runs alone): HBM traffic, VALU counters, clock
#   3. kernel trace + stats and PMC passes of the NTT kernels (tools/ntt_bench.py)
#   4. kernel trace of one 2^20 MSM (tools/prof_2p20.py)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 900 python $R/bench.py > $R/gpurun_out/bench_final.json 2> $R/gpurun_out/bench_final.err; tail -2 $R/gpurun_out/bench_final.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_* $R/gpurun_out/pmc_*
B="python $R/bench.py --steps 3 --warmup 1 --batches-per-step 4 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_headline -o r02 -- $B > $R/gpurun_out/prof_headline.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_streams1 -o r02 -- $B --streams 1 > $R/gpurun_out/prof_streams1.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o r02 -- $B --streams 1 > $R/gpurun_out/pmc_$tag.log 2>&1
done
NB="python $R/tools/ntt_bench.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ntt -o r02 -- $NB > $R/gpurun_out/prof_ntt.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ntt_$tag -o r02 -- $NB > $R/gpurun_out/pmc_ntt_$tag.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_2p20 -o r02 -- python $R/tools/prof_2p20.py > $R/gpurun_out/prof_2p20.log 2>&1
ls $R/gpurun_out | head -40
# 5. kernel trace of the FK20 cell-proof batches (tools/time_cells.py)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cells -o r02 -- python $R/tools/time_cells.py 256 > $R/gpurun_out/prof_cells.log 2>&1

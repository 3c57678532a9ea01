# Round profile collection (TAG below) (run on the MI355X box through gpurun).  rocprofv3 passes, each in its own run:
#   1. kernel trace + stats of the HEADLINE configuration (bench.py defaults: 4 streams) and of --streams 1
#   2. PMC passes over the headline kernel (--streams 1, so that a launch runs alone): HBM traffic, VALU counters, clock
#   3. kernel trace + stats and PMC passes of the NTT kernels (tools/ntt_bench.py)
#   4. kernel trace of one 2^20 MSM (tools/prof_2p20.py)
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r05}
mkdir -p $R/gpurun_out
timeout 1500 python $R/bench.py > $R/gpurun_out/bench_final.json 2> $R/gpurun_out/bench_final.err; tail -2 $R/gpurun_out/bench_final.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_* $R/gpurun_out/pmc_*
B="python $R/bench.py --steps 3 --warmup 1 --batches-per-step 4 --no-cpu-baseline --no-extras --no-counters"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_headline -o $TAG -- $B > $R/gpurun_out/prof_headline.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_streams1 -o $TAG -- $B --streams 1 > $R/gpurun_out/prof_streams1.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o $TAG -- $B --streams 1 > $R/gpurun_out/pmc_$tag.log 2>&1
done
NB="python $R/tools/ntt_bench.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ntt -o $TAG -- $NB 20 > $R/gpurun_out/prof_ntt.log 2>&1
# NTT counters, one bench shape per run (both shapes launch 256-tile grids: they cannot be told apart by grid size)
for shape in 4096x256 1048576x1; do
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ntt_${shape}_$tag -o $TAG -- $NB 20 $shape > $R/gpurun_out/pmc_ntt_${shape}_$tag.log 2>&1
done
done
# counters of one 2^20 variable-base MSM (msm_sweep[].roofline.traffic)
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_2p20_$tag -o $TAG -- python $R/tools/prof_2p20.py > $R/gpurun_out/pmc_2p20_$tag.log 2>&1
done
# HBM traffic of the other sizes of the sweep (msm_sweep[*].roofline.traffic): FETCH_SIZE / WRITE_SIZE passes only
for logn in 16 18 21 22; do
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_msm${logn}_$pass -o $TAG -- python $R/tools/prof_2p20.py $logn > $R/gpurun_out/pmc_msm${logn}_$pass.log 2>&1
done
done
# kernel stats of the device-resident proof pipeline (k_challenge_sha256, k_quotient, the MSM)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_proofs -o $TAG -- python $R/tools/prof_proof_dev.py > $R/gpurun_out/prof_proofs.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_2p20 -o $TAG -- python $R/tools/prof_2p20.py > $R/gpurun_out/prof_2p20.log 2>&1
ls $R/gpurun_out | head -40
# 5. kernel trace of the FK20 cell-proof batches (tools/time_cells.py)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cells -o $TAG -- python $R/tools/time_g1.py --only-cells 4096,16384,32768 > $R/gpurun_out/prof_cells.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fftg1 -o $TAG -- python $R/tools/time_g1.py --only-fft 4096,16384,32768 > $R/gpurun_out/prof_fftg1.log 2>&1
# counters of the latency-bound kernels: the G1 transform stages (FK20 batches of 16 .. 256 blobs) and the proof pipeline's
# k_quotient / k_challenge_sha256 / k_check_commitments
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cells_$tag -o $TAG -- python $R/tools/time_g1.py --only-cells 4096,16384,32768 > $R/gpurun_out/pmc_cells_$tag.log 2>&1
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_proofs_$tag -o $TAG -- python $R/tools/prof_proof_dev.py > $R/gpurun_out/pmc_proofs_$tag.log 2>&1
done
# 6. kernel stats of 16 concurrent callers on one settings object (lane batches: k_blob_to_scalars_ptrs, k_gather_blobs,
#    k_quotient_a/b, k_blocksum_hybrid)
LD_LIBRARY_PATH=$R/rust-kzg_amd/csrc:/opt/rocm/lib KZGAMD_FBW_MAX_GB=100 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_conc -o $TAG -- $R/tools/concurrent_bench $R/tests/golden/trusted_setup.txt 0.5 16 > $R/gpurun_out/prof_conc.log 2>&1
tail -1 $R/gpurun_out/prof_conc.log
# 7. kernel stats of the verification entry points (k_decode_check_g1, k_affpts_in_g1_wide, k_vcell_agg / _interp, the two-row MSM)
KZGAMD_FBW_MAX_GB=100 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_verify -o $TAG -- python $R/tools/time_verify.py > $R/gpurun_out/prof_verify.log 2>&1
tail -4 $R/gpurun_out/prof_verify.log

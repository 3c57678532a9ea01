# rocprofv3 counter passes over the headline kernel, GLV wide table vs the 255-bit table (same box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for g in 1 0; do
  rm -rf $R/gpurun_out/fbw_pmc_g$g $R/gpurun_out/fbw_pmc2_g$g
  KZGAMD_TUNING="fbw_glv=$g" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/fbw_pmc_g$g -o p -- python $R/bench.py --steps 2 --warmup 1 --streams 1 --batches-per-step 2 --no-extras --no-cpu-baseline > $R/gpurun_out/fbw_pmc_g$g.log 2>&1
  KZGAMD_TUNING="fbw_glv=$g" rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/fbw_pmc2_g$g -o p -- python $R/bench.py --steps 2 --warmup 1 --streams 1 --batches-per-step 2 --no-extras --no-cpu-baseline > $R/gpurun_out/fbw_pmc2_g$g.log 2>&1
done
ls $R/gpurun_out | grep fbw

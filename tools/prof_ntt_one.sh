# PMC passes over one NTT shape: tools/prof_ntt_one.sh <tag> <scale> <n>x<batch>
R=$GRAFT_REPO_ROOT
TAG=$1; SCALE=$2; SHAPE=$3
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$TAG*
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_1 -o p -- python $R/tools/ntt_bench.py $SCALE $SHAPE > $R/gpurun_out/pmc_${TAG}_1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_2 -o p -- python $R/tools/ntt_bench.py $SCALE $SHAPE > $R/gpurun_out/pmc_${TAG}_2.log 2>&1

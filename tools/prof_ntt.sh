# rocprofv3 passes over tools/ntt_bench.py: kernel stats, then SQ counters (own passes, no extra trace domains)
R=$GRAFT_REPO_ROOT
TAG=${1:-ntt}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -o s -- python $R/tools/ntt_bench.py > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_pmc1 -o p -- python $R/tools/ntt_bench.py > $R/gpurun_out/prof_${TAG}_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_pmc2 -o p -- python $R/tools/ntt_bench.py > $R/gpurun_out/prof_${TAG}_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -o p -- python $R/tools/ntt_bench.py > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_write -o p -- python $R/tools/ntt_bench.py > $R/gpurun_out/prof_${TAG}_write.log 2>&1
find $R/gpurun_out/prof_${TAG}* -name "*.csv" | head -20

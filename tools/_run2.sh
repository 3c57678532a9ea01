set -x
cd /root/repo
mkdir -p gpurun_out/r2
export KZGAMD_TEST_FLAVOURS=product
timeout 600 python -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "several_large or 2p20 or variable_base" 2>&1 | tail -15 > gpurun_out/r2/pytest_msm.log
for a in "20 4" "20 2" "20 8" "16 4" "18 4"; do
  set -- $a
  timeout 300 python tools/ab_batched.py $1 $2 sub_streams=0 sub_streams=2 sub_streams=3 > gpurun_out/r2/ab_$1_$2.log 2>&1
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r2/trace_b4 -o t --output-format csv -- python /root/repo/tools/ab_batched.py 20 4 sub_streams=3 trace > /root/repo/gpurun_out/r2/trace_b4.log 2>&1)
timeout 900 python -m pytest tests/test_multigpu.py -x -q -m gpu -k "eight or rccl" --durations=5 2>&1 | tail -30 > gpurun_out/r2/pytest_multi.log
timeout 900 python -m pytest tests/test_lifecycle_gpu.py -x -q -m gpu -s --durations=5 2>&1 | tail -40 > gpurun_out/r2/pytest_life.log
ls gpurun_out/r2

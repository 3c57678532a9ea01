set -x
cd /root/repo
mkdir -p gpurun_out/r6
export KZGAMD_TEST_FLAVOURS=product
timeout 900 python -m pytest tests/test_msm_gpu.py tests/test_ckzg_gpu.py tests/test_concurrent_handles_gpu.py -x -q -m gpu -k "few_commitments or trusted_setup or several_large or every_size or concurrent or sixteen or vectors" --durations=5 2>&1 | tail -15 > gpurun_out/r6/pytest_msm.log
for t in "" "quad_accum_max=0;no_wide_tree=1"; do
  echo "== $t" >> gpurun_out/r6/single.log
  KZGAMD_TUNING="$t" timeout 300 python tools/time_single.py 2>&1 | head -3 >> gpurun_out/r6/single.log
done
timeout 300 python tools/time_batches.py 2>&1 | head -12 > gpurun_out/r6/batches.log
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r6/trace_single -o t --output-format csv -- python /root/repo/tools/prof_single.py > /root/repo/gpurun_out/r6/trace_single.log 2>&1)
KZGAMD_SOAK_SECONDS=180 timeout 600 python -m pytest tests/test_lifecycle_gpu.py -x -q -m gpu -s -k soak 2>&1 | tail -8 > gpurun_out/r6/soak.log
ls gpurun_out/r6

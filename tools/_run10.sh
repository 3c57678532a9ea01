set -x
cd /root/repo
mkdir -p gpurun_out/r10
timeout 1500 python -m pytest tests/test_msm_gpu.py tests/test_ckzg_gpu.py tests/test_concurrent_handles_gpu.py -x -q -m gpu > gpurun_out/r10/tests.log 2>&1
tail -3 gpurun_out/r10/tests.log
timeout 300 python tools/time_single.py > gpurun_out/r10/single.log 2>&1; head -5 gpurun_out/r10/single.log
timeout 300 python tools/time_batches.py > gpurun_out/r10/batches.log 2>&1; head -12 gpurun_out/r10/batches.log
timeout 300 python tools/time_2p20.py > gpurun_out/r10/sweep.log 2>&1; tail -1 gpurun_out/r10/sweep.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r10/prof_2p16 -o t -- python /root/repo/tools/prof_2p20.py 16 > /root/repo/gpurun_out/r10/prof_2p16.log 2>&1
grep -h "tile_sums\|_wide" /root/repo/gpurun_out/r10/prof_2p16/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-

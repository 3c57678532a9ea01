set -x
cd /root/repo
mkdir -p gpurun_out/r8
timeout 1200 python -m pytest tests/test_msm_gpu.py -x -q -m gpu --durations=5 > gpurun_out/r8/msm_tests.log 2>&1
tail -4 gpurun_out/r8/msm_tests.log
for logn in 16 20; do timeout 300 python tools/ab_2p20.py tile_quad=0 $logn > gpurun_out/r8/ab_quad_$logn.log 2>&1; cat gpurun_out/r8/ab_quad_$logn.log; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r8/prof_2p16 -o t -- python /root/repo/tools/prof_2p20.py 16 > /root/repo/gpurun_out/r8/prof_2p16.log 2>&1
grep -h "tile_sums" /root/repo/gpurun_out/r8/prof_2p16/t_kernel_stats.csv

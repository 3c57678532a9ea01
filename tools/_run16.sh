cd /root/repo; mkdir -p gpurun_out/r16
timeout 600 python -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "tile_tree or variants or 2p20_matches" > gpurun_out/r16/tests.log 2>&1; tail -2 gpurun_out/r16/tests.log
for logn in 16 20; do timeout 300 python tools/ab_2p20.py wide_spread=0 $logn 2>&1 | tail -2; done | tee gpurun_out/r16/ab_spread.log
for v in 10 40; do KZGAMD_TUNING="wide_spread=$v" timeout 300 python tools/time_2p20.py 2>&1 | tail -1; done | tee -a gpurun_out/r16/ab_spread.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r16/prof_2p16 -o t -- python /root/repo/tools/prof_2p20.py 16 > /root/repo/gpurun_out/r16/prof_2p16.log 2>&1
grep -h "_wide" /root/repo/gpurun_out/r16/prof_2p16/t_kernel_stats.csv | awk -F'","' '{print substr($1,1,60), $2, $4}'

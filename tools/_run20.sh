cd /root/repo; mkdir -p gpurun_out/r20
timeout 900 python -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "sorts_ahead or several_large or 2p20_matches" > gpurun_out/r20/tests.log 2>&1; tail -3 gpurun_out/r20/tests.log
timeout 600 python tools/ab_batched.py 20 4 default sort_ahead=0 2>&1 | tail -2 | tee gpurun_out/r20/ab_20_4.log
timeout 600 python tools/ab_batched.py 20 8 default sort_ahead=0 2>&1 | tail -2 | tee gpurun_out/r20/ab_20_8.log
timeout 600 python tools/ab_batched.py 18 8 default sort_ahead=0 2>&1 | tail -2 | tee gpurun_out/r20/ab_18_8.log
timeout 600 python tools/ab_batched.py 21 4 default sort_ahead=0 2>&1 | tail -2 | tee gpurun_out/r20/ab_21_4.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/r20/trace -o t -- python /root/repo/tools/ab_batched.py 20 4 default trace > /root/repo/gpurun_out/r20/trace.log 2>&1
cd /root/repo; python tools/timeline.py gpurun_out/r20/trace 2>&1 | tail -50 > gpurun_out/r20/timeline.txt; tail -5 gpurun_out/r20/timeline.txt

cd /root/repo; mkdir -p gpurun_out/r21
timeout 900 python -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "sorts_ahead or several_large" > gpurun_out/r21/tests.log 2>&1; tail -2 gpurun_out/r21/tests.log
timeout 600 python tools/ab_batched.py 20 4 default sort_ahead=0 "sub_prio=0" 2>&1 | tail -3 | tee gpurun_out/r21/ab_20_4.log
timeout 600 python tools/ab_batched.py 20 8 default sort_ahead=0 "sub_prio=0" 2>&1 | tail -3 | tee gpurun_out/r21/ab_20_8.log
timeout 600 python tools/ab_batched.py 18 8 default sort_ahead=0 "sub_prio=0" 2>&1 | tail -3 | tee gpurun_out/r21/ab_18_8.log

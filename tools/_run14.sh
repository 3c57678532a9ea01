cd /root/repo; mkdir -p gpurun_out/r14
timeout 900 python -m pytest tests/test_ntt_gpu.py -x -q -m gpu > gpurun_out/r14/ntt_tests.log 2>&1; tail -2 gpurun_out/r14/ntt_tests.log
for i in 1 2; do timeout 300 python tools/ntt_bench.py 20 2>&1 | tail -3 | cut -c1-900; done | tee gpurun_out/r14/ntt_bench.log

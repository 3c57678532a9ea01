mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_msm_gpu.py tests/test_ckzg_gpu.py -x -q 2>&1 | tail -3
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r01_b.json 2> gpurun_out/bench_r01_b.err; tail -3 gpurun_out/bench_r01_b.err; cat gpurun_out/bench_r01_b.json

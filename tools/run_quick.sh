timeout 1200 python -m pytest tests/test_ckzg_gpu.py -x -q 2>&1 | tail -3
python tools/extra_bench.py 2>/dev/null | tail -12

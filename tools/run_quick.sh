timeout 900 python -m pytest tests/test_msm_gpu.py tests/test_ckzg_gpu.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-large 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), 'commit/s  ms/step', round(r['ms_per_step'],2), 'accum', round(r['roofline']['kernel_ms'],2), 'valu', round(r['valu']['frac'],3))"

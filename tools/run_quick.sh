timeout 900 python -m pytest tests/test_msm_gpu.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), 'commit/s  ms/step', round(r['ms_per_step'],2), '2^20:', round(r['msm_2p20_ms'],2))"

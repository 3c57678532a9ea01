timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-large 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), 'commit/s  ms/step', round(r['ms_per_step'],2), 'accum', round(r['roofline']['kernel_ms'],2), 'pipeline', round(r['roofline']['pipeline_ms'],2))"; done
python tools/extra_bench.py 2>/dev/null | tail -12

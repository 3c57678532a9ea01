for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-large 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), 'commit/s  ms/step', round(r['ms_per_step'],2), 'accum', round(r['roofline']['kernel_ms'],2), 'pipeline', round(r['roofline']['pipeline_ms'],2))"
done
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4

for spl in 1 2 4; do
  KZGAMD_SPL=$spl timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-large 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('spl=$spl', round(r['value']), 'commit/s  ms/step', round(r['ms_per_step'],2), 'accum', round(r['roofline']['kernel_ms'],2))"
done

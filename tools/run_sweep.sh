for B in 256 512 1024 2048; do
for c in 13 14; do
  KZGAMD_WINDOW_PREPARED=$c KZGAMD_FBW_MAX_GB=120 timeout 300 python bench.py --steps 10 --warmup 2 --batch $B --no-cpu-baseline --no-large 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('B=$B c=$c', round(r['value']), 'commit/s  ms/step', round(r['ms_per_step'],2), 'accum', round(r['roofline']['kernel_ms'],2))"
done; done

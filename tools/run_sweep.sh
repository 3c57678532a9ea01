for c in 10 11 12 13; do
  KZGAMD_WINDOW_PREPARED=$c timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-large 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c=$c', round(r['value']), 'commit/s  ms/step', round(r['ms_per_step'],2), 'accum', round(r['roofline']['kernel_ms'],2))"
done
bash tools/run_prof.sh 2>&1 | cut -c1-150 | grep -v "^W2026"

echo "outline:"; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-large 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), 'commit/s  ms/step', round(r['ms_per_step'],2), 'accum', round(r['roofline']['kernel_ms'],2))"
cp rust-kzg_amd/csrc/libkzg_mi355x.so /tmp/keep.so; cp gpurun_lib_inline.so rust-kzg_amd/csrc/libkzg_mi355x.so
echo "inline:"; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-large 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), 'commit/s  ms/step', round(r['ms_per_step'],2), 'accum', round(r['roofline']['kernel_ms'],2))"

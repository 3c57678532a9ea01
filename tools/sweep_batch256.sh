# ms per 256-blob kzgamd_compute_blob_kzg_proof_batch call (host buffers) against the pipeline's chunk sizes
export KZGAMD_FBW_MAX_GB=100
for cfg in "" "KZGAMD_TUNING=prove_first=64" "KZGAMD_TUNING=prove_first=16" "KZGAMD_TUNING=prove_first=32;prove_chunk=96" "KZGAMD_TUNING=prove_first=32;prove_chunk=112" "KZGAMD_TUNING=prove_first=48;prove_chunk=104" "KZGAMD_TUNING=prove_chunk=128;prove_first=128"; do
  echo "== $cfg"
  env $cfg python tools/trace_batch256.py ${1:-256} 2>&1 | grep call | tail -3
done

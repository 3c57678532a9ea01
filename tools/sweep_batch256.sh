# ms per 256-blob kzgamd_compute_blob_kzg_proof_batch call (host buffers) against the pipeline's chunk sizes
export KZGAMD_FBW_MAX_GB=100
for cfg in "" "KZGAMD_PROVE_FIRST=64" "KZGAMD_PROVE_FIRST=16" "KZGAMD_PROVE_FIRST=32 KZGAMD_PROVE_CHUNK=96" "KZGAMD_PROVE_FIRST=32 KZGAMD_PROVE_CHUNK=112" "KZGAMD_PROVE_FIRST=48 KZGAMD_PROVE_CHUNK=104" "KZGAMD_PROVE_CHUNK=128 KZGAMD_PROVE_FIRST=128"; do
  echo "== $cfg"
  env $cfg python tools/trace_batch256.py ${1:-256} 2>&1 | grep call | tail -3
done

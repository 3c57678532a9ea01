# ms per kzgamd_blob_to_kzg_commitment_batch call (host buffers) against the pipeline's chunk sizes
export KZGAMD_FBW_MAX_GB=100
N=${1:-256}
for cfg in "" "KZGAMD_TUNING=commit_first=64;commit_chunk=192" "KZGAMD_TUNING=commit_first=32;commit_chunk=224" "KZGAMD_TUNING=commit_first=64;commit_chunk=96" "KZGAMD_TUNING="commit_first=$N"" "KZGAMD_COMMIT_FIRST=$((N/4)) KZGAMD_COMMIT_CHUNK=$((N/4))" "KZGAMD_COMMIT_FIRST=$((N/8)) KZGAMD_COMMIT_CHUNK=$((N*7/16))"; do
  echo "== n=$N $cfg"
  env $cfg python tools/trace_commit256.py $N 2>&1 | grep call | tail -3 | tr '\n' ' '; echo
done

# ms per kzgamd_blob_to_kzg_commitment_batch call (host buffers) against the pipeline's chunk sizes.
# Every configuration is a KZGAMD_TUNING string (the only switch the library reads for these keys; csrc/config.h).
export KZGAMD_FBW_MAX_GB=100
N=${1:-256}
for tuning in "" "commit_first=64;commit_chunk=192" "commit_first=32;commit_chunk=224" "commit_first=64;commit_chunk=96" \
              "commit_first=$N" "commit_first=$((N/4));commit_chunk=$((N/4))" "commit_first=$((N/8));commit_chunk=$((N*7/16))"; do
  echo "== n=$N KZGAMD_TUNING='$tuning'"
  KZGAMD_TUNING="$tuning" python tools/trace_commit256.py $N 2>&1 | grep call | tail -3 | tr '\n' ' '; echo
done

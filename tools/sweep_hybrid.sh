# host-buffer batch calls of 1 .. 256 blobs (tools/time_batches.py) against the fold switches:
# hybrid_max (largest batch k_blocksum_hybrid takes), wide_fold_max (largest batch k_wide_fold64 takes;
# negative: never)
export KZGAMD_FBW_MAX_GB=100
for cfg in "KZGAMD_TUNING=wide_fold_max=1" "KZGAMD_TUNING=wide_fold_max=-1" "KZGAMD_TUNING=wide_fold_max=4;hybrid_max=16"; do
  echo "== $cfg"
  env $cfg python tools/time_batches.py 2>&1 | grep "n="
done

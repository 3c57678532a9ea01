export LD_LIBRARY_PATH=rust-kzg_amd/csrc:/opt/rocm/lib KZGAMD_FBW_MAX_GB=100
for cfg in "4 8" "8 8" "16 8" "16 16" "8 16"; do
  set -- $cfg
  echo "WIDE_FOLD_MAX=$1 SPL1_MAX=$2"
  KZGAMD_TUNING="wide_fold_max=$1;spl1_max=$2" timeout 100 tools/concurrent_bench tests/golden/trusted_setup.txt 0.6 16
  echo
done

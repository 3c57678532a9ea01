export LD_LIBRARY_PATH=rust-kzg_amd/csrc:/opt/rocm/lib
for cfg in "4 4 0" "4 4 40" "4 8 80" "2 8 60" "3 6 60" "6 4 40" "4 6 120"; do
  set -- $cfg
  echo "LEADERS=$1 GATHER_MIN=$2 GATHER_US=$3"
  KZGAMD_LEADERS=$1 KZGAMD_GATHER_MIN=$2 KZGAMD_GATHER_US=$3 timeout 100 tools/concurrent_bench tests/golden/trusted_setup.txt 0.6 16
  KZGAMD_LEADERS=$1 KZGAMD_GATHER_MIN=$2 KZGAMD_GATHER_US=$3 timeout 100 tools/concurrent_bench tests/golden/trusted_setup.txt 0.4 1
done

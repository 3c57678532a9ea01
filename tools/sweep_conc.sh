# 16-thread throughput of the coalesced entry points against the leader / gather settings (tuning keys leaders,
# gather_min, gather_us: read when the settings object is created)
export LD_LIBRARY_PATH=rust-kzg_amd/csrc:/opt/rocm/lib
for cfg in "3 6 60" "3 5 40" "3 8 100" "4 4 40" "4 5 60" "2 8 60" "2 8 120" "5 4 40" "3 6 0"; do
  set -- $cfg
  echo "LEADERS=$1 GATHER_MIN=$2 GATHER_US=$3"
  KZGAMD_TUNING="leaders=$1;gather_min=$2;gather_us=$3" timeout 100 tools/concurrent_bench tests/golden/trusted_setup.txt 0.6 16
  echo
done

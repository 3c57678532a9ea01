# 16 (and 32) threads of mult_pippenger_prepared on one prepared handle against the combining keys and the table budget
export LD_LIBRARY_PATH=rust-kzg_amd/csrc:/opt/rocm/lib
S=tests/golden/trusted_setup.txt
for gb in 24 100; do for lanes in 1 2 3 4; do for gm in 4 6; do
  echo "table $gb GB lanes $lanes gather_min $gm"; B1_TABLE_GB=$gb KZGAMD_TUNING="combine_lanes=$lanes;combine_gather_min=$gm" tools/concurrent_bench $S 0.8 16 2
done; done; done
for lanes in 2 3 4; do echo "32 threads lanes $lanes"; KZGAMD_TUNING="combine_lanes=$lanes" tools/concurrent_bench $S 0.8 32 2; done

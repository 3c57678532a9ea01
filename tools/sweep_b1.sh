export LD_LIBRARY_PATH=rust-kzg_amd/csrc:/opt/rocm/lib
S=tests/golden/trusted_setup.txt
for t in 1 16 32; do tools/concurrent_bench $S 1.0 $t 2; done
for lanes in 1 2; do for gm in 1 6 12; do for us in 60 150; do echo "lanes $lanes gather_min $gm us $us"; KZGAMD_TUNING="combine_lanes=$lanes;combine_gather_min=$gm;combine_gather_us=$us" tools/concurrent_bench $S 0.8 16 2; done; done; done

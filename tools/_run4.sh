set -x
cd /root/repo
mkdir -p gpurun_out/r4
export KZGAMD_TEST_FLAVOURS=product
timeout 900 python -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "few_commitments or several_large or 2p20 or small_sizes or every_size or wide_table or trusted_setup" 2>&1 | tail -15 > gpurun_out/r4/pytest_msm.log
timeout 900 python -m pytest tests/test_ckzg_gpu.py tests/test_fftg1_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4/pytest_ckzg.log
for t in "" "quad_accum_max=0" "no_wide_tree=1" "quad_accum_max=0;no_wide_tree=1"; do
  echo "== $t" >> gpurun_out/r4/single.log
  KZGAMD_TUNING="$t" timeout 300 python tools/time_single.py 2>&1 | head -3 >> gpurun_out/r4/single.log
done
timeout 300 python tools/time_proofs_dev.py > gpurun_out/r4/proofs_dev.log 2>&1
for a in "20 4" "20 2" "20 8" "16 4"; do
  set -- $a
  timeout 400 python tools/ab_batched.py $1 $2 sub_streams=0 "sub_streams=3" "sub_streams=3;tile_rows=32" "sub_streams=3;sub_prio=0" "sub_streams=6" "sub_streams=6;sub_prio=0" > gpurun_out/r4/ab_$1_$2.log 2>&1
done
timeout 300 python tools/ab_2p20.py tile_rows=16 20 > gpurun_out/r4/ab_tile16_20.log 2>&1
timeout 300 python tools/ab_2p20.py tile_rows=16 16 > gpurun_out/r4/ab_tile16_16.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r4/trace_b4 -o t --output-format csv -- python /root/repo/tools/ab_batched.py 20 4 sub_streams=6 trace > /root/repo/gpurun_out/r4/trace_b4.log 2>&1)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r4/trace_single -o t --output-format csv -- python /root/repo/tools/prof_single.py > /root/repo/gpurun_out/r4/trace_single.log 2>&1)
ls gpurun_out/r4

set -x
cd /root/repo
mkdir -p gpurun_out/r5
export KZGAMD_TEST_FLAVOURS=product
timeout 900 python -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "few_commitments or several_large or 2p20" 2>&1 | tail -15 > gpurun_out/r5/pytest_msm.log
for t in "" "quad_accum_max=0;no_wide_tree=1"; do
  echo "== $t" >> gpurun_out/r5/single.log
  KZGAMD_TUNING="$t" timeout 300 python tools/time_single.py 2>&1 | head -3 >> gpurun_out/r5/single.log
done
for t in "" "wide_fold_max=4" "wide_fold_max=4;quad_accum_max=4" "wide_fold_max=8;quad_accum_max=8" "wide_fold_max=2;quad_accum_max=1"; do
  echo "== $t" >> gpurun_out/r5/batches.log
  KZGAMD_TUNING="$t" timeout 300 python tools/time_batches.py 2>&1 | head -6 >> gpurun_out/r5/batches.log
done
for a in "16 4" "17 4" "16 8" "18 4"; do
  set -- $a
  timeout 400 python tools/ab_batched.py $1 $2 sub_streams=0 "sub_streams=3;sub_large=1" "sub_streams=6;sub_large=1" > gpurun_out/r5/ab_$1_$2.log 2>&1
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r5/trace_single -o t --output-format csv -- python /root/repo/tools/prof_single.py > /root/repo/gpurun_out/r5/trace_single.log 2>&1)
timeout 1800 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -40 > gpurun_out/r5/pytest_all_product.log
ls gpurun_out/r5

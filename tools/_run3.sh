set -x
cd /root/repo
mkdir -p gpurun_out/r3
export KZGAMD_TEST_FLAVOURS=product
timeout 600 python -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "several_large or 2p20" 2>&1 | tail -15 > gpurun_out/r3/pytest_msm.log
timeout 600 python -m pytest tests/test_ckzg_gpu.py tests/test_switch_forms_gpu.py tests/test_config_matrix_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3/pytest_ckzg.log
timeout 300 python tools/time_proofs_dev.py > gpurun_out/r3/proofs_dev.log 2>&1
for a in "20 4" "20 2" "16 4"; do
  set -- $a
  timeout 300 python tools/ab_batched.py $1 $2 sub_streams=0 sub_streams=2 sub_streams=3 > gpurun_out/r3/ab_$1_$2.log 2>&1
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r3/trace_b4 -o t --output-format csv -- python /root/repo/tools/ab_batched.py 20 4 sub_streams=3 trace > /root/repo/gpurun_out/r3/trace_b4.log 2>&1)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r3/trace_single -o t --output-format csv -- python /root/repo/tools/prof_single.py > /root/repo/gpurun_out/r3/trace_single.log 2>&1)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r3/trace_proofdev -o t --output-format csv -- python /root/repo/tools/prof_proof_dev.py > /root/repo/gpurun_out/r3/trace_proofdev.log 2>&1)
timeout 900 python -m pytest tests/test_multigpu.py -x -q -m gpu -k "eight" --durations=5 2>&1 | tail -30 > gpurun_out/r3/pytest_multi.log
ls gpurun_out/r3

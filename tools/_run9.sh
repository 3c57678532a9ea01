set -x
cd /root/repo
mkdir -p gpurun_out/r9
timeout 1200 python -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "variants or 2p20 or several_large or small_sizes or every_size" > gpurun_out/r9/msm_tests.log 2>&1
tail -3 gpurun_out/r9/msm_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r9/prof_proofs -o t -- python /root/repo/tools/prof_proof_dev.py > /root/repo/gpurun_out/r9/prof_proofs.log 2>&1
head -20 /root/repo/gpurun_out/r9/prof_proofs/t_kernel_stats.csv | cut -c1-200

set -x
cd /root/repo
mkdir -p gpurun_out/r1
export KZGAMD_TEST_FLAVOURS=product
timeout 900 python -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "every_size or several_large or 2p20 or split_property or small_sizes" 2>&1 | tail -15 > gpurun_out/r1/pytest_msm.log
timeout 300 python -m pytest tests/test_ntt_gpu.py -x -q -m gpu -k "every_length" 2>&1 | tail -15 > gpurun_out/r1/pytest_ntt.log
timeout 300 python tools/ab_batched.py 20 4 sub_streams=0 sub_streams=1 sub_streams=2 sub_streams=3 > gpurun_out/r1/ab_20_4.log 2>&1
timeout 300 python tools/ab_batched.py 20 2 sub_streams=0 sub_streams=1 sub_streams=2 > gpurun_out/r1/ab_20_2.log 2>&1
timeout 300 python tools/ab_batched.py 16 4 sub_streams=0 sub_streams=1 sub_streams=2 > gpurun_out/r1/ab_16_4.log 2>&1
timeout 300 python tools/ab_batched.py 18 4 sub_streams=0 sub_streams=2 > gpurun_out/r1/ab_18_4.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r1/trace_b4 -o t --output-format csv -- python /root/repo/tools/ab_batched.py 20 4 sub_streams=2 trace > /root/repo/gpurun_out/r1/trace_b4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r1/trace_16 -o t --output-format csv -- python /root/repo/tools/prof_2p20.py 16 > /root/repo/gpurun_out/r1/trace_16.log 2>&1
cd /root/repo
python tools/timeline.py gpurun_out/r1/trace_b4 > gpurun_out/r1/timeline_b4.txt 2>&1
python tools/timeline.py gpurun_out/r1/trace_16 > gpurun_out/r1/timeline_16.txt 2>&1
ls gpurun_out/r1

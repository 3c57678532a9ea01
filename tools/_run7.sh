set -x
cd /root/repo
mkdir -p gpurun_out/r7
timeout 1100 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r7/gpu_suite.log 2>&1
tail -5 gpurun_out/r7/gpu_suite.log
timeout 600 python bench.py > gpurun_out/r7/bench.json 2> gpurun_out/r7/bench.err
tail -c 600 gpurun_out/r7/bench.err
cd /tmp && export TMPDIR=/tmp
for logn in 16 20; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r7/prof_2p$logn -o t -- python /root/repo/tools/prof_2p20.py $logn > /root/repo/gpurun_out/r7/prof_2p$logn.log 2>&1
done
ls /root/repo/gpurun_out/r7

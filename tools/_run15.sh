# round 6 final: the whole GPU suite on both library flavours, then the round's profile collection
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r06_gpu_suite_both_flavours.log 2>&1
tail -4 gpurun_out/r06_gpu_suite_both_flavours.log
TAG=r06 bash tools/collect_round_profiles.sh > gpurun_out/collect.log 2>&1
tail -5 gpurun_out/collect.log

# round 6, final state of the sources: smoke, the whole GPU suite on both flavours, the round's profile collection, the bench line
cd /root/repo; mkdir -p gpurun_out
rm -rf gpurun_out/prof_* gpurun_out/pmc_*
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r06_gpu_suite_both_flavours.log 2>&1
tail -3 gpurun_out/r06_gpu_suite_both_flavours.log
TAG=r06 bash tools/collect_round_profiles.sh > gpurun_out/collect.log 2>&1
tail -2 gpurun_out/collect.log

# kernel + memory-copy trace of tools/concurrent_bench: trace_conc.sh [seconds] [threads] [0 commitments | 1 proofs]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/conc_trace
LD_LIBRARY_PATH=$R/rust-kzg_amd/csrc:/opt/rocm/lib rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/conc_trace -o t -- $R/tools/concurrent_bench $R/tests/golden/trusted_setup.txt ${1:-0.2} ${2:-16} ${3:-0} > $R/gpurun_out/conc_trace.log 2>&1
ls -la $R/gpurun_out/conc_trace | head
tail -1 $R/gpurun_out/conc_trace.log

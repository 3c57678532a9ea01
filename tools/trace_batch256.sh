cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/b256_trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/b256_trace -o t -- python $R/tools/${2:-trace_batch256.py} ${1:-256} > $R/gpurun_out/b256_trace.log 2>&1
grep "call" $R/gpurun_out/b256_trace.log

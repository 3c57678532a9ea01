#!/bin/bash
# kernel statistics of the fixed-base bucket path at 2^20 for a few windows: bash tools/prof_fixed.sh  (-> gpurun_out/fixed_c*.csv)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in ${WINDOWS:-16 18 20}; do
  rm -rf /tmp/pf_$c
  KZGAMD_FBW_MAX_GB=4 KZGAMD_TUNING="window_prepared=$c" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$c -o p -- python $R/tools/prof_2p20.py 20 fixed > /tmp/pf_$c.log 2>&1
  f=$(find /tmp/pf_$c -name '*kernel_stats.csv' | head -1)
  echo "== c=$c"; tail -3 /tmp/pf_$c.log; head -16 "$f" | cut -d, -f1-4
  cp "$f" $R/gpurun_out/fixed_c$c.csv 2>/dev/null
done

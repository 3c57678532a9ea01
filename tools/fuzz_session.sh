R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
SEED=${SEED:-41}
mkdir -p $R/gpurun_out
LOG=$R/gpurun_out/long_fuzz_$SEED.log
: > $LOG
# SEED, PB (seconds per tool, product build), EB (exact build):  SEED=51 PB=100 EB=60 bash tools/fuzz_session.sh
for spec in "product libkzg_mi355x.so ${PB:-150} $SEED" "exact libkzg_mi355x_exact.so ${EB:-90} $SEED"; do
  set -- $spec
  s=$4
  for f in fuzz_ckzg.py fuzz_msm.py fuzz_g1.py; do
    echo "== $1 $f seed $s budget $3" >> $LOG
    KZGAMD_LIB=$R/rust-kzg_amd/csrc/$2 timeout $(( $3 + 120 )) python $R/tools/$f $3 $s 2>&1 | tail -3 >> $LOG
    s=$((s+1))
  done
done
cat $LOG

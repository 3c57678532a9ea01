R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
LOG=$R/gpurun_out/long_fuzz3.log
: > $LOG
for spec in "product libkzg_mi355x.so 150 41" "exact libkzg_mi355x_exact.so 90 41"; do
  set -- $spec
  s=$4
  for f in fuzz_ckzg.py fuzz_msm.py fuzz_g1.py; do
    echo "== $1 $f seed $s budget $3" >> $LOG
    KZGAMD_LIB=$R/rust-kzg_amd/csrc/$2 timeout $(( $3 + 120 )) python $R/tools/$f $3 $s 2>&1 | tail -3 >> $LOG
    s=$((s+1))
  done
done
cat $LOG

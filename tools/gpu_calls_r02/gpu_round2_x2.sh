#!/bin/bash
# success curves with the shuffled visiting order (4 trials x 1024 chains), models given as arguments
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2x2
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for m in "$@"; do
  timeout -k 10 500 python -m annealing_sign_problem_amd.full_hilbert_space --model $m --output $OUT/fhs_shuffled_$m.csv --number-sweeps 100,200,400,800,1600,3200,6400,12800,25600 --repetitions 1024 --trials 4 --seed 435834 --sweep-order shuffled > $OUT/fhs_shuffled_$m.log 2>&1
  rc=$?; echo "$m rc=$rc" | tee -a $OUT/status.txt
  grep -v amdgpu $OUT/fhs_shuffled_$m.log
  [ $rc -eq 0 ] || exit $rc
done

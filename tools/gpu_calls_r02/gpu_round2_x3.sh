#!/bin/bash
# shuffled-order success probabilities at long anneals (4 trials x 1024 chains)
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2x3
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for m in heisenberg_kagome_16 j1j2_square_4x4; do
  timeout -k 10 560 python -m annealing_sign_problem_amd.full_hilbert_space --model $m --output $OUT/fhs_shuffled_long_$m.csv --number-sweeps 25600,51200,102400 --repetitions 1024 --trials 4 --seed 435834 --sweep-order shuffled > $OUT/fhs_shuffled_long_$m.log 2>&1
  rc=$?; echo "$m rc=$rc" | tee -a $OUT/status.txt
  grep -v amdgpu $OUT/fhs_shuffled_long_$m.log
  [ $rc -eq 0 ] || exit $rc
done

#!/bin/bash
# the reference's large targets on the real models: ground states regenerated on the GPU, then
# `make kagome_36` / `make pyrochlore_32` (greedy only, as the reference runs them) and kagome_36 with annealing
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2k36
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
t0=$(date +%s)
timeout -k 10 900 make kagome_36 NUMBER_SAMPLES=256 DATA=/tmp/data-large OUT=$OUT/experiments > $OUT/make_kagome_36.log 2>&1; rc=$?
echo "make kagome_36 rc=$rc, $(( $(date +%s) - t0 )) s" | tee -a $OUT/status.txt
grep -v amdgpu $OUT/make_kagome_36.log | tail -25 | cut -c1-220
[ $rc -eq 0 ] || exit $rc
t0=$(date +%s)
timeout -k 10 600 make pyrochlore_32 NUMBER_SAMPLES=256 DATA=/tmp/data-large OUT=$OUT/experiments > $OUT/make_pyrochlore_32.log 2>&1; rc=$?
echo "make pyrochlore_32 rc=$rc, $(( $(date +%s) - t0 )) s" | tee -a $OUT/status.txt
grep -v amdgpu $OUT/make_pyrochlore_32.log | tail -12 | cut -c1-220
[ $rc -eq 0 ] || exit $rc
t0=$(date +%s)
timeout -k 10 900 python -m annealing_sign_problem_amd.sampled_components --model heisenberg_kagome_36 --hdf5 /tmp/data-large/heisenberg_kagome_36.h5 \
  --seed 435834 --output $OUT/experiments/kagome_36_annealed.csv --order 2 --global-cutoff 1e-6 --number-samples 64 --jobs 8 > $OUT/kagome_36_annealed.log 2>&1; rc=$?
echo "kagome_36 with annealing rc=$rc, $(( $(date +%s) - t0 )) s" | tee -a $OUT/status.txt
grep -v amdgpu $OUT/kagome_36_annealed.log | tail -12 | cut -c1-220
head -5 $OUT/experiments/kagome_36_annealed.csv

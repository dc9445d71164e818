#!/bin/bash
# `make kagome_36`'s pipeline (256 clusters x orders 0-2, greedy) against the number of host threads
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2jobs
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m annealing_sign_problem_amd.sector_ed --model heisenberg_kagome_36 --output /tmp/k36.h5 > $OUT/ed.log 2>&1 || exit 5
for jobs in 1 4 8 16; do
  t0=$(date +%s%N)
  timeout -k 10 300 python -m annealing_sign_problem_amd.sampled_components --model heisenberg_kagome_36 --hdf5 /tmp/k36.h5 --seed 435834 \
    --output $OUT/k36_jobs$jobs.csv --order 2 --no-annealing --global-cutoff 1e-6 --number-samples 256 --jobs $jobs > $OUT/k36_jobs$jobs.log 2>&1 || exit 6
  echo "kagome_36, 256 clusters x 3 orders, greedy, --jobs $jobs: $(( ($(date +%s%N) - t0) / 1000000 )) ms (incl. start-up and reading the 504 MB ground-state file)" | tee -a $OUT/pipeline_jobs.txt
done
cmp $OUT/k36_jobs1.csv $OUT/k36_jobs8.csv && echo "outputs identical" | tee -a $OUT/pipeline_jobs.txt
t0=$(date +%s%N)
timeout -k 10 300 python -c "
import time, sys
t=time.time()
from annealing_sign_problem_amd import common
g,e,r=common.load_ground_state('/tmp/k36.h5'); print('load_ground_state %.2f s'%(time.time()-t))
" | tee -a $OUT/pipeline_jobs.txt

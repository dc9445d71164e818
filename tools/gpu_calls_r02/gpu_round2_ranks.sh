#!/bin/bash
# `make kagome_36`'s pipeline with several ranks on ONE GPU (gloo; every rank its own interpreter, clusters c mod world)
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2ranks
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m annealing_sign_problem_amd.sector_ed --model heisenberg_kagome_36 --output /tmp/k36.h5 > $OUT/ed.log 2>&1 || exit 5
ARGS="--model heisenberg_kagome_36 --hdf5 /tmp/k36.h5 --seed 435834 --order 2 --no-annealing --global-cutoff 1e-6"
t0=$(date +%s%N)
timeout -k 10 400 python -m annealing_sign_problem_amd.sampled_components $ARGS --number-samples 1024 --jobs 8 --output $OUT/one_process.csv > $OUT/one_process.log 2>&1 || exit 6
echo "1 process x 8 threads, 1024 clusters x 3 orders: $(( ($(date +%s%N) - t0) / 1000000 )) ms" | tee -a $OUT/ranks.txt
for ranks in 4; do
  t0=$(date +%s%N)
  ASP_DIST_BACKEND=gloo ASP_SINGLE_DEVICE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks --master-addr 127.0.0.1 --master-port 29541 \
    -m annealing_sign_problem_amd.sampled_components $ARGS --number-samples 1024 --jobs 4 --output $OUT/ranks$ranks.csv > $OUT/ranks$ranks.log 2>&1 || { tail -20 $OUT/ranks$ranks.log; exit 7; }
  echo "$ranks ranks x 4 threads on one GPU, 1024 clusters x 3 orders: $(( ($(date +%s%N) - t0) / 1000000 )) ms" | tee -a $OUT/ranks.txt
  cmp $OUT/one_process.csv $OUT/ranks$ranks.csv && echo "output identical to the single process" | tee -a $OUT/ranks.txt
done

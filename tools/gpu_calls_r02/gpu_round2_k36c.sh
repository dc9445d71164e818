#!/bin/bash
# `make kagome_36` at a tenth of the reference's production size on one GPU: 4096 clusters greedy (4 ranks),
# then 512 clusters with the batched anneal
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2k36c
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m annealing_sign_problem_amd.sector_ed --model heisenberg_kagome_36 --output /tmp/k36.h5 > $OUT/ed.log 2>&1 || exit 5
ARGS="--model heisenberg_kagome_36 --hdf5 /tmp/k36.h5 --seed 435834 --order 2 --global-cutoff 1e-6"
t0=$(date +%s%N)
ASP_DIST_BACKEND=gloo ASP_SINGLE_DEVICE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29543 \
  -m annealing_sign_problem_amd.sampled_components $ARGS --no-annealing --number-samples 4096 --jobs 4 --output $OUT/kagome_36_4096_greedy.csv > $OUT/greedy.log 2>&1 || { tail -20 $OUT/greedy.log; exit 7; }
echo "4096 clusters x 3 orders, greedy, 4 ranks x 4 threads on one GPU: $(( ($(date +%s%N) - t0) / 1000000 )) ms" | tee -a $OUT/timing.txt
t0=$(date +%s%N)
timeout -k 10 600 python -m annealing_sign_problem_amd.sampled_components $ARGS --number-samples 512 --jobs 8 --batch 64 --output $OUT/kagome_36_512_annealed.csv > $OUT/annealed.log 2>&1 || { tail -20 $OUT/annealed.log; exit 8; }
echo "512 clusters x 3 orders, greedy + batched anneal (64 chains x 5120 sweeps per model), 1 process x 8 threads: $(( ($(date +%s%N) - t0) / 1000000 )) ms" | tee -a $OUT/timing.txt
python tools/density_summary.py $OUT/kagome_36_4096_greedy.csv | tee $OUT/summary.txt
python tools/density_summary.py $OUT/kagome_36_512_annealed.csv sa | tee -a $OUT/summary.txt

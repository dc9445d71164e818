#!/bin/bash
# kagome_18 success curves with the canonical vector of its three-fold degenerate ground level
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2k18
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m annealing_sign_problem_amd.full_hilbert_space --model heisenberg_kagome_18 --output $OUT/fhs_heisenberg_kagome_18.csv --number-sweeps 100,200,400,800,1600,3200,6400 --repetitions 1024 --trials 10 --seed 435834 > $OUT/fhs_heisenberg_kagome_18.log 2>&1 || exit 1
grep -v amdgpu $OUT/fhs_heisenberg_kagome_18.log
timeout -k 10 500 python -m annealing_sign_problem_amd.full_hilbert_space --model heisenberg_kagome_18 --output $OUT/fhs_shuffled_heisenberg_kagome_18.csv --number-sweeps 100,200,400,800,1600,3200,6400 --repetitions 1024 --trials 4 --seed 435834 --sweep-order shuffled > $OUT/fhs_shuffled_heisenberg_kagome_18.log 2>&1 || exit 2
grep -v amdgpu $OUT/fhs_shuffled_heisenberg_kagome_18.log
timeout -k 10 300 python tools/kagome18_degeneracy_probe.py 2 2>&1 | grep -v amdgpu | tee $OUT/kagome18_degeneracy_probe.txt | tail -8

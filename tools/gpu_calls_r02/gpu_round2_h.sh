#!/bin/bash
# Round-2 GPU call H: suite, kagome_18 success curve, batch wavefront cap, build host call
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2h
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -8 $OUT/pytest.log
timeout -k 10 600 python -m annealing_sign_problem_amd.full_hilbert_space --model heisenberg_kagome_18 --output $OUT/fhs_heisenberg_kagome_18.csv --number-sweeps 100,200,400,800,1600,3200,6400 --repetitions 1024 --trials 10 --seed 435834 > $OUT/fhs_heisenberg_kagome_18.log 2>&1; echo "fhs rc=$?" | tee -a $OUT/status.txt
cat $OUT/fhs_heisenberg_kagome_18.log
timeout -k 10 600 python tools/tune_batch.py 128 512 > $OUT/tune_batch.log 2>&1; cat $OUT/tune_batch.log
timeout -k 10 600 python tools/tune_build.py > $OUT/tune_build.log 2>&1; tail -20 $OUT/tune_build.log

#!/bin/bash
# matrix-free product: tests, then the ground state of sk_32_1 (6.0e8 states)
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2sk
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_sector.py -m gpu -q -x -k "matrix_free" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/status.txt
tail -25 $OUT/pytest.log | cut -c1-250
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m annealing_sign_problem_amd.sector_ed --model sk_32_1 --output /tmp/sk_32_1.h5 --tol 1e-8 > $OUT/sk_32_1_ed.log 2>&1; rc=$?; echo "sk_32_1 ED rc=$rc" | tee -a $OUT/status.txt
grep -v amdgpu $OUT/sk_32_1_ed.log | tail -30

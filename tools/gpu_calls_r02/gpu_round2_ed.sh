#!/bin/bash
# sector enumeration / matrix / Lanczos on the device: tests, then the 36-site kagome sector
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2ed
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_sector.py -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/status.txt
tail -30 $OUT/pytest.log | cut -c1-250
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m annealing_sign_problem_amd.sector_ed --model heisenberg_kagome_36 --output /tmp/kagome_36.h5 > $OUT/kagome_36_ed.log 2>&1; rc=$?; echo "kagome_36 ED rc=$rc" | tee -a $OUT/status.txt
grep -v amdgpu $OUT/kagome_36_ed.log | tail -40
ls -l /tmp/kagome_36.h5

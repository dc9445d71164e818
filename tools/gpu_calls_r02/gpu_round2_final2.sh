#!/bin/bash
# state check after the plan-building change: full suite, smoke, the bench line
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2final2
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/status.txt
tail -5 $OUT/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc" | tee -a $OUT/status.txt; tail -1 $OUT/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2>&1; rc=$?; echo "bench rc=$rc" | tee -a $OUT/status.txt
tail -c 400 $OUT/bench.log

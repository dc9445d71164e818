#!/bin/bash
# Round-2 GPU call L: batch kernel without descriptor spills; register budgets for 5 / 6 waves per SIMD
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2l
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_sa.py -m gpu -q -k batch > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt; tail -3 $OUT/pytest.log
for round in 1 2; do
  echo "== default"; timeout -k 10 600 python tools/tune_batch.py 128 512 2>&1 | grep -v "^$"
  for tag in w5 w6; do echo "== $tag"; ASP_LIB_TAG=$tag ASP_NO_REBUILD=1 timeout -k 10 600 python tools/tune_batch.py 128 512 2>&1 | grep -v "^$"; done
done > $OUT/tune.log 2>&1
cat $OUT/tune.log

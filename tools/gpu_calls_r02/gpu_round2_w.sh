#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2w
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_sa.py tests/test_gpu_published.py -m gpu -q -k "shuffled" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -25 $OUT/pytest.log | cut -c1-220

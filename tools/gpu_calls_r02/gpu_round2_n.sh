#!/bin/bash
# Round-2 GPU call N: suite with the duplicate-keeping device build; symmetric model build timing
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2n
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -12 $OUT/pytest.log | cut -c1-220
timeout -k 10 600 python tools/time_symmetric.py 1000 10000 > $OUT/symmetric.log 2>&1; cat $OUT/symmetric.log | cut -c1-300

#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2cfg
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "config3 or config4" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/status.txt
tail -30 $OUT/pytest.log | cut -c1-250

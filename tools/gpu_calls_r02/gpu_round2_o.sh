#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2o
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/tune_cache.py > $OUT/cache.log 2>&1; cat $OUT/cache.log

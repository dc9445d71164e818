#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2t
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python tools/schedule_probe.py > $OUT/schedule.log 2>&1; cat $OUT/schedule.log | grep -v amdgpu.ids

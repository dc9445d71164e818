#!/bin/bash
# Round-2 GPU call F: batch tuning with per-class replicas per workgroup; threads scan at K=1e4/3e4
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2f
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/tune_batch.py 64 128 512 > $OUT/tune_batch.log 2>&1; echo "tune rc=$?" | tee -a $OUT/status.txt
cat $OUT/tune_batch.log
for round in 1 2; do timeout -k 10 300 python tools/tune_sweep.py --sizes 10000,30000 --groups 4 --threads 768,1024 --sweeps 128 | grep "M="; done > $OUT/threads.log 2>&1
cat $OUT/threads.log

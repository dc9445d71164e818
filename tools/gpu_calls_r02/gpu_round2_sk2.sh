#!/bin/bash
# `make sk_32_1` on the real model: ground state (matrix-free Lanczos), then the sampled-cluster pipeline
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2sk2
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
t0=$(date +%s)
timeout -k 10 1100 make sk_32_1 NUMBER_SAMPLES=16 ORDER=1 DATA=/tmp/data-large OUT=$OUT/experiments JOBS=8 > $OUT/make_sk_32_1.log 2>&1; rc=$?
echo "make sk_32_1 (16 clusters, order 1) rc=$rc, $(( $(date +%s) - t0 )) s" | tee -a $OUT/status.txt
grep -v amdgpu $OUT/make_sk_32_1.log | tail -25 | cut -c1-220
ls -l /tmp/data-large/
python tools/density_summary.py $OUT/experiments/sk_32_1/noise_0/cutoff_1e-6/sk_32_1.csv | tee $OUT/accuracy_summary.txt

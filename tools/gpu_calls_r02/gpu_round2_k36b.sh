#!/bin/bash
# kagome_36 with the parameters of the reference's published density figure (experiments/density.gnu:30:
# sampled_power 0.1, cutoff 2e-4, extended up to three times), greedy only
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2k36b
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
t0=$(date +%s)
timeout -k 10 1000 make kagome_36 NUMBER_SAMPLES=384 ORDER=3 CUTOFF=2e-4 DATA=/tmp/data-large OUT=$OUT/experiments > $OUT/make_kagome_36.log 2>&1; rc=$?
echo "make kagome_36 ORDER=3 CUTOFF=2e-4 rc=$rc, $(( $(date +%s) - t0 )) s" | tee -a $OUT/status.txt
grep -v amdgpu $OUT/make_kagome_36.log | tail -6 | cut -c1-220
python tools/density_summary.py $OUT/experiments/heisenberg_kagome_36/noise_0/cutoff_2e-4/heisenberg_kagome_36.csv | tee $OUT/density_summary.txt

#!/bin/bash
# kernel trace of the 36-site kagome sector ED, then the sector tests (incl. the literature pin)
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2edprof
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o sector_ed -- python3 -m annealing_sign_problem_amd.sector_ed --model heisenberg_kagome_36 --output /tmp/k36.h5 > $OUT/ed.log 2>&1; rc=$?
echo "rocprofv3 sector_ed rc=$rc" | tee -a $OUT/status.txt
grep -v amdgpu $OUT/ed.log | tail -5
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/sector_ed_kernel_stats.csv
rm -rf $OUT/prof
head -14 $OUT/sector_ed_kernel_stats.csv | cut -c1-200
[ $rc -eq 0 ] || exit $rc

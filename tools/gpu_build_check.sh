#!/bin/bash
# Runs on the GPU box: parity tests of the coupling build, its timing on the bench workload and
# the kernel trace + HBM counters of its three kernels (each counter set a run of its own).
#   tools/gpu_build_check.sh [tag] [pmc]
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-build}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_build.py -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
python3 tools/profile_build_matrix.py > $OUT/time.log 2>&1 || { tail -5 $OUT/time.log; exit 2; }
cat $OUT/time.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/profile_build_matrix.py > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 3; }
find $OUT/trace -name "*_kernel_stats.csv" -exec cut -d, -f1-4 {} \; | cut -c1-120
if [ "$2" = "pmc" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/profile_build_matrix.py > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 4; }
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python3 tools/profile_build_matrix.py > $OUT/l2.log 2>&1 || { tail -5 $OUT/l2.log; exit 5; }
fi
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete

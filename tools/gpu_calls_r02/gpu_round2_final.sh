#!/bin/bash
# Round-2 state check: full suite, smoke, bench as the driver runs it, then the rocprofv3 passes
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2final
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/status.txt
tail -5 $OUT/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc" | tee -a $OUT/status.txt; tail -1 $OUT/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2>&1; rc=$?; echo "bench rc=$rc" | tee -a $OUT/status.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $OUT/bench_default.log 2>&1; rc=$?; echo "bench (default flags) rc=$rc" | tee -a $OUT/status.txt
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_profile.sh r02 > $OUT/profile.log 2>&1; echo "profile rc=$?" | tee -a $OUT/status.txt
cp profiles/r02_bench_kernel_stats.csv profiles/r02_build_kernel_stats.csv profiles/r02_batch_kernel_stats.csv profiles/r02_summary.json profiles/traffic.json profiles/sweep_counters.json $OUT/ 2>/dev/null
tail -c 700 $OUT/bench.log

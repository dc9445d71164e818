#!/bin/bash
# Round-2 GPU call Z: state check after the shuffled-order work — full suite, smoke, bench
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2z
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/status.txt
tail -5 $OUT/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc" | tee -a $OUT/status.txt; tail -1 $OUT/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/status.txt
tail -c 1500 $OUT/bench.log

#!/bin/bash
# Round-2 GPU call V: state check — suite, smoke, bench as the driver runs it, rocprofv3 passes
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2v
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -5 $OUT/pytest.log | cut -c1-200
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/status.txt; tail -1 $OUT/smoke.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/status.txt
bash tools/gpu_profile.sh r02 > $OUT/profile.log 2>&1; echo "profile rc=$?" | tee -a $OUT/status.txt
cp profiles/r02_* profiles/traffic.json profiles/sweep_counters.json $OUT/ 2>/dev/null
tail -c 600 $OUT/bench.log

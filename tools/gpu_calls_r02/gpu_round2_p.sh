#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2p
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_sa.py tests/test_gpu_configs.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -8 $OUT/pytest.log | cut -c1-220
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-build > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/status.txt
tail -c 1200 $OUT/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-build > $OUT/trace.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r2p/trace/**/*_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:7]:
        print("%-60s calls %4s avg %9.1f us %6s%%" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY

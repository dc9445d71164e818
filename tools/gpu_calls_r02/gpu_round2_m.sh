#!/bin/bash
# Round-2 GPU call M: full suite; pipeline timing with annealing; bench
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2m
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -6 $OUT/pytest.log
timeout -k 10 900 python tools/time_pipeline.py heisenberg_kagome_16 64 > $OUT/pipeline.log 2>&1; echo "pipeline rc=$?" | tee -a $OUT/status.txt; cat $OUT/pipeline.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/status.txt
tail -c 900 $OUT/bench.log

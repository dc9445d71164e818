#!/bin/bash
# Round-2 GPU call G: full suite (symmetry, rccl, hdf5, batch), bench
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2g
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -15 $OUT/pytest.log
timeout -k 10 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/status.txt
tail -c 2500 $OUT/bench.log

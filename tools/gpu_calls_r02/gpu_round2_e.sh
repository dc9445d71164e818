#!/bin/bash
# Round-2 GPU call E: parity suite on the new accept phase / sign construction, A/B of the variants
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2e
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -5 $OUT/pytest.log
AB_ARGS="--sizes 10000,30000,100000 --groups 4 --threads 768,1024 --sweeps 128 --cache 1,0" bash tools/ab_tags.sh base shr skip > $OUT/ab.log 2>&1
for round in 1 2 3; do echo "== round $round both"; timeout -k 10 200 python tools/tune_sweep.py --sizes 10000,30000,100000 --groups 4 --threads 768,1024 --sweeps 128 --cache 1,0 | grep "M="; done >> $OUT/ab.log 2>&1
cat $OUT/ab.log

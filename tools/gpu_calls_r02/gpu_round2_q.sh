#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2q
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TUNE_WAVES=4 timeout -k 10 900 python tools/tune_batch.py 2048 > $OUT/tune2048.log 2>&1; cat $OUT/tune2048.log
python - > $OUT/pipe.log 2>&1 <<'PY'
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from annealing_sign_problem_amd import sampled_components
with tempfile.TemporaryDirectory() as d:
    base = ["--model", "heisenberg_kagome_16", "--order", "2", "--number-samples", "512", "--seed", "435834", "--global-cutoff", "1e-6"]
    sampled_components.main(base[:5] + ["2"] + base[6:] + ["--output", os.path.join(d, "warm.csv")])
    for batch in ("64", "256", "512"):
        t0 = time.time()
        sampled_components.main(base + ["--output", os.path.join(d, "b%s.csv" % batch), "--batch", batch])
        print("512 clusters x 3 orders with annealing, --batch %s: %.2f s" % (batch, time.time() - t0), flush=True)
    t0 = time.time()
    sampled_components.main(base + ["--output", os.path.join(d, "g.csv"), "--no-annealing"])
    print("512 clusters x 3 orders greedy only: %.2f s" % (time.time() - t0), flush=True)
    a = open(os.path.join(d, "b64.csv")).read(); b = open(os.path.join(d, "b512.csv")).read()
    assert a == b
PY
cat $OUT/pipe.log

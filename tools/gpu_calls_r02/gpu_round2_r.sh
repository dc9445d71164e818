#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2r
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt; tail -3 $OUT/pytest.log
python - > $OUT/pipe.log 2>&1 <<'PY'
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from annealing_sign_problem_amd import sampled_components
with tempfile.TemporaryDirectory() as d:
    base = ["--model", "heisenberg_kagome_16", "--order", "2", "--number-samples", "512", "--seed", "435834", "--global-cutoff", "1e-6"]
    sampled_components.main(base[:5] + ["2"] + base[6:] + ["--output", os.path.join(d, "warm.csv")])
    outs = []
    for jobs in ("1", "4", "8", "16"):
        t0 = time.time()
        out = os.path.join(d, "j%s.csv" % jobs)
        sampled_components.main(base + ["--output", out, "--batch", "128", "--jobs", jobs])
        print("512 clusters x 3 orders with annealing, --batch 128 --jobs %s: %.2f s" % (jobs, time.time() - t0), flush=True)
        outs.append(open(out).read())
    assert all(o == outs[0] for o in outs)
    for jobs in ("1", "8"):
        t0 = time.time()
        sampled_components.main(base + ["--output", os.path.join(d, "g%s.csv" % jobs), "--no-annealing", "--jobs", jobs])
        print("512 clusters x 3 orders greedy only --jobs %s: %.2f s" % (jobs, time.time() - t0), flush=True)
PY
cat $OUT/pipe.log

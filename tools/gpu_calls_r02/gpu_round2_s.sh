#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2s
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python - > $OUT/profile.log 2>&1 <<'PY'
import cProfile, pstats, os, sys, tempfile, io
sys.path.insert(0, os.getcwd())
from annealing_sign_problem_amd import sampled_components
with tempfile.TemporaryDirectory() as d:
    base = ["--model", "heisenberg_kagome_16", "--order", "2", "--number-samples", "256", "--seed", "435834", "--global-cutoff", "1e-6"]
    sampled_components.main(base[:5] + ["2"] + base[6:] + ["--output", os.path.join(d, "warm.csv")])
    pr = cProfile.Profile(); pr.enable()
    sampled_components.main(base + ["--output", os.path.join(d, "a.csv"), "--batch", "128"])
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:6000])
PY
cat $OUT/profile.log | cut -c1-180

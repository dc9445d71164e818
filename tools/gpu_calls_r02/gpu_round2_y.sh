#!/bin/bash
# shuffled kernel with 8 chains per workgroup: parity tests, then its rate at K=12870
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2y
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_sa.py tests/test_gpu_published.py -m gpu -q -k "shuffled" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/status.txt
tail -8 $OUT/pytest.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python - > $OUT/rate.txt 2>&1 <<'PY'
import time, numpy as np
from annealing_sign_problem_amd import annealer, synthetic, _lib
for n, deg in ((12870, 20.0), (4000, 12.0), (500, 8.0)):
    J, h, _ = synthetic.planted_cluster(n, seed=11, mean_degree=deg)
    H = annealer.Hamiltonian(J, h)
    for reps, sweeps in ((2048, 400), (1024, 400), (64, 2000)):
        betas = np.geomspace(0.1, 10.0, sweeps)
        t0 = time.time()
        x, e = annealer.anneal_raw(H, 7, betas, repetitions=reps, shuffled=True)
        dt = time.time() - t0
        print("K=%d chains=%d sweeps=%d: %.3f s wall, %.2f G flips/s, kernel+host-order %.1f ms" % (
            n, reps, sweeps, dt, n * reps * sweeps / dt / 1e9, _lib.load().asp_sa_last_sweep_ms(H.plan())), flush=True)
PY
echo "rate rc=$?" | tee -a $OUT/status.txt
cat $OUT/rate.txt | grep -v amdgpu

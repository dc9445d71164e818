"""Batched colour-order anneal on the production mix (128 problems, K log-uniform in [1e2, 1e4], 64
chains x 5120 sweeps): the size above which a problem's workgroups take up to 16 wavefronts instead
of 4 (ASP_BATCH_SMALL_MAX; development aid).  Usage: python tools/tune_batch_tiers.py [problems]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SEED = 783494
rng = np.random.default_rng(SEED)
sizes = [int(round(np.exp(rng.uniform(np.log(1e2), np.log(1e4))))) for _ in range(n)]
hams = []
for i, k in enumerate(sizes):
    J, h, _ = synthetic.planted_cluster(k, seed=SEED + i)
    ham = sa.Hamiltonian(J, h)
    ham.info()
    hams.append(ham)
flips = float(sum(sizes)) * 64 * 5120
sa.anneal_batch(hams[:4], seed=1, number_sweeps=8, repetitions=64, sweep_order="colour")
for small_max in os.environ.get("TUNE_SMALL_MAX", "10000,7000,5000,3500,2500,1500").split(","):
    os.environ["ASP_BATCH_SMALL_MAX"] = small_max
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        sa.anneal_batch(hams, seed=12345, number_sweeps=5120, repetitions=64, sweep_order="colour")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("4 wavefronts up to %6s spins, 16 beyond: %.3f s wall, sweep kernels %.1f ms, %.1f G flips/s "
          "(%d problems above)" % (small_max, best, lib.asp_sa_batch_last_ms(), flips / best / 1e9,
                                   sum(1 for k in sizes if k > int(small_max))), flush=True)
os.environ.pop("ASP_BATCH_SMALL_MAX", None)
_lib.shutdown()

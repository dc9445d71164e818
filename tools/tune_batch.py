"""Batched anneal on the production mix (K log-uniform in [1e2, 1e4], 64 chains x 5120 sweeps):
time per replicas-per-workgroup setting (ASP_BATCH_M) and per batch size.
Usage: python tools/tune_batch.py [problems ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

lib = _lib.load()
counts = [int(a) for a in sys.argv[1:]] or [128, 512]
SEED = 783494
for n in counts:
    rng = np.random.default_rng(SEED)
    sizes = [int(round(np.exp(rng.uniform(np.log(1e2), np.log(1e4))))) for _ in range(n)]
    hams = []
    t0 = time.perf_counter()
    for i, k in enumerate(sizes):
        J, h, _ = synthetic.planted_cluster(k, seed=SEED + i)
        ham = sa.Hamiltonian(J, h)
        ham.info()
        hams.append(ham)
    print("%d problems, sum K = %d: instances + plans %.2f s" % (n, sum(sizes), time.perf_counter() - t0),
          flush=True)
    flips = float(sum(sizes)) * 64 * 5120
    sa.anneal_batch(hams[:4], seed=1, number_sweeps=8, repetitions=64)
    for waves in os.environ.get("TUNE_WAVES", "8,4").split(","):
        if waves == "auto":  # the library's own choice (by LDS footprint)
            os.environ.pop("ASP_BATCH_WAVES", None)
        else:
            os.environ["ASP_BATCH_WAVES"] = waves
        t0 = time.perf_counter()
        sa.anneal_batch(hams, seed=12345, number_sweeps=5120, repetitions=64)
        dt = time.perf_counter() - t0
        print("  wavefront cap %-3s  %.3f s wall, sweep kernels %.1f ms, %.1f G flips/s, %.0f problems/s" % (
            waves, dt, lib.asp_sa_batch_last_ms(), flips / dt / 1e9, n / dt), flush=True)
    os.environ.pop("ASP_BATCH_WAVES", None)
    os.environ.pop("ASP_BATCH_BIG_M", None)
    os.environ.pop("ASP_BATCH_SMALL_M", None)
    os.environ.pop("ASP_BATCH_M", None)
    for ham in hams:
        ham.release()

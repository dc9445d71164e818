"""The production shape with the reference annealer's visiting order: N planted clusters, K log-uniform
in [1e2, 1e4], 64 chains x 5120 sweeps each, one asp_sa_anneal_batch call with ASP_SA_BATCH_SHUFFLED
against the colour-ordered batch and (on a sample) the per-problem shuffled calls.  (Development aid; GPU.)

    python tools/time_shuffled_batch.py [N=64] [sweeps=5120]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 5120
rng = np.random.default_rng(783494)
sizes = [int(round(np.exp(rng.uniform(np.log(1e2), np.log(1e4))))) for _ in range(n)]
hams = []
for i, k in enumerate(sizes):
    J, h, _ = synthetic.planted_cluster(k, seed=783494 + i)
    ham = sa.Hamiltonian(J, h)
    ham.info()
    hams.append(ham)
flips = float(sum(sizes)) * 64 * sweeps
sa.anneal_batch(hams[:2], seed=1, number_sweeps=8, repetitions=64, sweep_order="shuffled")
for order in ("shuffled", "colour"):
    t0 = time.perf_counter()
    out = sa.anneal_batch(hams, seed=12345, number_sweeps=sweeps, repetitions=64, sweep_order=order)
    t = time.perf_counter() - t0
    print("%-8s batch of %d problems (sum K = %d), 64 chains x %d sweeps: %.2f s = %.1f G flips/s, %.1f problems/s" % (
        order, n, sum(sizes), sweeps, t, flips / t / 1e9, n / t), flush=True)
    if order == "shuffled":
        shuffled = out
sample = list(range(0, n, 8))
t0 = time.perf_counter()
single = [sa.anneal(hams[i], seed=12345, number_sweeps=sweeps, repetitions=64, sweep_order="shuffled") for i in sample]
t = time.perf_counter() - t0
same = all(np.array_equal(x, shuffled[i][0]) and e == shuffled[i][1] for i, (x, e) in zip(sample, single))
print("shuffled, one call per problem (every 8th): %.2f s = %.1f G flips/s; identical to the batch: %s" % (
    t, sum(sizes[i] for i in sample) * 64.0 * sweeps / t / 1e9, same), flush=True)

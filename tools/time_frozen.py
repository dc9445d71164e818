"""Cost of a fully frozen sweep with the field cache on/off (development aid)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

lib = _lib.load()
for k in (10000, 100000):
    J, h, planted = synthetic.planted_cluster(k, seed=783494)
    ham = sa.Hamiltonian(J, h)
    x0, _ = sa.greedy_solve(ham)          # a local minimum: nothing flips at huge beta
    betas = np.full(64, 1e15)
    for cache in (0, 1):
        _lib.check(lib.asp_sa_set_field_cache(ham.plan(), cache))
        _lib.check(lib.asp_sa_set_launch(ham.plan(), 4, 1024))
        for _ in range(2):
            sa.anneal_raw(ham, 1, betas, 1024, 0, x0)
            ms = lib.asp_sa_last_sweep_ms(ham.plan())
        acc = np.zeros(1024, np.uint64)
        lib.asp_sa_last_stats(ham.plan(), 1024, None, _lib.ptr(acc))
        print("K=%d cache=%d: %.3f ms per frozen sweep (flips total %d)" % (k, cache, ms / 64, acc.sum()), flush=True)

"""Field-cache switch-over threshold (ASP_CACHE_FACTOR: enter cached mode below factor * blocks /
degree flips per sweep and workgroup) on the bench workload.  (Development aid.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

lib = _lib.load()
factors = [float(a) for a in sys.argv[1:]] or [0.0, 0.2, 0.4, 0.7, 1.0, 1.5, 2.5, 4.0]
for k in (10000, 30000, 100000):
    J, h, _ = synthetic.planted_cluster(k, seed=783494)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 128)
    row = []
    for f in factors:
        os.environ["ASP_CACHE_FACTOR"] = repr(f)
        best = None
        for _ in range(3):
            sa.anneal_raw(ham, 12345, betas, 1024)
            ms = lib.asp_sa_last_sweep_ms(ham.plan())
            best = ms if best is None else min(best, ms)
        row.append(k * 1024 * 128 / best / 1e6)
    print("K=%6d  " % k + "  ".join("f=%.1f: %6.1f" % (f, g) for f, g in zip(factors, row)) + "  G flips/s", flush=True)
os.environ.pop("ASP_CACHE_FACTOR", None)

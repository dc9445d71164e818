"""Team sweep (one chain over several workgroups) against the single-workgroup kernel: many
shapes, long ladders, repeated — looking for rare ordering bugs in the exchange (development aid)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

lib = _lib.load()
bad = 0
for k, degree in ((100000, 23.0), (40000, 8.0), (250000, 6.0)):
    J, h, _ = synthetic.planted_cluster(k, seed=k, mean_degree=degree)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 300)
    for chains in (1, 7, 32, 64, 128):
        _lib.check(lib.asp_sa_set_team(ham.plan(), 0))
        ref_x, ref_e = sa.anneal_raw(ham, 99, betas, chains)
        solo = lib.asp_sa_last_sweep_ms(ham.plan())
        for team in (2, 4, 8, -1):
            if team > 0 and team * chains > 256:
                continue
            _lib.check(lib.asp_sa_set_team(ham.plan(), team))
            for rep in range(2):
                x, e = sa.anneal_raw(ham, 99, betas, chains)
                ok = np.array_equal(x, ref_x) and e.tobytes() == ref_e.tobytes()
                if not ok:
                    bad += 1
            print("K=%d chains=%3d team=%2d layout=%d: %.2f ms (single workgroup %.2f ms) %s" % (
                k, chains, team, lib.asp_sa_last_layout(ham.plan()),
                lib.asp_sa_last_sweep_ms(ham.plan()), solo, "ok" if ok else "MISMATCH"), flush=True)
print("done: %d mismatches" % bad)
J, h, _ = synthetic.planted_cluster(100000, seed=1, mean_degree=8.0)
ham = sa.Hamiltonian(J, h)
for team in (0, -1):
    _lib.check(lib.asp_sa_set_team(ham.plan(), team))
    t0 = time.time()
    x, e = sa.anneal(ham, seed=12345, number_sweeps=5120, repetitions=64, sweep_order="colour")
    print("anneal(5120 sweeps x 64) team=%d: %.3f s, E = %.12g" % (team, time.time() - t0, e), flush=True)

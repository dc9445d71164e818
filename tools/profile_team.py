"""The reference-default anneal call (5120 sweeps x 64 chains) on a 1e5-spin cluster, once as
teams and once one workgroup per chain — run under `rocprofv3 --kernel-trace --stats`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

lib = _lib.load()
J, h, _ = synthetic.planted_cluster(100000, seed=1, mean_degree=8.0)
ham = sa.Hamiltonian(J, h)
sa.anneal(ham, seed=1, number_sweeps=16, repetitions=64, sweep_order="colour")  # warm-up
for team in (-1, 0):
    _lib.check(lib.asp_sa_set_team(ham.plan(), team))
    t0 = time.time()
    x, e = sa.anneal(ham, seed=12345, number_sweeps=5120, repetitions=64, sweep_order="colour")
    print("team=%2d: %.3f s wall, sweep kernel %.1f ms, E = %.12g" % (
        team, time.time() - t0, lib.asp_sa_last_sweep_ms(ham.plan()), e), flush=True)
ham.release()  # handles are destroyed before the interpreter (and the profiler) wind down
_lib.shutdown()

"""Fixed small driver for rocprofv3 runs of the sweep kernel (development aid).

    rocprofv3 --kernel-trace --stats ... -- python tools/profile_sweep.py --size 100000
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--size", type=int, default=100000)
p.add_argument("--replicas", type=int, default=1024)
p.add_argument("--sweeps", type=int, default=32)
p.add_argument("--group", type=int, default=0)
p.add_argument("--threads", type=int, default=0)
p.add_argument("--runs", type=int, default=2)
p.add_argument("--kind", default="planted")
a = p.parse_args()
lib = _lib.load()
if a.kind == "sk":
    J, h = synthetic.sk_cluster(a.size)
else:
    J, h, _ = synthetic.planted_cluster(a.size, seed=783494)
ham = sa.Hamiltonian(J, h)
info = ham.info()
_lib.check(lib.asp_sa_set_launch(ham.plan(), a.group, a.threads))
betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, a.sweeps)
for _ in range(a.runs):
    sa.anneal_raw(ham, 1, betas, a.replicas)
    ms = lib.asp_sa_last_sweep_ms(ham.plan())
    print("K=%d R=%d sweeps=%d: sweep kernel %.3f ms, %.2f Gflips/s" % (
        a.size, a.replicas, a.sweeps, ms, a.size * a.replicas * a.sweeps / ms / 1e6), flush=True)

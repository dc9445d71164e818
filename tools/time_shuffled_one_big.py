"""One large sparse cluster in the shuffled order (development aid; GPU): K spins, mean degree D, 64 chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _lib, synthetic
from annealing_sign_problem_amd import annealer as sa
k = int(sys.argv[1]) if len(sys.argv) > 1 else 177000
d = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
J, h, _ = synthetic.planted_cluster(k, seed=3, mean_degree=d)
ham = sa.Hamiltonian(J, h)
info = ham.info()
betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, sweeps)
lib = _lib.load()
if len(sys.argv) > 5:
    lib.asp_sa_set_shuffled_launch(ham.plan(), int(sys.argv[4]), int(sys.argv[5]))
for _ in range(2):
    t0 = time.perf_counter()
    sa.anneal_raw(ham, 12345, betas, 64, shuffled=True)
    print("K = %d, d = %.1f: %d sweeps x 64 chains in %.1f ms (kernels %.1f ms)" % (k, d, sweeps, (time.perf_counter() - t0) * 1e3, lib.asp_sa_last_sweep_ms(ham.plan())), flush=True)

"""A round of the sampled-cluster pipeline's LARGE models in the shuffled order, without the pipeline: N
planted clusters, K log-uniform in [3e4, 2e5] (dbar = 24; the order-2 models of kagome_36 are 3.5e4 .. 3e5
spins), 64 chains x SWEEPS sweeps, one asp_sa_anneal_batch call.  (Development aid; GPU.)

    python tools/time_shuffled_big_batch.py [N=24] [sweeps=256]
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(4711)
sizes = [int(round(np.exp(rng.uniform(np.log(3e4), np.log(2e5))))) for _ in range(n)]
hams = []
for i, k in enumerate(sizes):
    J, h, _ = synthetic.planted_cluster(k, seed=4711 + i)
    ham = sa.Hamiltonian(J, h)
    ham.info()
    hams.append(ham)
flips = float(sum(sizes)) * 64 * sweeps
lib = _lib.load()
sa.anneal_batch(hams[:2], seed=1, number_sweeps=8, repetitions=64, sweep_order="shuffled")
for _ in range(2):
    t0 = time.perf_counter()
    sa.anneal_batch(hams, seed=12345, number_sweeps=sweeps, repetitions=64, sweep_order="shuffled")
    t = time.perf_counter() - t0
    print("shuffled batch of %d large problems (sum K = %d), 64 chains x %d sweeps: %.2f s = %.1f G flips/s (device %.0f ms)" % (
        n, sum(sizes), sweeps, t, flips / t / 1e9, lib.asp_sa_batch_last_ms()), flush=True)
for ham, k in sorted(zip(hams, sizes), key=lambda x: x[1])[::6]:
    spins, wgs = ctypes.c_uint32(0), ctypes.c_uint32(0)
    lib.asp_sa_last_shuffled_blocks(ham.plan(), ctypes.byref(spins), ctypes.byref(wgs))
    m, th, gr = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    lib.asp_sa_last_launch(ham.plan(), ctypes.byref(m), ctypes.byref(th), ctypes.byref(gr))
    print("  K = %6d: M = %d, %d wavefronts, %d workgroups, layout %d" % (
        k, m.value, th.value // 64, wgs.value, lib.asp_sa_last_layout(ham.plan())), flush=True)

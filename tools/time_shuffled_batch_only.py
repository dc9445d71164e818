"""The production mix with the shuffled order only (for rocprofv3): N planted clusters, K log-uniform in
[1e2, 1e4], 64 chains x SWEEPS sweeps, one asp_sa_anneal_batch call.  (Development aid; GPU.)

    python tools/time_shuffled_batch_only.py [N=128] [sweeps=5120] [repeat=1]

ASP_SHUFFLED_ABLATE=1 (sweeps through stale orders: the sweep kernels alone) / =2 (no sweep launches after
the first chunk: the order kernels alone) are honoured by a library built with -DASP_SHUF_ABLATE_ENV=1 only
(ASP_LIB_TAG=abl ASP_EXTRA_FLAGS=-DASP_SHUF_ABLATE_ENV=1): the results are WRONG, the times are the point.
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 5120
repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 1
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
lib = _lib.load()
for _ in range(repeat):
    t0 = time.perf_counter()
    sa.anneal_batch(hams, seed=12345, number_sweeps=sweeps, repetitions=64, sweep_order="shuffled")
    t = time.perf_counter() - t0
    print("shuffled batch of %d problems (sum K = %d), 64 chains x %d sweeps: %.2f s = %.1f G flips/s (device %.0f ms)" % (
        n, sum(sizes), sweeps, t, flips / t / 1e9, lib.asp_sa_batch_last_ms()), flush=True)
shapes = {}
for ham, k in zip(hams, sizes):
    spins, wgs = ctypes.c_uint32(0), ctypes.c_uint32(0)
    lib.asp_sa_last_shuffled_blocks(ham.plan(), ctypes.byref(spins), ctypes.byref(wgs))
    m, th, gr = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    lib.asp_sa_last_launch(ham.plan(), ctypes.byref(m), ctypes.byref(th), ctypes.byref(gr))
    key = (spins.value, th.value // 64, m.value)
    shapes.setdefault(key, []).append((k, wgs.value))
for key in sorted(shapes):
    ks = [k for k, _ in shapes[key]]
    print("  blocks of %2d spins, %d wavefronts, M=%d: %3d problems, K %d..%d, %d workgroups" % (
        key[0], key[1], key[2], len(ks), min(ks), max(ks), sum(w for _, w in shapes[key])), flush=True)

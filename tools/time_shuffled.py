"""Shuffled-order sweep (asp_sa_anneal_shuffled): rate and launch-geometry scan (development aid).

    python tools/time_shuffled.py [--sizes 12870] [--chains 1024] [--sweeps 128] [--groups 0] [--waves 0]
                                  [--degree 20] [--check]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--sizes", default="12870")
    p.add_argument("--chains", default="1024")
    p.add_argument("--sweeps", type=int, default=128)
    p.add_argument("--groups", default="0", help="comma list of chains per workgroup (0 = automatic)")
    p.add_argument("--waves", default="0", help="comma list of wavefronts per workgroup (0 = automatic)")
    p.add_argument("--teams", default="0", help="comma list: 0 automatic, 1, 2 teams per workgroup")
    p.add_argument("--degree", type=float, default=23.0)
    p.add_argument("--repeat", type=int, default=2)
    p.add_argument("--colour", action="store_true", help="also time the colour-ordered sweep")
    p.add_argument("--check", action="store_true", help="compare 4 chains x 6 sweeps with the oracle")
    a = p.parse_args()
    lib = _lib.load()
    for k in [int(s) for s in a.sizes.split(",")]:
        J, h, _ = synthetic.planted_cluster(k, seed=783494, mean_degree=a.degree)
        ham = sa.Hamiltonian(J, h)
        info = ham.info()
        betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, a.sweeps)
        print("K=%d nnz/K=%.1f max degree %d" % (k, J.nnz / k, info.max_degree), flush=True)
        if a.check:
            import oracle

            xs, es = sa.anneal_raw(ham, 7, betas[:6], 4, 1, None, shuffled=True)
            oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, 7, betas[:6], 4, 1, None, info.energy_scale_exp,
                                                       num_threads=4)
            print("  oracle parity:", bool(np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()), flush=True)
        for chains in [int(s) for s in a.chains.split(",")]:
            if a.colour:
                sa.anneal_raw(ham, 1, betas, chains)
                sa.anneal_raw(ham, 1, betas, chains)
                ms = lib.asp_sa_last_sweep_ms(ham.plan())
                print("  colour order: chains=%d  %8.2f ms  %7.2f Gflips/s" % (
                    chains, ms, k * chains * a.sweeps / ms / 1e6), flush=True)
            for m in [int(s) for s in a.groups.split(",")]:
                for w, teams in [(int(s), int(t)) for s in a.waves.split(",") for t in a.teams.split(",")]:
                    _lib.check(lib.asp_sa_set_shuffled_launch(ham.plan(), m, w))
                    _lib.check(lib.asp_sa_set_shuffled_teams(ham.plan(), teams))
                    best, wall = None, None
                    for _ in range(a.repeat):
                        t0 = time.perf_counter()
                        sa.anneal_raw(ham, 1, betas, chains, shuffled=True)
                        t1 = time.perf_counter()
                        ms = lib.asp_sa_last_sweep_ms(ham.plan())
                        best = ms if best is None else min(best, ms)
                        wall = (t1 - t0) if wall is None else min(wall, t1 - t0)
                    levels = ctypes.c_uint32(0)
                    lib.asp_sa_last_shuffled(ham.plan(), ctypes.byref(levels), None)
                    mm, th, gr = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
                    lib.asp_sa_last_launch(ham.plan(), ctypes.byref(mm), ctypes.byref(th), ctypes.byref(gr))
                    flips = k * chains * a.sweeps
                    print("  chains=%d M=%d teams=%d waves(all)=%d groups=%d levels<=%d: %8.2f ms  %7.2f Gflips/s  (wall %.1f ms)" % (
                        chains, mm.value, teams, th.value // 64, gr.value, levels.value, best, flips / best / 1e6,
                        wall * 1e3), flush=True)


if __name__ == "__main__":
    main()

"""Sweep-kernel launch-geometry scan (development aid, not part of the product).

    python tools/tune_sweep.py [--sizes 10000,30000,100000] [--replicas 1024] [--sweeps 32]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--sizes", default="10000,30000,100000")
    p.add_argument("--replicas", type=int, default=1024)
    p.add_argument("--sweeps", type=int, default=32)
    p.add_argument("--groups", default="1,2,4,8")
    p.add_argument("--threads", default="256,512,1024")
    p.add_argument("--kind", default="planted")
    p.add_argument("--cache", default="1", help="comma list of field-cache settings to time (0/1)")
    p.add_argument("--wide", default="1", help="comma list: 0 = byte layout only, 1 = word layout allowed")
    a = p.parse_args()
    lib = _lib.load()
    for k in [int(s) for s in a.sizes.split(",")]:
        if a.kind == "sk":
            J, h = synthetic.sk_cluster(k)
        else:
            J, h, _ = synthetic.planted_cluster(k, seed=783494)
        ham = sa.Hamiltonian(J, h)
        info = ham.info()
        betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, a.sweeps)
        print("K=%d nnz/K=%.1f colors=%d blocks=%d ell_pad=%.3f" % (
            k, J.nnz / k, info.num_colors, info.num_blocks, info.ell_entries / max(1, info.nnz_offdiag)),
            flush=True)
        for m in [int(s) for s in a.groups.split(",")]:
            for th in [int(s) for s in a.threads.split(",")]:
                _lib.check(lib.asp_sa_set_launch(ham.plan(), m, th))
                for cache in [int(c) for c in a.cache.split(",")]:
                    _lib.check(lib.asp_sa_set_field_cache(ham.plan(), cache))
                    for wide in [int(c) for c in a.wide.split(",")]:
                        _lib.check(lib.asp_sa_set_wide(ham.plan(), wide))
                        best = None
                        for _ in range(2):
                            sa.anneal_raw(ham, 1, betas, a.replicas)
                            ms = lib.asp_sa_last_sweep_ms(ham.plan())
                            best = ms if best is None else min(best, ms)
                        flips = k * a.replicas * a.sweeps
                        print("  M=%d threads=%4d cache=%d layout=%d  sweep %8.2f ms  %7.2f Gflips/s  total %.2f ms" % (
                            m, th, cache, lib.asp_sa_last_layout(ham.plan()), best, flips / best / 1e6,
                            lib.asp_sa_last_total_ms(ham.plan())), flush=True)


if __name__ == "__main__":
    main()

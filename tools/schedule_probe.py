"""Which annealing ladder reproduces the reference's published success probabilities?
The library behind the reference (ising_glass_annealer) is absent; its schedule, beta range and
sweep order are unknown.  This probe repeats a few points of `make small` (1024 chains x TRIALS)
under different ladders between the same automatic beta estimates and prints the z-score against
the published value (tests/golden/published_sa_curves.json).  (Development aid.)"""
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import annealer as sa  # noqa: E402
from annealing_sign_problem_amd import full_hilbert_space  # noqa: E402

TRIALS = int(os.environ.get("PROBE_TRIALS", "4"))
published = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                        "tests", "golden", "published_sa_curves.json")))["models"]


def ladders(b0, b1, n):
    t = np.arange(n) / max(n - 1, 1)
    return {
        "geometric": np.geomspace(b0, b1, n),
        "linear": np.linspace(b0, b1, n),
        "geometric, beta1 / 10": np.geomspace(b0, b1 / 10, n),
        "geometric, beta0 x 10": np.geomspace(b0 * 10, b1, n),
        "linear, beta1 / 10": np.linspace(b0, b1 / 10, n),
        "linear in T": 1.0 / np.linspace(1.0 / b0, 1.0 / b1, n),
        "quadratic": b0 + (b1 - b0) * t ** 2,
    }


points = [("heisenberg_kagome_16", 400), ("heisenberg_kagome_16", 3200), ("sk_16_3", 200),
          ("sk_16_3", 12800), ("heisenberg_kagome_18", 400), ("j1j2_square_4x4", 100)]
if len(sys.argv) > 1:
    points = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]]
sims = {}
for name, sweeps in points:
    if name not in sims:
        sims[name] = full_hilbert_space.Simulation(name)
    sim = sims[name]
    h = sim.exact_model.ising_hamiltonian
    info = h.info()
    row = published[name][str(sweeps)]
    print("%s @ %d sweeps: published %.4f +- %.4f" % (name, sweeps, row["acc_prob_mean"], row["acc_prob_std"]),
          flush=True)
    only = os.environ.get("PROBE_LADDERS")
    for label, betas in ladders(info.beta0_auto, info.beta1_auto, sweeps).items():
        if only and label not in only.split(";"):
            continue
        probs = []
        for trial in range(TRIALS):
            xs, es = sa.anneal_raw(h, 435834 + 1000003 * trial + sweeps, betas, 1024)
            probs.append(sim.analyze(xs, es)[0])
        mean, std = float(np.mean(probs)), float(np.std(probs))
        se = math.sqrt(row["acc_prob_std"] ** 2 / 10 + max(std, 0.005) ** 2 / TRIALS)
        print("    %-24s %.4f +- %.4f   z = %+5.1f" % (label, mean, std, (mean - row["acc_prob_mean"]) / se),
              flush=True)

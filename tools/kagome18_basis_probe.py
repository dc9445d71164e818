"""Is the reference's published kagome_18 curve (experiments/heisenberg_kagome_18.csv) that of the
inversion-symmetric basis its YAML names (24 310 representatives) or of the plain Sz = 0 basis
(48 620 states)?  Anneals both with both visiting orders and prints P(accuracy > 0.995) next to
the published values.  (Development aid; GPU.)"""
import copy
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import full_hilbert_space, synthetic  # noqa: E402

published = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden",
                                        "published_sa_curves.json")))["models"]["heisenberg_kagome_18"]
models = synthetic.load_models()
plain = copy.deepcopy(models["heisenberg_kagome_18"])
plain["basis"].pop("spin_inversion")
models["heisenberg_kagome_18_plain"] = plain
synthetic.load_models = lambda: models
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for name in ("heisenberg_kagome_18_plain", "heisenberg_kagome_18"):
    sim = full_hilbert_space.Simulation(name)
    print("%s: K = %d, E0 = %.12f" % (name, sim.exact_model.size, sim.energy), flush=True)
    for order in ("shuffled", "colour"):
        for sweeps in (100, 200, 400, 800):
            p = [sim.run(sweeps, 1024, seed=435834 + 1000003 * t + sweeps, sweep_order=order)[0]
                 for t in range(trials)]
            print("  %-8s %4d sweeps: P(acc>0.995) = %.4f +- %.4f   published %.4f" % (
                order, sweeps, np.mean(p), np.std(p), published[str(sweeps)]["acc_prob_mean"]), flush=True)

"""Writes tests/golden/published_sa_curves.json (build container only: reads /root/reference).

Data, not source: for every model with a published `experiments/<model>.csv` in the reference
(the output of `make small`, Makefile:27-35: 1024 repetitions x 10 trials per number of sweeps)
the rows (number_sweeps, acc_prob_mean, acc_prob_std, residual_prob_mean) together with the
line of the reference file each one comes from, and next to them this repository's round-1
MI355X measurement of the same experiment (profiles/full_hilbert_space/fhs_<model>.csv,
10 trials x 1024, 5 trials for the sk models).  tests/test_gpu_published.py asserts bands
around both."""
import csv
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
MODELS = ["heisenberg_kagome_16", "j1j2_square_4x4", "sk_16_1", "sk_16_2", "sk_16_3",
          "heisenberg_kagome_18"]  # (kagome_18: symmetry-adapted basis, no round-1 measurement)
TRIALS_R1 = {"heisenberg_kagome_16": 10, "j1j2_square_4x4": 10, "sk_16_1": 5, "sk_16_2": 5,
             "sk_16_3": 5, "heisenberg_kagome_18": 10}
ROUND = {"heisenberg_kagome_18": 2}  # measured in round 2 (symmetry-adapted basis); others round 1

out = {"source": "experiments/<model>.csv of twesterhout/annealing-sign-problem (make small)",
       "repetitions": 1024, "trials_published": 10, "models": {}}
for name in MODELS:
    path = "/root/reference/experiments/%s.csv" % name
    rows = {}
    with open(path) as f:
        for lineno, r in enumerate(csv.DictReader(f), start=2):
            rows[int(r["number_sweeps"])] = {
                "reference_line": "experiments/%s.csv:%d" % (name, lineno),
                "acc_prob_mean": float(r["acc_prob_mean"]),
                "acc_prob_std": float(r["acc_prob_std"]),
                "residual_prob_mean": float(r["residual_prob_mean"]),
            }
    mine = os.path.join(ROOT, "profiles", "full_hilbert_space", "fhs_%s.csv" % name)
    with open(mine if os.path.exists(mine) else os.devnull) as f:
        for r in csv.DictReader(f):
            k = int(r["number_sweeps"])
            if k in rows:
                rows[k]["mi355x_acc_prob_mean"] = float(r["acc_prob_mean"])
                rows[k]["mi355x_acc_prob_std"] = float(r["acc_prob_std"])
                rows[k]["mi355x_trials"] = TRIALS_R1[name]
                rows[k]["mi355x_round"] = ROUND.get(name, 1)
    out["models"][name] = {str(k): rows[k] for k in sorted(rows)}
with open(os.path.join(HERE, "published_sa_curves.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print("wrote", len(MODELS), "models")

"""Markdown tables for DESIGN.md §6.1 from tests/golden/published_sa_curves.json: the reference's
published P(accuracy > 0.995) next to this repository's MI355X measurement, with the z-score of
the difference (standard errors of the two means combined)."""
import json
import math
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = json.load(open(os.path.join(root, "tests", "golden", "published_sa_curves.json")))
for name in sys.argv[1:] or sorted(data["models"]):
    rows = data["models"][name]
    print("| %s: sweeps | reference ± std (10 trials) | MI355X ± std (trials) | difference | z |" % name)
    print("|---|---|---|---|---|")
    for k in sorted(rows, key=int):
        r = rows[k]
        if "mi355x_acc_prob_mean" not in r:
            continue
        n = r["mi355x_trials"]
        se = math.sqrt(r["acc_prob_std"] ** 2 / 10 + r["mi355x_acc_prob_std"] ** 2 / n)
        diff = r["mi355x_acc_prob_mean"] - r["acc_prob_mean"]
        z = "%+.1f" % (diff / se) if se > 0 else "—"
        print("| %s | %.4f ± %.4f | %.4f ± %.4f (%d) | %+.4f | %s |" % (
            k, r["acc_prob_mean"], r["acc_prob_std"], r["mi355x_acc_prob_mean"],
            r["mi355x_acc_prob_std"], n, diff, z))
    print()

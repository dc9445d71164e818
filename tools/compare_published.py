"""Markdown table: P(acc > 0.995) of this repo's full-Hilbert-space runs next to the reference's
published experiments/*.csv (build container only: reads /root/reference)."""
import csv
import sys

for name in sys.argv[1:]:
    mine = {int(r["number_sweeps"]): r for r in csv.DictReader(open("profiles/full_hilbert_space/fhs_%s.csv" % name))}
    ref = {int(r["number_sweeps"]): r for r in csv.DictReader(open("/root/reference/experiments/%s.csv" % name))}
    print("| %s: sweeps | reference P(acc>0.995) ± std | MI355X P(acc>0.995) ± std | reference = P(residual) | MI355X P(residual≤1e-12) |" % name)
    print("|---|---|---|---|---|")
    for k in sorted(mine):
        r, m = ref.get(k), mine[k]
        print("| %d | %s | %.4f ± %.4f | %s | %.4f |" % (
            k, "%.4f ± %.4f" % (float(r["acc_prob_mean"]), float(r["acc_prob_std"])) if r else "-",
            float(m["acc_prob_mean"]), float(m["acc_prob_std"]),
            "%.4f" % float(r["residual_prob_mean"]) if r else "-", float(m["residual_prob_mean"])))
    print()

"""Accuracy distribution per extension order of a sampled-cluster CSV (the quantity the
reference plots in experiments/density.gnu:30-34, figure experiments/density_kagome.png: kernel
density of the greedy accuracy over [0.8, 1] for "not extended" ... "extended three times").
Prints quantiles, the mass above 0.8 / 0.95 and a Gaussian kernel density on a grid.
(Development aid.)"""
import sys

import numpy as np
import scipy.stats

path = sys.argv[1]
column = {"greedy": 1, "sa": 3}[sys.argv[2] if len(sys.argv) > 2 else "greedy"]
rows = [line.strip().split(",") for line in open(path) if not line.startswith(("#", "size"))]
data = np.array(rows, dtype=float)
orders = data.shape[1] // 6
grid = np.array([0.80, 0.85, 0.90, 0.95, 0.97, 0.98, 0.99, 1.00])
print("%s: %d clusters, %d orders, column %s" % (path, data.shape[0], orders, sys.argv[2] if len(sys.argv) > 2 else "greedy"))
print("order  median size   quantiles 10/25/50/75/90            P(>0.8)  P(>0.95)  density at " +
      " ".join("%.2f" % g for g in grid))
for o in range(orders):
    block = data[:, 6 * o: 6 * o + 6]
    v = block[:, column]
    v = v[~np.isnan(v)]
    kde = scipy.stats.gaussian_kde(v)
    print("%5d  %11d   %s   %.3f    %.3f     %s" % (
        o, np.median(block[:, 0]), " ".join("%.3f" % q for q in np.quantile(v, [.1, .25, .5, .75, .9])),
        (v > 0.8).mean(), (v > 0.95).mean(), " ".join("%5.2f" % d for d in kde(grid))))

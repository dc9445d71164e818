"""Repeats the GPU component extraction on chain-like graphs (long parent chains, the hard case
for a concurrent union-find) and compares with scipy every time (development aid)."""
import os
import sys

import numpy as np
import scipy.sparse

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from annealing_sign_problem_amd import common  # noqa: E402

rng = np.random.default_rng(1)
bad = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    n = int(rng.integers(50, 60000))
    # a few long paths in random vertex order + sparse random edges
    perm = rng.permutation(n)
    cut = rng.random(n - 1) < 0.98
    rows = np.concatenate([perm[:-1][cut], rng.integers(0, n, n // 20)])
    cols = np.concatenate([perm[1:][cut], rng.integers(0, n, n // 20)])
    vals = rng.normal(size=rows.size) * np.exp(rng.normal(size=rows.size) * 3)
    a = scipy.sparse.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    a = (a + a.T).tocsr() if trial % 2 == 0 else a
    a.sum_duplicates()
    a.sort_indices()
    frozen = np.zeros(n, dtype=bool)
    anchor = int(rng.integers(n))
    frozen[anchor] = True
    reltol = [0.0, 1e-4, 1e-2][trial % 3]
    keep_o, block_o = oracle.sparsify_component(a, frozen, reltol, anchor)
    keep, block = common.sparsify_component(a, frozen, reltol, anchor)
    bo = scipy.sparse.csr_matrix(block_o)
    bo.sort_indices()
    ok = (np.array_equal(keep, keep_o) and np.array_equal(block.indices, bo.indices)
          and block.data.tobytes() == bo.data.tobytes())
    if not ok:
        bad += 1
        print("MISMATCH trial", trial, "n", n, "kept", keep.sum(), "expected", keep_o.sum(), flush=True)
print("done: %d mismatches" % bad)

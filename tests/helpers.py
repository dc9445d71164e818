"""Shared input generators for the tests (and for tests/golden/generate_golden.py)."""
import numpy as np

import oracle


def random_build_case(rng, num_spins, mean_other, multiword, miss_rate, repeat_counts):
    if multiword:
        table = rng.integers(0, 4, size=(num_spins * 2, 8), dtype=np.uint64)
        table[:, 0] = rng.integers(0, 6, size=num_spins * 2, dtype=np.uint64)
        table = np.unique(table, axis=0)  # lexicographic by column 0 first = ls_bits512_cmp
        table = table[np.sort(rng.choice(table.shape[0], size=min(num_spins, table.shape[0]),
                                         replace=False))]
    else:
        keys = np.unique(rng.integers(0, 1 << 40, size=num_spins * 2, dtype=np.uint64))[:num_spins]
        table = oracle.as_keys512(np.sort(keys))
    n = table.shape[0]
    other_counts = rng.poisson(mean_other, size=n).astype(np.int64)
    other_counts[rng.random(n) < 0.1] = 0
    m = int(other_counts.sum())
    pick = rng.integers(0, max(n, 1), size=m)
    other = table[pick].copy() if n else np.zeros((0, 8), np.uint64)
    miss = rng.random(m) < miss_rate
    which = rng.integers(0, 8 if multiword else 1, size=m)
    other[miss, which[miss]] ^= np.uint64(1) << rng.integers(0, 40, size=int(miss.sum())).astype(np.uint64)
    counts = (rng.integers(1, 4, size=n) if repeat_counts else np.ones(n)).astype(np.int64)
    psi = rng.normal(size=n)
    psi[rng.random(n) < 0.05] = 0.0
    return dict(spins=table, counts=counts, psi=psi, other_spins=other,
                other_coeffs=rng.normal(size=m) * 2, other_counts=other_counts,
                other_psi=rng.normal(size=m))

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


def random_operator(seed, kind, number_spins=14, num_bonds=20, num_keys=700):
    """A random two-site operator, a sorted cluster of random states and unit-norm amplitudes.

    kind: "exchange" (symmetric c*sigma.sigma-like, distinct bonds), "general" (all 16 real
    entries, symmetric pattern, distinct bonds incl. pair flips |00><11|), "one_way" (some
    off-diagonal elements present in one direction only), "single_flip" (elements that flip
    one site: bonds sharing a site then reach the same state twice; some one-directional),
    "double_reach" (the same with every element mirrored)."""
    from annealing_sign_problem_amd import operators

    rng = np.random.default_rng(seed)
    pairs = set()
    while len(pairs) < num_bonds:
        a, b = rng.choice(number_spins, size=2, replace=False)
        if (min(a, b), max(a, b)) not in {(min(x, y), max(x, y)) for x, y in pairs}:
            pairs.add((int(a), int(b)))
    pairs = sorted(pairs)
    terms = []
    for a, b in pairs:
        m = np.zeros((4, 4))
        m[np.diag_indices(4)] = rng.normal(size=4)
        if kind == "exchange":
            m[1, 2] = m[2, 1] = rng.normal()
        elif kind == "general":
            m[1, 2], m[2, 1] = rng.normal(size=2)
            m[0, 3], m[3, 0] = rng.normal(size=2)
        elif kind == "one_way":
            m[1, 2] = rng.normal()
            if rng.random() < 0.5:
                m[2, 1] = rng.normal()
            m[3, 0] = rng.normal()
        elif kind == "double_reach":
            # symmetric single-site flips: bonds sharing a site reach the same state twice, and
            # every element has its mirror (the device build with duplicate targets applies)
            m[1, 2] = m[2, 1] = rng.normal()
            m[0, 1] = m[1, 0] = rng.normal()            # flips site b
            m[2, 3] = m[3, 2] = rng.normal()            # flips site b
        elif kind == "single_flip":
            m[1, 2] = m[2, 1] = rng.normal()
            m[0, 1], m[1, 0] = rng.normal(size=2)      # flips site b
            m[3, 1] = rng.normal()                      # flips site a, one way
        else:
            raise ValueError(kind)
        terms.append(operators.Term(m, [(a, b)]))
    op = operators.Operator(operators.SpinBasis(number_spins), terms)
    # a cluster with many internal connections: a random walk through the operator's targets
    keys = {int(rng.integers(0, 2 ** number_spins))}
    frontier = list(keys)
    while len(keys) < num_keys:
        other, _, _ = op.batched_apply(np.array(frontier, dtype=np.uint64))
        cand = rng.permutation(np.unique(other[:, 0]))[: max(8, num_keys // 10)]
        fresh = [int(x) for x in cand if int(x) not in keys]
        if not fresh:
            fresh = [int(rng.integers(0, 2 ** number_spins))]
        keys.update(fresh[: num_keys - len(keys)])
        frontier = fresh
    keys = np.array(sorted(keys), dtype=np.uint64)
    psi = rng.normal(size=keys.shape[0]) * np.exp(rng.normal(size=keys.shape[0]) * 1.5)
    psi /= np.linalg.norm(psi)
    return op, keys, psi


def reference_route_ising(op, keys, psi):
    """make_ising_model's arithmetic with numpy + scipy (common.py:71-82, 116-128, 172-173,
    190-196), from the numpy operator's connections: COO sorted by (row, col)."""
    import scipy.sparse

    other, coeffs, counts = op.batched_apply(keys)
    other = other[:, 0]
    coeffs = np.ascontiguousarray(coeffs.real)
    idx = np.clip(np.searchsorted(keys, other), 0, keys.size - 1)
    member = keys[idx] == other
    offsets = np.concatenate([[0], np.cumsum(counts)])
    elements = coeffs * np.abs(np.where(member, psi[idx], 0))
    elements = elements * np.abs(psi[np.repeat(np.arange(keys.size), counts)])
    m = scipy.sparse.csr_matrix((elements, idx, offsets), shape=(keys.size, keys.size))
    m = 0.5 * (m + m.T)
    m.sort_indices()
    return m.tocoo()

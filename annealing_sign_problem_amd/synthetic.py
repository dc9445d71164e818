"""Synthetic instances of the sizes named in BASELINE.json.

The reference's inputs are exact-diagonalisation ground states that its Makefile
downloads (Makefile:143-153); they are not available offline, so benchmarks use
planted random instances with the degree statistics of the named models
 and the amplitude-weighted coupling structure
``J_ij = H_ij |psi_i| |psi_j|`` of annealing_sign_problem/common.py:71-82,194.
All randomness is ``numpy.random.default_rng(seed)``.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Tuple

import numpy as np
import scipy.sparse

_HERE = os.path.dirname(os.path.abspath(__file__))


def random_symmetric_graph(num_spins: int, mean_degree: float, max_degree: int,
                           rng: np.random.Generator) -> Tuple[np.ndarray, np.ndarray]:
    """Edges (i < j) of a random graph with ~Poisson(mean_degree) degrees, capped."""
    n = int(num_spins)
    m = int(round(n * mean_degree / 2))
    a = rng.integers(0, n, size=m, dtype=np.int64)
    b = rng.integers(0, n, size=m, dtype=np.int64)
    keep = a != b
    lo, hi = np.minimum(a[keep], b[keep]), np.maximum(a[keep], b[keep])
    code = np.unique(lo * n + hi)
    lo, hi = code // n, code % n
    # cap: rank every edge among the edges of each of its endpoints (random order) and
    # drop it when either rank reaches the cap, so no vertex keeps more than max_degree
    order = rng.permutation(lo.shape[0])
    lo, hi = lo[order], hi[order]
    ends = np.concatenate([lo, hi])
    by_vertex = np.argsort(ends, kind="stable")
    starts = np.concatenate([[0], np.cumsum(np.bincount(ends, minlength=n))])
    rank = np.empty(ends.shape[0], dtype=np.int64)
    rank[by_vertex] = np.arange(ends.shape[0]) - starts[ends[by_vertex]]
    m_edges = lo.shape[0]
    admit = (rank[:m_edges] < max_degree) & (rank[m_edges:] < max_degree)
    return lo[admit], hi[admit]


def planted_cluster(num_spins: int, mean_degree: float = 23.0, max_degree: int = 37,
                    seed: int = 783494, frustrated_fraction: float = 0.05,
                    diagonal_range: int = 18, amplitude_sigma: float = 1.5):
    """kagome_36-/pyrochlore-sized cluster.

    Returns ``(J csr, field, planted_signs)``: amplitudes ``a = exp(N(0, sigma^2))``
    L2-normalised, ``J_ij = -2 a_i a_j s*_i s*_j f_ij`` with ``f = -1`` on a small
    fraction of bonds, ``J_ii = a_i^2 c_i`` with integer ``c_i``.
    """
    rng = np.random.default_rng(seed)
    n = int(num_spins)
    lo, hi = random_symmetric_graph(n, mean_degree, max_degree, rng)
    a = np.exp(rng.normal(0.0, amplitude_sigma, size=n))
    a /= np.linalg.norm(a)
    planted = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    f = np.where(rng.random(lo.shape[0]) < frustrated_fraction, -1.0, 1.0)
    w = -2.0 * a[lo] * a[hi] * planted[lo] * planted[hi] * f
    c = rng.integers(-diagonal_range, diagonal_range + 1, size=n).astype(np.float64)
    rows = np.concatenate([lo, hi, np.arange(n)])
    cols = np.concatenate([hi, lo, np.arange(n)])
    vals = np.concatenate([w, w, a * a * c])
    matrix = scipy.sparse.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    matrix.sum_duplicates()
    matrix.sort_indices()
    return matrix, np.zeros(n, dtype=np.float64), planted


def sk_cluster(num_spins: int = 8192, degree: int = 256, noise: float = 0.79,
               seed: int = 783494, amplitude_sigma: float = 1.5):
    """Dense random-J cluster of the sk_32_1 + NOISE shape:
    ``J_ij = 2 g_ij a_i a_j``, ``g ~ N(0, 1)`` symmetric, amplitudes multiplied by
    ``exp(noise * U(-1, 1))`` and renormalised (common.py:832-834)."""
    rng = np.random.default_rng(seed)
    n = int(num_spins)
    lo, hi = random_symmetric_graph(n, float(degree), degree + 64, rng)
    a = np.exp(rng.normal(0.0, amplitude_sigma, size=n))
    a *= np.exp(noise * 2 * (rng.random(n) - 0.5))
    a /= np.linalg.norm(a)
    g = rng.normal(0.0, 1.0, size=lo.shape[0])
    w = 2.0 * g * a[lo] * a[hi]
    rows = np.concatenate([lo, hi])
    cols = np.concatenate([hi, lo])
    matrix = scipy.sparse.coo_matrix((np.concatenate([w, w]), (rows, cols)), shape=(n, n)).tocsr()
    matrix.sum_duplicates()
    matrix.sort_indices()
    return matrix, np.zeros(n, dtype=np.float64)


def build_inputs_from_matrix(matrix, miss_fraction: float = 0.3, seed: int = 1,
                             key_bits: int = 36):
    """Inputs of ``build_matrix`` whose hits reproduce the off-diagonal structure of
    ``matrix``: sorted random ``key_bits``-bit keys, per row its neighbours' keys
    plus ``miss_fraction`` extra keys outside the table, shuffled; counts = 1.
    Returns the seven input arrays (keys as plain uint64)."""
    rng = np.random.default_rng(seed)
    m = scipy.sparse.csr_matrix(matrix)
    n = m.shape[0]
    pool = np.unique(rng.integers(0, 1 << key_bits, size=int(n * 2.5) + 64, dtype=np.uint64))
    pool = rng.permutation(pool)
    keys = np.sort(pool[:n])
    outside = pool[n:]
    degree = np.diff(m.indptr)
    extra = rng.binomial(np.maximum(degree, 1), miss_fraction)
    other_counts = (degree + extra).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(other_counts)])
    total = int(offsets[-1])
    other = np.empty(total, dtype=np.uint64)
    is_extra = np.zeros(total, dtype=bool)
    # hits first, extras after, then a per-row shuffle via random sort keys
    row_of = np.repeat(np.arange(n), other_counts)
    within = np.arange(total) - offsets[row_of]
    hit = within < degree[row_of]
    other[hit] = keys[m.indices]
    is_extra[~hit] = True
    other[~hit] = outside[rng.integers(0, outside.shape[0], size=int((~hit).sum()))]
    shuffle = np.lexsort((rng.random(total), row_of))
    other = other[shuffle]
    psi = rng.normal(size=n)
    psi /= np.linalg.norm(psi)
    other_psi = rng.normal(size=total) * 0.1
    other_coeffs = np.where(rng.random(total) < 0.5, 2.0, -1.0) * rng.integers(1, 4, size=total)
    counts = np.ones(n, dtype=np.int64)
    return keys, counts, psi, other, other_coeffs.astype(np.float64), other_counts, other_psi


def load_models() -> Dict[str, dict]:
    """Model definitions (bonds and two-site matrices) of the symmetry-free systems
    in physical_systems/*.yaml, converted to JSON by tests/golden/generate_golden.py."""
    with open(os.path.join(_HERE, "models.json")) as f:
        return json.load(f)


def kagome_lattice(l1: int = 4, l2: int = 3, coupling: float = 1.0) -> dict:
    """Heisenberg model on an l1 x l2 periodic kagome lattice as a models.json-style config
    (3 sites and 6 bonds per unit cell; 4 x 3 gives the 36 sites / 72 bonds of
    physical_systems/heisenberg_kagome_36.yaml:30-37, here WITHOUT its lattice symmetries).
    Site (x, y, s) has index 3 * (y * l1 + x) + s."""
    def site(x, y, s):
        return 3 * ((y % l2) * l1 + (x % l1)) + s

    bonds = []
    for y in range(l2):
        for x in range(l1):
            bonds += [(site(x, y, 0), site(x, y, 1)), (site(x, y, 0), site(x, y, 2)),
                      (site(x, y, 1), site(x, y, 2)),                      # up triangle
                      (site(x, y, 1), site(x + 1, y, 0)),
                      (site(x + 1, y, 0), site(x + 1, y - 1, 2)),
                      (site(x, y, 1), site(x + 1, y - 1, 2))]              # down triangle
    unique = sorted({(min(a, b), max(a, b)) for a, b in bonds})
    if len(unique) != len(bonds):
        raise ValueError("lattice too small: periodic images coincide")
    c = float(coupling)
    matrix = [[c, 0, 0, 0], [0, -c, 2 * c, 0], [0, 2 * c, -c, 0], [0, 0, 0, c]]
    n = 3 * l1 * l2
    return {"basis": {"number_spins": n, "hamming_weight": n // 2, "symmetries": []},
            "hamiltonian": {"terms": [{"matrix": matrix, "sites": [list(b) for b in unique]}]}}


def hashed_log_amplitudes(spins, sigma: float = 1.5, seed: int = 0) -> np.ndarray:
    """Deterministic stand-in for a wavefunction on states that cannot be enumerated:
    ``log psi(s) = sigma * N(0,1) + i*pi*bit`` with the normal and the bit drawn from a
    splitmix64 hash of the state (reproducible for any subset, like an NN amplitude)."""
    x = np.asarray(spins, dtype=np.uint64)
    if x.ndim > 1:
        x = x[:, 0]

    def mix(z):
        z = (z + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))

    with np.errstate(over="ignore"):
        a = mix(x ^ np.uint64(seed))
        b = mix(a)
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = ((b >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    normal = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    negative = (b & np.uint64(1)).astype(np.float64)
    return sigma * normal + 1j * np.pi * negative


def grow_cluster(hamiltonian, start: int, size: int, keep_probability: float = 0.5,
                 seed: int = 0) -> np.ndarray:
    """Breadth-first cluster around ``start`` with whole frontiers applied at once (a batched
    version of common.py:481-513 for clusters of 1e4..1e5 states)."""
    rng = np.random.default_rng(seed)
    members = np.array([start], dtype=np.uint64)
    frontier = members
    while members.shape[0] < size and frontier.shape[0] > 0:
        other, _, _ = hamiltonian.batched_apply(frontier)
        cand = np.setdiff1d(np.unique(other[:, 0]), members)
        cand = cand[rng.random(cand.shape[0]) <= keep_probability]
        if members.shape[0] + cand.shape[0] > size:
            cand = rng.permutation(cand)[: size - members.shape[0]]
        members = np.union1d(members, cand)
        frontier = cand
    return members

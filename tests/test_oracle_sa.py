"""CPU: pin what can be pinned of the annealer specification (ASP-SA-1).

The reference's annealer (ising_glass_annealer) is unavailable, so parity with
it is UNPINNED; these tests pin the building blocks against independent known
answers: Philox4x32-10 vs the Random123 KAT vectors, exp vs libm, the energy vs
numpy, the colouring vs its defining property, and the chain vs first principles
on tiny systems."""
import math

import numpy as np
import pytest
import scipy.sparse

import oracle


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, out in kat:
        assert oracle.philox4x32_10(ctr, key).tolist() == out


def test_expneg_accuracy_and_cutoff():
    xs = np.concatenate([np.linspace(1e-9, 22.999, 5001), [0.5, 1.0, math.log(2), 22.99999]])
    for x in xs:
        ref = math.exp(-x)
        assert abs(oracle.expneg(float(x)) - ref) <= 4e-16 * ref
    assert oracle.expneg(23.0) == 0.0 and oracle.expneg(1e300) == 0.0
    assert oracle.expneg(float("inf")) == 0.0 and oracle.expneg(float("nan")) == 0.0
    assert oracle.expneg(0.0) == 1.0


def _random_problem(n, density, seed, symmetric=True, field=True):
    rng = np.random.default_rng(seed)
    m = scipy.sparse.random(n, n, density=density, random_state=seed, format="csr")
    m.data = rng.normal(size=m.data.shape)
    if symmetric:
        m = (m + m.T).tocsr()
    m = (m + scipy.sparse.diags(rng.normal(size=n))).tocsr()
    h = rng.normal(size=n) if field else np.zeros(n)
    return m, h


@pytest.mark.parametrize("symmetric", [True, False])
def test_energy_equals_numpy(symmetric):
    J, h = _random_problem(777, 0.02, 5, symmetric)
    rng = np.random.default_rng(6)
    words = (777 + 63) // 64
    xs = rng.integers(0, 2**63, size=(5, words), dtype=np.uint64)
    es = oracle.sa_energy(J, h, xs)
    for x, e in zip(xs, es):
        s = 2.0 * ((x[np.arange(777) // 64] >> (np.arange(777) % 64).astype(np.uint64)) & np.uint64(1)) - 1.0
        ref = s @ (J @ s) + h @ s
        assert abs(e - ref) <= 1e-12 * abs(ref)


def test_layout_is_a_proper_colouring_and_permutation():
    J, _ = _random_problem(1500, 0.01, 7)
    colors, order, ncol, nnz, diag = oracle.sa_layout(J)
    A = (J + J.T).tocsr()
    A.setdiag(0)
    A.eliminate_zeros()
    coo = A.tocoo()
    assert np.all(colors[coo.row] != colors[coo.col])            # proper
    assert sorted(order.tolist()) == list(range(1500))            # permutation
    assert np.all(np.diff(colors[order]) >= 0) and ncol == colors.max() + 1
    assert nnz == A.nnz and abs(diag - J.diagonal().sum()) < 1e-12
    # DSATUR, restated in Python with the same tie-breaks: most distinct
    # neighbour colours, then larger degree, then smaller index; smallest free colour
    import heapq

    indptr, idx = A.indptr, A.indices
    n = 1500
    deg = np.diff(indptr)
    col = np.full(n, -1)
    seen = [set() for _ in range(n)]
    heap = [(0, -int(deg[i]), i) for i in range(n)]
    heapq.heapify(heap)
    while heap:
        s_neg, _, v = heapq.heappop(heap)
        if col[v] >= 0 or -s_neg != len(seen[v]):
            continue
        c = 0
        while c in seen[v]:
            c += 1
        col[v] = c
        for u in idx[indptr[v]:indptr[v + 1]]:
            if col[u] < 0 and c not in seen[u]:
                seen[u].add(c)
                heapq.heappush(heap, (-len(seen[u]), -int(deg[u]), int(u)))
    assert np.array_equal(col, colors)


def test_chain_single_spin_first_principles():
    """One spin, field h: dE = -2 s h; at beta = 0 every proposal is accepted, so the spin
    alternates; the tracked best energy is min over visited states in units of 2^-S."""
    J = scipy.sparse.csr_matrix((1, 1))
    h = np.array([0.75])
    xs, es, tracked, accepted = oracle.sa_anneal(J, h, 1, np.zeros(5), 4, 0, None, 10)
    assert accepted.tolist() == [5, 5, 5, 5]
    assert set(es.tolist()) == {-0.75}
    # chains that start at s=+1 (E=+0.75) improve by 1.5 = 1536 * 2^-10; the others never improve
    assert set(tracked.tolist()) <= {0, -1536}
    # frozen: beta huge, start from the minimum -> nothing is accepted
    x0 = np.array([0], dtype=np.uint64)  # s = -1, E = -0.75
    xs, es, tracked, accepted = oracle.sa_anneal(J, h, 1, np.full(5, 1e9), 3, 0, x0, 10)
    assert accepted.tolist() == [0, 0, 0] and es.tolist() == [-0.75] * 3 and xs.ravel().tolist() == [0, 0, 0]


def test_chain_finds_ground_state_of_unfrustrated_ring():
    n = 64
    rows = np.arange(n)
    J = scipy.sparse.coo_matrix((np.full(n, -0.5), (rows, (rows + 1) % n)), shape=(n, n))
    J = (J + J.T).tocsr()  # ferromagnetic ring, E_min = -2 * 0.5 * n = -n
    _, _, ncol, _, _ = oracle.sa_layout(J)
    assert ncol <= 3
    betas = np.geomspace(0.1, 20.0, 300)
    xs, es, _, _ = oracle.sa_anneal(J, np.zeros(n), 42, betas, 8, 0, None, 40, num_threads=4)
    assert es.min() == -float(n)
    best = xs[int(np.argmin(es))][0]
    assert best in (0, 2**64 - 1)


def test_chains_depend_only_on_global_replica_id_and_seed():
    J, h = _random_problem(300, 0.03, 9)
    betas = np.geomspace(0.2, 30, 20)
    full = oracle.sa_anneal(J, h, 77, betas, 12, 0, None, 30, num_threads=3)
    part = oracle.sa_anneal(J, h, 77, betas, 5, 4, None, 30, num_threads=1)
    for a, b in zip(full, part):
        assert np.array_equal(a[4:9], b)
    other = oracle.sa_anneal(J, h, 78, betas, 12, 0, None, 30)
    assert not np.array_equal(full[0], other[0])
    # tracked best (fixed point) is consistent with the exact energies: E_best - E_start
    x0 = np.zeros((300 + 63) // 64, dtype=np.uint64)
    xs, es, tracked, _ = oracle.sa_anneal(J, h, 5, betas, 4, 0, x0, 30)
    e_start = oracle.sa_energy(J, h, x0.reshape(1, -1))[0]
    assert np.allclose(tracked * 2.0**-30, es - e_start, rtol=0, atol=1e-6)


def test_shuffled_orders_of_the_product_are_the_oracles_sequential_order():
    """the product visits LEVELS of the priority graph (host code of libasp_hip,
    csrc/sa_plan.cpp) where the oracle visits spins one by one in ascending (priority, index).
    The two are the same Markov step iff the level-major order is a linear extension of the
    priority order on every edge and a level holds no two neighbours."""
    import ctypes

    from annealing_sign_problem_amd import _lib, synthetic

    lib = _lib.load()
    olib = oracle.lib()
    for n, seed, degree in [(1, 3, 1.0), (65, 4, 5.0), (700, 5, 12.0)]:
        J, h, _ = synthetic.planted_cluster(n, seed=seed, mean_degree=min(degree, max(n / 3, 1.0)))
        m = J.tocsr()
        indptr, indices = m.indptr.astype(np.int64), m.indices.astype(np.int32)
        for sweep in (0, 7, 4000):
            order = np.zeros(n, np.uint32)
            level = np.zeros(n, np.uint32)
            levels = ctypes.c_uint32(0)
            _lib.check(lib.asp_sa_shuffled_order_host(
                n, _lib.ptr(indptr), _lib.ptr(indices), _lib.ptr(m.data), _lib.ptr(h), 999, sweep,
                _lib.ptr(order), _lib.ptr(level), ctypes.byref(levels)))
            assert sorted(order.tolist()) == list(range(n))
            # priorities as the oracle draws them
            prio = np.zeros(n, np.uint64)
            for i in range(n):
                ctr = (ctypes.c_uint32 * 4)(i, sweep, 0xFFFFFFFE, 0)
                key = (ctypes.c_uint32 * 2)(999, 0)
                out = (ctypes.c_uint32 * 4)()
                olib.oracle_philox4x32_10(ctr, key, out)
                prio[i] = (int(out[0]) << 32) | i
            level_of_spin = np.zeros(n, np.int64)
            level_of_spin[order] = level
            a = (m + m.T).tocoo()
            off = a.row != a.col
            r, c = a.row[off], a.col[off]
            assert np.all(level_of_spin[r] != level_of_spin[c])            # a level: no two neighbours
            earlier = prio[r] < prio[c]
            assert np.all(level_of_spin[r][earlier] < level_of_spin[c][earlier])  # linear extension
            assert int(level.max()) + 1 == levels.value

"""CPU: host-side logic of the package and the C-ABI surface (no device compute)."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse

import oracle
from conftest import ROOT, golden


def test_c_abi_exports_every_declared_symbol():
    from annealing_sign_problem_amd import _lib

    header = open(os.path.join(ROOT, "include", "asp.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*\(", header)) - {"defined"}
    declared = {d for d in declared if d.startswith("asp_") or d in ("build_matrix", "extract_signs")}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()  # resolves every symbol or raises
    raw = ctypes.CDLL(_lib.library_path())
    for name in declared:
        assert getattr(raw, name) is not None
    assert lib.asp_version().decode().count(".") == 2


def test_fails_loudly_without_gpu():
    from annealing_sign_problem_amd import _build_matrix, _lib
    from annealing_sign_problem_amd import annealer as sa

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.AspError) as err:
        _build_matrix.build_matrix(np.arange(3, dtype=np.uint64), np.ones(3, np.int64), np.ones(3),
                                   np.arange(3, dtype=np.uint64), np.ones(3),
                                   np.ones(3, np.int64), np.ones(3))
    assert err.value.code == -1
    ham = sa.Hamiltonian(scipy.sparse.identity(4, format="csr"), np.zeros(4))
    with pytest.raises(_lib.AspError):
        sa.anneal(ham, seed=1, number_sweeps=2, repetitions=1)
    with pytest.raises(_lib.AspError):
        _build_matrix.extract_signs(np.ones(5))


def test_large_bases_need_the_device_and_say_so(models):
    """Beyond the host limits the representatives, the index lookup and the ground state come from
    the GPU (sector_ed.py, csrc/sector_basis.hip, csrc/key_table.hip): no silent host fallback."""
    from annealing_sign_problem_amd import _lib, operators, sector_ed

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    op = operators.Operator.from_config(models["heisenberg_kagome_36"])
    with pytest.raises(RuntimeError, match="needs a GPU"):
        op.basis.build()
    with pytest.raises(RuntimeError, match="needs a GPU"):
        sector_ed.ground_state(op)
    big = operators.SpinBasis(40)
    big.build(np.arange(10, dtype=np.uint64))
    big.DEVICE_INDEX_LIMIT = 5
    with pytest.raises(_lib.AspError):
        big.batched_index(np.array([3], dtype=np.uint64))
    # the small models keep the host route
    small = operators.Operator.from_config(models["heisenberg_kagome_16"])
    small.basis.build()
    assert small.basis.number_states == 12870 and small.basis.index(int(small.basis.states[5])) == 5
    assert sector_ed.binomial(36, 18) == 9075135300


def test_bit_packing_conventions():
    from annealing_sign_problem_amd import annealer as sa

    rng = np.random.default_rng(0)
    for n in [1, 63, 64, 65, 1000]:
        signs = rng.choice([-1.0, 1.0], size=n)
        bits = sa.signs_to_bits(signs)
        assert bits.dtype == np.uint64 and bits.shape == ((n + 63) // 64,)
        assert np.array_equal(bits, oracle.extract_signs(signs))  # cbits/build_matrix.c:72-74
        back = sa.bits_to_signs(bits, n)
        assert back.dtype == np.float64 and np.array_equal(back, signs)
    assert sa.signs_to_bits(np.array([0.0, 1.0, -1.0, 0.0]))[0] == 0b0010  # sign(0) -> bit clear
    with pytest.raises(ValueError):
        sa.bits_to_signs(np.zeros(1, np.uint64), 65)


def test_schedule():
    from annealing_sign_problem_amd import annealer as sa

    b = sa.make_schedule(0.5, 50.0, 11)
    assert b.shape == (11,) and b[0] == 0.5 and b[-1] == 50.0
    assert np.allclose(b[1:] / b[:-1], (50.0 / 0.5) ** 0.1)
    assert sa.make_schedule(1.0, 2.0, 0).shape == (0,)
    assert sa.make_schedule(1.0, 2.0, 1).tolist() == [2.0]
    with pytest.raises(ValueError):
        sa.make_schedule(0.0, 1.0, 5)


def test_accuracy_and_overlap_match_reference_golden():
    from annealing_sign_problem_amd import common

    g = golden("accuracy_overlap.npz")
    n = int(g["number_spins"])
    for pred, acc, ov in zip(g["predicted"], g["accuracy"], g["overlap"]):
        a, o = common.compute_accuracy_and_overlap(pred, g["exact"], g["weights"])
        assert a == acc[0] and o == ov[0]
        a, o = common.compute_accuracy_and_overlap(pred, g["exact"], number_spins=n)
        assert a == acc[1] and o == ov[1]
    with pytest.raises(ValueError):
        common.compute_accuracy_and_overlap(g["predicted"][0], g["exact"])


def test_sparsify_oracle_matches_reference_golden():
    """oracle/sparsify_oracle.py (numpy/scipy restatement of common.py:634-692) on the
    reference's own output: mask and kept block equal what the reference returned."""
    import oracle

    g = golden("make_ising_kagome16_cluster.npz")
    n = g["ext_spins"].shape[0]
    ext = scipy.sparse.coo_matrix((g["ext_data"], (g["ext_row"], g["ext_col"])), shape=(n, n))
    frozen = np.zeros(n, dtype=bool)
    at = np.searchsorted(g["ext_spins"], g["spins"])
    frozen[at] = True
    keep, block = oracle.sparsify_component(ext, frozen, float(g["sp_reltol"]), int(at[0]))
    assert np.array_equal(g["ext_spins"][keep], g["sp_spins"])
    m = scipy.sparse.coo_matrix(block)
    assert np.array_equal(m.row, g["sp_row"]) and np.array_equal(m.col, g["sp_col"])
    assert m.data.tobytes() == g["sp_data"].tobytes()
    # a cutoff that cuts a frozen spin off trips the reference's assertion
    lonely = scipy.sparse.csr_matrix(np.array([[0, 1.0, 0], [1.0, 0, 1e-9], [0, 1e-9, 0]]))
    with pytest.raises(AssertionError):
        oracle.sparsify_component(lonely, np.array([True, False, True]), 1e-3, 0)


def test_strongest_off_diag_and_binary_search():
    from annealing_sign_problem_amd import common

    rng = np.random.default_rng(3)
    m = scipy.sparse.random(200, 200, density=0.05, random_state=3, format="csr")
    m = (m + scipy.sparse.diags(rng.normal(size=200) * 10)).tocsr()
    got = common.get_strongest_off_diag(m)
    dense = np.abs(m.toarray())
    np.fill_diagonal(dense, 0)
    assert np.array_equal(got, dense.max(axis=1))
    hay = np.array([2, 5, 9, 11], dtype=np.uint64)
    assert common.binary_search(hay, np.array([9, 2], dtype=np.uint64)).tolist() == [2, 0]
    with pytest.raises(AssertionError):
        common.binary_search(hay, np.array([3], dtype=np.uint64))


def test_plan_layout_matches_oracle_layout():
    """Host preprocessing of the product (csrc/sa_plan.cpp) vs the oracle's restatement."""
    from annealing_sign_problem_amd import _lib, synthetic

    lib = _lib.load()
    for n, seed in [(1, 1), (70, 2), (3000, 3)]:
        J, h, _ = synthetic.planted_cluster(n, seed=seed, mean_degree=min(23.0, n / 3))
        indptr = J.indptr.astype(np.int64)
        indices = J.indices.astype(np.int32)
        info = _lib.SaInfo()
        colors = np.zeros(n, np.int32)
        pos = np.zeros(n, np.uint32)
        _lib.check(lib.asp_sa_layout_host(n, _lib.ptr(indptr), _lib.ptr(indices), _lib.ptr(J.data),
                                          _lib.ptr(h), ctypes.byref(info), _lib.ptr(colors),
                                          _lib.ptr(pos)))
        ocolors, order, ncol, nnz, diag = oracle.sa_layout(J)
        assert np.array_equal(colors, ocolors) and info.num_colors == ncol
        assert info.nnz_offdiag == nnz and info.diag_sum == diag
        assert np.all(np.diff(pos[order].astype(np.int64)) > 0)   # same visiting order
        assert len(set(pos.tolist())) == n and pos.max() < info.num_blocks * 64
        assert info.ell_entries % 256 == 0 and info.ell_entries >= nnz
        assert info.beta0_auto > 0 and info.beta1_auto >= info.beta0_auto or nnz == 0
    # an exactly symmetric J takes a short cut (A = 2 offdiag(J) without transposing); the same
    # couplings handed over as an upper triangle, or with ONE mirror element off by an ulp, go
    # through the general merge: same layout, same greedy tree, bit for bit / as the oracle says
    J, h, _ = synthetic.planted_cluster(3000, seed=3)
    J = (J + scipy.sparse.diags(np.random.default_rng(0).normal(size=3000))).tocsr()
    J.sort_indices()
    assert abs(J - J.T).max() == 0
    upper = (2.0 * scipy.sparse.triu(J, 1) + scipy.sparse.diags(J.diagonal())).tocsr()
    upper.sort_indices()
    nudged = J.copy()
    k = int(np.flatnonzero(nudged.indices != np.repeat(np.arange(3000), np.diff(nudged.indptr)))[5])
    nudged.data[k] = np.nextafter(nudged.data[k], np.inf)
    results = []
    for m in (J, upper, nudged):
        info = _lib.SaInfo()
        colors, pos, x = np.zeros(3000, np.int32), np.zeros(3000, np.uint32), np.zeros(47, np.uint64)
        args = (3000, _lib.ptr(m.indptr.astype(np.int64)), _lib.ptr(m.indices.astype(np.int32)),
                _lib.ptr(m.data), _lib.ptr(h))
        _lib.check(lib.asp_sa_layout_host(*args, ctypes.byref(info), _lib.ptr(colors), _lib.ptr(pos)))
        _lib.check(lib.asp_sa_greedy_tree_host(*args, _lib.ptr(x)))
        ocolors, _, ncol, nnz, diag = oracle.sa_layout(m)
        assert np.array_equal(colors, ocolors) and info.num_colors == ncol
        assert info.nnz_offdiag == nnz and info.diag_sum == diag
        assert np.array_equal(x, oracle.greedy_solve(m, h, relax=False)[0])
        results.append((colors, pos, x, info.beta0_auto, info.beta1_auto, info.energy_scale_exp))
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b)
    # the two 64-bit hashes are only the fast negative test.  With the hashes forced to
    # say "symmetric" (a collision), the exact mirror search must still send the upper triangle,
    # the nudged matrix and a matrix whose lower half misses one element through the general merge
    holed = J.tolil()
    i, j = [(i, j) for i, j in zip(*J.nonzero()) if i > j][7]
    holed[i, j] = 0.0
    holed = holed.tocsr()
    holed.eliminate_zeros()
    holed.sort_indices()
    os.environ["ASP_PLAN_HASHES_SAY_SYMMETRIC"] = "1"
    try:
        for m, expected in ((J, results[0]), (upper, results[1]), (nudged, results[2]), (holed, None)):
            info = _lib.SaInfo()
            colors, pos = np.zeros(3000, np.int32), np.zeros(3000, np.uint32)
            _lib.check(lib.asp_sa_layout_host(3000, _lib.ptr(m.indptr.astype(np.int64)),
                                              _lib.ptr(m.indices.astype(np.int32)), _lib.ptr(m.data), _lib.ptr(h),
                                              ctypes.byref(info), _lib.ptr(colors), _lib.ptr(pos)))
            ocolors, _, ncol, nnz, diag = oracle.sa_layout(m)
            assert np.array_equal(colors, ocolors) and info.nnz_offdiag == nnz and info.diag_sum == diag
            if expected is not None:
                assert np.array_equal(colors, expected[0]) and np.array_equal(pos, expected[1])
                assert (info.beta0_auto, info.beta1_auto, info.energy_scale_exp) == expected[3:]
    finally:
        del os.environ["ASP_PLAN_HASHES_SAY_SYMMETRIC"]
    # rejects non-canonical input loudly
    bad_indices = np.array([1, 0], np.int32)
    rc = lib.asp_sa_layout_host(2, _lib.ptr(np.array([0, 2, 2], np.int64)), _lib.ptr(bad_indices),
                                _lib.ptr(np.ones(2)), _lib.ptr(np.zeros(2)), None, None, None)
    assert rc == -3 and b"canonical" in lib.asp_last_error()


def test_operators_small_systems(models):
    from annealing_sign_problem_amd import operators

    ring = operators.Operator(operators.SpinBasis(4, 2), [operators.Term(
        operators.SIGMA_DOT_SIGMA, [(0, 1), (1, 2), (2, 3), (3, 0)])])
    ring.basis.build()
    e0, psi = ring.ground_state()
    assert abs(e0 + 8.0) < 1e-10 and ring.basis.number_states == 6
    kag = operators.Operator.from_config(models["heisenberg_kagome_16"])
    kag.basis.build()
    assert kag.basis.number_states == 12870
    other, coeffs, counts = kag.batched_apply(kag.basis.states[:500])
    assert other.shape == (counts.sum(), 8) and coeffs.dtype == np.complex128
    # 24 bonds: off-diagonal connections = number of antiparallel bonds, plus one diagonal entry
    s = kag.basis.states[:500]
    bonds = models["heisenberg_kagome_16"]["hamiltonian"]["terms"][0]["sites"]
    anti = sum(((s >> np.uint64(a)) ^ (s >> np.uint64(b))) & np.uint64(1) for a, b in bonds)
    assert np.array_equal(counts, anti.astype(np.int64) + 1)
    h = kag.to_sparse()
    assert abs(h - h.T).max() < 1e-14


def test_shard_range_covers_everything():
    from annealing_sign_problem_amd import distributed

    for total in [0, 1, 7, 64, 1000]:
        for world in [1, 2, 3, 8]:
            parts = [distributed.shard_range(total, world, k) for k in range(world)]
            assert sum(c for _, c in parts) == total
            assert all(parts[k][0] + parts[k][1] == parts[k + 1][0] for k in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_greedy_cluster_merging_matches_oracle():
    """Host half of asp_sa_greedy (csrc/greedy.cpp) vs the oracle's restatement, incl. fields,
    isolated spins and a non-symmetric J."""
    from annealing_sign_problem_amd import _lib, synthetic

    lib = _lib.load()
    rng = np.random.default_rng(11)
    cases = []
    for n, seed in [(2, 1), (300, 2), (4000, 3)]:
        J, h, _ = synthetic.planted_cluster(n, seed=seed, mean_degree=min(12.0, n / 2))
        cases.append((J, h))
        cases.append((J, rng.normal(size=n) * 0.01))
    m = scipy.sparse.random(500, 500, density=0.004, random_state=5, format="csr")
    m.data = rng.normal(size=m.data.shape)
    cases.append((m, rng.normal(size=500)))           # non-symmetric, isolated spins, field
    cases.append((scipy.sparse.csr_matrix((70, 70)), rng.normal(size=70)))  # field only
    for J, h in cases:
        J = scipy.sparse.csr_matrix(J)
        J.sum_duplicates()
        J.sort_indices()
        n = J.shape[0]
        x = np.zeros((n + 63) // 64, np.uint64)
        _lib.check(lib.asp_sa_greedy_tree_host(n, _lib.ptr(J.indptr.astype(np.int64)),
                                               _lib.ptr(J.indices.astype(np.int32)),
                                               _lib.ptr(J.data.astype(np.float64)), _lib.ptr(h),
                                               _lib.ptr(x)))
        ox, _ = oracle.greedy_solve(J, h, relax=False)
        assert np.array_equal(x, ox)


def test_greedy_solves_unfrustrated_instance_exactly():
    from annealing_sign_problem_amd import synthetic

    J, h, planted = synthetic.planted_cluster(3000, seed=16, frustrated_fraction=0.0,
                                              diagonal_range=0)
    x, e = oracle.greedy_solve(J, h)
    assert abs(e - planted @ (J @ planted)) <= 1e-12 * abs(e)


def _state_by_state_growth(op, start, required_size, keep_probability):
    """The reference's loop (common.py:481-513), restated."""
    members = {start}

    def children_of(state):
        kept = []
        for x in op.apply(state)[0][:, 0]:
            if x in members:
                continue
            if np.random.rand() <= keep_probability:
                kept.append(int(x))
        return kept

    frontier = children_of(start)
    while len(members) < required_size and len(frontier) > 0:
        upcoming = set()
        for child in frontier:
            members.add(child)
            if len(members) >= required_size:
                break
            upcoming |= set(children_of(child))
        frontier = upcoming
    return sorted(members)


def test_cluster_growth_is_connected_and_sized(models):
    from annealing_sign_problem_amd import operators, sampled_components

    op = operators.Operator.from_config(models["heisenberg_kagome_16"])
    op.basis.build()

    class Foreign:  # not this package's Operator type: growth uses its batched_apply on the host
        basis = op.basis

        def batched_apply(self, x):
            return op.batched_apply(x)

    # the pass-at-once growth consumes the random stream exactly like the reference's
    # state-by-state loop: same clusters AND the same stream position afterwards, for clusters
    # that fill up, for a frontier that dies out first (keep probability 0.02) and for a cluster
    # that wants more states than the basis has
    for seed, index, size, keep in [(5, 1234, 120, 0.5), (6, 17, 50, 0.5), (7, 9000, 1000, 0.5),
                                    (8, 400, 300, 0.02), (9, 77, 2, 0.5), (10, 5000, 1, 0.5),
                                    (11, 3, 20000, 0.9), (12, 12869, 700, 0.3)]:
        start = int(op.basis.states[index])
        np.random.seed(seed)
        got = sampled_components.create_small_cluster_around_point(start, Foreign(), required_size=size,
                                                                   keep_probability=keep)
        after = np.random.rand()
        np.random.seed(seed)
        want = _state_by_state_growth(op, start, size, keep)
        assert got == want and after == np.random.rand(), (seed, index, size, keep)
        assert len(got) <= max(size, 1) or size <= 1
    np.random.seed(5)
    start = int(op.basis.states[1234])
    cluster = sampled_components.create_small_cluster_around_point(start, Foreign(), required_size=120)
    assert cluster == sorted(cluster) and start in cluster and len(set(cluster)) == len(cluster)
    assert 60 <= len(cluster) <= 120
    # connected under the Hamiltonian's off-diagonal action
    members = set(cluster)
    seen, stack = {start}, [start]
    while stack:
        other, _ = op.apply(stack.pop())
        for x in other[:, 0]:
            x = int(x)
            if x in members and x not in seen:
                seen.add(x)
                stack.append(x)
    assert seen == members
    sizes = [sampled_components.random_cluster_size(50, 1000) for _ in range(200)]
    assert min(sizes) >= 50 and max(sizes) <= 1000 and np.median(sizes) < 400  # log-uniform


def test_load_hamiltonian_from_yaml_and_invert_permutation(tmp_path, models):
    from annealing_sign_problem_amd import common, operators

    text = """basis:
  number_spins: 4
  hamming_weight: 2
  symmetries: []
hamiltonian:
  name: "ring"
  terms:
    - matrix: [[1,  0,  0, 0],
               [0, -1,  2, 0],
               [0,  2, -1, 0],
               [0,  0,  0, 1]]
      sites: [[0, 1], [1, 2], [2, 3], [3, 0]]
observables: []
"""
    path = tmp_path / "ring.yaml"
    path.write_text(text)
    op = common.load_hamiltonian(str(path))
    assert isinstance(op, operators.Operator) and op.basis.number_spins == 4
    op.basis.build()
    e0, _ = op.ground_state()
    assert abs(e0 + 8.0) < 1e-10          # 4-site Heisenberg ring in sigma.sigma units
    symmetric = text.replace("symmetries: []", "symmetries: [{permutation: [1, 2, 3, 0], sector: 0}]")
    path.write_text(symmetric)
    sym = common.load_hamiltonian(str(path))  # translations in sector 0: representatives only
    sym.basis.build()
    assert sym.basis.states.tolist() == [0b0011, 0b0101]
    assert abs(sym.ground_state()[0] + 8.0) < 1e-10  # the ring's ground state is translation-invariant
    path.write_text(symmetric.replace("sector: 0", "sector: 1"))
    with pytest.raises(ValueError):  # other sectors need complex characters: not supported
        common.load_hamiltonian(str(path))
    p = np.array([2, 0, 3, 1])
    assert common.invert_permutation(p).tolist() == [1, 3, 0, 2]


def test_hamiltonian_arrays_are_frozen_against_stale_plans():
    """the plan cache is keyed on object identity, so an in-place edit of
    exchange.data would silently anneal the old couplings.  The arrays are read-only instead."""
    from annealing_sign_problem_amd import annealer as sa

    m = scipy.sparse.random(30, 30, density=0.2, random_state=1, format="csr")
    field = np.zeros(30)
    ham = sa.Hamiltonian(m + m.T, field)
    with pytest.raises(ValueError):
        ham.exchange.data[0] = 5.0
    with pytest.raises(ValueError):
        ham.field[3] = 1.0
    field[3] = 1.0  # the caller's own array is a different object (copied on construction)
    assert ham.field[3] == 0.0
    # the usual read-only uses of the reference keep working (common.py:444,654,674)
    assert ham.exchange.tocoo().nnz == ham.exchange.nnz and ham.exchange[:5][:, :5].shape == (5, 5)


def test_bench_roofline_refuses_counters_of_other_kernel_sources(tmp_path, monkeypatch):
    """bench.py multiplied the live rate by constants from a committed JSON, so a
    changed kernel with a forgotten PMC pass still printed the old fraction.  Every case of the JSON
    now names the sources its kernel was built from when it was measured (kernel file, plan builder,
    headers, flags: build.KERNEL_SOURCE_SETS); a case measured on other sources gets no fraction
    but a reason, while the cases of an untouched kernel file stay."""
    import json

    import bench

    profiles = tmp_path / "profiles"
    profiles.mkdir()
    counters = {"cycles_per_valu_inst": 4.35,
                "cases": {"colour_10000": {"valu_insts_per_flip": 1.5, "kernel_source_set": "colour",
                                           "source_set_fingerprint": "a" * 64},
                          "shuffled_10000": {"valu_insts_per_flip": 4.0, "kernel_source_set": "shuffled",
                                             "source_set_fingerprint": "b" * 64}}}
    (profiles / "sweep_counters.json").write_text(json.dumps(counters))
    (profiles / "traffic.json").write_text(json.dumps({"cases": {
        "colour_10000": {"hbm_bytes_per_flip": 4.5, "source_set_fingerprint": "0" * 64}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got, traffic, reason = bench.profiled_counters({"colour": "a" * 64, "shuffled": "b" * 64})
    assert got == counters and reason is None
    assert traffic == {"cases": {}}  # measured on yet other sources: dropped on its own
    frac = bench.issue_fraction("colour_10000", got, 200e9)
    assert abs(frac - 1.5 * 4.35 * 200 / (1024 * 2.4)) < 1e-12
    assert bench.issue_fraction("colour_30000", got, 200e9) is None  # a case that was not profiled
    # sa_shuffled.hip changed: the shuffled case goes, the colour case stays
    got, _, reason = bench.profiled_counters({"colour": "a" * 64, "shuffled": "c" * 64})
    assert sorted(got["cases"]) == ["colour_10000"] and "stale" in reason and "shuffled_10000" in reason
    assert bench.issue_fraction("shuffled_10000", got, 200e9) is None
    stale, traffic, reason = bench.profiled_counters({"colour": "c" * 64, "shuffled": "c" * 64})
    assert stale is None and traffic is None and "stale" in reason and "colour_10000" in reason
    assert bench.issue_fraction("colour_10000", stale, 200e9) is None
    assert bench.profiled_counters(None)[2] is not None  # no build stamp: no fraction either
    (profiles / "sweep_counters.json").write_text(json.dumps({"cases": {"colour_10000": {}}}))  # no fingerprint
    assert bench.profiled_counters({"colour": "a" * 64, "shuffled": "b" * 64})[0] is None


def test_build_records_the_fingerprints_of_the_kernel_source_sets():
    """The stamp beside the library names what its sweep kernels were built from; it follows the
    sources (a change to another kernel file leaves both sets alone)."""
    from annealing_sign_problem_amd import build

    now = build.source_set_fingerprints()
    assert sorted(now) == ["colour", "shuffled"] and now["colour"] != now["shuffled"]
    built = build.built_source_set_fingerprints()
    if build.built_fingerprint() == build.fingerprint():  # (the library on disk is of these sources)
        assert built == now
    # the committed counters: every case names its set, and a fingerprint if it has one is 64 hex digits
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                           "sweep_counters.json")) as f:
        for case, entry in json.load(f)["cases"].items():
            assert entry["kernel_source_set"] == ("shuffled" if "shuffled" in case else "colour")
            assert entry["source_set_fingerprint"] is None or len(entry["source_set_fingerprint"]) == 64

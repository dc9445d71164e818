"""Whole symmetry sectors on the GPU (csrc/sector_basis.hip, annealing_sign_problem_amd/sector_ed.py):
enumeration of the representatives, the sector's Hamiltonian as a resident ELL matrix and the
Lanczos ground state, against the numpy / scipy host route (operators.SpinBasis.build,
Operator.to_sparse, scipy eigsh) where that route is feasible, and through size-independent
properties (symmetry of the matrix, eigenvalue residual) on the 32-site pyrochlore sector of
heisenberg_pyrochlore_2x2x2.yaml:1-17 where it is not."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ring(n, inversion, weight="half", reflection=True):
    """Heisenberg ring of n sites with translations (+ reflection) and spin inversion."""
    from annealing_sign_problem_amd import operators, symmetry

    generators = [[(i + 1) % n for i in range(n)]]
    if reflection:
        generators.append([(n - i) % n for i in range(n)])
    group = symmetry.SymmetryGroup(n, generators, inversion)
    basis = operators.SpinBasis(n, n // 2 if weight == "half" else weight, group)
    term = operators.Term(operators.SIGMA_DOT_SIGMA, [(i, (i + 1) % n) for i in range(n)])
    return operators.Operator(basis, [term])


def _cases(models):
    from annealing_sign_problem_amd import operators

    yield "ring16 even", _ring(16, 1)
    yield "ring16 odd", _ring(16, -1)          # orbits of zero norm are dropped
    yield "ring20 no inversion", _ring(20, None)
    yield "ring12 any magnetisation", _ring(12, None, weight=None)
    yield "kagome_18", operators.Operator.from_config(models["heisenberg_kagome_18"])
    yield "kagome_16 (no symmetry)", operators.Operator.from_config(models["heisenberg_kagome_16"])
    yield "j1j2 (asymmetric J2 matrix)", operators.Operator.from_config(models["j1j2_square_4x4"])


def test_enumeration_matrix_and_ground_state_equal_the_host_route(models):
    import scipy.sparse.linalg
    import torch

    from annealing_sign_problem_amd import sector_ed

    for name, op in _cases(models):
        reps, norms = sector_ed.enumerate_sector(op)
        op.basis.build()
        want = op.basis.states
        got = reps.cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want), name
        if op.basis.group is not None:
            _, _, want_norm = op.basis.group.state_info(want)
            assert norms.cpu().numpy().tobytes() == want_norm.tobytes(), name
        else:
            assert np.all(norms.cpu().numpy() == 1.0)
        matrix = sector_ed.SectorMatrix(op, reps, norms)
        h = op.to_sparse()
        assert abs(h.imag).max() == 0
        h = h.real.tocsr()
        rng = np.random.default_rng(3)
        x = rng.standard_normal(matrix.n)
        y = matrix.matvec(torch.from_numpy(x).cuda()).cpu().numpy()
        # rows of the ELL matrix are columns of `to_sparse` (entry = <target| H |source>); the
        # sector's matrix is symmetric, which this comparison also asserts
        assert np.allclose(y, h @ x, rtol=0, atol=1e-11 * np.abs(h).sum(axis=1).max()), name
        assert np.allclose(y, h.T @ x, rtol=0, atol=1e-11 * np.abs(h).sum(axis=1).max()), name
        energy, vector, info = sector_ed.lanczos_ground_state(matrix, tol=1e-11, max_iterations=300)
        want_energy = scipy.sparse.linalg.eigsh(h, k=1, which="SA", tol=1e-13)[0][0]
        assert abs(energy - want_energy) < 1e-9 * abs(want_energy), (name, energy, want_energy)
        assert info["residual"] < 1e-7, (name, info)


def test_ground_state_of_the_pyrochlore_sector_and_the_reference_file_layout(models, tmp_path):
    """heisenberg_pyrochlore_2x2x2.yaml:1-17: 32 sites, C(32,16) = 6.0e8 states — beyond the host
    route; checked through the eigenvalue residual, spot checks of the representatives and the
    round trip through the SpinED file layout the reference's drivers read (common.py:772-780)."""
    from annealing_sign_problem_amd import common, operators, sector_ed

    op = operators.Operator.from_config(models["heisenberg_pyrochlore_2x2x2"])
    energy, psi, representatives, info = sector_ed.ground_state(op, tol=1e-10)
    group = op.basis.group
    k = representatives.shape[0]
    assert 601080390 / group.order <= k <= 601080390 / group.order * 1.2   # total / |G| + short orbits
    assert np.all(np.diff(representatives.astype(np.int64)) > 0)
    # spot check: the listed states are representatives of norm > 0 and of the right magnetisation
    pick = np.random.default_rng(1).choice(k, size=2000, replace=False)
    rep, _, norm = group.state_info(representatives[pick])
    assert np.array_equal(rep, representatives[pick]) and np.all(norm > 0)
    assert all(bin(int(s)).count("1") == 16 for s in representatives[pick[:200]])
    assert info["residual"] < 1e-7 and abs(np.linalg.norm(psi) - 1.0) < 1e-12
    # 96 bonds of sigma.sigma: the spectrum lies in [-3 * 96, 96]; an antiferromagnet is negative
    assert -288.0 < energy < -96.0 * 0.5
    filename = str(tmp_path / "pyrochlore.h5")
    sector_ed.write_spined_hdf5(filename, energy, psi, representatives)
    back, e_back, reps_back = common.load_ground_state(filename)
    assert e_back == energy and np.array_equal(reps_back, representatives)
    assert back.tobytes() == psi.tobytes()


def test_kagome_36_sector_dimension_and_ground_state_energy_match_the_literature(models):
    """The one EXTERNAL pin of the symmetry-adapted construction (representatives, characters,
    norms, matrix elements; lattice_symmetries itself is absent): the fully symmetric, inversion-
    even Sz = 0 sector of the 36-site kagome cluster of heisenberg_kagome_36.yaml:7-29 has
    31 527 894 states and the ground-state energy per site is -0.43837653 J (Waldtmann et al.,
    Eur. Phys. J. B 2, 501 (1998); Laeuchli, Sudan, Moessner, Phys. Rev. B 100, 155142 (2019),
    table of the N = 36 cluster) — in the YAML's units, sum of sigma.sigma = 4 S.S over 72 bonds."""
    from annealing_sign_problem_amd import operators, sector_ed

    op = operators.Operator.from_config(models["heisenberg_kagome_36"])
    energy, psi, representatives, info = sector_ed.ground_state(op, tol=1e-9)
    assert representatives.shape[0] == 31527894
    assert abs(energy / 36.0 / 4.0 - (-0.43837653)) < 1e-8
    assert info["residual"] < 1e-6 and abs(np.linalg.norm(psi) - 1.0) < 1e-12


def test_device_index_of_a_large_basis_equals_searchsorted():
    """csrc/key_table.hip behind SpinBasis.batched_index (common.py:813-818's index lookup)."""
    import ctypes

    from annealing_sign_problem_amd import _lib, operators

    rng = np.random.default_rng(5)
    keys = np.unique(rng.integers(0, 1 << 40, size=300_000, dtype=np.uint64))
    basis = operators.SpinBasis(40)
    basis.build(keys)
    basis.DEVICE_INDEX_LIMIT = 1000            # (instance attribute: this basis only)
    queries = keys[rng.integers(0, keys.shape[0], size=50_000)]
    assert np.array_equal(basis.batched_index(queries), np.searchsorted(keys, queries).astype(np.uint64))
    assert basis.index(int(keys[-1])) == keys.shape[0] - 1 and basis.index(int(keys[0])) == 0
    assert basis.batched_index(np.zeros(0, np.uint64)).shape == (0,)
    absent = np.setdiff1d(rng.integers(0, 1 << 40, size=100, dtype=np.uint64), keys)
    with pytest.raises(ValueError):
        basis.batched_index(np.concatenate([queries[:10], absent[:1]]))
    # the C entry point itself: -1 for absent keys, ascending keys required
    lib = _lib.load()
    out = np.zeros(3, np.int64)
    probe = np.array([keys[7], absent[0], keys[123]], dtype=np.uint64)
    _lib.check(lib.asp_table_index(basis._table, 3, _lib.ptr(probe), _lib.ptr(out)))
    assert out.tolist() == [7, -1, 123]
    handle = ctypes.c_void_p()
    unsorted = np.array([3, 2, 5], dtype=np.uint64)
    assert lib.asp_table_create(3, _lib.ptr(unsorted), ctypes.byref(handle)) != 0
    basis.release_table()


def test_matrix_free_product_of_plain_bases_equals_the_host_route(models):
    """csrc/plain_basis.hip (the route sk_32_1.yaml's 6.0e8 states take): states, y = Hx and the
    three-vector Lanczos against numpy / scipy on the 16-site models and an 18-site chain whose
    sites straddle the two index words unevenly."""
    import scipy.sparse.linalg
    import torch

    from annealing_sign_problem_amd import operators, sector_ed

    chain = operators.Operator(operators.SpinBasis(18, 7), [
        operators.Term(operators.SIGMA_DOT_SIGMA, [(i, (i + 1) % 18) for i in range(18)]),
        operators.Term(0.37 * operators.SIGMA_DOT_SIGMA, [(i, (i + 5) % 18) for i in range(18)])])
    cases = [(name, operators.Operator.from_config(models[name]))
             for name in ("heisenberg_kagome_16", "j1j2_square_4x4", "sk_16_1")] + [("chain18", chain)]
    for name, op in cases:
        op.basis.build()
        matrix = sector_ed.PlainBasisMatrix(op)
        assert matrix.n == op.basis.number_states, name
        assert np.array_equal(matrix.states().cpu().numpy().view(np.uint64), op.basis.states), name
        h = op.to_sparse().real.tocsr()
        x = np.random.default_rng(2).standard_normal(matrix.n)
        y = matrix.matvec(torch.from_numpy(x).cuda()).cpu().numpy()
        assert np.allclose(y, h @ x, rtol=0, atol=1e-11 * abs(h).sum(axis=1).max()), name
        energy, vector, info = sector_ed.lanczos_two_pass(matrix, tol=1e-10)
        want = scipy.sparse.linalg.eigsh(h, k=1, which="SA", tol=1e-13)[0][0]
        assert abs(energy - want) < 1e-8 * abs(want), (name, energy, want, info)
        assert info["residual"] < 1e-6, (name, info)
        matrix.release()
    # lattice symmetries and unconstrained magnetisation are other routes
    with pytest.raises(ValueError):
        sector_ed.PlainBasisMatrix(operators.Operator.from_config(models["heisenberg_kagome_18"]))

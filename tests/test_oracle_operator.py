"""Operator oracle (oracle/operator_oracle.c) pinned against the vectors the reference's own
make_ising_model / make_hamiltonian_extension produced (tests/golden/generate_golden.py) and
against the numpy stand-in for lattice_symmetries + scipy, on general 4x4 bond matrices."""
import numpy as np
import pytest
import scipy.sparse

import oracle
from conftest import golden
from helpers import random_operator, reference_route_ising


@pytest.mark.parametrize("case,name", [("kagome16_cluster", "heisenberg_kagome_16"),
                                       ("sk16_cluster", "sk_16_1")])
def test_operator_oracle_matches_reference_golden(case, name, models):
    from annealing_sign_problem_amd import operators

    g = golden("make_ising_%s.npz" % case)
    op = operators.Operator.from_config(models[name])
    table = op.bond_table()
    other, coeffs, counts = oracle.operator_apply(table, g["spins"])
    assert np.array_equal(other, g["other_spins"])
    assert coeffs.tobytes() == g["other_coeffs"].tobytes()
    assert np.array_equal(counts, g["other_counts"])
    psi = np.ascontiguousarray(np.exp(g["log_psi"]).real)
    psi /= np.linalg.norm(psi)
    row, col, val = oracle.operator_ising(table, g["spins"], psi)
    assert np.array_equal(row, g["row"]) and np.array_equal(col, g["col"])
    assert val.tobytes() == g["data"].tobytes()
    if "ext_spins" in g.files:
        assert np.array_equal(oracle.operator_extend(table, g["spins"]), g["ext_spins"])


@pytest.mark.parametrize("seed,kind", [(1, "exchange"), (2, "general"), (3, "one_way"),
                                       (4, "single_flip")])
def test_operator_oracle_matches_numpy_and_scipy(seed, kind):
    """General matrices: apply equals the numpy operator entry for entry, and the oracle's
    J equals csr(M); 0.5 * (M + M.T); sort_indices(); tocoo() done by scipy itself."""
    op, keys, psi = random_operator(seed, kind, number_spins=14, num_bonds=20, num_keys=700)
    table = op.bond_table()
    other, coeffs, counts = oracle.operator_apply(table, keys)
    ref_other, ref_coeffs, ref_counts = op.batched_apply(keys)
    assert np.array_equal(other, ref_other[:, 0])
    assert coeffs.tobytes() == np.ascontiguousarray(ref_coeffs.real).tobytes()
    assert np.array_equal(counts, ref_counts)
    row, col, val = oracle.operator_ising(table, keys, psi)
    m = reference_route_ising(op, keys, psi)
    assert np.array_equal(row, m.row) and np.array_equal(col, m.col)
    assert val.tobytes() == m.data.tobytes()
    ext = oracle.operator_extend(table, keys)
    assert np.array_equal(ext, np.unique(ref_other[:, 0]))


def test_operator_oracle_degenerate():
    from annealing_sign_problem_amd import operators

    op = operators.Operator(operators.SpinBasis(5), [])
    table = op.bond_table()
    keys = np.array([3, 9], dtype=np.uint64)
    other, coeffs, counts = oracle.operator_apply(table, keys)
    assert np.array_equal(other, keys) and np.all(coeffs == 0) and np.array_equal(counts, [1, 1])
    row, col, val = oracle.operator_ising(table, keys, np.array([0.6, 0.8]))
    assert row.size == 0 and col.size == 0 and val.size == 0
    none = np.zeros(0, np.uint64)
    assert oracle.operator_apply(table, none)[0].size == 0
    assert oracle.operator_extend(table, none).size == 0

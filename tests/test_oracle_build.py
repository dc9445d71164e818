"""CPU: pin the oracle's coupling-build restatement against the REFERENCE —
oracle/_ref (cbits/build_matrix.c compiled in place, present in the build
container) and the committed golden vectors that binary produced."""
import numpy as np
import pytest

import oracle
from conftest import golden
from helpers import random_build_case

INPUTS = ["spins", "counts", "psi", "other_spins", "other_coeffs", "other_counts", "other_psi"]


def _bits(a):
    return np.ascontiguousarray(a).tobytes()


@pytest.mark.parametrize("case", ["hand", "word0", "multiword", "allmiss"])
def test_oracle_matches_golden(case):
    g = golden("build_matrix_%s.npz" % case)
    nnz, row, col, elements, field = oracle.build_matrix(*[g[k] for k in INPUTS])
    assert nnz == int(g["nnz"])
    assert np.array_equal(row, g["row"]) and np.array_equal(col, g["col"])
    assert _bits(elements) == _bits(g["elements"]) and _bits(field) == _bits(g["field"])


def test_hand_example_values():
    """SURVEY appendix A.1 (checked by hand against cbits/build_matrix.c:22-65)."""
    g = golden("build_matrix_hand.npz")
    assert int(g["nnz"]) == 4
    assert g["row"].tolist() == [0, 1, 2, 3] and g["col"].tolist() == [1, 0, 2, 0]
    assert g["elements"].tolist() == [0.5, 1.0, -0.25, 0.5]
    assert g["field"].tolist() == [0.1, 0.0, 0.0, -0.2]


def test_extract_signs_golden_and_edge_values():
    g = golden("extract_signs.npz")
    assert np.array_equal(oracle.extract_signs(g["psi"]), g["signs"])
    psi = np.array([1.0, -1.0, 0.0, -0.0, np.nan, np.inf, -np.inf, 5e-324])
    assert oracle.extract_signs(psi)[0] == 0b10100001  # only strictly positive values


@pytest.mark.skipif(oracle.ref_lib() is None, reason="reference checkout/_ref build not present")
@pytest.mark.parametrize("seed", range(12))
def test_oracle_matches_reference_binary_random(seed):
    rng = np.random.default_rng(1000 + seed)
    case = random_build_case(rng, int(rng.integers(1, 600)), float(rng.uniform(0.5, 20)),
                             bool(seed % 3 == 0), float(rng.uniform(0, 1)), bool(seed % 2))
    args = [case[k] for k in INPUTS]
    a = oracle.build_matrix(*args)
    b = oracle.ref_build_matrix(*args)
    assert a[0] == b[0]
    for x, y in zip(a[1:], b[1:]):
        assert _bits(x) == _bits(y)
    assert np.array_equal(oracle.extract_signs(case["psi"]), oracle.ref_extract_signs(case["psi"]))


def test_key_order_is_word0_major():
    """ls_bits512_cmp orders by words[0] first (cbits/build_matrix.c:11-18), which is
    NOT little-endian 512-bit numeric order."""
    table = np.zeros((3, 8), np.uint64)
    table[0] = [1, 9, 0, 0, 0, 0, 0, 0]
    table[1] = [2, 0, 0, 0, 0, 0, 0, 0]
    table[2] = [2, 0, 5, 0, 0, 0, 0, 0]
    needles = table[[2, 0, 1]].copy()
    nnz, row, col, _, _ = oracle.build_matrix(table, np.ones(3, np.int64), np.ones(3), needles,
                                              np.ones(3), np.array([3, 0, 0]), np.ones(3))
    assert nnz == 3 and col.tolist() == [2, 0, 1]

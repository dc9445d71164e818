"""HIP Hamiltonian action, fused coupling build and extension (csrc/operator_apply.hip) through
the C ABI, bit for bit against the oracle and the reference-generated golden vectors."""
import numpy as np
import pytest
import scipy.sparse

import oracle
from conftest import golden
from helpers import random_operator, reference_route_ising

pytestmark = pytest.mark.gpu


def _norm2(x):
    """The norm make_ising_model divides by (common.norm2: np.linalg.norm up to 10 000 elements,
    numpy's pairwise sum beyond, where BLAS would split the sum over its threads)."""
    from annealing_sign_problem_amd import common

    return common.norm2(x)


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("case,name", [("kagome16_cluster", "heisenberg_kagome_16"),
                                       ("sk16_cluster", "sk_16_1")])
def test_device_operator_matches_reference_golden(case, name, models):
    from annealing_sign_problem_amd import common, operators

    g = golden("make_ising_%s.npz" % case)
    op = operators.Operator.from_config(models[name])
    dev = op.device()
    assert dev.unique_targets
    other, coeffs, counts = dev.apply(g["spins"])
    _same(other, g["other_spins"])
    _same(coeffs, g["other_coeffs"])
    _same(counts, g["other_counts"])
    # the package's _batched_apply takes the same route
    flat = common._batched_apply(op, g["spins"])
    _same(flat[0], g["other_spins"])
    psi = np.ascontiguousarray(np.exp(g["log_psi"]).real)
    psi /= _norm2(psi)
    row, col, val = dev.ising(g["spins"], psi)
    _same(row, g["row"].astype(np.int32))
    _same(col, g["col"].astype(np.int32))
    _same(val, g["data"])
    # make_ising_model end to end on the fused route
    model = common.make_ising_model(g["spins"], op, log_psi=g["log_psi"])
    m = scipy.sparse.coo_matrix(model.ising_hamiltonian.exchange)
    _same(m.row.astype(np.int32), g["row"].astype(np.int32))
    _same(m.col.astype(np.int32), g["col"].astype(np.int32))
    _same(m.data, g["data"])
    _same(model.initial_signs, g["x0"])
    if "ext_spins" in g.files:
        _same(dev.extend(g["spins"]), g["ext_spins"])
        log_fn = common.ground_state_to_log_coeff_fn(g["ground_state"], _Basis(g["basis_states"]))
        ext = common.make_hamiltonian_extension(model, log_fn)
        e = scipy.sparse.coo_matrix(ext.ising_hamiltonian.exchange)
        _same(ext.spins, g["ext_spins"])
        _same(e.row.astype(np.int32), g["ext_row"].astype(np.int32))
        _same(e.col.astype(np.int32), g["ext_col"].astype(np.int32))
        _same(e.data, g["ext_data"])


class _Basis:
    def __init__(self, states):
        self.states = states

    def batched_index(self, spins):
        return np.searchsorted(self.states, np.asarray(spins, dtype=np.uint64))


@pytest.mark.parametrize("seed,kind,unique", [(1, "exchange", True), (2, "general", True),
                                              (3, "one_way", True), (4, "single_flip", False),
                                              (5, "double_reach", False)])
def test_device_operator_general_matrices(seed, kind, unique):
    from annealing_sign_problem_amd import _lib, common

    op, keys, psi = random_operator(seed, kind, number_spins=14, num_bonds=20, num_keys=900)
    log_psi = np.log(psi.astype(np.complex128))
    psi = np.ascontiguousarray(np.exp(log_psi).real)  # what make_ising_model will see
    psi /= _norm2(psi)
    dev = op.device()
    assert dev.unique_targets == unique
    table = op.bond_table()
    other, coeffs, counts = dev.apply(keys)
    o_other, o_coeffs, o_counts = oracle.operator_apply(table, keys)
    _same(other, o_other)
    _same(coeffs, o_coeffs)
    _same(counts, o_counts)
    _same(dev.extend(keys), oracle.operator_extend(table, keys))
    o_row, o_col, o_val = oracle.operator_ising(table, keys, psi)
    if unique or kind == "double_reach":
        # distinct targets: the pair-fused build; rows reaching a state twice with mirrored
        # elements: the build that keeps scipy's duplicate arithmetic (k_merge_rows / k_sym_rows)
        row, col, val = dev.ising(keys, psi)
        _same(row, o_row)
        _same(col, o_col)
        _same(val, o_val)
    else:
        # duplicates AND one-directional elements: refused, make_ising_model takes the host route
        with pytest.raises(_lib.AspError):
            dev.ising(keys, psi)
    # make_ising_model picks the route itself; either way it is the oracle's J
    model = common.make_ising_model(keys, op, log_psi=log_psi)
    m = scipy.sparse.coo_matrix(model.ising_hamiltonian.exchange)
    assert np.array_equal(m.row, o_row) and np.array_equal(m.col, o_col)
    _same(m.data, o_val)


def test_fused_route_equals_reference_route_on_kagome36_cluster():
    """A 20 000-state cluster of the 36-site kagome model (72 bonds, two bond chunks per
    wavefront): fused J == numpy/scipy restatement of the reference route == generic route
    of this package (foreign operator object -> batched_apply + ising_elements + scipy)."""
    from annealing_sign_problem_amd import common, operators, synthetic

    op = operators.Operator.from_config(synthetic.kagome_lattice())
    keys = synthetic.grow_cluster(op, int("01" * 18, 2), 20000, seed=3)
    log_psi = synthetic.hashed_log_amplitudes(keys)
    model = common.make_ising_model(keys, op, log_psi=log_psi)
    psi = np.ascontiguousarray(np.exp(log_psi).real)
    psi /= _norm2(psi)
    ref = reference_route_ising(op, keys, psi)
    m = scipy.sparse.coo_matrix(model.ising_hamiltonian.exchange)
    assert np.array_equal(m.row, ref.row) and np.array_equal(m.col, ref.col)
    _same(m.data, ref.data)

    class Foreign:  # hides the type, so common.py takes the reference's route
        basis = op.basis

        def batched_apply(self, x):
            return op.batched_apply(x)

    generic = common.make_ising_model(keys, Foreign(), log_psi=log_psi)
    gm = scipy.sparse.coo_matrix(generic.ising_hamiltonian.exchange)
    assert np.array_equal(m.row, gm.row) and np.array_equal(m.col, gm.col)
    _same(m.data, gm.data)
    _same(model.initial_signs, generic.initial_signs)
    # extension: 20 000 -> every connected state, equals numpy's unique
    ext = op.device().extend(keys)
    other, _, _ = op.batched_apply(keys)
    _same(ext, np.unique(other[:, 0]))


def test_device_operator_dense_sk32_rows():
    """sk_32-like: 496 bonds (8 bond chunks), rows with ~250 couplings."""
    from annealing_sign_problem_amd import operators

    rng = np.random.default_rng(12)
    terms = []
    for a in range(32):
        for b in range(a + 1, 32):
            terms.append(operators.Term(rng.normal() * operators.SIGMA_DOT_SIGMA, [(a, b)]))
    op = operators.Operator(operators.SpinBasis(32, 16), terms)
    start = np.uint64(int("01" * 16, 2))
    first, _, _ = op.batched_apply(np.array([start], dtype=np.uint64))
    second, _, _ = op.batched_apply(first[:80, 0])
    keys = np.unique(np.concatenate([first[:, 0], second[:, 0]]))[:3000]
    psi = rng.normal(size=keys.size)
    psi /= _norm2(psi)
    dev = op.device()
    table = op.bond_table()
    row, col, val = dev.ising(keys, psi)
    o_row, o_col, o_val = oracle.operator_ising(table, keys, psi)
    _same(row, o_row)
    _same(col, o_col)
    _same(val, o_val)
    other, coeffs, counts = dev.apply(keys[:500])
    o = oracle.operator_apply(table, keys[:500])
    _same(other, o[0])
    _same(coeffs, o[1])
    _same(counts, o[2])


def test_device_operator_degenerate_and_errors():
    from annealing_sign_problem_amd import _lib, operators

    op = operators.Operator(operators.SpinBasis(5), [])
    dev = op.device()
    keys = np.array([3, 9], dtype=np.uint64)
    other, coeffs, counts = dev.apply(keys)
    _same(other, keys)
    assert np.all(coeffs == 0) and np.array_equal(counts, [1, 1])
    row, col, val = dev.ising(keys, np.array([0.6, 0.8]))
    assert row.size == 0 and col.size == 0 and val.size == 0
    none = np.zeros(0, np.uint64)
    assert dev.apply(none)[0].size == 0 and dev.extend(none).size == 0
    assert dev.ising(none, np.zeros(0))[0].size == 0
    _same(dev.extend(keys), keys)
    # diagonal-only operator keeps its diagonal couplings
    diag = operators.Operator(operators.SpinBasis(4), [operators.Term(
        np.diag([1.0, -1.0, -1.0, 1.0]), [(0, 1), (2, 3)])])
    row, col, val = diag.device().ising(np.array([1, 2, 7], dtype=np.uint64),
                                        np.array([0.6, 0.0, 0.8]))
    o = oracle.operator_ising(diag.bond_table(), np.array([1, 2, 7], dtype=np.uint64),
                              np.array([0.6, 0.0, 0.8]))
    _same(row, o[0])
    _same(col, o[1])
    _same(val, o[2])
    # unsorted keys are rejected, not silently mis-searched
    two = operators.Operator(operators.SpinBasis(4), [operators.Term(
        operators.SIGMA_DOT_SIGMA, [(0, 1)])])
    with pytest.raises(_lib.AspError, match="sorted"):
        two.device().ising(np.array([5, 3], dtype=np.uint64), np.array([0.6, 0.8]))
    lib = _lib.load()
    import ctypes
    handle = ctypes.c_void_p()
    a = np.array([0], np.uint8)
    rc = lib.asp_operator_create(4, 1, _lib.ptr(a), _lib.ptr(a), _lib.ptr(np.zeros(16)),
                                 ctypes.byref(handle))
    assert rc == -3 and "invalid bond" in _lib.last_error()
    rc = lib.asp_operator_create(65, 0, None, None, None, ctypes.byref(handle))
    assert rc == -3


def test_raw_ctypes_binding_as_documented(models):
    """The reference-side stub of INTEGRATION.md §2b, executed literally: plain ctypes on the
    shared object, no wrapper from this package."""
    import ctypes

    from annealing_sign_problem_amd import build

    lib = ctypes.CDLL(build.LIB_PATH)
    lib.asp_last_error.restype = ctypes.c_char_p
    config = models["heisenberg_kagome_16"]
    a, b, m = [], [], []
    for term in config["hamiltonian"]["terms"]:
        for (i, j) in term["sites"]:
            a.append(i)
            b.append(j)
            m.append(np.asarray(term["matrix"], dtype=np.float64).reshape(16))
    a, b, m = np.asarray(a, np.uint8), np.asarray(b, np.uint8), np.ascontiguousarray(m)
    op = ctypes.c_void_p()
    rc = lib.asp_operator_create(ctypes.c_uint32(config["basis"]["number_spins"]),
                                 ctypes.c_uint32(len(a)), a.ctypes, b.ctypes, m.ctypes,
                                 ctypes.byref(op))
    assert rc == 0, lib.asp_last_error()
    g = golden("make_ising_kagome16_cluster.npz")
    spins = np.ascontiguousarray(g["spins"], dtype=np.uint64)
    psi = np.ascontiguousarray(np.exp(g["log_psi"]).real)
    psi /= _norm2(psi)
    capacity = g["data"].shape[0]
    row = np.zeros(capacity, np.int32)
    col = np.zeros(capacity, np.int32)
    val = np.zeros(capacity, np.float64)
    nnz = ctypes.c_uint64(0)
    rc = lib.asp_operator_ising(op, ctypes.c_uint64(spins.shape[0]), spins.ctypes, psi.ctypes,
                                ctypes.c_uint64(capacity), row.ctypes, col.ctypes, val.ctypes,
                                ctypes.byref(nnz))
    assert rc == 0, lib.asp_last_error()
    assert nnz.value == capacity
    _same(row, g["row"].astype(np.int32))
    _same(col, g["col"].astype(np.int32))
    _same(val, g["data"])
    out = np.zeros(g["ext_spins"].shape[0], np.uint64)
    count = ctypes.c_uint64(0)
    rc = lib.asp_operator_extend(op, ctypes.c_uint64(spins.shape[0]), spins.ctypes,
                                 ctypes.c_uint64(out.shape[0]), out.ctypes, ctypes.byref(count))
    assert rc == 0 and count.value == out.shape[0]
    _same(out, g["ext_spins"])
    lib.asp_operator_destroy.argtypes = [ctypes.c_void_p]
    lib.asp_operator_destroy(op)


def test_device_operator_sixty_four_sites():
    """The largest basis the reference allows (common.py:86): site 63 is the top key bit, so
    flips, hashing, the radix sort of the extension and the sorted-key check all see keys above
    2^63."""
    from annealing_sign_problem_amd import operators

    rng = np.random.default_rng(64)
    pairs = set()
    while len(pairs) < 90:
        a, b = rng.choice(64, size=2, replace=False)
        pairs.add((int(min(a, b)), int(max(a, b))))
    pairs.update({(62, 63), (0, 63), (31, 63)})
    terms = [operators.Term(rng.normal() * operators.SIGMA_DOT_SIGMA, [p]) for p in sorted(pairs)]
    op = operators.Operator(operators.SpinBasis(64, 32), terms)
    start = np.uint64(int("10" * 32, 2))          # bit 63 set
    first, _, _ = op.batched_apply(np.array([start], dtype=np.uint64))
    second, _, _ = op.batched_apply(first[:, 0])
    keys = np.unique(np.concatenate([first[:, 0], second[:, 0]]))[:4000]
    assert keys.max() >= np.uint64(1) << np.uint64(63) and keys.min() < np.uint64(1) << np.uint64(63)
    psi = rng.normal(size=keys.size)
    psi /= _norm2(psi)
    dev, table = op.device(), op.bond_table()
    row, col, val = dev.ising(keys, psi)
    o_row, o_col, o_val = oracle.operator_ising(table, keys, psi)
    _same(row, o_row)
    _same(col, o_col)
    _same(val, o_val)
    other, coeffs, counts = dev.apply(keys[:300])
    o = oracle.operator_apply(table, keys[:300])
    _same(other, o[0])
    _same(coeffs, o[1])
    _same(counts, o[2])
    _same(dev.extend(keys[:500]), oracle.operator_extend(table, keys[:500]))

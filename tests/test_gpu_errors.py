"""GPU: error behaviour at the C ABI — violated preconditions are reported, never silently
computed on, and the reference-signature symbols leave their outputs untouched."""
import ctypes

import numpy as np
import pytest
import scipy.sparse

pytestmark = pytest.mark.gpu


def test_build_matrix_rejects_negative_counts_and_keeps_outputs():
    from annealing_sign_problem_amd import _build_matrix as bm, _lib

    spins = bm.as_bits512(np.arange(4, dtype=np.uint64))
    other = bm.as_bits512(np.arange(2, dtype=np.uint64))
    oc = np.array([1, -1, 2, 0], dtype=np.int64)
    row = np.full(4, 9, np.uint32)
    field = np.full(4, 7.0)
    with pytest.raises(_lib.AspError) as err:
        bm.lib.build_matrix(4, spins, np.ones(4, np.int64), np.ones(4), other, np.ones(2), oc,
                            np.ones(2), row, np.zeros(4, np.uint32), np.zeros(4), field)
    assert err.value.code == -3
    assert np.all(row == 9) and np.all(field == 7.0)
    # the library recovers: a valid call right after works
    r, c, e, f = bm.build_matrix(np.arange(4, dtype=np.uint64), np.ones(4, np.int64), np.ones(4),
                                 np.array([2, 9], np.uint64), np.ones(2), np.array([1, 0, 1, 0]),
                                 np.array([0.5, -0.25]))
    assert r.tolist() == [0] and c.tolist() == [2] and f.tolist() == [0.0, 0.0, -0.25, 0.0]


def test_ising_elements_rejects_unsorted_keys_and_bad_lengths():
    from annealing_sign_problem_amd import _lib, common

    with pytest.raises(_lib.AspError) as err:
        common.ising_elements(np.array([3, 1, 2], np.uint64), np.ones(3), np.array([1], np.uint64),
                              np.ones(1), np.array([1, 0, 0]))
    assert err.value.code == -3 and "sorted" in str(err.value)
    with pytest.raises(_lib.AspError):
        common.ising_elements(np.array([1, 2, 3], np.uint64), np.ones(3), np.array([1], np.uint64),
                              np.ones(1), np.array([1, 1, 0]))  # sum(counts) != len


def test_plan_rejects_bad_matrices_and_oversized_problems():
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    with pytest.raises(ValueError):
        sa.Hamiltonian(scipy.sparse.csr_matrix((3, 4)), np.zeros(3))
    ham = sa.Hamiltonian(scipy.sparse.identity(3, format="csr"), np.array([0.0, np.inf, 0.0]))
    with pytest.raises(_lib.AspError) as err:
        ham.plan()
    assert err.value.code == -3
    # more spins than one workgroup's LDS can hold even bit-packed: the sign words move to HBM
    n = 1500000
    big = sa.Hamiltonian(scipy.sparse.identity(n, format="csr"), np.zeros(n))
    x, e = sa.anneal(big, seed=1, number_sweeps=1, repetitions=1, sweep_order="colour")
    assert e == float(n) and _lib.load().asp_sa_last_layout(big.plan()) == 3
    # arguments of anneal
    ok = sa.Hamiltonian(scipy.sparse.identity(5, format="csr"), np.zeros(5))
    with pytest.raises(ValueError):
        sa.anneal(ok, seed=1, number_sweeps=2, repetitions=0)
    with pytest.raises(ValueError):
        sa.anneal(ok, seed=1, number_sweeps=2, repetitions=1, x0=np.zeros(3, np.uint64))
    lib = _lib.load()
    assert lib.asp_sa_set_launch(ok.plan(), 3, 0) == -3
    assert lib.asp_sa_set_launch(ok.plan(), 0, 100) == -3
    bad_betas = np.array([1.0, -2.0])
    out_x = np.zeros(1, np.uint64)
    out_e = np.zeros(1)
    rc = lib.asp_sa_anneal(ok.plan(), 1, _lib.ptr(bad_betas), 2, 1, 0, None, _lib.ptr(out_x),
                           _lib.ptr(out_e))
    assert rc == -3 and b"betas" in lib.asp_last_error()


def test_sector_and_table_entry_points_report_violated_preconditions(models):
    """csrc/sector_basis.hip, plain_basis.hip, key_table.hip: capacity, null pointers, wrong kind
    of basis — an error code and a message, and the library keeps working afterwards."""
    import torch

    from annealing_sign_problem_amd import _lib, operators, sector_ed

    lib = _lib.load()
    ring = operators.Operator.from_config(models["heisenberg_kagome_18"])
    handle = ring.device()._handle
    count = ctypes.c_uint64(0)
    # sizing call: nothing written, the count returned
    _lib.check(lib.asp_sector_enumerate(handle, 9, 0, None, None, ctypes.byref(count)))
    assert count.value == 24310
    small = torch.zeros(100, dtype=torch.int64, device="cuda")
    rc = lib.asp_sector_enumerate(handle, 9, 100, ctypes.c_void_p(small.data_ptr()), None, ctypes.byref(count))
    assert rc == -4 and count.value == 24310 and "capacity" in _lib.last_error()
    assert int(small.abs().sum()) == 0                       # untouched
    assert lib.asp_sector_enumerate(handle, 19, 0, None, None, ctypes.byref(count)) == -3   # weight > n
    assert lib.asp_sector_enumerate(None, 9, 0, None, None, ctypes.byref(count)) == -3
    assert lib.asp_sector_enumerate(handle, 9, 0, None, None, None) == -3
    # rows: width below the operator's, null arrays
    reps, norms = sector_ed.enumerate_sector(ring)
    n = reps.shape[0]
    idx = torch.zeros((4, n), dtype=torch.int32, device="cuda")
    val = torch.zeros((4, n), dtype=torch.float64, device="cuda")
    diag = torch.zeros(n, dtype=torch.float64, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    assert lib.asp_sector_rows(handle, n, p(reps), p(norms), 4, p(idx), p(val), p(diag)) == -3
    assert "width" in _lib.last_error()
    assert lib.asp_sector_rows(handle, n, p(reps), None, 36, p(idx), p(val), p(diag)) == -3
    assert lib.asp_sector_matvec(n, 4, p(idx), p(val), p(diag), p(diag), p(diag)) == -3      # x aliases y
    # the matrix-free product: symmetric bases and magnetisation-violating bonds are refused
    out = ctypes.c_void_p()
    assert lib.asp_plain_basis_create(handle, 9, ctypes.byref(out)) == -3 and "symmetries" in _lib.last_error()
    flip = np.zeros((4, 4))
    flip[0, 1] = flip[1, 0] = 1.0                      # sigma^x on the second site: changes Sz
    bad = operators.Operator(operators.SpinBasis(6, 3), [operators.Term(flip, [(0, 1)])])
    assert lib.asp_plain_basis_create(bad.device()._handle, 3, ctypes.byref(out)) == -3
    assert "magnetisation" in _lib.last_error()
    plain = operators.Operator.from_config(models["heisenberg_kagome_16"])
    assert lib.asp_plain_basis_create(plain.device()._handle, -1, ctypes.byref(out)) == -3
    # and everything still works
    matrix = sector_ed.SectorMatrix(ring, reps, norms)
    energy, _, info = sector_ed.lanczos_ground_state(matrix, tol=1e-10)
    assert abs(energy + 31.054814383595) < 1e-8 and info["residual"] < 1e-6

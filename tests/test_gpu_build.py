"""GPU parity: coupling build (build_matrix / extract_signs / ising_elements /
make_ising_model) through the C ABI vs the oracle and the reference's golden vectors."""
import numpy as np
import pytest
import scipy.sparse

import oracle
from conftest import golden

pytestmark = pytest.mark.gpu


def _norm2(x):
    """The norm make_ising_model divides by (common.norm2: np.linalg.norm up to 10 000 elements,
    numpy's pairwise sum beyond, where BLAS would split the sum over its threads)."""
    from annealing_sign_problem_amd import common

    return common.norm2(x)

BUILD_CASES = ["hand", "word0", "multiword", "allmiss"]
INPUTS = ["spins", "counts", "psi", "other_spins", "other_coeffs", "other_counts", "other_psi"]


def _same(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    assert a.tobytes() == b.tobytes()  # bit-exact, NaN- and signed-zero-proof


@pytest.mark.parametrize("case", BUILD_CASES)
def test_build_matrix_matches_reference_golden(case):
    from annealing_sign_problem_amd import _build_matrix

    g = golden("build_matrix_%s.npz" % case)
    row, col, elements, field = _build_matrix.build_matrix(*[g[k] for k in INPUTS])
    assert row.shape[0] == int(g["nnz"])
    _same(row, g["row"])
    _same(col, g["col"])
    _same(elements, g["elements"])
    _same(field, g["field"])


def test_build_matrix_raw_cffi_style_call():
    """The exact call shape of the reference's cffi user: caller-allocated outputs."""
    from annealing_sign_problem_amd import _build_matrix as bm

    g = golden("build_matrix_word0.npz")
    arrays = [np.ascontiguousarray(g[k]) for k in INPUTS]
    m = arrays[3].shape[0]
    row = np.full(m, 77, np.uint32)
    col = np.full(m, 77, np.uint32)
    el = np.full(m, -1.0)
    field = np.full(arrays[0].shape[0], -1.0)
    nnz = bm.lib.build_matrix(arrays[0].shape[0], *[bm.ffi.from_buffer(a) for a in arrays],
                              bm.ffi.from_buffer(row), bm.ffi.from_buffer(col),
                              bm.ffi.from_buffer(el), bm.ffi.from_buffer(field))
    assert nnz == int(g["nnz"])
    _same(row[:nnz], g["row"])
    _same(el[:nnz], g["elements"])
    _same(field, g["field"])
    assert np.all(row[nnz:] == 77) and np.all(el[nnz:] == -1.0)  # only a prefix is written


@pytest.mark.parametrize("seed,n,mean,multi,miss", [
    (1, 1, 3.0, False, 0.5), (2, 65, 0.5, False, 0.0), (3, 3000, 30.0, False, 0.5),
    (4, 500, 12.0, True, 0.3), (5, 4097, 2.0, False, 0.9),
    # the flat search counts hits per 64 / 2048 / 131072 connections: several super-chunks, multi-word
    # keys; and rows far longer than the 32 lanes that emit them
    (6, 12000, 30.0, True, 0.4), (7, 300, 1000.0, False, 0.5)])
def test_build_matrix_matches_oracle_random(seed, n, mean, multi, miss):
    from annealing_sign_problem_amd import _build_matrix

    from helpers import random_build_case

    c = random_build_case(np.random.default_rng(seed), n, mean, multi, miss, True)
    args = [c[k] for k in INPUTS]
    nnz, row, col, elements, field = oracle.build_matrix(*args)
    g_row, g_col, g_el, g_field = _build_matrix.build_matrix(*args)
    _same(g_row, row)
    _same(g_col, col)
    _same(g_el, elements)
    _same(g_field, field)


def test_build_handle_is_reusable_run_after_run():
    """The device-resident form (asp_build_create / upload / run / download: what bench.py times):
    one build is three launches and leaves its hash slots, flag and super-chunk totals clean for the next
    one — three runs on one upload, then other data of the same shape through the same handle,
    are each the oracle's result (sizes that span several chunks of the flat search)."""
    import ctypes

    from annealing_sign_problem_amd import _build_matrix, _lib

    from helpers import random_build_case

    lib = _lib.load()
    rng = np.random.default_rng(11)
    first = random_build_case(rng, 5000, 9.0, False, 0.4, True)
    k, n = first["spins"].shape[0], first["other_spins"].shape[0]
    # the same shape with other keys, amplitudes and hit pattern (same counts per row)
    second = dict(first)
    second["psi"] = rng.normal(size=k)
    second["other_psi"] = rng.normal(size=n)
    second["other_coeffs"] = rng.normal(size=n)
    second["other_spins"] = first["other_spins"][::-1].copy()
    handle = lib.asp_build_create(ctypes.c_uint64(k), ctypes.c_uint64(n))
    assert handle
    try:
        for case, runs in ((first, 3), (second, 2), (first, 1)):
            arrays = [np.ascontiguousarray(case[key]) for key in INPUTS]
            spins512 = _build_matrix.as_bits512(arrays[0])
            others512 = _build_matrix.as_bits512(arrays[3])
            _lib.check(lib.asp_build_upload(handle, _lib.ptr(spins512), _lib.ptr(arrays[1]), _lib.ptr(arrays[2]),
                                            _lib.ptr(others512), _lib.ptr(arrays[4]), _lib.ptr(arrays[5]),
                                            _lib.ptr(arrays[6])))
            want_nnz, row, col, elements, field = oracle.build_matrix(*arrays)
            for _ in range(runs):
                nnz = ctypes.c_uint64(0)
                _lib.check(lib.asp_build_run(handle, ctypes.byref(nnz)))
                assert nnz.value == want_nnz
                g_row, g_col = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
                g_el, g_field = np.zeros(n), np.zeros(k)
                _lib.check(lib.asp_build_download(handle, _lib.ptr(g_row), _lib.ptr(g_col), _lib.ptr(g_el),
                                                  _lib.ptr(g_field)))
                _same(g_row[:want_nnz], row)
                _same(g_col[:want_nnz], col)
                _same(g_el[:want_nnz], elements)
                _same(g_field, field)
    finally:
        lib.asp_build_destroy(handle)


def test_build_matrix_empty():
    from annealing_sign_problem_amd import _build_matrix

    z = np.zeros
    row, col, el, field = _build_matrix.build_matrix(z((0, 8), np.uint64), z(0, np.int64), z(0),
                                                     z((0, 8), np.uint64), z(0), z(0, np.int64), z(0))
    assert row.size == col.size == el.size == field.size == 0
    # rows but no connections: field is zero-filled (memset at cbits/build_matrix.c:29)
    row, col, el, field = _build_matrix.build_matrix(np.arange(5, dtype=np.uint64), np.ones(5, np.int64),
                                                     np.ones(5), z(0, np.uint64), z(0),
                                                     z(5, np.int64), z(0))
    assert row.size == 0 and np.array_equal(field, np.zeros(5))


def test_extract_signs_matches_reference_golden():
    from annealing_sign_problem_amd import _build_matrix

    g = golden("extract_signs.npz")
    _same(_build_matrix.extract_signs(g["psi"]), g["signs"])
    for n in [1, 63, 64, 65, 1000]:
        psi = np.random.default_rng(n).normal(size=n)
        _same(_build_matrix.extract_signs(psi), oracle.extract_signs(psi))


def test_build_matrix_large_properties():
    """kagome_36-sized build: checked through size-independent properties."""
    from annealing_sign_problem_amd import _build_matrix, synthetic

    J, _, _ = synthetic.planted_cluster(100000, seed=5)
    keys, counts, psi, other, coeffs, other_counts, other_psi = synthetic.build_inputs_from_matrix(J)
    row, col, el, field = _build_matrix.build_matrix(keys, counts, psi, other, coeffs,
                                                     other_counts, other_psi)
    offsets = np.concatenate([[0], np.cumsum(other_counts)])
    row_of = np.repeat(np.arange(keys.shape[0]), other_counts)
    idx = np.searchsorted(keys, other)
    hit = keys[np.minimum(idx, keys.shape[0] - 1)] == other
    assert row.shape[0] == int(hit.sum())
    assert np.array_equal(row, row_of[hit].astype(np.uint32))          # stable order
    assert np.array_equal(col, idx[hit].astype(np.uint32))
    expect = ((counts[row_of] * coeffs) * np.abs(psi[row_of]))[hit] * np.abs(other_psi[hit])
    assert np.array_equal(el, expect)
    # field: rows without misses are exactly zero; totals agree to rounding
    miss_rows = np.unique(row_of[~hit])
    mask = np.ones(keys.shape[0], bool)
    mask[miss_rows] = False
    assert np.all(field[mask] == 0)
    contrib = np.where(hit, 0.0, (counts[row_of] * coeffs) * np.abs(psi[row_of]) * other_psi)
    ref = np.add.reduceat(contrib, offsets[:-1][other_counts > 0])
    assert np.allclose(field[other_counts > 0], ref, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("case", ["ring4", "kagome16_cluster", "sk16_cluster"])
def test_ising_elements_and_model_match_reference_golden(case):
    from annealing_sign_problem_amd import common
    from annealing_sign_problem_amd import annealer as sa

    g = golden("make_ising_%s.npz" % case)
    spins = g["spins"]
    psi = np.exp(g["log_psi"]).real
    psi = np.ascontiguousarray(psi)
    psi /= _norm2(psi)
    idx, member, elements, offsets = common.ising_elements(
        spins, psi, g["other_spins"], g["other_coeffs"], g["other_counts"])
    # numpy restatement of common.py:71-82,116-128,173
    ref_idx = np.clip(np.searchsorted(spins, g["other_spins"]), 0, spins.size - 1)
    ref_member = g["other_spins"] == spins[ref_idx]
    ref_off = np.concatenate([[0], np.cumsum(g["other_counts"])])
    ref_el = g["other_coeffs"] * np.abs(np.where(ref_member, psi[ref_idx], 0))
    ref_el *= np.abs(psi[np.repeat(np.arange(spins.size), g["other_counts"])])
    assert np.array_equal(idx, ref_idx)
    assert np.array_equal(member, ref_member)
    assert np.array_equal(offsets, ref_off)
    _same(elements, ref_el)

    class Basis:
        number_spins = 16

    class Op:
        basis = Basis()

        def batched_apply(self, x):
            lo = np.searchsorted(spins, x[:, 0])
            parts = [np.arange(ref_off[i], ref_off[i + 1]) for i in lo]
            sel = np.concatenate(parts) if parts else np.zeros(0, np.int64)
            out = np.zeros((sel.size, 8), np.uint64)
            out[:, 0] = g["other_spins"][sel]
            return out, g["other_coeffs"][sel].astype(np.complex128), g["other_counts"][lo]

    model = common.make_ising_model(spins, Op(), log_psi=g["log_psi"])
    m = scipy.sparse.coo_matrix(model.ising_hamiltonian.exchange)
    _same(m.row.astype(np.int32), g["row"].astype(np.int32))
    _same(m.col.astype(np.int32), g["col"].astype(np.int32))
    _same(m.data, g["data"])
    _same(model.initial_signs, g["x0"])
    _same(model.spins, g["spins"])
    # energy identity (common.py:757-760): s^T J s at the exact signs
    s = sa.bits_to_signs(model.initial_signs, spins.size)
    dense = s @ (model.ising_hamiltonian.exchange @ s)
    assert abs(model.ising_hamiltonian.energy(model.initial_signs) - dense) <= 1e-12 * abs(dense)

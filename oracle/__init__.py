"""ORACLE — test infrastructure only (see oracle/README.md).

Python face of the CPU oracle: ctypes bindings for ``liboracle.so`` (this
directory's C restatements) and ``_ref/libbuild_matrix_ref.so`` (the reference's
own ``cbits/build_matrix.c`` compiled in place), plus numpy restatements of the
reference's live coupling build (``ising_oracle.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np

from . import build_oracle as _build

_HERE = os.path.dirname(os.path.abspath(__file__))

_c_void_p = ctypes.c_void_p
_u64 = ctypes.c_uint64
_u32 = ctypes.c_uint32
_i32 = ctypes.c_int32


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_c_void_p)


_lib = None
_ref = None


def lib() -> ctypes.CDLL:
    """liboracle.so (built on first use when a compiler is present)."""
    global _lib
    if _lib is None:
        path = _build.ORACLE_LIB
        try:
            path = _build.build_oracle()
        except Exception:
            if not os.path.exists(path):
                raise
        _lib = ctypes.CDLL(path)
        _lib.oracle_build_matrix.restype = _u64
        _lib.oracle_expneg.restype = ctypes.c_double
        _lib.oracle_expneg.argtypes = [ctypes.c_double]
    return _lib


def ref_lib() -> Optional[ctypes.CDLL]:
    """The reference's build_matrix.c compiled in place, or None if unavailable."""
    global _ref
    if _ref is None:
        path = _build.build_reference()
        if path is None or not os.path.exists(path):
            return None
        _ref = ctypes.CDLL(path)
        _ref.build_matrix.restype = _u64
    return _ref


# ----------------------------------------------------------------------------
# coupling build (cbits/build_matrix.c)
# ----------------------------------------------------------------------------

def as_keys512(keys) -> np.ndarray:
    """1-D uint64 keys -> C-contiguous (n, 8) zero-padded; (n, 8) passes through."""
    keys = np.asarray(keys, dtype=np.uint64)
    if keys.ndim == 1:
        out = np.zeros((keys.shape[0], 8), dtype=np.uint64)
        out[:, 0] = keys
        return out
    assert keys.ndim == 2 and keys.shape[1] == 8
    return np.ascontiguousarray(keys)


def _call_build(fn, spins, counts, psi, other_spins, other_coeffs, other_counts, other_psi):
    spins = as_keys512(spins)
    other_spins = as_keys512(other_spins)
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    psi = np.ascontiguousarray(psi, dtype=np.float64)
    other_coeffs = np.ascontiguousarray(other_coeffs, dtype=np.float64)
    other_counts = np.ascontiguousarray(other_counts, dtype=np.int64)
    other_psi = np.ascontiguousarray(other_psi, dtype=np.float64)
    n = spins.shape[0]
    m = other_spins.shape[0]
    assert int(other_counts.sum()) == m == other_coeffs.shape[0] == other_psi.shape[0]
    row = np.full(max(m, 1), 0xFFFFFFFF, dtype=np.uint32)
    col = np.full(max(m, 1), 0xFFFFFFFF, dtype=np.uint32)
    elements = np.full(max(m, 1), np.nan, dtype=np.float64)
    field = np.full(max(n, 1), np.nan, dtype=np.float64)
    nnz = fn(
        _u64(n), _ptr(spins), _ptr(counts), _ptr(psi), _ptr(other_spins), _ptr(other_coeffs),
        _ptr(other_counts), _ptr(other_psi), _ptr(row), _ptr(col), _ptr(elements), _ptr(field),
    )
    nnz = int(nnz)
    return nnz, row[:nnz].copy(), col[:nnz].copy(), elements[:nnz].copy(), field[:n].copy()


def build_matrix(*args):
    """This repo's C restatement. Returns (nnz, row, col, elements, field)."""
    return _call_build(lib().oracle_build_matrix, *args)


def ref_build_matrix(*args):
    """The reference itself (oracle/_ref). Raises if it could not be built/found."""
    r = ref_lib()
    if r is None:
        raise RuntimeError("reference build (oracle/_ref) unavailable")
    return _call_build(r.build_matrix, *args)


def _call_signs(fn, psi):
    psi = np.ascontiguousarray(psi, dtype=np.float64)
    n = psi.shape[0]
    out = np.full(max((n + 63) // 64, 1), 0xDEADBEEFDEADBEEF, dtype=np.uint64)
    fn(_u64(n), _ptr(psi), _ptr(out))
    return out[: (n + 63) // 64].copy()


def extract_signs(psi):
    return _call_signs(lib().oracle_extract_signs, psi)


def ref_extract_signs(psi):
    r = ref_lib()
    if r is None:
        raise RuntimeError("reference build (oracle/_ref) unavailable")
    return _call_signs(r.extract_signs, psi)


# ----------------------------------------------------------------------------
# annealer specification ASP-SA-1 (sa_oracle.c)
# ----------------------------------------------------------------------------

def philox4x32_10(ctr, key) -> np.ndarray:
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().oracle_philox4x32_10(_ptr(ctr), _ptr(key), _ptr(out))
    return out


def expneg(x: float) -> float:
    return float(lib().oracle_expneg(float(x)))


def _csr_args(matrix):
    import scipy.sparse

    m = scipy.sparse.csr_matrix(matrix)
    m.sum_duplicates()
    m.sort_indices()
    indptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(m.indices, dtype=np.int32)
    data = np.ascontiguousarray(m.data, dtype=np.float64)
    return m.shape[0], indptr, indices, data


def sa_anneal(matrix, field, seed: int, betas, repetitions: int, replica_offset: int = 0,
              x0=None, energy_scale_exp: int = 0, num_threads: int = 1):
    """Returns (x[R, words] uint64, e[R] float64, tracked[R] int64, accepted[R] uint64)."""
    n, indptr, indices, data = _csr_args(matrix)
    field = np.ascontiguousarray(field, dtype=np.float64)
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    words = (n + 63) // 64
    out_x = np.zeros((repetitions, max(words, 1)), dtype=np.uint64)
    out_e = np.zeros(repetitions, dtype=np.float64)
    tracked = np.zeros(repetitions, dtype=np.int64)
    accepted = np.zeros(repetitions, dtype=np.uint64)
    if x0 is not None:
        x0 = np.ascontiguousarray(x0, dtype=np.uint64)
        assert x0.shape[0] == words
    rc = lib().oracle_sa_anneal(
        _u64(n), _ptr(indptr), _ptr(indices), _ptr(data), _ptr(field), _u64(seed & (2**64 - 1)),
        _ptr(betas), _u32(betas.shape[0]), _u32(repetitions), _u32(replica_offset), _ptr(x0),
        _i32(energy_scale_exp), _ptr(out_x), _ptr(out_e), _ptr(tracked), _ptr(accepted),
        ctypes.c_int(num_threads),
    )
    if rc != 0:
        raise RuntimeError("oracle_sa_anneal failed")
    return out_x[:, :words], out_e, tracked, accepted


def sa_anneal_shuffled(matrix, field, seed: int, betas, repetitions: int, replica_offset: int = 0,
                       x0=None, energy_scale_exp: int = 0, num_threads: int = 1):
    """`sa_anneal` with a fresh visiting order every sweep."""
    n, indptr, indices, data = _csr_args(matrix)
    field = np.ascontiguousarray(field, dtype=np.float64)
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    words = (n + 63) // 64
    out_x = np.zeros((repetitions, max(words, 1)), dtype=np.uint64)
    out_e = np.zeros(repetitions, dtype=np.float64)
    tracked = np.zeros(repetitions, dtype=np.int64)
    accepted = np.zeros(repetitions, dtype=np.uint64)
    if x0 is not None:
        x0 = np.ascontiguousarray(x0, dtype=np.uint64)
        assert x0.shape[0] == words
    rc = lib().oracle_sa_anneal_shuffled(
        _u64(n), _ptr(indptr), _ptr(indices), _ptr(data), _ptr(field), _u64(seed & (2**64 - 1)),
        _ptr(betas), _u32(betas.shape[0]), _u32(repetitions), _u32(replica_offset), _ptr(x0),
        _i32(energy_scale_exp), _ptr(out_x), _ptr(out_e), _ptr(tracked), _ptr(accepted),
        ctypes.c_int(num_threads),
    )
    if rc != 0:
        raise RuntimeError("oracle_sa_anneal_shuffled failed")
    return out_x[:, :words], out_e, tracked, accepted


def sa_anneal_trace(matrix, field, seed: int, betas, repetitions: int, replica_offset: int = 0,
                    x0=None, energy_scale_exp: int = 0, num_threads: int = 1):
    """Returns (x[R, words], e[R], trace int64[R, T+1]): trace[r, t] = tracked fixed-point energy
    of chain r after t sweeps, relative to its initial configuration."""
    n, indptr, indices, data = _csr_args(matrix)
    field = np.ascontiguousarray(field, dtype=np.float64)
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    words = (n + 63) // 64
    out_x = np.zeros((repetitions, max(words, 1)), dtype=np.uint64)
    out_e = np.zeros(repetitions, dtype=np.float64)
    trace = np.zeros((repetitions, betas.shape[0] + 1), dtype=np.int64)
    if x0 is not None:
        x0 = np.ascontiguousarray(x0, dtype=np.uint64)
        assert x0.shape[0] == words
    rc = lib().oracle_sa_anneal_trace(
        _u64(n), _ptr(indptr), _ptr(indices), _ptr(data), _ptr(field), _u64(seed & (2**64 - 1)),
        _ptr(betas), _u32(betas.shape[0]), _u32(repetitions), _u32(replica_offset), _ptr(x0),
        _i32(energy_scale_exp), _ptr(out_x), _ptr(out_e), _ptr(trace), ctypes.c_int(num_threads),
    )
    if rc != 0:
        raise RuntimeError("oracle_sa_anneal_trace failed")
    return out_x[:, :words], out_e, trace


def sa_energy(matrix, field, x) -> np.ndarray:
    n, indptr, indices, data = _csr_args(matrix)
    field = np.ascontiguousarray(field, dtype=np.float64)
    words = (n + 63) // 64
    x = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, max(words, 1))
    out = np.zeros(x.shape[0], dtype=np.float64)
    rc = lib().oracle_sa_energy(_u64(n), _ptr(indptr), _ptr(indices), _ptr(data), _ptr(field),
                                _u32(x.shape[0]), _ptr(x), _ptr(out))
    if rc != 0:
        raise RuntimeError("oracle_sa_energy failed")
    return out


def sa_layout(matrix):
    """(colors[K] int32, order[K] int64, num_colors, nnz_offdiag, diag_sum)."""
    n, indptr, indices, data = _csr_args(matrix)
    colors = np.zeros(max(n, 1), dtype=np.int32)
    order = np.zeros(max(n, 1), dtype=np.int64)
    ncol = _i32(0)
    nnz = ctypes.c_int64(0)
    diag = ctypes.c_double(0.0)
    rc = lib().oracle_sa_layout(_u64(n), _ptr(indptr), _ptr(indices), _ptr(data), _ptr(colors),
                                _ptr(order), ctypes.byref(ncol), ctypes.byref(nnz), ctypes.byref(diag))
    if rc != 0:
        raise RuntimeError("oracle_sa_layout failed")
    return colors[:n], order[:n], ncol.value, nnz.value, diag.value


def greedy_solve(matrix, field, max_sweeps: int = 10000, relax: bool = True):
    """ASP-GREEDY-1 on the CPU: (x uint64[words], e float)."""
    n, indptr, indices, data = _csr_args(matrix)
    field = np.ascontiguousarray(field, dtype=np.float64)
    words = (n + 63) // 64
    x = np.zeros(max(words, 1), dtype=np.uint64)
    e = ctypes.c_double(0.0)
    rc = lib().oracle_greedy_solve(_u64(n), _ptr(indptr), _ptr(indices), _ptr(data), _ptr(field),
                                   _u32(max_sweeps), ctypes.c_int(1 if relax else 0), _ptr(x),
                                   ctypes.byref(e))
    if rc != 0:
        raise RuntimeError("oracle_greedy_solve failed")
    return x[:words], e.value


# ----------------------------------------------------------------------------
# Hamiltonian action + fused coupling build (operator_oracle.c)
# ----------------------------------------------------------------------------

def _bond_args(bond_table):
    a, b, m = bond_table
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    m = np.ascontiguousarray(m, dtype=np.float64).reshape(a.shape[0], 16)
    return a, b, m


def operator_apply(bond_table, keys):
    """Flat batched_apply: (other_keys u64[N], other_coeffs f64[N], other_counts i64[n])."""
    a, b, m = _bond_args(bond_table)
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    n = keys.shape[0]
    per = 1 + 3 * a.shape[0]
    other = np.zeros(max(n * per, 1), dtype=np.uint64)
    coeff = np.zeros(max(n * per, 1), dtype=np.float64)
    counts = np.zeros(max(n, 1), dtype=np.int64)
    fn = lib().oracle_operator_apply
    fn.restype = _u64
    total = fn(_u32(a.shape[0]), _ptr(a), _ptr(b), _ptr(m), _u64(n), _ptr(keys), _ptr(other),
               _ptr(coeff), _ptr(counts))
    return other[:total].copy(), coeff[:total].copy(), counts[:n]


def operator_ising(bond_table, keys, psi):
    """COO (row i32, col i32, val f64) of 0.5 * (M + M^T), the reference's way."""
    a, b, m = _bond_args(bond_table)
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    psi = np.ascontiguousarray(psi, dtype=np.float64)
    k = keys.shape[0]
    cap = max(2 * k * (1 + 3 * a.shape[0]), 1)
    row = np.zeros(cap, dtype=np.int32)
    col = np.zeros(cap, dtype=np.int32)
    val = np.zeros(cap, dtype=np.float64)
    fn = lib().oracle_operator_ising
    fn.restype = _u64
    nnz = fn(_u32(a.shape[0]), _ptr(a), _ptr(b), _ptr(m), _u64(k), _ptr(keys), _ptr(psi),
             _ptr(row), _ptr(col), _ptr(val))
    if nnz == 2 ** 64 - 1:
        raise MemoryError("oracle_operator_ising")
    return row[:nnz].copy(), col[:nnz].copy(), val[:nnz].copy()


def operator_extend(bond_table, keys):
    a, b, m = _bond_args(bond_table)
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    n = keys.shape[0]
    out = np.zeros(max(n * (1 + 3 * a.shape[0]), 1), dtype=np.uint64)
    fn = lib().oracle_operator_extend
    fn.restype = _u64
    count = fn(_u32(a.shape[0]), _ptr(a), _ptr(b), _ptr(m), _u64(n), _ptr(keys), _ptr(out))
    if count == 2 ** 64 - 1:
        raise MemoryError("oracle_operator_extend")
    return out[:count].copy()


# ----------------------------------------------------------------------------
# global-cutoff sparsification (sparsify_oracle.py)
# ----------------------------------------------------------------------------

from .sparsify_oracle import sparsify_component  # noqa: E402,F401

"""Stand-in for the reference's cffi extension module ``_build_matrix``
(annealing_sign_problem/build_extension.py:22-30): ``lib.build_matrix`` and
``lib.extract_signs`` with the C signatures of cbits/build_matrix.h:7-14, bound
with ctypes to the HIP implementation in libasp_hip.so.

``lib.*`` accept what cffi callers pass — numpy arrays, ``ffi.from_buffer``
results, ctypes pointers or integer addresses.  The array-level helpers
(:func:`build_matrix`, :func:`extract_signs`) allocate the outputs the way the
reference's (missing) ``extract_classical_ising_model`` did
(call sites experiments/sampled_connected_components.py:507-513).
"""
from __future__ import annotations

import ctypes
from typing import Tuple

import numpy as np

from . import _lib


def _address(obj):
    if obj is None:
        return None
    if isinstance(obj, np.ndarray):
        if not obj.flags["C_CONTIGUOUS"]:
            raise ValueError("arrays passed to the C ABI must be C-contiguous")
        return obj.ctypes.data_as(ctypes.c_void_p)
    if isinstance(obj, int):
        return ctypes.c_void_p(obj)
    return ctypes.cast(obj, ctypes.c_void_p)


class _Lib:
    """``_build_matrix.lib`` of the reference."""

    @staticmethod
    def build_matrix(num_spins, spins, counts, psi, other_spins, other_coeffs, other_counts,
                     other_psi, row_indices, col_indices, elements, field) -> int:
        lib = _lib.load()
        nnz = lib.build_matrix(
            ctypes.c_uint64(int(num_spins)), _address(spins), _address(counts), _address(psi),
            _address(other_spins), _address(other_coeffs), _address(other_counts),
            _address(other_psi), _address(row_indices), _address(col_indices), _address(elements),
            _address(field),
        )
        _lib.check_recorded()
        return int(nnz)

    @staticmethod
    def extract_signs(num_spins, psi, signs) -> None:
        lib = _lib.load()
        lib.extract_signs(ctypes.c_uint64(int(num_spins)), _address(psi), _address(signs))
        _lib.check_recorded()


class _FFI:
    """The two cffi calls a caller of the reference module needs."""

    NULL = None

    @staticmethod
    def from_buffer(*args):
        array = args[-1]
        return np.ascontiguousarray(array).ctypes.data_as(ctypes.c_void_p)

    @staticmethod
    def cast(_ctype: str, pointer):
        return _address(pointer)


lib = _Lib()
ffi = _FFI()


def as_bits512(keys) -> np.ndarray:
    """``_normalize_spins`` (annealing_sign_problem/common.py:58-68)."""
    keys = np.asarray(keys, dtype=np.uint64, order="C")
    if keys.ndim <= 1:
        keys = keys.reshape(-1)
        out = np.zeros((keys.shape[0], 8), dtype=np.uint64)
        out[:, 0] = keys
        return out
    if keys.ndim == 2:
        if keys.shape[1] != 8:
            raise ValueError("'spins' has wrong shape: {}; expected (?, 8)".format(keys.shape))
        return np.ascontiguousarray(keys)
    raise ValueError("'spins' has wrong shape: {}; expected a 2D array".format(keys.shape))


def build_matrix(spins, counts, psi, other_spins, other_coeffs, other_counts,
                 other_psi) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Array-level call: returns (row u32[nnz], col u32[nnz], elements f64[nnz], field f64[K])."""
    spins = as_bits512(spins)
    other_spins = as_bits512(other_spins)
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    psi = np.ascontiguousarray(psi, dtype=np.float64)
    other_coeffs = np.ascontiguousarray(other_coeffs, dtype=np.float64)
    other_counts = np.ascontiguousarray(other_counts, dtype=np.int64)
    other_psi = np.ascontiguousarray(other_psi, dtype=np.float64)
    n = spins.shape[0]
    m = other_spins.shape[0]
    if not (counts.shape[0] == psi.shape[0] == other_counts.shape[0] == n):
        raise ValueError("per-row arrays must have one entry per spin")
    if not (other_coeffs.shape[0] == other_psi.shape[0] == m) or int(other_counts.sum()) != m:
        raise ValueError("flat connection arrays must have sum(other_counts) entries")
    row = np.empty(max(m, 1), dtype=np.uint32)
    col = np.empty(max(m, 1), dtype=np.uint32)
    elements = np.empty(max(m, 1), dtype=np.float64)
    field = np.zeros(max(n, 1), dtype=np.float64)
    nnz = lib.build_matrix(n, spins, counts, psi, other_spins, other_coeffs, other_counts,
                           other_psi, row, col, elements, field)
    return row[:nnz].copy(), col[:nnz].copy(), elements[:nnz].copy(), field[:n].copy()


def extract_signs(psi) -> np.ndarray:
    psi = np.ascontiguousarray(psi, dtype=np.float64)
    n = psi.shape[0]
    signs = np.zeros(max((n + 63) // 64, 1), dtype=np.uint64)
    lib.extract_signs(n, psi, signs)
    return signs[: (n + 63) // 64]

"""``sa.greedy_solve`` (annealing_sign_problem/common.py:250) on the MI355X path:
strongest-coupling-first cluster merging on the host, strict-descent relaxation sweeps on
the GPU (``asp_sa_greedy``; specification in DESIGN.md §4.8)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

MAX_RELAXATION_SWEEPS = 10000


def greedy_solve(hamiltonian, max_sweeps: int = MAX_RELAXATION_SWEEPS):
    """Returns ``(x, e)``: packed configuration (uint64[ceil(K/64)]) and its energy."""
    lib = _lib.load()
    words = (hamiltonian.size + 63) // 64
    x = np.zeros(max(words, 1), dtype=np.uint64)
    e = ctypes.c_double(0.0)
    sweeps = ctypes.c_uint32(0)
    energy = np.zeros(1, dtype=np.float64)
    _lib.check(lib.asp_sa_greedy(hamiltonian.plan(), ctypes.c_uint32(int(max_sweeps)), _lib.ptr(x),
                                 _lib.ptr(energy), ctypes.byref(sweeps)))
    return x[:words], float(energy[0])

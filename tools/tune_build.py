"""Coupling-build timing on device-resident inputs (development aid)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _build_matrix, _lib, synthetic  # noqa: E402

lib = _lib.load()
for k, kind in [(10000, "planted"), (100000, "planted"), (8192, "sk")]:
    if kind == "sk":
        J, _ = synthetic.sk_cluster(k)
    else:
        J, _, _ = synthetic.planted_cluster(k, seed=783494)
    keys, counts, psi, other, coeffs, oc, opsi = synthetic.build_inputs_from_matrix(J)
    spins = _build_matrix.as_bits512(keys)
    others = _build_matrix.as_bits512(other)
    h = ctypes.c_void_p(lib.asp_build_create(spins.shape[0], others.shape[0]))
    _lib.check(lib.asp_build_upload(h, _lib.ptr(spins), _lib.ptr(counts), _lib.ptr(psi), _lib.ptr(others),
                                    _lib.ptr(coeffs), _lib.ptr(oc), _lib.ptr(opsi)))
    nnz = ctypes.c_uint64(0)
    ts = []
    for _ in range(8):
        _lib.check(lib.asp_build_run(h, ctypes.byref(nnz)))
        ts.append(lib.asp_build_last_ms(h))
    lib.asp_build_destroy(h)
    ms = float(np.median(ts[2:]))
    m = others.shape[0]
    print("%s K=%d connections=%d nnz=%d: %.3f ms  %.2f Gconn/s  %.0f GB/s algorithmic (96 B/conn)" % (
        kind, k, m, nnz.value, ms, m / ms / 1e6, m * 96 / ms / 1e6), flush=True)
    # the drop-in symbol itself with host pointers (ctypes call on prepared 512-bit arrays)
    import time
    n_rows = spins.shape[0]
    row = np.empty(max(m, 1), np.uint32)
    col = np.empty(max(m, 1), np.uint32)
    el = np.empty(max(m, 1), np.float64)
    fld = np.empty(max(n_rows, 1), np.float64)
    times = []
    for _ in range(6):
        t0 = time.perf_counter()
        _build_matrix.lib.build_matrix(n_rows, spins, counts, psi, others, coeffs, oc, opsi, row, col, el, fld)
        times.append((time.perf_counter() - t0) * 1e3)
    up = spins.nbytes + others.nbytes + 8 * (3 * n_rows + 2 * m)
    down = 16 * int(nnz.value) + 8 * n_rows
    print("   build_matrix(host pointers): %.2f ms median of 5 (first call %.1f ms); %.0f MB up + %.0f MB down "
          "= %.1f ms at 63 GB/s" % (float(np.median(times[1:])), times[0], up / 1e6, down / 1e6,
                                    (up + down) / 63e9 * 1e3), flush=True)

"""Only the three kernels of `build_matrix` (device-resident handle, K = 1e5 kagome_36-sized planted
cluster, the workload of bench.py's build leg), for rocprofv3 passes of their own:
  rocprofv3 --kernel-trace --stats -- python3 tools/profile_build_matrix.py
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/profile_build_matrix.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python3 tools/profile_build_matrix.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _build_matrix, _lib, synthetic  # noqa: E402

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = _lib.load()
J, _, _ = synthetic.planted_cluster(100000, seed=783494)
keys, counts, psi, other, coeffs, oc, opsi = synthetic.build_inputs_from_matrix(J)
spins = _build_matrix.as_bits512(keys)
others = _build_matrix.as_bits512(other)
h = ctypes.c_void_p(lib.asp_build_create(spins.shape[0], others.shape[0]))
_lib.check(lib.asp_build_upload(h, _lib.ptr(spins), _lib.ptr(counts), _lib.ptr(psi), _lib.ptr(others),
                                _lib.ptr(coeffs), _lib.ptr(oc), _lib.ptr(opsi)))
nnz = ctypes.c_uint64(0)
ts = []
for _ in range(RUNS):
    _lib.check(lib.asp_build_run(h, ctypes.byref(nnz)))
    ts.append(lib.asp_build_last_ms(h))
lib.asp_build_destroy(h)
m = others.shape[0]
ms = float(np.median(ts[1:]))
print("build_matrix K=%d connections=%d nnz=%d: %.4f ms (median of %d runs; all: %s) = %.2f G connections/s = "
      "%.0f GB/s algorithmic at 96 B per connection" % (
          spins.shape[0], m, nnz.value, ms, RUNS - 1, " ".join("%.3f" % t for t in ts), m / ms / 1e6,
          m * 96 / ms / 1e6), flush=True)
_lib.shutdown()

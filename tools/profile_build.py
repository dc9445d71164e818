"""The build-side kernels under rocprofv3 --kernel-trace --stats: build_matrix (512-bit ABI),
asp_ising_elements, the fused operator build, the extension, the
symmetric-basis action and the sparsification, at the sizes bench.py quotes (K = 1e5)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import _build_matrix, _lib, common, operators, synthetic  # noqa: E402

lib = _lib.load()
J, _, _ = synthetic.planted_cluster(100000, seed=783494)
keys, counts, psi, other, coeffs, oc, opsi = synthetic.build_inputs_from_matrix(J)
for _ in range(3):
    _build_matrix.build_matrix(keys, counts, psi, other, coeffs, oc, opsi)
for _ in range(3):
    common.ising_elements(keys, psi, other, coeffs, oc)
op = operators.Operator.from_config(synthetic.kagome_lattice())
cluster = synthetic.grow_cluster(op, int("01" * 18, 2), 100000, seed=1)
amp = np.ascontiguousarray(np.exp(synthetic.hashed_log_amplitudes(cluster)).real)
amp /= np.linalg.norm(amp)
dev = op.device()
for _ in range(3):
    dev.ising(cluster, amp)
    dev.extend(cluster[:20000])
sym = operators.Operator.from_config(synthetic.load_models()["heisenberg_kagome_36"])
reps = np.unique(sym.basis.group.state_info(cluster[:20000])[0])
for _ in range(3):
    sym.device().apply(reps)
model = common.make_ising_model(cluster[:30000], op, log_psi=np.log(amp[:30000]) + 0j)
bigger = common.make_hamiltonian_extension(model, lambda s: synthetic.hashed_log_amplitudes(np.asarray(s)))
for _ in range(3):  # (a cutoff small enough to keep the hashed-amplitude cluster connected)
    common.sparsify_using_global_cutoff(bigger, 1e-12, model.spins)
print("profile_build done: K=%d connections=%d, extension %d states" % (keys.shape[0], other.shape[0], bigger.size))
_lib.shutdown()

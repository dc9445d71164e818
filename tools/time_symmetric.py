"""Stage times of one model build on the symmetry-adapted 36-site kagome basis
(heisenberg_kagome_36.yaml: 144 lattice maps x spin inversion): cluster of representatives grown
through the device action, make_ising_model (generic route: device action -> asp_ising_elements
-> scipy symmetrisation), extension, plan.  (Development aid.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import common, operators, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

op = operators.Operator.from_config(synthetic.load_models()["heisenberg_kagome_36"])
dev = op.device()
start = int(op.basis.group.state_info(np.array([int("01" * 18, 2)], dtype=np.uint64))[0][0])
for size in [int(a) for a in sys.argv[1:]] or [1000, 10000]:
    rng = np.random.default_rng(1)
    members = np.array([start], dtype=np.uint64)
    frontier = members
    while members.shape[0] < size:
        other, _, _ = dev.apply(frontier)
        cand = np.setdiff1d(np.unique(other), members)
        cand = cand[rng.random(cand.shape[0]) <= 0.5]
        if members.shape[0] + cand.shape[0] > size:
            cand = rng.permutation(cand)[: size - members.shape[0]]
        members = np.union1d(members, cand)
        frontier = cand
    log_psi = synthetic.hashed_log_amplitudes(members)
    for rep in range(2):
        t0 = time.perf_counter()
        other, coeffs, counts = dev.apply(members)
        t1 = time.perf_counter()
        model = common.make_ising_model(members, op, log_psi=log_psi)
        t2 = time.perf_counter()
        bigger = dev.extend(members)
        t3 = time.perf_counter()
        model.ising_hamiltonian.info()
        t4 = time.perf_counter()
    print("K=%d representatives, %d connections, J nnz %d: action %.2f ms (device %.2f), "
          "make_ising_model %.2f ms, extension to %d states %.2f ms, plan %.2f ms" % (
              size, other.shape[0], model.ising_hamiltonian.exchange.nnz, (t1 - t0) * 1e3, dev.last_ms,
              (t2 - t1) * 1e3, bigger.shape[0], (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)

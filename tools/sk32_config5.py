"""BASELINE config 5 on the real model: sk_32_1 + NOISE = 0.79, dense random-J SK Ising, 4096
replicas.  Ground state of the 6.0e8-state basis on the GPU (matrix-free Lanczos, ~4 min), one
sampled cluster extended once with noisy amplitudes, then
  * parity: 8 chains x 12 sweeps against the CPU oracle, bit for bit;
  * throughput: 4096 chains x 128 sweeps.
(Development aid; GPU.  oracle/ is used as the checker only.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from annealing_sign_problem_amd import _lib, common, operators, sampled_components, sector_ed, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402

log = lambda s: print(s, flush=True)  # noqa: E731
op = operators.Operator.from_config(synthetic.load_models()["sk_32_1"])
energy, psi, states, info = sector_ed.ground_state(op, tol=1e-8, log=log)
op.basis.build(states)
np.random.seed(435834)
noisy = common.add_noise_to_amplitudes(psi, 0.79)
fn = common.ground_state_to_log_coeff_fn(noisy, op.basis)
cluster = sampled_components.generate_clusters(op, psi, 1, 0.1, 200, 400, 0.5)[0]
t0 = time.time()
h = common.make_ising_model(cluster, op, log_psi_fn=fn)
h = common.make_hamiltonian_extension(h, fn)
h = common.sparsify_using_global_cutoff(h, 1e-6, cluster)
ham = h.ising_hamiltonian
hinfo = ham.info()
j = ham.exchange
log("cluster of %d states -> order-1 model: K = %d, nnz = %d (dbar = %.1f), %d colours, max degree %d  [%.2f s]" % (
    len(cluster), h.size, j.nnz, j.nnz / j.shape[0], hinfo.num_colors, hinfo.max_degree, time.time() - t0))
betas = sa.make_schedule(hinfo.beta0_auto, hinfo.beta1_auto, 12)
xs, es = sa.anneal_raw(ham, 435834, betas, 8)
oxs, oes, _, _ = oracle.sa_anneal(j, ham.field, 435834, betas, 8, 0, None, hinfo.energy_scale_exp, num_threads=8)
assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
log("parity: 8 chains x 12 sweeps equal the oracle bit for bit")
betas = sa.make_schedule(hinfo.beta0_auto, hinfo.beta1_auto, 128)
sa.anneal_raw(ham, 435834, betas, 4096)
t0 = time.time()
for _ in range(3):
    sa.anneal_raw(ham, 435834, betas, 4096)
dt = (time.time() - t0) / 3
log("throughput: 4096 chains x 128 sweeps on K = %d: %.3f s per call = %.1f G flips/s (kernel %.1f ms)" % (
    h.size, dt, h.size * 4096 * 128 / dt / 1e9, _lib.load().asp_sa_last_sweep_ms(ham.plan())))

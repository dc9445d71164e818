"""Fixtures of BASELINE configs 4 and 5 on the REAL models (run once on a GPU box; the outputs are
committed under tests/golden/ and read by tests/test_gpu_configs.py):

  * config 5: sk_32_1 + NOISE = 0.79 (reference Makefile:129-141, physical_systems/sk_32_1.yaml):
    ground state of the 6.0e8-state basis (matrix-free Lanczos, ~4 min), amplitudes with noise,
    one sampled cluster extended once and cut at 1e-6 as the pipeline does;
  * config 4: heisenberg_pyrochlore_2x2x2, CUTOFF = 2e-6 (Makefile:101-113, slurm-tcm-big.sh:7):
    ground state of the symmetric sector, one sampled cluster extended twice.

A fixture holds the model's INPUTS — basis states u64[K] and the (noisy, cluster-normalised)
amplitudes f64[K] the couplings were built from — plus sizes and the sha256 of the couplings this
run produced from them.  The test rebuilds J from the inputs with asp_operator_ising and compares
with the oracle; nothing of the reference travels.

    python tests/golden/generate_config_fixtures.py [sk_32_1] [pyrochlore]
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from annealing_sign_problem_amd import common, operators, sampled_components, sector_ed, synthetic  # noqa: E402

log = lambda s: print(s, flush=True)  # noqa: E731
OUT = os.path.join(ROOT, "gpurun_out", "fixtures")
os.makedirs(OUT, exist_ok=True)


def make(name, model, noise, order, cutoff, sizes, target, seed=435834):
    t0 = time.time()
    op = operators.Operator.from_config(synthetic.load_models()[model])
    energy, psi, states, info = sector_ed.ground_state(op, tol=1e-8, log=log)
    op.basis.build(states)
    log("%s: ground state E = %.12f in %.1f s" % (model, energy, time.time() - t0))
    np.random.seed(seed)
    noisy = common.add_noise_to_amplitudes(psi, noise) if noise > 0 else psi
    fn = common.ground_state_to_log_coeff_fn(noisy, op.basis)
    clusters = sampled_components.generate_clusters(op, psi, 6, 0.1, sizes[0], sizes[1], 0.5)
    best = None
    for cluster in clusters:
        h = common.make_ising_model(cluster, op, log_psi_fn=fn)
        for _ in range(order):
            h = common.make_hamiltonian_extension(h, fn)
            h = common.sparsify_using_global_cutoff(h, cutoff, cluster)
        log("  cluster of %d states -> order-%d model of %d spins, nnz %d" % (
            len(cluster), order, h.size, h.ising_hamiltonian.exchange.nnz))
        if best is None or abs(np.log(h.size / target)) < abs(np.log(best[1].size / target)):
            best = (cluster, h)
    cluster, h = best
    spins = np.ascontiguousarray(h.spins, dtype=np.uint64)
    # the amplitudes the couplings are products of: J_ij = H_ij |psi_i| |psi_j| (common.py:173-196);
    # recover them the way make_ising_model computed them for the LAST build and re-derive J
    # from (spins, psi) below, so that the fixture is self-consistent
    log_psi = fn(spins)
    psi_model = np.ascontiguousarray(np.exp(log_psi).real)
    psi_model /= np.linalg.norm(psi_model)
    row, col, val = op.device().ising(spins, psi_model)
    digest = hashlib.sha256(row.tobytes() + col.tobytes() + val.tobytes()).hexdigest()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, spins=spins, psi=psi_model, seed_cluster=np.asarray(cluster, dtype=np.uint64),
                        nnz=np.int64(val.shape[0]), sha256=np.array(digest), model=np.array(model),
                        noise=np.float64(noise), order=np.int64(order), cutoff=np.float64(cutoff),
                        ground_state_energy=np.float64(energy))
    log("%s: K = %d, nnz = %d, sha256 %s..., %.0f KiB -> %s" % (
        name, spins.shape[0], val.shape[0], digest[:16], os.path.getsize(path) / 1024, path))
    op.release_device()


if __name__ == "__main__":
    which = sys.argv[1:] or ["pyrochlore", "sk_32_1"]
    if "pyrochlore" in which:
        make("config4_pyrochlore_cluster", "heisenberg_pyrochlore_2x2x2", 0.0, 2, 2e-6, (50, 1000), 40000)
    if "sk_32_1" in which:
        make("config5_sk32_cluster", "sk_32_1", 0.79, 1, 1e-6, (200, 400), 50000)

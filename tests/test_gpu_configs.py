"""GPU parity at BASELINE.json's configurations — the full 16-site Hilbert spaces
(K = C(16,8) = 12 870) the reference's `make small` anneals (Makefile:27-35) and sampled clusters
of the REAL 36- and 32-site models, whose ground states are computed on the GPU in the test
(sector_ed.py; the reference downloads them, Makefile:143-153):

  config 1  j1j2_square_4x4 (physical_systems/j1j2_square_4x4.yaml:18-41, incl. the (3,3) = 1
            entry of the J2 matrix): coupling build, energy identity, 1 and few replicas
  config 2  heisenberg_kagome_16: 256 replicas, fixed seed, chains vs the oracle
  config 3  heisenberg_kagome_36: a sampled cluster extended twice (K ~ 1e4), 1024 replicas
  config 4  heisenberg_pyrochlore_2x2x2: cluster extension with CUTOFF = 2e-6, replicas in two
            shards (global replica ids) equal to one call
  config 5  sk_32_1 + NOISE = 0.79: a sampled cluster of the real model extended once — its
            basis states and noisy amplitudes are a committed fixture (the 6.0e8-state ground
            state takes four minutes of GPU time: tests/golden/generate_config_fixtures.py) —, couplings
            rebuilt here, 4096 replicas

Each test goes model YAML -> exact ground state -> make_ising_model (GPU) -> sa.anneal (GPU)
through the package's reference-named entry points and compares with the CPU oracle bit for
bit."""
import numpy as np
import pytest

import oracle
from helpers import reference_route_ising

pytestmark = pytest.mark.gpu

_cache = {}


def _full_space(models, name):
    """(operator, ground state, exact IsingModel) of the whole basis."""
    from annealing_sign_problem_amd import common, operators

    if name not in _cache:
        op = operators.Operator.from_config(models[name])
        op.basis.build()
        energy, psi = op.ground_state()
        fn = common.ground_state_to_log_coeff_fn(psi, op.basis)
        model = common.make_ising_model(op.basis.states, op, log_psi_fn=fn)
        _cache[name] = (op, energy, psi, model)
    return _cache[name]


@pytest.mark.parametrize("name", ["j1j2_square_4x4", "heisenberg_kagome_16"])
def test_full_space_coupling_build_and_energy_identity(models, name):
    """J of the WHOLE basis equals the reference route (common.py:71-82,116-128,190-196 in
    numpy/scipy) bit for bit, and E(sign psi) = <psi|H|psi> to 1e-12 (common.py:757-760,
    experiments/full_hilbert_space.py:142-145)."""
    import scipy.sparse

    op, energy, psi, model = _full_space(models, name)
    assert model.size == 12870
    if name == "j1j2_square_4x4":
        j2 = np.asarray(models[name]["hamiltonian"]["terms"][1]["matrix"], dtype=float)
        assert j2[3, 3] == 1.0 and j2[0, 0] == 0.55  # the YAML's quirk is part of the config
    # the amplitudes exactly as make_ising_model forms them (common.py:178-181): exp of the
    # log-amplitudes, real part, L2-normalised (common.norm2: np.linalg.norm up to 10 000
    # elements, numpy's pairwise sum beyond, where BLAS would split the sum over its threads)
    from annealing_sign_problem_amd import common

    log_psi = common.ground_state_to_log_coeff_fn(psi, op.basis)(op.basis.states)
    amp = np.ascontiguousarray(np.exp(log_psi, dtype=np.complex128).real)
    amp /= common.norm2(amp)
    want = reference_route_ising(op, op.basis.states, amp)
    got = scipy.sparse.coo_matrix(model.ising_hamiltonian.exchange)
    assert np.array_equal(got.row, want.row) and np.array_equal(got.col, want.col)
    assert got.data.tobytes() == want.data.tobytes()
    h = model.ising_hamiltonian
    rayleigh = float(amp @ (op.to_sparse().real @ amp))
    e_signs = h.energy(model.initial_signs)
    assert abs(e_signs - rayleigh) <= 1e-12 * abs(rayleigh)
    assert abs(e_signs - energy) <= 1e-9 * abs(energy)  # eigensolver tolerance
    e_oracle = oracle.sa_energy(h.exchange, h.field, model.initial_signs)[0]
    assert abs(e_signs - e_oracle) <= 1e-12 * abs(e_signs)


def _chains_vs_oracle(h, seed, sweeps, reps, threads):
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    info = h.info()
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, sweeps)
    xs, es = sa.anneal_raw(h, seed, betas, reps)
    tracked = np.zeros(reps, np.int64)
    accepted = np.zeros(reps, np.uint64)
    _lib.check(_lib.load().asp_sa_last_stats(h.plan(), reps, _lib.ptr(tracked), _lib.ptr(accepted)))
    oxs, oes, otracked, oaccepted = oracle.sa_anneal(h.exchange, h.field, seed, betas, reps, 0, None,
                                                    info.energy_scale_exp, num_threads=threads)
    assert np.array_equal(accepted, oaccepted) and np.array_equal(tracked, otracked)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    return xs, es


def test_config1_j1j2_full_space_one_and_few_replicas(models):
    """BASELINE config 1: `make small` on j1j2_square_4x4, seed 435834 (Makefile:14): one
    replica (the CPU reference path's shape) and a few, automatic beta ladder."""
    _, _, _, model = _full_space(models, "j1j2_square_4x4")
    h = model.ising_hamiltonian
    xs, es = _chains_vs_oracle(h, 435834, 400, 1, 1)
    assert es[0] <= 0.5 * h.energy(model.initial_signs)  # (negative) well on the way to E0
    _chains_vs_oracle(h, 435834, 150, 5, 5)
    # the public entry point with only_best: the first minimum of the chains
    from annealing_sign_problem_amd import annealer as sa

    x, e = sa.anneal(h, seed=435834, number_sweeps=150, repetitions=5, sweep_order="colour")
    xs5, es5 = sa.anneal(h, seed=435834, number_sweeps=150, repetitions=5, only_best=False,
                       sweep_order="colour")
    k = int(np.argmin(es5))
    assert np.array_equal(x, xs5[k]) and e == es5[k]


def test_config2_kagome16_full_space_256_replicas(models):
    """BASELINE config 2: heisenberg_kagome_16 full space, 256 replicas on one GPU, fixed seed,
    every chain's signs and energy equal to the oracle's."""
    from annealing_sign_problem_amd import common

    _, _, _, model = _full_space(models, "heisenberg_kagome_16")
    xs, es = _chains_vs_oracle(model.ising_hamiltonian, 435834, 60, 256, 16)
    # 256 different Markov chains (many end in the same minimum), and the sign metric works
    assert np.unique(xs, axis=0).shape[0] > 16
    weights = np.ones(model.size) / model.size
    acc, overlap = common.compute_accuracy_and_overlap(xs[int(np.argmin(es))], model.initial_signs,
                                                       weights)
    assert 0.5 <= acc <= 1.0 and 0.0 <= overlap <= 1.0 + 1e-12


def _sampled_model(models, name, samples, order, cutoff, seed=435834):
    """Ground state of the model's whole sector (GPU), `samples` clusters grown as the driver
    grows them, each extended `order` times with the global cutoff; returns the models."""
    from annealing_sign_problem_amd import common, operators, sampled_components

    if name not in _cache:
        op = operators.Operator.from_config(models[name])
        op.basis.build()
        energy, psi = op.ground_state()
        _cache[name] = (op, energy, psi)
    op, energy, psi = _cache[name]
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        clusters = sampled_components.generate_clusters(op, psi, samples, 0.1, 50, 1000, 0.5)
    finally:
        np.random.set_state(state)
    fn = common.ground_state_to_log_coeff_fn(psi, op.basis)
    out = []
    for cluster in clusters:
        h = common.make_ising_model(cluster, op, log_psi_fn=fn)
        for _ in range(order):
            h = common.make_hamiltonian_extension(h, fn)
            h = common.sparsify_using_global_cutoff(h, cutoff, cluster)
        out.append((cluster, h))
    return op, energy, psi, out


def test_config3_kagome36_real_cluster_1024_replicas(models):
    """BASELINE config 3 on the real model: a sampled cluster of heisenberg_kagome_36 extended
    twice (`make kagome_36`'s order 2, cutoff 1e-6), 1024 replicas, every chain equal to the
    oracle's; the cluster model carries the energy identity of the construction."""
    op, energy, psi, built = _sampled_model(models, "heisenberg_kagome_36", 6, 2, 1e-6)
    assert op.basis.number_states == 31527894 and abs(energy / 144.0 + 0.43837653) < 1e-8
    cluster, model = min(built, key=lambda item: item[1].size)
    assert 2000 < model.size < 60000
    h = model.ising_hamiltonian
    j = h.exchange
    assert abs(j - j.T).max() == 0.0
    _chains_vs_oracle(h, 435834, 12, 1024, 16)


def test_config4_pyrochlore_real_cluster_cutoff_and_replica_shards(models):
    """BASELINE config 4 on the real model: heisenberg_pyrochlore_2x2x2, extension with
    CUTOFF = 2e-6; the chains of two shards (global replica ids 0..255 and 256..511, what two
    ranks run) are those of one 512-replica call and of the oracle."""
    from annealing_sign_problem_amd import annealer as sa

    op, energy, psi, built = _sampled_model(models, "heisenberg_pyrochlore_2x2x2", 4, 1, 2e-6)
    assert op.basis.number_states == 789438
    cluster, model = min(built, key=lambda item: item[1].size)
    h = model.ising_hamiltonian
    xs, es = _chains_vs_oracle(h, 435834, 20, 512, 16)
    info = h.info()
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 20)
    lo = sa.anneal_raw(h, 435834, betas, 256, 0)
    hi = sa.anneal_raw(h, 435834, betas, 256, 256)
    assert np.array_equal(np.concatenate([lo[0], hi[0]]), xs)
    assert np.concatenate([lo[1], hi[1]]).tobytes() == es.tobytes()


def test_config5_sk32_real_cluster_noise_4096_replicas(models):
    """BASELINE config 5 on the REAL model (reference Makefile:129-141, physical_systems/
    sk_32_1.yaml:1-4, NOISE = 0.79): the order-1 model of a sampled cluster, K = 5e4 spins with up
    to 256 couplings each.  The fixture holds the model's inputs (basis states, noisy
    cluster-normalised amplitudes); the couplings are rebuilt from them with asp_operator_ising
    and must be the ones of the run that made the fixture and the oracle's, the colour-ordered
    and the shuffled chains equal the oracle's bit for bit, and 4096 replicas satisfy the
    size-independent properties."""
    import hashlib

    import scipy.sparse

    from annealing_sign_problem_amd import annealer as sa
    from annealing_sign_problem_amd import operators
    from conftest import golden

    data = golden("config5_sk32_cluster.npz")
    spins, psi = data["spins"], data["psi"]
    k = spins.shape[0]
    assert k > 20000 and abs(np.linalg.norm(psi) - 1.0) < 1e-12 and float(data["noise"]) == 0.79
    op = operators.Operator.from_config(models["sk_32_1"])
    assert op.basis.number_spins == 32 and len(op.bond_table()[0]) == 496
    row, col, val = op.device().ising(spins, psi)
    assert val.shape[0] == int(data["nnz"])
    assert hashlib.sha256(row.tobytes() + col.tobytes() + val.tobytes()).hexdigest() == str(data["sha256"])
    qrow, qcol, qval = oracle.operator_ising(op.bond_table(), spins, psi)
    assert np.array_equal(row, qrow) and np.array_equal(col, qcol) and val.tobytes() == qval.tobytes()
    J = scipy.sparse.coo_matrix((val, (row, col)), shape=(k, k)).tocsr()
    h = sa.Hamiltonian(J, np.zeros(k))
    info = h.info()
    assert info.max_degree >= 200 and J.nnz / k > 30  # the dense rows of an all-to-all model
    _chains_vs_oracle(h, 435834, 12, 8, 8)
    # the reference annealer's visiting order on rows of 64 quads (they stream past the registers)
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 6)
    xs, es = sa.anneal_raw(h, 435834, betas, 4, 0, None, shuffled=True)
    oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h.field, 435834, betas, 4, 0, None,
                                               info.energy_scale_exp, num_threads=4)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    # 4096 replicas (the configuration's size): reported energies are those of the returned
    # configurations, no chain ends above its random start on average, shards are the one call
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 64)
    xs, es = sa.anneal_raw(h, 783494, betas, 4096)
    signs = [sa.bits_to_signs(x, k) for x in xs[:64]]
    again = np.array([v @ (J @ v) for v in signs])  # E = s^T J s (common.py:757-760), numpy
    assert np.allclose(again, es[:64], rtol=1e-12, atol=0)
    lo = sa.anneal_raw(h, 783494, betas, 64, 0)
    hi = sa.anneal_raw(h, 783494, betas, 64, 4032)
    assert np.array_equal(lo[0], xs[:64]) and np.array_equal(hi[0], xs[4032:])
    assert lo[1].tobytes() == es[:64].tobytes() and hi[1].tobytes() == es[4032:].tobytes()
    assert es.min() < es.mean() <= 0.0  # sign-problem couplings: annealed energies are negative

"""Symmetry-adapted bases on the GPU (csrc/operator_apply.hip: k_source_norms / k_symmetrise):
representatives, characters, norms, the symmetric batched_apply and the one-hop extension against
the numpy restatement (annealing_sign_problem_amd/symmetry.py, itself checked from first
principles in tests/test_symmetry.py), and the whole path on the basis of
heisenberg_kagome_18.yaml:4."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _norm2(x):
    """The norm make_ising_model divides by (common.norm2: np.linalg.norm up to 10 000 elements,
    numpy's pairwise sum beyond, where BLAS would split the sum over its threads)."""
    from annealing_sign_problem_amd import common

    return common.norm2(x)


def _random_representatives(op, count, seed):
    rng = np.random.default_rng(seed)
    n, w = op.basis.number_spins, op.basis.hamming_weight
    states = np.array([sum(1 << int(b) for b in rng.choice(n, size=w, replace=False))
                       for _ in range(count)], dtype=np.uint64)
    rep, _, norm = op.basis.group.state_info(states)
    return np.unique(rep[norm > 0])


@pytest.mark.parametrize("name", ["heisenberg_kagome_36", "heisenberg_pyrochlore_2x2x2",
                                  "heisenberg_kagome_18"])
def test_state_info_apply_and_extension_equal_the_numpy_restatement(models, name):
    from annealing_sign_problem_amd import common, operators

    op = operators.Operator.from_config(models[name])
    dev = op.device()
    assert not dev.unique_targets
    rng = np.random.default_rng(7)
    n, w = op.basis.number_spins, op.basis.hamming_weight
    anything = np.array([sum(1 << int(b) for b in rng.choice(n, size=w, replace=False))
                         for _ in range(300)] + [int("01" * (n // 2), 2)], dtype=np.uint64)
    rep, character, norm = dev.state_info(anything)
    want_rep, want_character, want_norm = op.basis.group.state_info(anything)
    assert np.array_equal(rep, want_rep) and norm.tobytes() == want_norm.tobytes()
    assert np.array_equal(character[want_norm > 0], want_character[want_norm > 0])

    keys = _random_representatives(op, 400, 11)
    other, coeffs, counts = dev.apply(keys)
    want_other, want_coeffs, want_counts = op.batched_apply(keys)
    assert np.array_equal(counts, want_counts) and np.array_equal(other, want_other[:, 0])
    assert coeffs.tobytes() == np.ascontiguousarray(want_coeffs.real).tobytes()
    assert not np.any(want_coeffs.imag)
    # the extension: sorted unique representatives of all targets
    assert np.array_equal(dev.extend(keys), np.unique(want_other[:, 0]))
    # and through the reference-named entry points: a cluster model in the symmetric basis
    log_psi = np.log(np.abs(np.sin(keys.astype(np.float64) * 1e-7)) + 0.1) + 0j
    model = common.make_ising_model(keys, op, log_psi=log_psi)
    j = model.ising_hamiltonian.exchange
    assert abs(j - j.T).max() == 0.0 and model.size == keys.shape[0]
    # the device build that keeps scipy's duplicate arithmetic (several connections of a row end
    # in the same representative) against the reference route in numpy / scipy, bit for bit
    import scipy.sparse

    from helpers import reference_route_ising

    psi = np.ascontiguousarray(np.exp(log_psi).real)
    psi /= _norm2(psi)
    want = reference_route_ising(op, keys, psi)
    row, col, val = dev.ising(keys, psi)
    assert np.array_equal(row, want.row) and np.array_equal(col, want.col)
    assert val.tobytes() == want.data.tobytes()
    got = scipy.sparse.coo_matrix(j)
    assert np.array_equal(got.row, want.row) and got.data.tobytes() == want.data.tobytes()
    bigger = common.make_hamiltonian_extension(model, lambda s: np.zeros(len(s), dtype=complex))
    assert np.array_equal(bigger.spins, np.unique(want_other[:, 0]))


def test_kagome_18_full_sector_energy_identity_and_chains(models):
    """heisenberg_kagome_18.yaml:4: the whole symmetric sector (24 310 representatives) through
    make_ising_model; E(sign psi) = <psi|H|psi> to 1e-12 (common.py:757-760) and annealing chains
    equal to the oracle's."""
    import oracle
    from annealing_sign_problem_amd import common, operators
    from annealing_sign_problem_amd import annealer as sa

    op = operators.Operator.from_config(models["heisenberg_kagome_18"])
    op.basis.build()
    energy, psi = op.ground_state()
    fn = common.ground_state_to_log_coeff_fn(psi, op.basis)
    model = common.make_ising_model(op.basis.states, op, log_psi_fn=fn)
    assert model.size == 24310
    h = model.ising_hamiltonian
    amp = np.exp(fn(op.basis.states)).real
    amp /= _norm2(amp)
    rayleigh = float(amp @ (op.to_sparse().real @ amp))
    e_signs = h.energy(model.initial_signs)
    assert abs(e_signs - rayleigh) <= 1e-12 * abs(rayleigh)
    assert abs(e_signs - energy) <= 1e-9 * abs(energy)
    info = h.info()
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 80)
    xs, es = sa.anneal_raw(h, 435834, betas, 8)
    oxs, oes, _, _ = oracle.sa_anneal(h.exchange, h.field, 435834, betas, 8, 0, None,
                                      info.energy_scale_exp, num_threads=8)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()


@pytest.mark.parametrize("name", ["heisenberg_kagome_36", "heisenberg_pyrochlore_2x2x2"])
def test_row_wise_symmetric_action_equals_the_entry_wise_kernels(models, name, monkeypatch):
    """k_symmetrise_rows (a wavefront per row: the images of the source state once, an XOR per
    target and group element) against k_source_norms + k_symmetrise (a full bit permutation per
    target and element), on the two production models (144 and 384 permutations, with inversion):
    same representatives, same coefficients, same extension, bit for bit; rows of every length
    incl. single states."""
    import time

    from annealing_sign_problem_amd import operators

    op = operators.Operator.from_config(models[name])
    dev = op.device()
    keys = _random_representatives(op, 6000, 3)
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ASP_SYMMETRISE_ROWS", mode)
        dev.apply(keys[:64])  # warm-up
        t0 = time.perf_counter()
        results[mode] = dev.apply(keys)
        seconds = time.perf_counter() - t0
        print("%s ASP_SYMMETRISE_ROWS=%s: apply(%d states, %d connections) %.1f ms (device %.2f ms)" % (
            name, mode, keys.shape[0], results[mode][0].shape[0], seconds * 1e3, dev.last_ms))
        results[mode] += (dev.extend(keys), dev.apply(keys[:1]), dev.apply(keys[17:20]))
    for a, b in zip(results["1"][:4], results["0"][:4]):
        assert a.dtype == b.dtype and a.tobytes() == b.tobytes()
    for a, b in zip(results["1"][4] + results["1"][5], results["0"][4] + results["0"][5]):
        assert a.tobytes() == b.tobytes()

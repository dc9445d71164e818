"""GPU parity: the annealing sweep through the C ABI vs the CPU oracle (bit-exact
packed spins, energies, tracked fixed-point energies and accepted-flip counts)."""
import ctypes
import os

import numpy as np
import pytest
import scipy.sparse

import oracle

pytestmark = pytest.mark.gpu


def _stats(h, count):
    from annealing_sign_problem_amd import _lib

    tracked = np.zeros(count, np.int64)
    accepted = np.zeros(count, np.uint64)
    _lib.check(_lib.load().asp_sa_last_stats(h.plan(), count, _lib.ptr(tracked), _lib.ptr(accepted)))
    return tracked, accepted


def _set_launch(h, m, threads):
    from annealing_sign_problem_amd import _lib

    _lib.check(_lib.load().asp_sa_set_launch(h.plan(), m, threads))


def _compare(J, field, seed, betas, reps, offset=0, x0=None, m=0, threads=0):
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    h = sa.Hamiltonian(J, field)
    _set_launch(h, m, threads)
    xs, es = sa.anneal_raw(h, seed, betas, reps, offset, x0)
    tracked, accepted = _stats(h, reps)
    lib = _lib.load()
    h.layout_used = lib.asp_sa_last_layout(h.plan())
    if h.layout_used == 2:
        # the word-per-position layout ran (four replicas per workgroup): the byte
        # layout must give the very same chains
        _lib.check(lib.asp_sa_set_wide(h.plan(), 0))
        xs_b, es_b = sa.anneal_raw(h, seed, betas, reps, offset, x0)
        tracked_b, accepted_b = _stats(h, reps)
        assert lib.asp_sa_last_layout(h.plan()) == 0
        assert np.array_equal(xs, xs_b) and es.tobytes() == es_b.tobytes()
        assert np.array_equal(tracked, tracked_b) and np.array_equal(accepted, accepted_b)
        _lib.check(lib.asp_sa_set_wide(h.plan(), 1))
    S = h.info().energy_scale_exp
    oxs, oes, otracked, oaccepted = oracle.sa_anneal(J, field, seed, betas, reps, offset, x0, S,
                                                    num_threads=8)
    assert np.array_equal(accepted, oaccepted), "accepted-flip counts differ"
    assert np.array_equal(tracked, otracked), "tracked energies differ"
    assert np.array_equal(xs, oxs), "best configurations differ"
    assert es.tobytes() == oes.tobytes(), "energies differ"
    return h, xs, es


def _planted(n, seed, **kw):
    from annealing_sign_problem_amd import synthetic

    return synthetic.planted_cluster(n, seed=seed, **kw)


@pytest.mark.parametrize("m", [1, 2, 4, 8])
def test_sweep_bit_exact_every_group_width(m):
    from annealing_sign_problem_amd import _lib

    J, h, _ = _planted(1500, 11)
    betas = np.geomspace(0.5, 2e4, 40)
    ham, _, _ = _compare(J, h, 12345, betas, 16, m=m, threads=256)
    # layouts: words for 4 replicas per workgroup at this size, bytes otherwise
    assert ham.layout_used == (2 if m == 4 else 0)


def test_wide_layout_limits_and_initial_configuration():
    """The word layout stops at ~4e4 spins (LDS) and is never used by the descent kernel; with
    a given x0 and a ragged last group it still follows the oracle."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    lib = _lib.load()
    J, h, planted = _planted(36000, 21)
    x0 = sa.signs_to_bits(np.where(np.random.default_rng(3).random(36000) < 0.5, 1.0, -1.0))
    betas = np.geomspace(2.0, 1e6, 12)
    ham, _, _ = _compare(J, h, 5, betas, 7, offset=3, x0=x0, m=4, threads=512)
    assert ham.layout_used == 2
    big = sa.Hamiltonian(*_planted(45000, 22)[:2])
    _set_launch(big, 4, 512)
    sa.anneal_raw(big, 5, betas[:3], 4)
    assert lib.asp_sa_last_layout(big.plan()) == 0
    sa.greedy_solve(ham)
    assert lib.asp_sa_last_layout(ham.plan()) == 0


@pytest.mark.parametrize("threads", [64, 192, 1024])
def test_sweep_independent_of_workgroup_size(threads):
    J, h, _ = _planted(3000, 12, mean_degree=9.0, max_degree=20)
    betas = np.geomspace(1.0, 1e5, 25)
    _compare(J, h, 99, betas, 8, m=4, threads=threads)


def test_sweep_auto_launch_ragged_replicas_and_offset():
    J, h, _ = _planted(700, 13)
    betas = np.geomspace(0.3, 3e4, 30)
    # 13 chains starting at global id 6: groups straddle the Philox 4-replica blocks
    hh, xs, es = _compare(J, h, 2**63 + 17, betas, 13, offset=6)
    # the same global ids computed as part of a bigger run give the same chains
    from annealing_sign_problem_amd import annealer as sa
    xs_all, es_all = sa.anneal_raw(hh, 2**63 + 17, betas, 32, 0)
    assert np.array_equal(xs_all[6:19], xs) and es_all[6:19].tobytes() == es.tobytes()


def test_sweep_with_field_nonsymmetric_J_and_x0():
    rng = np.random.default_rng(3)
    n = 900
    J = scipy.sparse.random(n, n, density=0.01, random_state=4, format="csr", dtype=np.float64)
    J.data = rng.normal(size=J.data.shape)          # not symmetric, has a diagonal
    J = (J + scipy.sparse.diags(rng.normal(size=n))).tocsr()
    field = rng.normal(size=n) * 0.3
    x0 = rng.integers(0, 2**63, size=(n + 63) // 64, dtype=np.uint64)
    _compare(J, field, 7, np.geomspace(0.2, 50.0, 35), 9, x0=x0, m=2, threads=128)


def test_sweep_dense_rows_sk_like():
    from annealing_sign_problem_amd import synthetic

    J, h = synthetic.sk_cluster(600, degree=120, seed=5)
    _compare(J, h, 5, np.geomspace(1.0, 1e4, 12), 8, m=8, threads=512)


def test_sweep_degenerate_shapes():
    # isolated spins only (no couplings): every proposal has dE = +-2h
    n = 130
    J = scipy.sparse.csr_matrix((n, n), dtype=np.float64)
    _compare(J, np.linspace(-1, 1, n), 3, np.geomspace(0.1, 10, 10), 5)
    # one spin
    _compare(scipy.sparse.csr_matrix(np.array([[2.5]])), np.array([0.7]), 3, np.ones(4), 3)
    # zero sweeps: best = initial configuration
    J, h, _ = _planted(200, 14)
    _compare(J, h, 1, np.zeros(0), 4)


def test_energy_matches_oracle_and_numpy():
    from annealing_sign_problem_amd import annealer as sa

    J, h, _ = _planted(5000, 15)
    h = np.random.default_rng(1).normal(size=5000) * 1e-3
    ham = sa.Hamiltonian(J, h)
    rng = np.random.default_rng(2)
    xs = rng.integers(0, 2**63, size=(7, (5000 + 63) // 64), dtype=np.uint64)
    xs[:, -1] &= np.uint64((1 << (5000 % 64)) - 1)
    got = ham.energies(xs)
    assert got.tobytes() == oracle.sa_energy(J, h, xs).tobytes()
    for x, e in zip(xs, got):
        s = sa.bits_to_signs(x, 5000)
        ref = s @ (J @ s) + h @ s
        assert abs(e - ref) <= 1e-12 * max(abs(ref), 1e-300)


def test_anneal_api_and_quality_on_planted_ferromagnet():
    """Unfrustrated planted instance: SA must reach the planted ground state."""
    from annealing_sign_problem_amd import annealer as sa

    J, h, planted = _planted(2000, 16, frustrated_fraction=0.0, diagonal_range=0)
    ham = sa.Hamiltonian(J, h)
    x, e = sa.anneal(ham, seed=12345, number_sweeps=400, repetitions=16, only_best=True, sweep_order="colour")
    assert x.dtype == np.uint64 and x.shape == ((2000 + 63) // 64,)
    s = sa.bits_to_signs(x, 2000)
    e_planted = planted @ (J @ planted)
    assert e <= e_planted + 1e-12 * abs(e_planted)
    xs, es = sa.anneal(ham, seed=12345, number_sweeps=400, repetitions=16, only_best=False,
                     sweep_order="colour")
    assert xs.shape == (16, 32) and es.shape == (16,)
    assert float(es.min()) == e and np.array_equal(xs[int(np.argmin(es))], x)
    # determinism
    x2, e2 = sa.anneal(ham, seed=12345, number_sweeps=400, repetitions=16, sweep_order="colour")
    assert np.array_equal(x, x2) and e == e2
    assert abs(ham.energy(x) - s @ (J @ s)) <= 1e-12 * abs(e)


def test_kagome16_full_space_signs(models):
    """BASELINE config 'heisenberg_kagome_16: full-space SA': K = 12870, the real
    coupling structure; chains vs oracle bit for bit, and SA recovers the exact
    ground-state energy <psi|H|psi> = E(sign psi) within 1e-12."""
    from annealing_sign_problem_amd import annealer as sa, common, operators

    op = operators.Operator.from_config(models["heisenberg_kagome_16"])
    op.basis.build()
    e0, psi = op.ground_state()
    model = common.make_ising_model(op.basis.states, op, log_psi_fn=common.ground_state_to_log_coeff_fn(psi, op.basis))
    ham = model.ising_hamiltonian
    assert abs(ham.energy(model.initial_signs) - e0) <= 1e-12 * abs(e0)  # full_hilbert_space.py:142-145
    info = ham.info()
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 60)
    _compare(ham.exchange, ham.field, 435834, betas, 6, m=2, threads=1024)


def test_full_size_properties_kagome36_sized():
    """K = 1e5 (kagome_36-sized), 64 chains: too big for the oracle in seconds, so
    check size-independent properties: reported energies equal E(x) recomputed
    from the returned bits (numpy, 1e-12), different launch geometries agree
    bit for bit, chains differ from each other, energies decrease with sweeps."""
    from annealing_sign_problem_amd import annealer as sa

    J, h, planted = _planted(100000, 17)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    betas = sa.make_schedule(info.beta0_auto, min(info.beta1_auto, 1e8), 24)
    _set_launch(ham, 8, 1024)
    xs, es = sa.anneal_raw(ham, 42, betas, 64)
    _set_launch(ham, 1, 512)
    xs1, es1 = sa.anneal_raw(ham, 42, betas, 64)
    assert np.array_equal(xs, xs1) and es.tobytes() == es1.tobytes()
    for r in [0, 31, 63]:
        s = sa.bits_to_signs(xs[r], 100000)
        ref = s @ (J @ s)
        assert abs(es[r] - ref) <= 1e-12 * abs(ref)
    assert len({x.tobytes() for x in xs}) == 64
    xs_short, es_short = sa.anneal_raw(ham, 42, betas[:2], 64)
    assert es.mean() < es_short.mean()


def test_greedy_solve_matches_oracle_and_improves():
    """sa.greedy_solve (common.py:250): host cluster merging + device strict-descent
    relaxation vs the oracle, bit for bit; and the drop-in call through solve_ising_model."""
    from annealing_sign_problem_amd import annealer as sa, common, synthetic

    rng = np.random.default_rng(4)
    cases = []
    for n, seed in [(1, 1), (700, 2), (6000, 3)]:
        J, h, _ = _planted(n, seed, mean_degree=min(14.0, max(n / 2, 1.0)))
        cases.append((J, h))
    J, h, _ = _planted(2500, 5)
    cases.append((J, rng.normal(size=2500) * 1e-3))
    Jsk, hsk = synthetic.sk_cluster(500, degree=100, seed=6)
    cases.append((Jsk, hsk))
    for J, h in cases:
        ham = sa.Hamiltonian(J, h)
        x, e = sa.greedy_solve(ham)
        ox, oe = oracle.greedy_solve(J, h)
        assert np.array_equal(x, ox) and e == oe
        x_tree, e_tree = oracle.greedy_solve(J, h, relax=False)
        assert e <= e_tree
        n = J.shape[0]
        s = sa.bits_to_signs(x, n)
        ref = s @ (J @ s) + h @ s
        assert abs(e - ref) <= 1e-12 * max(abs(ref), 1e-300)
        # local minimum: no single flip lowers the energy
        A = (J + J.T).tocsr()
        A.setdiag(0)
        de = -2.0 * s * (A @ s + h)
        assert np.all(de >= -1e-12 * np.abs(de).max())
    model = common.IsingModel(np.arange(J.shape[0], dtype=np.uint64), None, sa.Hamiltonian(J, h), None)
    xg = common.solve_ising_model(model, mode="greedy")
    assert np.array_equal(xg, oracle.greedy_solve(J, h)[0])


def _set_packed(h, packed):
    from annealing_sign_problem_amd import _lib

    _lib.check(_lib.load().asp_sa_set_packed(h.plan(), int(packed)))


@pytest.mark.parametrize("threads,where", [(64, 1), (256, 1), (1024, 1), (64, 2), (512, 2)])
def test_bit_packed_layout_bit_exact(threads, where):
    """One bit per spin, flips by wavefront ballot, forced on a small instance: same chains as
    the oracle.  where = 1: words in LDS (the layout used beyond ~1.4e5 spins); 2: words in HBM
    (the layout used beyond ~1.3e6 spins, when not even the bits fit the LDS)."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    J, h, _ = _planted(1500, 21)
    field = np.random.default_rng(5).normal(size=1500) * 0.01
    betas = np.geomspace(0.5, 2e4, 40)
    ham = sa.Hamiltonian(J, field)
    _set_packed(ham, where)
    _set_launch(ham, 0, threads)
    xs, es = sa.anneal_raw(ham, 4242, betas, 9, 3)
    assert _lib.load().asp_sa_last_layout(ham.plan()) == (1 if where == 1 else 3)
    tracked, accepted = _stats(ham, 9)
    oxs, oes, otr, oacc = oracle.sa_anneal(J, field, 4242, betas, 9, 3, None,
                                           ham.info().energy_scale_exp, num_threads=8)
    assert np.array_equal(accepted, oacc) and np.array_equal(tracked, otr)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    x0 = np.random.default_rng(6).integers(0, 2**63, size=(1500 + 63) // 64, dtype=np.uint64)
    xs, es = sa.anneal_raw(ham, 1, betas[:10], 2, 0, x0)
    oxs, oes, _, _ = oracle.sa_anneal(J, field, 1, betas[:10], 2, 0, x0, ham.info().energy_scale_exp)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()


def test_beyond_lds_capacity_spins_in_hbm():
    """K = 1.6e6 does not fit the LDS even at one bit per spin: the sign words move to HBM by
    themselves.  Chains against a short oracle run, energies against numpy, and the greedy
    solver (descent kernel in the same layout) against its oracle."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    n = 1600000
    J, h, _ = _planted(n, 23, mean_degree=3.0, max_degree=8)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    betas = sa.make_schedule(info.beta0_auto, min(info.beta1_auto, 1e6), 3)
    xs, es = sa.anneal_raw(ham, 78, betas, 3)
    assert _lib.load().asp_sa_last_layout(ham.plan()) == 3
    for r in range(3):
        s = sa.bits_to_signs(xs[r], n)
        ref = s @ (J @ s)
        assert abs(es[r] - ref) <= 1e-12 * abs(ref)
    oxs, oes, _, _ = oracle.sa_anneal(J, h, 78, betas, 2, 0, None, info.energy_scale_exp,
                                      num_threads=2)
    assert np.array_equal(xs[:2], oxs) and es[:2].tobytes() == oes.tobytes()
    xg, eg = sa.greedy_solve(ham)
    oxg, oeg = oracle.greedy_solve(J, h)
    assert np.array_equal(xg, oxg) and eg == oeg


def test_beyond_byte_layout_capacity():
    """K = 3e5 does not fit one byte per spin in LDS: the bit-packed kernel takes over by
    itself; energies check against numpy, a short oracle run checks the chains."""
    from annealing_sign_problem_amd import annealer as sa

    J, h, _ = _planted(300000, 22, mean_degree=6.0, max_degree=14)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    betas = sa.make_schedule(info.beta0_auto, min(info.beta1_auto, 1e6), 3)
    xs, es = sa.anneal_raw(ham, 77, betas, 4)
    for r in range(4):
        s = sa.bits_to_signs(xs[r], 300000)
        ref = s @ (J @ s)
        assert abs(es[r] - ref) <= 1e-12 * abs(ref)
    oxs, oes, _, _ = oracle.sa_anneal(J, h, 77, betas, 2, 0, None, info.energy_scale_exp, num_threads=2)
    assert np.array_equal(xs[:2], oxs) and es[:2].tobytes() == oes.tobytes()


def _set_cache(h, enable):
    from annealing_sign_problem_amd import _lib

    _lib.check(_lib.load().asp_sa_set_field_cache(h.plan(), int(enable)))


@pytest.mark.parametrize("m,threads", [(1, 128), (4, 512), (8, 1024)])
def test_field_cache_in_frozen_tail_is_exact(m, threads):
    """A ladder with a long frozen tail drives the workgroups into cached mode (local fields
    kept in HBM, re-evaluated only after a neighbour flipped).  Chains must equal the oracle's
    and the cache-off run bit for bit, including chains that still flip occasionally."""
    from annealing_sign_problem_amd import annealer as sa

    J, h, _ = _planted(4000, 31)
    field = np.random.default_rng(7).normal(size=4000) * 1e-4
    betas = np.concatenate([np.geomspace(0.5, 3e3, 25), np.geomspace(3e3, 1e11, 150)])
    ham = sa.Hamiltonian(J, field)
    _set_launch(ham, m, threads)
    xs, es = sa.anneal_raw(ham, 2024, betas, 16, 5)
    tracked, accepted = _stats(ham, 16)
    _set_cache(ham, False)
    xs_off, es_off = sa.anneal_raw(ham, 2024, betas, 16, 5)
    tracked_off, accepted_off = _stats(ham, 16)
    assert np.array_equal(xs, xs_off) and es.tobytes() == es_off.tobytes()
    assert np.array_equal(tracked, tracked_off) and np.array_equal(accepted, accepted_off)
    oxs, oes, otr, oacc = oracle.sa_anneal(J, field, 2024, betas, 16, 5, None,
                                           ham.info().energy_scale_exp, num_threads=8)
    assert np.array_equal(accepted, oacc) and np.array_equal(tracked, otr)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    # the tail really is frozen for most chains but not dead: some flips happen after sweep 100
    _, _, _, acc_head = oracle.sa_anneal(J, field, 2024, betas[:100], 16, 5, None,
                                         ham.info().energy_scale_exp, num_threads=8)
    late = oacc.astype(np.int64) - acc_head.astype(np.int64)
    assert late.sum() > 0 and late.max() < 4000


@pytest.mark.parametrize("m,threads", [(2, 256), (4, 1024)])
def test_field_cache_survives_reheating(m, threads):
    """Blocks whose proposals were all certain rejections are skipped while their fields are
    clean — valid only for non-decreasing beta.  A schedule that freezes, reheats and freezes
    again must still follow the oracle bit for bit."""
    from annealing_sign_problem_amd import annealer as sa

    J, h, _ = _planted(3000, 5)
    field = np.random.default_rng(11).normal(size=3000) * 1e-4
    freeze = np.geomspace(1.0, 1e12, 60)
    betas = np.concatenate([freeze, np.full(10, 1e12), np.geomspace(1e12, 2e2, 6),
                            np.geomspace(2e2, 1e12, 40), [1e3, 1e12, 1e12, 1e5, 1e12]])
    ham = sa.Hamiltonian(J, field)
    _set_launch(ham, m, threads)
    xs, es = sa.anneal_raw(ham, 99, betas, 8, 3)
    tracked, accepted = _stats(ham, 8)
    oxs, oes, otr, oacc = oracle.sa_anneal(J, field, 99, betas, 8, 3, None,
                                           ham.info().energy_scale_exp, num_threads=8)
    assert np.array_equal(accepted, oacc) and np.array_equal(tracked, otr)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    # the reheating really unfreezes chains that were dead at the end of the first freeze
    _, _, _, acc_a = oracle.sa_anneal(J, field, 99, betas[:70], 8, 3, None,
                                      ham.info().energy_scale_exp, num_threads=8)
    _, _, _, acc_b = oracle.sa_anneal(J, field, 99, betas[:60], 8, 3, None,
                                      ham.info().energy_scale_exp, num_threads=8)
    assert np.array_equal(acc_a, acc_b)            # sweeps 60..69: nothing moves
    assert (oacc.astype(np.int64) - acc_a.astype(np.int64)).min() > 100


def test_full_size_properties_sk32_sized():
    """BASELINE config 5 (sk_32_1 + NOISE 0.79 shape): K = 8192, 256 neighbours per spin.  The
    first chains are checked against the oracle on a short ladder; at full width (512 chains,
    every group size) the size-independent properties: energies equal E(x) recomputed with
    numpy (1e-12), launch geometries agree bit for bit, any shard of the chains equals the same
    chains of the full run (what the multi-GPU split relies on)."""
    from annealing_sign_problem_amd import annealer as sa, synthetic

    J, h = synthetic.sk_cluster(8192, degree=256, seed=783494)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    assert info.max_degree >= 200
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 12)
    xs, es = sa.anneal_raw(ham, 783494, betas, 512)
    oxs, oes, _, _ = oracle.sa_anneal(J, h, 783494, betas[:4], 8, 0, None, info.energy_scale_exp,
                                      num_threads=8)
    xs4, es4 = sa.anneal_raw(ham, 783494, betas[:4], 8)
    assert np.array_equal(xs4, oxs) and es4.tobytes() == oes.tobytes()
    _set_launch(ham, 1, 256)
    xs1, es1 = sa.anneal_raw(ham, 783494, betas, 512)
    assert np.array_equal(xs, xs1) and es.tobytes() == es1.tobytes()
    _set_launch(ham, 0, 0)
    part_x, part_e = sa.anneal_raw(ham, 783494, betas, 100, 300)   # chains 300..399
    assert np.array_equal(part_x, xs[300:400]) and part_e.tobytes() == es[300:400].tobytes()
    Js = J.tocsr()
    for r in [0, 255, 511]:
        s = sa.bits_to_signs(xs[r], 8192)
        ref = s @ (Js @ s) + h @ s
        assert abs(es[r] - ref) <= 1e-12 * abs(ref)
    assert len({x.tobytes() for x in xs}) == 512


def test_full_size_properties_pyrochlore_sized_with_cutoff():
    """BASELINE config 4 shape: a pyrochlore-sized cluster (mean degree 40, capped at 49),
    sparsified with CUTOFF = 2e-6 around a frozen core (common.py:634-692), then annealed:
    the kept component holds every frozen spin, its couplings are the un-pruned block
    (common.py:674), energies equal numpy's, launch geometries agree."""
    from annealing_sign_problem_amd import annealer as sa, common, synthetic

    J, h, planted = synthetic.planted_cluster(40000, mean_degree=40.0, max_degree=49, seed=674385)
    spins = np.arange(40000, dtype=np.uint64) * np.uint64(3) + np.uint64(7)   # sorted unique keys
    model = common.IsingModel(spins, None, sa.Hamiltonian(J, h),
                              sa.signs_to_bits(np.ones(40000)))
    # frozen core: spin 0 and its strongest chain of neighbours
    csr = J.tocsr()
    core, at = [0], 0
    for _ in range(30):
        lo, hi = csr.indptr[at], csr.indptr[at + 1]
        cand = [(abs(v), j) for v, j in zip(csr.data[lo:hi], csr.indices[lo:hi]) if j not in core]
        if not cand:
            break
        at = max(cand)[1]
        core.append(int(at))
    frozen = spins[np.sort(np.array(core))]
    small = common.sparsify_using_global_cutoff(model, 2e-6, frozen)
    keep = np.searchsorted(spins, small.spins)
    assert np.all(np.isin(frozen, small.spins)) and 100 < small.size <= 40000
    block = csr[keep][:, keep]
    assert (abs(small.ising_hamiltonian.exchange - block)).nnz == 0
    ham = small.ising_hamiltonian
    info = ham.info()
    betas = sa.make_schedule(info.beta0_auto, min(info.beta1_auto, 1e9), 16)
    xs, es = sa.anneal_raw(ham, 674385, betas, 128)
    _set_launch(ham, 2, 512)
    xs2, es2 = sa.anneal_raw(ham, 674385, betas, 128)
    assert np.array_equal(xs, xs2) and es.tobytes() == es2.tobytes()
    for r in [0, 64, 127]:
        s = sa.bits_to_signs(xs[r], small.size)
        ref = s @ (block @ s) + ham.field @ s
        assert abs(es[r] - ref) <= 1e-12 * abs(ref)


@pytest.mark.parametrize("m,threads", [(1, 256), (4, 512), (8, 1024)])
def test_energy_traces_match_oracle(m, threads):
    """asp_sa_anneal_trace: the tracked energy after every sweep, as integers, equals the
    oracle's; the legacy 3-tuple API built on it returns energies consistent with numpy."""
    from annealing_sign_problem_amd import annealer as sa

    J, h, _ = _planted(1200, 41)
    field = np.random.default_rng(4).normal(size=1200) * 0.02
    betas = np.geomspace(0.3, 5e3, 30)
    ham = sa.Hamiltonian(J, field)
    _set_launch(ham, m, threads)
    xs, es, trace = sa.anneal_trace_raw(ham, 31337, betas, 11, 2)
    S = ham.info().energy_scale_exp
    oxs, oes, otrace = oracle.sa_anneal_trace(J, field, 31337, betas, 11, 2, None, S, num_threads=8)
    assert np.array_equal(trace, otrace)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    assert np.all(trace[:, 0] == 0) and trace.min() < 0
    _set_launch(ham, 0, 0)
    x0 = sa.signs_to_bits(np.ones(1200))
    x, e_current, e_best = sa.anneal_with_traces(ham, x0, seed=5, number_sweeps=40, beta0=0.3,
                                                 beta1=5e3)
    assert e_current.shape == (41,) and e_best.shape == (41,)
    s0 = np.ones(1200)
    e_start = s0 @ (J @ s0) + field @ s0
    assert abs(e_current[0] - e_start) <= 1e-9 * abs(e_start)          # anchored consistently
    assert abs(e_best[-1] - ham.energy(x)) == 0 and np.all(np.diff(e_best) <= 0)
    assert np.all(e_best <= e_current + 1e-12)


def _set_team(h, team):
    from annealing_sign_problem_amd import _lib

    _lib.check(_lib.load().asp_sa_set_team(h.plan(), int(team)))


@pytest.mark.parametrize("team", [2, 4, 8])
def test_team_sweep_bit_exact(team):
    """One chain spread over a team of workgroups (flip words exchanged per colour step behind a
    device-scope barrier): same chains, tracked energies and flip counts as the oracle — with
    random start, with x0 and a replica offset."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    lib = _lib.load()
    J, h, _ = _planted(6000, 51)
    field = np.random.default_rng(9).normal(size=6000) * 0.01
    betas = np.geomspace(0.5, 5e4, 30)
    ham = sa.Hamiltonian(J, field)
    _set_team(ham, team)
    xs, es = sa.anneal_raw(ham, 2718, betas, 5, 7)
    assert lib.asp_sa_last_layout(ham.plan()) == 4
    tracked, accepted = _stats(ham, 5)
    S = ham.info().energy_scale_exp
    oxs, oes, otr, oacc = oracle.sa_anneal(J, field, 2718, betas, 5, 7, None, S, num_threads=8)
    assert np.array_equal(accepted, oacc) and np.array_equal(tracked, otr)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    x0 = sa.signs_to_bits(np.where(np.random.default_rng(1).random(6000) < 0.5, 1.0, -1.0))
    xs, es = sa.anneal_raw(ham, 3, betas[:12], 3, 0, x0)
    oxs, oes, _, _ = oracle.sa_anneal(J, field, 3, betas[:12], 3, 0, x0, S, num_threads=4)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    # switched off: the ordinary kernel, same answer
    _set_team(ham, 0)
    xs2, es2 = sa.anneal_raw(ham, 3, betas[:12], 3, 0, x0)
    assert lib.asp_sa_last_layout(ham.plan()) != 4
    assert np.array_equal(xs2, oxs) and es2.tobytes() == oes.tobytes()


def test_greedy_relaxation_in_team_mode_matches_oracle():
    """The greedy solver's descent sweeps on one chain also run as a team (forced here on a
    mid-sized cluster, automatic on large ones): same result as the oracle and as team-off."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    J, h, _ = _planted(9000, 61, frustrated_fraction=0.2)
    field = np.random.default_rng(2).normal(size=9000) * 1e-3
    ham = sa.Hamiltonian(J, field)
    ox, oe = oracle.greedy_solve(J, field)
    for team in (4, 0):
        _set_team(ham, team)
        x, e = sa.greedy_solve(ham)
        assert _lib.load().asp_sa_last_layout(ham.plan()) == (4 if team else 0)
        assert np.array_equal(x, ox) and e == oe


def test_team_sweep_chosen_automatically_for_few_chains_on_a_large_cluster():
    """64 chains at K = 1e5 (the reference's default repetitions on a kagome_36-sized cluster):
    the launcher forms teams by itself; energies check against numpy, chains against the
    single-workgroup kernel."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    lib = _lib.load()
    J, h, _ = _planted(100000, 17)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    betas = sa.make_schedule(info.beta0_auto, min(info.beta1_auto, 1e8), 20)
    xs, es = sa.anneal_raw(ham, 42, betas, 64)
    assert lib.asp_sa_last_layout(ham.plan()) == 4
    team_ms = lib.asp_sa_last_sweep_ms(ham.plan())
    _set_team(ham, 0)
    xs1, es1 = sa.anneal_raw(ham, 42, betas, 64)
    solo_ms = lib.asp_sa_last_sweep_ms(ham.plan())
    assert np.array_equal(xs, xs1) and es.tobytes() == es1.tobytes()
    for r in [0, 63]:
        s = sa.bits_to_signs(xs[r], 100000)
        ref = s @ (J @ s)
        assert abs(es[r] - ref) <= 1e-12 * abs(ref)
    print("team %.2f ms, single workgroup %.2f ms" % (team_ms, solo_ms))


def test_team_sweeps_from_concurrent_host_threads():
    """sampled_components --jobs runs several Hamiltonians at once from host threads: their
    cooperative team launches must take turns (two resident at once could starve each other)
    and still return the single-threaded results."""
    from concurrent.futures import ThreadPoolExecutor

    from annealing_sign_problem_amd import annealer as sa

    problems = []
    for k in range(6):
        J, h, _ = _planted(40000 + 5000 * k, 70 + k, mean_degree=8.0)
        problems.append(sa.Hamiltonian(J, h))

    def solve(ham):
        info = ham.info()
        betas = sa.make_schedule(info.beta0_auto, min(info.beta1_auto, 1e7), 40)
        return sa.anneal_raw(ham, 5, betas, 8)

    serial = [solve(h) for h in problems]
    with ThreadPoolExecutor(max_workers=6) as pool:
        threaded = list(pool.map(solve, problems))
    for (xs, es), (xt, et) in zip(serial, threaded):
        assert np.array_equal(xs, xt) and es.tobytes() == et.tobytes()


@pytest.mark.parametrize("team", [2, 8])
def test_team_sweep_tracks_untouched_blocks_in_the_frozen_tail(team):
    """In the frozen tail a team skips blocks whose neighbourhood has not moved and whose
    proposals were all certain rejections (tracking switches on when a sweep flips few spins,
    also learning about the other members' flips).  A ladder that freezes, reheats and freezes
    again must follow the oracle bit for bit, with and without the tracking."""
    from annealing_sign_problem_amd import annealer as sa

    J, h, _ = _planted(5000, 81)
    field = np.random.default_rng(12).normal(size=5000) * 1e-4
    betas = np.concatenate([np.geomspace(1.0, 1e12, 70), np.full(10, 1e12), np.geomspace(1e12, 3e2, 5),
                            np.geomspace(3e2, 1e12, 50), [1e3, 1e12, 1e12, 1e5, 1e12]])
    ham = sa.Hamiltonian(J, field)
    _set_team(ham, team)
    xs, es = sa.anneal_raw(ham, 123, betas, 4, 1)
    tracked, accepted = _stats(ham, 4)
    S = ham.info().energy_scale_exp
    oxs, oes, otr, oacc = oracle.sa_anneal(J, field, 123, betas, 4, 1, None, S, num_threads=4)
    assert np.array_equal(accepted, oacc) and np.array_equal(tracked, otr)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    _set_cache(ham, False)   # the same switch turns the tracking off
    xs2, es2 = sa.anneal_raw(ham, 123, betas, 4, 1)
    assert np.array_equal(xs2, oxs) and es2.tobytes() == oes.tobytes()
    # the chains really freeze and thaw: flips during the reheating, none in sweeps 70..79
    _, _, _, acc80 = oracle.sa_anneal(J, field, 123, betas[:80], 4, 1, None, S, num_threads=4)
    _, _, _, acc70 = oracle.sa_anneal(J, field, 123, betas[:70], 4, 1, None, S, num_threads=4)
    assert np.array_equal(acc80, acc70) and (oacc.astype(np.int64) - acc80.astype(np.int64)).min() > 100


def test_randomised_configurations_match_oracle():
    """Two hundred random small problems x launch shapes x spin layouts x team sizes x start modes:
    every combination must reproduce the oracle's chains, tracked energies and flip counts."""
    from annealing_sign_problem_amd import _lib, synthetic
    from annealing_sign_problem_amd import annealer as sa

    lib = _lib.load()
    rng = np.random.default_rng(20261004)
    for case in range(200):
        n = int(rng.choice([1, 2, 63, 64, 65, 128, 500, 1024, 1500, 2500]))
        degree = float(rng.choice([0.0, 1.0, 3.0, 9.0, 24.0]))
        if degree == 0.0 or n < 4:
            J = scipy.sparse.diags(rng.normal(size=n)).tocsr()
            h = rng.normal(size=n)
        else:
            J, h, _ = synthetic.planted_cluster(n, seed=int(rng.integers(1 << 30)),
                                                mean_degree=min(degree, n / 3),
                                                frustrated_fraction=float(rng.random() * 0.5))
            h = h + rng.normal(size=n) * float(rng.choice([0.0, 1e-3, 1.0]))
        sweeps = int(rng.choice([0, 1, 7, 30]))
        betas = np.geomspace(0.2, float(rng.choice([5.0, 1e4, 1e12])), sweeps) if sweeps else np.zeros(0)
        if sweeps > 3 and rng.random() < 0.3:
            betas[sweeps // 2] = 0.0          # an infinite-temperature sweep in the middle
        reps = int(rng.choice([1, 3, 8, 17]))
        offset = int(rng.choice([0, 1, 5, 1000]))
        x0 = None
        if rng.random() < 0.4:
            x0 = sa.signs_to_bits(np.where(rng.random(n) < 0.5, 1.0, -1.0))
        ham = sa.Hamiltonian(J, h)
        mode = int(rng.integers(5))
        m = int(rng.choice([0, 1, 2, 4, 8]))
        threads = int(rng.choice([0, 64, 256, 1024]))
        label = "case %d: n=%d degree=%s sweeps=%d reps=%d mode=%d m=%d threads=%d" % (
            case, n, degree, sweeps, reps, mode, m, threads)
        if mode == 0:
            _set_launch(ham, m, threads)                    # bytes / words, any group size
        elif mode == 1:
            _set_packed(ham, 1)
            _set_launch(ham, 0, threads)
        elif mode == 2:
            _set_packed(ham, 2)
            _set_launch(ham, 0, threads)
        elif mode == 3:
            _set_team(ham, int(rng.choice([2, 4, 8])))
        else:
            _set_launch(ham, m, threads)
            _lib.check(lib.asp_sa_set_wide(ham.plan(), 0))
            _set_cache(ham, False)
        xs, es = sa.anneal_raw(ham, 77 + case, betas, reps, offset, x0)
        tracked, accepted = _stats(ham, reps)
        S = ham.info().energy_scale_exp
        oxs, oes, otr, oacc = oracle.sa_anneal(J, h, 77 + case, betas, reps, offset, x0, S,
                                               num_threads=4)
        assert np.array_equal(xs, oxs), label
        assert es.tobytes() == oes.tobytes(), label
        assert np.array_equal(tracked, otr) and np.array_equal(accepted, oacc), label


# ---------------------------------------------------------------------------------------------
# Batched anneal: many problems in shared launches (asp_sa_anneal_batch)
# ---------------------------------------------------------------------------------------------

def _batch_problems(sizes, seed0, degree=6.0):
    from annealing_sign_problem_amd import annealer as sa
    from annealing_sign_problem_amd import synthetic

    problems = []
    for i, k in enumerate(sizes):
        J, h, _ = synthetic.planted_cluster(k, seed=seed0 + i, mean_degree=degree)
        ham = sa.Hamiltonian(J, h)
        info = ham.info()
        sweeps = 20 + 3 * (i % 5)
        problems.append(dict(J=J, h=h, ham=ham, seed=1000 + 17 * i, reps=3 + (5 * i) % 13,
                             offset=(i % 3) * 5,
                             betas=sa.make_schedule(info.beta0_auto, min(info.beta1_auto, 1e6), sweeps),
                             S=info.energy_scale_exp))
    return problems


@pytest.mark.parametrize("forced_m", [None, 1, 2, 4, 8])
def test_batch_equals_single_launches_and_oracle(forced_m, monkeypatch):
    """Every problem of a batch — different sizes, chain counts, ladders, seeds and replica
    offsets, spread over several wavefront classes — returns exactly what its own
    asp_sa_anneal call returns and what the oracle computes (spins, energies, tracked energies,
    accepted flips), for every number of replicas per workgroup."""
    from annealing_sign_problem_amd import annealer as sa

    if forced_m is None:
        monkeypatch.delenv("ASP_BATCH_M", raising=False)
    else:
        monkeypatch.setenv("ASP_BATCH_M", str(forced_m))
    sizes = [70, 130, 64, 400, 900, 1500, 2600, 5200, 3, 11000, 333, 1]
    problems = _batch_problems(sizes, 50)
    results = sa.anneal_batch_raw([p["ham"] for p in problems], [p["seed"] for p in problems],
                                  [p["betas"] for p in problems], [p["reps"] for p in problems],
                                  [p["offset"] for p in problems])
    stats = [_stats(p["ham"], p["reps"]) for p in problems]
    for p, (xs, es), (tracked, accepted) in zip(problems, results, stats):
        oxs, oes, otracked, oaccepted = oracle.sa_anneal(p["J"], p["h"], p["seed"], p["betas"],
                                                        p["reps"], p["offset"], None, p["S"],
                                                        num_threads=8)
        assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
        assert np.array_equal(tracked, otracked) and np.array_equal(accepted, oaccepted)
        sxs, ses = sa.anneal_raw(p["ham"], p["seed"], p["betas"], p["reps"], p["offset"])
        assert np.array_equal(xs, sxs) and es.tobytes() == ses.tobytes()


def test_batch_many_small_clusters_fill_the_chip_and_match():
    """The production shape: many clusters, 64 chains each (common.py:238), launcher's own choice
    of replicas per workgroup; chains still equal the oracle's."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    rng = np.random.default_rng(5)
    sizes = [int(round(np.exp(rng.uniform(np.log(100), np.log(3000))))) for _ in range(96)]
    problems = _batch_problems(sizes, 900, degree=10.0)
    results = sa.anneal_batch_raw([p["ham"] for p in problems], [12345] * len(problems),
                                  [p["betas"] for p in problems], [64] * len(problems))
    assert _lib.load().asp_sa_batch_last_ms() > 0
    for i in (0, 17, 40, 95):
        p = problems[i]
        oxs, oes, _, _ = oracle.sa_anneal(p["J"], p["h"], 12345, p["betas"], 64, 0, None, p["S"],
                                          num_threads=16)
        assert np.array_equal(results[i][0], oxs) and results[i][1].tobytes() == oes.tobytes()
    # the public form: one (x, e) per problem = the first minimum of its chains
    best = sa.anneal_batch([p["ham"] for p in problems[:5]], seed=7, number_sweeps=30, repetitions=8,
                           sweep_order="colour")
    for p, (x, e) in zip(problems[:5], best):
        x1, e1 = sa.anneal(p["ham"], seed=7, number_sweeps=30, repetitions=8, sweep_order="colour")
        assert np.array_equal(x, x1) and e == e1


@pytest.mark.parametrize("mode", ["default", "bits", "alone"])
def test_batch_mixes_the_spin_layouts(mode, monkeypatch):
    """One batch with problems of every LDS layout — a word per spin (small), a byte per spin
    (60 000 spins), four bits per spin (200 000 spins: the largest order-2 models of the
    sampled-cluster pipeline) and beyond that (260 000 spins) a bit per spin.  The last kind runs
    by itself as a team sweep by default, or as a class of the shared launches (ASP_BATCH_BITS=1);
    ASP_BATCH_NIBBLES=0 sends the third kind the same way.  Every chain is the single-problem
    call's and the oracle's."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa
    from annealing_sign_problem_amd import synthetic

    if mode != "default":
        monkeypatch.setenv("ASP_BATCH_NIBBLES", "0")
    if mode == "bits":
        monkeypatch.setenv("ASP_BATCH_BITS", "1")
    lib = _lib.load()
    problems = []
    for k, degree, sweeps, reps in ((200000, 6.0, 6, 5), (60000, 8.0, 8, 6), (9000, 10.0, 12, 8),
                                    (300, 6.0, 15, 5), (260000, 5.0, 5, 2)):
        J, h, _ = synthetic.planted_cluster(k, seed=k, mean_degree=degree)
        ham = sa.Hamiltonian(J, h)
        info = ham.info()
        problems.append(dict(J=J, h=h, ham=ham, reps=reps, S=info.energy_scale_exp,
                             betas=sa.make_schedule(info.beta0_auto, min(info.beta1_auto, 1e6), sweeps)))
    results = sa.anneal_batch_raw([p["ham"] for p in problems], [31, 32, 33, 34, 35],
                                  [p["betas"] for p in problems], [p["reps"] for p in problems])
    layouts = [lib.asp_sa_last_layout(p["ham"].plan()) for p in problems]
    # 0 bytes, 1 bits, 2 words (when four chains share a workgroup), 4 team sweep of a problem
    # that ran by itself, 6 nibbles
    assert layouts[0] == {"default": 6, "bits": 1, "alone": 4}[mode]
    assert layouts[4] == (1 if mode == "bits" else 4)
    assert layouts[1] == 0 and set(layouts[2:4]) <= {0, 2}
    stats = [_stats(p["ham"], p["reps"]) for p in problems]
    for seed, p, (xs, es), (tracked, accepted) in zip((31, 32, 33, 34, 35), problems, results, stats):
        oxs, oes, otracked, oaccepted = oracle.sa_anneal(p["J"], p["h"], seed, p["betas"], p["reps"], 0,
                                                        None, p["S"], num_threads=8)
        assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
        assert np.array_equal(tracked, otracked) and np.array_equal(accepted, oaccepted)
        sxs, ses = sa.anneal_raw(p["ham"], seed, p["betas"], p["reps"])
        assert np.array_equal(xs, sxs) and es.tobytes() == ses.tobytes()


def test_nibble_layout_single_problem_matches_oracle():
    """170 000 spins — more than a byte per spin holds in LDS — with four chains per workgroup:
    four bits per spin (layout 6), flips as LDS atomics, through a ladder that ends frozen (field
    cache and inert blocks in use); chains, tracked energies and accepted flips equal the oracle's,
    and the chain-per-workgroup bit layout's."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa
    from annealing_sign_problem_amd import synthetic

    lib = _lib.load()
    J, h, _ = synthetic.planted_cluster(170000, seed=21, mean_degree=7.0)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 40)
    _set_launch(ham, 4, 1024)
    xs, es = sa.anneal_raw(ham, 777, betas, 9, 3)
    assert lib.asp_sa_last_layout(ham.plan()) == 6
    tracked, accepted = _stats(ham, 9)
    oxs, oes, otracked, oaccepted = oracle.sa_anneal(J, h, 777, betas, 9, 3, None, info.energy_scale_exp,
                                                    num_threads=8)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    assert np.array_equal(tracked, otracked) and np.array_equal(accepted, oaccepted)
    _set_launch(ham, 0, 0)
    bxs, bes = sa.anneal_raw(ham, 777, betas, 9, 3)  # few chains: teams / a bit per spin
    assert lib.asp_sa_last_layout(ham.plan()) in (1, 4)
    assert np.array_equal(xs, bxs) and es.tobytes() == bes.tobytes()
    # many chains: the launcher picks four per workgroup by itself
    xs2, es2 = sa.anneal_raw(ham, 777, betas[:6], 1030)
    assert lib.asp_sa_last_layout(ham.plan()) == 6
    o2, oe2, _, _ = oracle.sa_anneal(J, h, 777, betas[:6], 6, 1024, None, info.energy_scale_exp, num_threads=8)
    assert np.array_equal(xs2[1024:], o2) and es2[1024:].tobytes() == oe2.tobytes()


def test_batch_rejects_bad_items():
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    problems = _batch_problems([100, 200], 3)
    ham = problems[0]["ham"]
    with pytest.raises(ValueError):
        sa.anneal_batch_raw([ham, ham], [1, 2], [problems[0]["betas"]] * 2, [2, 2])
    bad = problems[1]["betas"].copy()
    bad[3] = -1.0
    with pytest.raises(_lib.AspError):
        sa.anneal_batch_raw([p["ham"] for p in problems], [1, 2], [problems[0]["betas"], bad], [2, 2])
    assert sa.anneal_batch_raw([], [], [], []) == []


def test_team_watchdog_timeout_reruns_without_teams(monkeypatch):
    """a barrier time-out used to be a hard error.  Provoked here (watchdog limit of one
    poll): the call must come back with the correct chains through the one-workgroup-per-chain
    kernel, and the plan must stay off teams afterwards."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    lib = _lib.load()
    J, h, _ = _planted(6000, 77)
    betas = np.geomspace(0.5, 5e4, 25)
    ham = sa.Hamiltonian(J, h)
    S = ham.info().energy_scale_exp
    oxs, oes, _, _ = oracle.sa_anneal(J, h, 99, betas, 4, 0, None, S, num_threads=4)
    _set_team(ham, 8)
    trips, all_trips = ctypes.c_uint32(9), ctypes.c_uint64(0)
    _lib.check(lib.asp_sa_team_watchdog_trips(ham.plan(), ctypes.byref(trips), ctypes.byref(all_trips)))
    before = all_trips.value
    assert trips.value == 0
    monkeypatch.setenv("ASP_TEAM_SPIN_LIMIT", "0")
    xs, es = sa.anneal_raw(ham, 99, betas, 4)
    assert lib.asp_sa_last_layout(ham.plan()) != 4          # answered by the fallback
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    # the trip is no longer silent
    _lib.check(lib.asp_sa_team_watchdog_trips(ham.plan(), ctypes.byref(trips), ctypes.byref(all_trips)))
    assert trips.value == 1 and all_trips.value == before + 1
    monkeypatch.delenv("ASP_TEAM_SPIN_LIMIT")
    xs, es = sa.anneal_raw(ham, 99, betas, 4)                # teams stay off for this plan
    assert lib.asp_sa_last_layout(ham.plan()) != 4
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()


def test_team_is_refused_for_single_colour_plans():
    """with one colour class (a diagonal-only J: no couplings at all) the team's single
    barrier per sweep does not order a fast member's next flip words against a slow member's
    read.  The launcher must not form teams there, forced or not; results equal the oracle's."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    n = 64 * 80  # 80 blocks in the one colour class: wide enough for the automatic rule
    rng = np.random.default_rng(12)
    J = scipy.sparse.diags(rng.normal(size=n)).tocsr()
    field = rng.normal(size=n)
    ham = sa.Hamiltonian(J, field)
    assert ham.info().num_colors == 1
    betas = np.geomspace(0.1, 50.0, 20)
    S = ham.info().energy_scale_exp
    oxs, oes, _, _ = oracle.sa_anneal(J, field, 5, betas, 3, 0, None, S, num_threads=3)
    for team in (-1, 8):
        _set_team(ham, team)
        xs, es = sa.anneal_raw(ham, 5, betas, 3)
        assert _lib.load().asp_sa_last_layout(ham.plan()) != 4
        assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()


def test_team_exchange_stress_against_single_workgroup_kernel():
    """The team barrier orders its exchange with `s_waitcnt vmcnt(0)` + relaxed device-scope
    atomics on fine-grained memory instead of release/acquire fences (a fence costs
    20 us on this chip).  That rests on hardware behaviour, so it is stressed in the suite (a
    reduced tools/stress_team.py): shapes x chain counts x team sizes x repeats, long ladders,
    every result compared bit for bit with the one-workgroup-per-chain kernel."""
    from annealing_sign_problem_amd import _lib, synthetic
    from annealing_sign_problem_amd import annealer as sa

    lib = _lib.load()
    launches = 0
    for k, degree, sweeps in ((60000, 12.0, 200), (25000, 6.0, 300)):
        J, h, _ = synthetic.planted_cluster(k, seed=k, mean_degree=degree)
        ham = sa.Hamiltonian(J, h)
        info = ham.info()
        betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, sweeps)
        for chains in (1, 7, 32):
            _set_team(ham, 0)
            ref_x, ref_e = sa.anneal_raw(ham, 99, betas, chains)
            for team in (2, 4, 8):
                _set_team(ham, team)
                for _ in range(2):
                    x, e = sa.anneal_raw(ham, 99, betas, chains)
                    assert lib.asp_sa_last_layout(ham.plan()) == 4
                    assert np.array_equal(x, ref_x) and e.tobytes() == ref_e.tobytes(), (k, chains, team)
                    launches += 1
    assert launches == 36


# ---------------------------------------------------------------------------------------------
# Shuffled sweep: a fresh visiting order every sweep (asp_sa_anneal_shuffled)
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n,degree,sweeps,reps", [(1, 1.0, 5, 2), (70, 5.0, 30, 5), (900, 12.0, 40, 7),
                                                  (6000, 23.0, 25, 4)])
def test_shuffled_sweep_bit_exact(n, degree, sweeps, reps):
    """Levels of the priority graph on the device == one spin after another in priority order in
    the oracle: spins, energies, tracked energies and accepted-flip counts, with and without x0
    and a replica offset; and the colour-ordered sweep gives DIFFERENT chains."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    J, h, _ = _planted(n, 70 + n, mean_degree=min(degree, max(n / 3, 1.0)))
    field = np.random.default_rng(n).normal(size=n) * 0.01
    ham = sa.Hamiltonian(J, field)
    info = ham.info()
    betas = sa.make_schedule(max(info.beta0_auto, 1e-3), min(max(info.beta1_auto, 1.0), 1e6), sweeps)
    S = info.energy_scale_exp
    xs, es = sa.anneal_raw(ham, 31337, betas, reps, 3, None, shuffled=True)
    assert _lib.load().asp_sa_last_layout(ham.plan()) == 5
    tracked, accepted = _stats(ham, reps)
    oxs, oes, otracked, oaccepted = oracle.sa_anneal_shuffled(J, field, 31337, betas, reps, 3, None, S,
                                                             num_threads=4)
    assert np.array_equal(accepted, oaccepted) and np.array_equal(tracked, otracked)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    x0 = sa.signs_to_bits(np.where(np.random.default_rng(2).random(n) < 0.5, 1.0, -1.0))
    xs, es = sa.anneal_raw(ham, 5, betas, 2, 0, x0, shuffled=True)
    oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, field, 5, betas, 2, 0, x0, S, num_threads=2)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    if n >= 900:
        cxs, _ = sa.anneal_raw(ham, 31337, betas, reps, 3)
        assert not np.array_equal(cxs, oxs if reps == 2 else sa.anneal_raw(ham, 31337, betas, reps, 3, None, shuffled=True)[0])


def test_shuffled_sweep_runs_long_ladders_in_chunks():
    """More sweeps than one chunk of visiting orders holds (at most 256): the chain state
    survives between the chunks' launches, and the order kernel of a chunk runs beside the sweep
    kernel of the chunk before; compared with the oracle."""
    from annealing_sign_problem_amd import annealer as sa

    n = 40000
    J, h, _ = _planted(n, 8, mean_degree=4.0)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 900)
    xs, es = sa.anneal_raw(ham, 11, betas, 2, 0, None, shuffled=True)
    oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, 11, betas, 2, 0, None, info.energy_scale_exp,
                                               num_threads=2)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()


def _shuffled_case(n, degree, sweeps, seed=5, **kw):
    from annealing_sign_problem_amd import annealer as sa

    J, h, _ = _planted(n, seed, mean_degree=degree, **kw)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    betas = sa.make_schedule(max(info.beta0_auto, 1e-3), min(max(info.beta1_auto, 1.0), 1e6), sweeps)
    return J, h, ham, info, betas


@pytest.mark.parametrize("m,waves", [(1, 1), (1, 8), (2, 3), (4, 1), (4, 5), (8, 2), (8, 8)])
def test_shuffled_sweep_does_not_depend_on_the_launch_geometry(m, waves):
    """Chains per workgroup (word layout for <= 4, bytes for 8) and wavefronts per workgroup:
    the same chains as the oracle for every choice; 11 chains leave a ragged last group."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    J, h, ham, info, betas = _shuffled_case(1300, 9.0, 14)
    lib = _lib.load()
    _lib.check(lib.asp_sa_set_shuffled_launch(ham.plan(), m, waves))
    xs, es = sa.anneal_raw(ham, 77, betas, 11, 6, None, shuffled=True)
    got_m, got_threads, got_groups = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.asp_sa_last_launch(ham.plan(), ctypes.byref(got_m), ctypes.byref(got_threads),
                                      ctypes.byref(got_groups)))
    assert (got_m.value, got_threads.value, got_groups.value) == (m, 64 * waves, (11 + m - 1) // m)
    tracked, accepted = _stats(ham, 11)
    oxs, oes, otracked, oaccepted = oracle.sa_anneal_shuffled(J, h, 77, betas, 11, 6, None,
                                                             info.energy_scale_exp, num_threads=4)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    assert np.array_equal(tracked, otracked) and np.array_equal(accepted, oaccepted)


@pytest.mark.parametrize("m,waves", [(2, 2), (4, 4), (4, 1), (8, 3)])
def test_shuffled_sweep_two_teams_per_workgroup(m, waves):
    """asp_sa_set_shuffled_teams(2): two teams of wavefronts share a workgroup's blocks, each for
    half of its chains (half of every LDS spin word, or four bits of every spin byte).  Same
    chains as the oracle."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    J, h, ham, info, betas = _shuffled_case(1700, 10.0, 12)
    lib = _lib.load()
    _lib.check(lib.asp_sa_set_shuffled_launch(ham.plan(), m, waves))
    _lib.check(lib.asp_sa_set_shuffled_teams(ham.plan(), 2))
    xs, es = sa.anneal_raw(ham, 41, betas, 13, 2, None, shuffled=True)
    got_m, got_threads, got_groups = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.asp_sa_last_launch(ham.plan(), ctypes.byref(got_m), ctypes.byref(got_threads),
                                      ctypes.byref(got_groups)))
    assert (got_m.value, got_threads.value) == (m, 128 * waves)
    tracked, accepted = _stats(ham, 13)
    oxs, oes, otracked, oaccepted = oracle.sa_anneal_shuffled(J, h, 41, betas, 13, 2, None,
                                                             info.energy_scale_exp, num_threads=4)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    assert np.array_equal(tracked, otracked) and np.array_equal(accepted, oaccepted)


def test_shuffled_sweep_rows_wider_than_the_held_quads():
    """Rows of more than 48 couplings (a dense SK-like cluster: 120 per row) do not fit the
    registers a wavefront keeps a block in: the rest of the row streams.  Same bits."""
    from annealing_sign_problem_amd import annealer as sa
    from annealing_sign_problem_amd import synthetic

    J, h = synthetic.sk_cluster(400, degree=120, seed=3)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    assert info.max_degree > 60
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 10)
    for m in (4, 8):
        from annealing_sign_problem_amd import _lib

        _lib.check(_lib.load().asp_sa_set_shuffled_launch(ham.plan(), m, 0))
        xs, es = sa.anneal_raw(ham, 3, betas, 9, 0, None, shuffled=True)
        oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, 3, betas, 9, 0, None, info.energy_scale_exp,
                                                   num_threads=4)
        assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()


def test_shuffled_sweep_grows_its_capacities_and_repeats(monkeypatch):
    """The order kernel works within capacities guessed from the degree; a sweep with more levels
    than provided makes it raise a flag, the sweep kernels of the attempt stand down, and the call
    is repeated with the capacities the flag words ask for.  Provoked here with a cap of 2 levels."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    J, h, ham, info, betas = _shuffled_case(2500, 12.0, 9)
    reference = sa.anneal_raw(ham, 21, betas, 5, 0, None, shuffled=True)
    levels = ctypes.c_uint32(0)
    _lib.check(_lib.load().asp_sa_last_shuffled(ham.plan(), ctypes.byref(levels), None))
    assert levels.value > 8
    monkeypatch.setenv("ASP_SHUFFLED_LEVEL_CAP", "2")
    again = sa.anneal_raw(ham, 21, betas, 5, 0, None, shuffled=True)
    assert np.array_equal(again[0], reference[0]) and again[1].tobytes() == reference[1].tobytes()
    oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, 21, betas, 5, 0, None, info.energy_scale_exp,
                                               num_threads=4)
    assert np.array_equal(again[0], oxs) and again[1].tobytes() == oes.tobytes()


def test_shuffled_sweep_byte_layout_of_large_clusters():
    """Beyond ~4e4 spins a word per spin no longer fits the LDS: a byte per spin (bit m = chain
    m), also for four chains per workgroup.  70 000 spins, the oracle on 3 chains x 5 sweeps."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    J, h, ham, info, betas = _shuffled_case(70000, 6.0, 5)
    _lib.check(_lib.load().asp_sa_set_shuffled_launch(ham.plan(), 4, 0))
    xs, es = sa.anneal_raw(ham, 9, betas, 3, 2, None, shuffled=True)
    oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, 9, betas, 3, 2, None, info.energy_scale_exp,
                                               num_threads=3)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()


@pytest.mark.parametrize("k,forced,expect_m", [(170000, 4, 4), (170000, 0, 1), (300000, 4, 1)])
def test_shuffled_sweep_beyond_a_byte_per_spin(k, forced, expect_m):
    """Clusters that do not fit the LDS with a byte per spin — the largest order-2 models of the
    sampled-cluster pipeline — keep four bits per spin (up to four chains per workgroup, 170 000
    spins) or one bit (one chain, 300 000 spins); a request for more chains per workgroup than fit
    is cut down.  Flips are LDS atomics there.  Chains as the oracle's."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    J, h, ham, info, betas = _shuffled_case(k, 6.0, 4)
    lib = _lib.load()
    _lib.check(lib.asp_sa_set_shuffled_launch(ham.plan(), forced, 0))
    xs, es = sa.anneal_raw(ham, 4321, betas, 5, 2, None, shuffled=True)
    got_m = ctypes.c_int(0)
    _lib.check(lib.asp_sa_last_launch(ham.plan(), ctypes.byref(got_m), None, None))
    assert got_m.value == expect_m
    oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, 4321, betas, 5, 2, None, info.energy_scale_exp,
                                               num_threads=8)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()


def test_shuffled_batch_with_a_cluster_beyond_a_byte_per_spin():
    """A batch of shuffled items, one of them too large for a byte per spin: it shares the
    launches in a class of its own layout (or runs by itself when its chains per workgroup had
    to be cut); every item is its own single call."""
    from annealing_sign_problem_amd import annealer as sa

    cases = [_shuffled_case(k, d, 6, seed=k) for k, d in ((500, 5.0), (170000, 5.0), (2000, 8.0), (60000, 6.0))]
    hams = [c[2] for c in cases]
    batch = sa.anneal_batch_raw(hams, [3, 4, 5, 6], [c[4] for c in cases], [4, 4, 4, 4], shuffled=True)
    for (J, h, ham, info, betas), seed, (bx, be) in zip(cases, (3, 4, 5, 6), batch):
        sx, se = sa.anneal_raw(ham, seed, betas, 4, 0, None, shuffled=True)
        assert np.array_equal(bx, sx) and be.tobytes() == se.tobytes()
    J, h, ham, info, betas = cases[1]
    ox, oe, _, _ = oracle.sa_anneal_shuffled(J, h, 4, betas, 4, 0, None, info.energy_scale_exp, num_threads=8)
    assert np.array_equal(batch[1][0], ox) and batch[1][1].tobytes() == oe.tobytes()
    # many chains: four per workgroup for the whole batch, four bits per spin for the large one
    big = sa.anneal_batch_raw(hams[:2], [1, 2], [c[4] for c in cases[:2]], [600, 600], shuffled=True)
    for (J, h, ham, info, betas), seed, (bx, be) in zip(cases[:2], (1, 2), big):
        ox, oe, _, _ = oracle.sa_anneal_shuffled(J, h, seed, betas, 3, 597, None, info.energy_scale_exp, num_threads=8)
        assert np.array_equal(bx[597:], ox) and be[597:].tobytes() == oe.tobytes()


def test_shuffled_batch_equals_the_single_calls():
    """asp_sa_anneal_batch with ASP_SA_BATCH_SHUFFLED: every problem is its own
    asp_sa_anneal_shuffled call, chain for chain (and the oracle's), while their kernels overlap
    on the plans' streams; mixed with colour-ordered items in one batch through two calls."""
    from annealing_sign_problem_amd import annealer as sa

    problems = []
    # four problems with one ladder length share their launches (two wavefront classes among
    # them), the fifth has another length and runs by itself inside the same call
    for k, deg, sweeps in ((90, 4.0, 40), (700, 8.0, 40), (2600, 14.0, 40), (1, 1.0, 40), (300, 6.0, 13)):
        J, h, ham, info, betas = _shuffled_case(k, min(deg, max(k / 3, 1.0)), sweeps, seed=k)
        problems.append((J, h, ham, info, betas))
    hams = [p[2] for p in problems]
    seeds = [5, 6, 7, 8, 9]
    reps = [6, 3, 9, 2, 5]
    offsets = [0, 4, 1, 0, 2]
    batch = sa.anneal_batch_raw(hams, seeds, [p[4] for p in problems], reps, offsets, shuffled=True)
    for (J, h, ham, info, betas), seed, r, off, (bx, be) in zip(problems, seeds, reps, offsets, batch):
        sx, se = sa.anneal_raw(ham, seed, betas, r, off, None, shuffled=True)
        assert np.array_equal(bx, sx) and be.tobytes() == se.tobytes()
        ox, oe, _, _ = oracle.sa_anneal_shuffled(J, h, seed, betas, r, off, None, info.energy_scale_exp,
                                                 num_threads=4)
        assert np.array_equal(bx, ox) and be.tobytes() == oe.tobytes()
    # many chains per problem: the batch packs four chains into a workgroup (the single call one)
    big = sa.anneal_batch_raw(hams[:3], [1, 2, 3], [p[4] for p in problems[:3]], [300, 300, 300], shuffled=True)
    for (J, h, ham, info, betas), seed, (bx, be) in zip(problems[:3], (1, 2, 3), big):
        sx, se = sa.anneal_raw(ham, seed, betas, 300, 0, None, shuffled=True)
        assert np.array_equal(bx, sx) and be.tobytes() == se.tobytes()
    # the public entry points: anneal_batch(sweep_order=...) == [anneal(..., sweep_order=...)]
    best = sa.anneal_batch(hams[:3], seed=12345, number_sweeps=30, repetitions=5, sweep_order="shuffled")
    for ham, (x, e) in zip(hams[:3], best):
        sx, se = sa.anneal(ham, seed=12345, number_sweeps=30, repetitions=5, sweep_order="shuffled")
        assert np.array_equal(x, sx) and e == se
    colour = sa.anneal_batch(hams[:3], seed=12345, number_sweeps=30, repetitions=5, sweep_order="colour")
    assert any(not np.array_equal(x, cx) for (x, _), (cx, _) in zip(best, colour))


def test_sweep_order_reaches_the_solver_entry_points(monkeypatch):
    """common.solve_ising_model(s)(sweep_order=...) and $ASP_SWEEP_ORDER select the annealer."""
    from annealing_sign_problem_amd import annealer as sa
    from annealing_sign_problem_amd import common

    J, h, ham, info, betas = _shuffled_case(600, 7.0, 8)
    model = common.IsingModel(np.arange(600, dtype=np.uint64), None, ham, sa.signs_to_bits(np.ones(600)))
    x_colour = common.solve_ising_model(model, seed=3, number_sweeps=50, repetitions=4,
                                        sweep_order="colour")
    x_shuffled = common.solve_ising_model(model, seed=3, number_sweeps=50, repetitions=4,
                                          sweep_order="shuffled")
    # the drop-in default is the reference's law
    monkeypatch.delenv("ASP_SWEEP_ORDER", raising=False)
    assert np.array_equal(common.solve_ising_model(model, seed=3, number_sweeps=50, repetitions=4), x_shuffled)
    expected, _ = sa.anneal(ham, seed=3, number_sweeps=50, repetitions=4, sweep_order="shuffled")
    assert np.array_equal(x_shuffled, expected) and not np.array_equal(x_shuffled, x_colour)
    assert np.array_equal(common.solve_ising_models([model], seed=3, number_sweeps=50, repetitions=4,
                                                    sweep_order="shuffled")[0], expected)
    monkeypatch.setenv("ASP_SWEEP_ORDER", "colour")
    assert np.array_equal(common.solve_ising_model(model, seed=3, number_sweeps=50, repetitions=4), x_colour)
    monkeypatch.setenv("ASP_SWEEP_ORDER", "shuffled")
    assert np.array_equal(common.solve_ising_model(model, seed=3, number_sweeps=50, repetitions=4), expected)
    with pytest.raises(ValueError):
        sa.anneal(ham, seed=3, number_sweeps=5, sweep_order="typewriter")


def _shuffled_blocks(ham):
    from annealing_sign_problem_amd import _lib

    spins, wgs = ctypes.c_uint32(0), ctypes.c_uint32(0)
    _lib.check(_lib.load().asp_sa_last_shuffled_blocks(ham.plan(), ctypes.byref(spins), ctypes.byref(wgs)))
    return spins.value, wgs.value


@pytest.mark.parametrize("n,degree,reps,offset,m,log_s", [
    (90, 20.0, 64, 0, 4, None),    # one workgroup holds all 64 chains: 16 groups x 4 spins per wavefront
    (90, 20.0, 13, 2, 4, None),    # ragged: 4 groups (one of a single chain), idle groups in the wavefront
    (400, 22.0, 64, 0, 4, None),
    (1000, 23.0, 64, 7, 4, None),  # replica offset not a multiple of four: two Philox calls per lane
    (1000, 23.0, 20, 1, 2, 3),     # M = 2, blocks of 8 forced: 8 groups per workgroup, 10 groups in the call
    (700, 9.0, 19, 0, 1, 2),       # M = 1, blocks of 4: 16 chains per wavefront, 19 in the call
    (2500, 23.0, 33, 5, 4, 5),     # blocks of 32, two groups per workgroup, an odd number of groups
    (300, 30.0, 40, 0, 4, 4),      # rows wider than... a level smaller than the forced block
])
def test_shuffled_sweep_lane_packing(monkeypatch, n, degree, reps, offset, m, log_s):
    """levels of fewer than 64 spins are cut into blocks of S = 4 .. 32 spins and
    a wavefront visits a block for 64 / S groups of chains at once (lane = (group, spin)).  Every
    chain must be the oracle's, whatever S, M, the number of chains and the replica offset; x0 too."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    J, h, ham, info, betas = _shuffled_case(n, degree, 12, seed=n)
    S = info.energy_scale_exp
    lib = _lib.load()
    if log_s is not None:
        monkeypatch.setenv("ASP_SHUFFLED_LOG_S", str(log_s))
    _lib.check(lib.asp_sa_set_shuffled_launch(ham.plan(), m, 0))
    xs, es = sa.anneal_raw(ham, 4711, betas, reps, offset, None, shuffled=True)
    spins_per_block, wgs = _shuffled_blocks(ham)
    assert spins_per_block < 64 and wgs == -(-(-(-reps // m)) // (64 // spins_per_block))
    if log_s is not None:
        assert spins_per_block == 1 << log_s
    tracked, accepted = _stats(ham, reps)
    oxs, oes, otracked, oaccepted = oracle.sa_anneal_shuffled(J, h, 4711, betas, reps, offset, None, S,
                                                             num_threads=8)
    assert np.array_equal(accepted, oaccepted) and np.array_equal(tracked, otracked)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    x0 = sa.signs_to_bits(np.where(np.random.default_rng(n).random(n) < 0.5, 1.0, -1.0))
    xs, es = sa.anneal_raw(ham, 5, betas, 6, 1, x0, shuffled=True)
    oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, 5, betas, 6, 1, x0, S, num_threads=6)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    # the same chains without packing (blocks of 64 spins, one group per workgroup)
    monkeypatch.setenv("ASP_SHUFFLED_NO_PACKING", "1")
    plain_x, plain_e = sa.anneal_raw(ham, 5, betas, 6, 1, x0, shuffled=True)
    assert _shuffled_blocks(ham)[0] == 64
    assert np.array_equal(plain_x, xs) and plain_e.tobytes() == es.tobytes()


def test_shuffled_lane_packing_over_many_chunks_and_in_a_batch():
    """Packed problems keep their chain state between the chunks of a long ladder, and a batch
    mixes block sizes (one launch per class of workgroup shape): every problem is its own call."""
    from annealing_sign_problem_amd import annealer as sa

    cases = [_shuffled_case(k, d, 600 if k == 150 else 40, seed=k) for k, d in
             ((150, 15.0), (800, 23.0), (60, 10.0), (5000, 23.0), (2000, 23.0))]
    J, h, ham, info, betas = cases[0]
    xs, es = sa.anneal_raw(ham, 21, betas, 10, 0, None, shuffled=True)
    assert _shuffled_blocks(ham)[0] < 64
    oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, 21, betas, 10, 0, None, info.energy_scale_exp, num_threads=8)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    batch = cases[1:]
    results = sa.anneal_batch_raw([c[2] for c in batch], [3, 4, 5, 6], [c[4] for c in batch], [64, 64, 17, 64],
                                  shuffled=True)
    sizes = [_shuffled_blocks(c[2])[0] for c in batch]
    assert len(set(sizes)) >= 3 and max(sizes) == 64
    for (J, h, ham, info, betas), seed, reps, (bx, be) in zip(batch, (3, 4, 5, 6), (64, 64, 17, 64), results):
        oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, seed, betas, reps, 0, None, info.energy_scale_exp,
                                                   num_threads=16)
        assert np.array_equal(bx, oxs) and be.tobytes() == oes.tobytes()


_LEVEL_LAUNCHES = {"ASP_SHUFFLED_ORDER_IN_HBM": "1", "ASP_SHUFFLED_COUNTERS_IN_HBM": "1"}


@pytest.mark.parametrize("n,degree,reps,sweeps,env", [
    (3000, 12.0, 6, 40, _LEVEL_LAUNCHES),                                        # the level launches on a small cluster
    (3000, 12.0, 6, 40, {"ASP_SHUFFLED_ORDER_IN_HBM": "1"}),                     # counters in LDS, the rest in HBM
    (3000, 12.0, 6, 40, {"ASP_SHUFFLED_ORDER_IN_HBM": "1", "ASP_SHUFFLED_ORDER_FUSED": "1"}),  # one workgroup per sweep, arrays in HBM
    (700, 9.0, 20, 300, dict(_LEVEL_LAUNCHES, ASP_SHUFFLED_BYTES="400000")),     # many chunks, lane packing
    (701, 9.0, 20, 300, {"ASP_SHUFFLED_ORDER_IN_HBM": "1", "ASP_SHUFFLED_BYTES": "400000"}),   # the same, counters in LDS
    (70000, 6.0, 3, 24, {}),                                                     # beyond 16-bit indices: counters in LDS by itself
    (70000, 6.0, 3, 24, {"ASP_SHUFFLED_COUNTERS_IN_HBM": "1"}),                  # ... and the level launches
    (3000, 40.0, 4, 30, {"ASP_SHUFFLED_ORDER_IN_HBM": "1", "ASP_SHUFFLED_COUNTER_NIBBLES": "1"}),  # 4-bit counters, most spins on the HBM escape
    (3001, 9.0, 4, 30, {"ASP_SHUFFLED_ORDER_IN_HBM": "1", "ASP_SHUFFLED_COUNTER_NIBBLES": "1"}),   # ... few of them
    (170000, 5.0, 2, 12, {}),                                                    # beyond byte counters: nibbles by itself
    (330000, 4.0, 2, 8, {}),                                                     # beyond those: level launches by itself
])
def test_shuffled_order_build_paths(monkeypatch, n, degree, reps, sweeps, env):
    """The visiting orders are built by one of four device paths — a workgroup per sweep with the
    peel's arrays in LDS, the same with the arrays in HBM, or (large clusters) priorities, counts and
    stream as grids over all sweeps of the chunk and the peel either in the per-sweep workgroup with
    its counters in LDS (bytes, or nibbles with an HBM escape for spins of 15 or more earlier
    neighbours) or as one launch per level — and every path must give the oracle's chains."""
    from annealing_sign_problem_amd import annealer as sa

    for key, value in env.items():
        monkeypatch.setenv(key, value)
    J, h, ham, info, betas = _shuffled_case(n, degree, sweeps, seed=n + 1)
    xs, es = sa.anneal_raw(ham, 99, betas, reps, 2, None, shuffled=True)
    oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, 99, betas, reps, 2, None, info.energy_scale_exp,
                                               num_threads=min(reps, 8))
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()


@pytest.mark.parametrize("wide_levels", [1, 7, 500])
def test_shuffled_wide_orders_hand_over_to_the_sweep_workgroup(monkeypatch, wide_levels):
    """The wide path peels the first levels of every sweep of a chunk as grids (one launch per level)
    and the per-sweep workgroup peels the rest from where the launches stop: any split gives the
    oracle's chains — one launch (nearly everything in the workgroup), a few, or more launches than
    the sweeps have levels (the peel completes inside the launches); alone and inside a batch."""
    from annealing_sign_problem_amd import annealer as sa

    monkeypatch.setenv("ASP_SHUFFLED_ORDER_IN_HBM", "1")
    monkeypatch.setenv("ASP_SHUFFLED_COUNTERS_IN_HBM", "1")
    monkeypatch.setenv("ASP_SHUFFLED_BYTES", "2000000")  # several chunks
    monkeypatch.setenv("ASP_SHUFFLED_WIDE_LEVELS", str(wide_levels))
    J, h, ham, info, betas = _shuffled_case(2500, 10.0, 60, seed=17)
    xs, es = sa.anneal_raw(ham, 5, betas, 4, 0, None, shuffled=True)
    oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, 5, betas, 4, 0, None, info.energy_scale_exp, num_threads=4)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    other = _shuffled_case(400, 8.0, 60, seed=18)
    results = sa.anneal_batch_raw([ham, other[2]], [5, 6], [betas, other[4]], [4, 4], shuffled=True)
    assert np.array_equal(results[0][0], oxs) and results[0][1].tobytes() == oes.tobytes()
    o2, e2, _, _ = oracle.sa_anneal_shuffled(other[0], other[1], 6, other[4], 4, 0, None, other[3].energy_scale_exp,
                                             num_threads=4)
    assert np.array_equal(results[1][0], o2) and results[1][1].tobytes() == e2.tobytes()


def test_shuffled_batch_mixes_the_order_build_paths():
    """One batch with a cluster of every order-build path — arrays in LDS, byte and nibble counters in LDS
    (peel in the per-sweep workgroup), level launches — shares its launches; every problem is its own
    call's chain."""
    from annealing_sign_problem_amd import annealer as sa

    cases = [_shuffled_case(400, 8.0, 10, seed=31), _shuffled_case(60000, 6.0, 10, seed=32),
             _shuffled_case(165000, 5.0, 10, seed=33), _shuffled_case(30000, 7.0, 10, seed=34),
             _shuffled_case(330000, 4.0, 10, seed=35)]
    seeds, reps_of = (1, 2, 3, 4, 5), (4, 2, 2, 3, 1)
    results = sa.anneal_batch_raw([c[2] for c in cases], list(seeds), [c[4] for c in cases], list(reps_of),
                                  shuffled=True)
    for (J, h, ham, info, betas), seed, reps, (bx, be) in zip(cases, seeds, reps_of, results):
        oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, seed, betas, reps, 0, None, info.energy_scale_exp,
                                                   num_threads=4)
        assert np.array_equal(bx, oxs) and be.tobytes() == oes.tobytes()


def test_shuffled_fill_statistics():
    """asp_sa_last_shuffled_fill: spins / lane slots and couplings / coupling slots of the level-major
    blocks — what lane packing improves for small clusters and what no exec-mask counter shows."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    lib = _lib.load()
    J, h, ham, info, betas = _shuffled_case(400, 22.0, 10, seed=3)
    fills = {}
    for packing in (True, False):
        if not packing:
            os.environ["ASP_SHUFFLED_NO_PACKING"] = "1"
        try:
            sa.anneal_raw(ham, 1, betas, 64, 0, None, shuffled=True)
        finally:
            os.environ.pop("ASP_SHUFFLED_NO_PACKING", None)
        lane, row = ctypes.c_double(0), ctypes.c_double(0)
        _lib.check(lib.asp_sa_last_shuffled_fill(ham.plan(), ctypes.byref(lane), ctypes.byref(row)))
        fills[packing] = (lane.value, row.value)
    assert 0.5 < fills[True][0] <= 1.0 and 0.3 < fills[True][1] <= 1.0
    assert fills[False][0] < 0.3 < fills[True][0]  # ~9 spins per level: 64-lane blocks are mostly padding


def test_shuffled_sweep_with_the_spins_in_hbm(monkeypatch):
    """Beyond a bit per spin in LDS (~6e5 spins) the default visiting order keeps a chain's spins as
    words in HBM (one chain per workgroup, every gather an L2 access): slow, but `sa.anneal` then
    accepts a cluster of any size in its default order, as it does in the colour order.  Parity on
    a cluster that needs it (sparse, so that the oracle finishes), and forced on a small one."""
    from annealing_sign_problem_amd import _lib
    from annealing_sign_problem_amd import annealer as sa

    n = 700000
    J, h, _ = _planted(n, 5, mean_degree=3.0)
    ham = sa.Hamiltonian(J, h)
    info = ham.info()
    betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 6)
    xs, es = sa.anneal_raw(ham, 3, betas, 2, 1, None, shuffled=True)
    oxs, oes, _, _ = oracle.sa_anneal_shuffled(J, h, 3, betas, 2, 1, None, info.energy_scale_exp, num_threads=2)
    assert np.array_equal(xs, oxs) and es.tobytes() == oes.tobytes()
    # the public entry point in its default order
    x, e = sa.anneal(ham, seed=3, number_sweeps=4, repetitions=1)
    assert x.shape == ((n + 63) // 64,) and np.isfinite(e)

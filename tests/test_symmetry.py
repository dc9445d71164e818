"""Sector-0 symmetry-adapted bases (heisenberg_kagome_18.yaml:4,
heisenberg_kagome_36.yaml:7-29, heisenberg_pyrochlore_2x2x2.yaml:1-17).  lattice_symmetries is
absent, so everything is checked from first principles: brute-force orbits, the symmetric-sector
eigenvector embedded in the full space is an eigenvector of the full operator, symmetric matrix."""
import numpy as np
import pytest

from annealing_sign_problem_amd import operators, symmetry


def _ring(n, inversion):
    """Heisenberg ring of n sites: translation + reflection (+ spin inversion)."""
    return {
        "basis": {"number_spins": n, "hamming_weight": n // 2, "spin_inversion": inversion,
                  "symmetries": [{"permutation": [(i + 1) % n for i in range(n)], "sector": 0},
                                 {"permutation": [(n - i) % n for i in range(n)], "sector": 0}]},
        "hamiltonian": {"terms": [{"matrix": operators.SIGMA_DOT_SIGMA.tolist(),
                                   "sites": [[i, (i + 1) % n] for i in range(n)]}]},
    }


def _plain(config):
    basis = {k: v for k, v in config["basis"].items() if k in ("number_spins", "hamming_weight")}
    return {"basis": dict(basis, symmetries=[]), "hamiltonian": config["hamiltonian"]}


def test_group_orders_of_the_shipped_models(models):
    orders = {}
    for name in ("heisenberg_kagome_18", "heisenberg_kagome_36", "heisenberg_pyrochlore_2x2x2"):
        g = symmetry.group_from_config(models[name]["basis"])
        orders[name] = (g.num_permutations, g.order)
    # 36-site kagome: 12 translations x C6v; the 32-site pyrochlore cluster: 384 lattice maps
    assert orders == {"heisenberg_kagome_18": (1, 2), "heisenberg_kagome_36": (144, 288),
                      "heisenberg_pyrochlore_2x2x2": (384, 768)}
    assert symmetry.group_from_config(models["heisenberg_kagome_16"]["basis"]) is None


def test_symmetries_of_the_shipped_models_leave_the_hamiltonian_invariant(models):
    """Every generator must map the bond list onto itself — otherwise the YAML was misread."""
    for name in ("heisenberg_kagome_36", "heisenberg_pyrochlore_2x2x2"):
        cfg = models[name]
        bonds = {tuple(sorted(b)) for t in cfg["hamiltonian"]["terms"] for b in t["sites"]}
        for s in cfg["basis"]["symmetries"]:
            p = s["permutation"]
            assert {tuple(sorted((p[a], p[b]))) for a, b in bonds} == bonds


def test_state_info_against_brute_force_orbits(models):
    rng = np.random.default_rng(3)
    g = symmetry.group_from_config(models["heisenberg_kagome_36"]["basis"])
    states = []
    for _ in range(40):
        up = rng.choice(36, size=18, replace=False)
        states.append(sum(1 << int(b) for b in up))
    states += [int("01" * 18, 2), int("0" * 18 + "1" * 18, 2)]  # highly symmetric ones
    rep, character, norm = g.state_info(np.array(states, dtype=np.uint64))
    mask = (1 << 36) - 1
    for s, r, c, nrm in zip(states, rep, character, norm):
        orbit = set()
        for p in g.permutations:
            image = 0
            for i in range(36):
                if (s >> i) & 1:
                    image |= 1 << int(p[i])
            orbit.add(image)
            orbit.add(~image & mask)
        assert int(r) == min(orbit) and c == 1.0
        assert abs(nrm * nrm - 1.0 / len(orbit)) < 1e-15  # |Stab| / |G| = 1 / orbit size


@pytest.mark.parametrize("inversion", [1, -1, None])
def test_sector_eigenvector_is_an_eigenvector_of_the_full_operator(inversion):
    """The lowest state of the symmetric sector, expanded over the orbits with the characters and
    norms of symmetry.py, must satisfy H psi = E psi in the FULL basis: this pins the matrix
    elements c * chi * norm(r') / norm(r) of the symmetric batched_apply from first principles."""
    cfg = _ring(12, inversion)
    if inversion is None:
        cfg["basis"].pop("spin_inversion")
    sym = operators.Operator.from_config(cfg)
    sym.basis.build()
    full = operators.Operator.from_config(_plain(cfg))
    full.basis.build()
    assert sym.basis.number_states < full.basis.number_states // 10
    h = sym.to_sparse()
    assert abs(h - h.T).max() < 1e-12  # the symmetrised operator is symmetric
    energy, psi = sym.ground_state()
    group = sym.basis.group
    rep, character, norm = group.state_info(full.basis.states)
    where = np.searchsorted(sym.basis.states, rep)
    inside = (norm > 0) & (where < sym.basis.number_states)
    inside[inside] &= sym.basis.states[where[inside]] == rep[inside]
    # amplitude of s in |r~>: chi(g: s -> r) * norm(r) (each of the 1/norm^2 orbit members)
    vector = np.zeros(full.basis.number_states)
    vector[inside] = psi[where[inside]] * character[inside] * norm[inside]
    assert abs(np.linalg.norm(vector) - 1.0) < 1e-10
    hv = full.to_sparse().real @ vector
    assert np.abs(hv - energy * vector).max() < 1e-8
    if inversion != -1:  # the ring's ground state is the singlet: even under inversion for n = 12
        e_full, _ = full.ground_state()
        assert abs(e_full - energy) < 1e-8


def test_kagome_18_sector_of_the_yaml(models):
    """heisenberg_kagome_18.yaml:4 asks for spin inversion +1: 24 310 representatives (SURVEY
    appendix B); its lowest state is an eigenstate of the full 48 620-dimensional operator, and the
    global ground state sits in the -1 sector (9 up spins: odd)."""
    cfg = models["heisenberg_kagome_18"]
    op = operators.Operator.from_config(cfg)
    op.basis.build()
    assert op.basis.number_states == 24310
    energy, psi = op.ground_state()
    full = operators.Operator.from_config(_plain(cfg))
    full.basis.build()
    rep, character, norm = op.basis.group.state_info(full.basis.states)
    where = np.searchsorted(op.basis.states, rep)
    vector = psi[where] * character * norm
    hv = full.to_sparse().real @ vector
    assert np.abs(hv - energy * vector).max() < 1e-8
    odd = dict(cfg, basis=dict(cfg["basis"], spin_inversion=-1))
    op_odd = operators.Operator.from_config(odd)
    op_odd.basis.build()
    assert op_odd.basis.number_states == 24310
    assert abs(op_odd.ground_state()[0] - full.ground_state()[0]) < 1e-8


def test_degenerate_ground_level_gets_a_vector_fixed_by_construction(models):
    """heisenberg_kagome_18.yaml's sector: the lowest level is three-fold degenerate, and which
    of its vectors an eigensolver returns depends on its arithmetic.  Operator.ground_state
    returns the projection of its start vector onto the eigenspace — the same whatever basis of
    that eigenspace the solver delivers."""
    import scipy.sparse.linalg

    from annealing_sign_problem_amd import operators

    op = operators.Operator.from_config(models["heisenberg_kagome_18"])
    op.basis.build()
    energy, psi = op.ground_state()
    h = op.to_sparse().real.tocsr()
    assert np.linalg.norm(h @ psi - energy * psi) < 1e-9 and abs(np.linalg.norm(psi) - 1) < 1e-12
    # another run of the eigensolver, another basis of the eigenspace, the same projection
    n = h.shape[0]
    values, vectors = scipy.sparse.linalg.eigsh(h, k=6, which="SA", tol=1e-12, ncv=40,
                                                v0=np.random.default_rng(77).standard_normal(n))
    order = np.argsort(values)
    values, vectors = values[order], vectors[:, order]
    level = np.abs(values - values[0]) < 1e-8
    assert level.sum() == 3 and abs(values[0] - energy) < 1e-9
    space = np.linalg.qr(vectors[:, level])[0]
    again = space @ (space.T @ np.random.default_rng(0).standard_normal(n))
    again /= np.linalg.norm(again)
    again *= np.sign(again[np.argmax(np.abs(again))])
    assert np.abs(again - psi).max() < 1e-10

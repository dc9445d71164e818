"""Generate the committed golden vectors under tests/golden/ from the REFERENCE.

Run in the build container only (needs /root/reference and gcc):

    python tests/golden/generate_golden.py

What is executed from the reference:
  * cbits/build_matrix.c, compiled in place by oracle/build_oracle.py into
    oracle/_ref/ (git-ignored) -> build_matrix_*.npz, extract_signs.npz;
  * annealing_sign_problem/common.py, loaded with importlib after registering
    stub modules for the packages this image lacks (numba, loguru, h5py,
    lattice_symmetries, ising_glass_annealer, networkx) -> make_ising_*.npz,
    accuracy_overlap.npz, sparsify.npz.  The stubs carry no algorithm of the hot
    path: numba.njit is the identity decorator, the annealer stub only stores
    (exchange, field) and packs bits with the convention of
    cbits/build_matrix.c:72-74.
Nothing of the reference's text is written to disk: the .npz files hold inputs
and outputs only, models.json holds the bond lists / two-site matrices of the
symmetry-free physical_systems/*.yaml models as plain data.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np
import scipy.sparse

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"

import oracle  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
from helpers import random_build_case  # noqa: E402
from annealing_sign_problem_amd import operators  # noqa: E402


# ---------------------------------------------------------------------------
# reference loader
# ---------------------------------------------------------------------------

def _signs_to_bits(signs):
    signs = np.asarray(signs)
    n = signs.shape[0]
    out = np.zeros((n + 63) // 64, dtype=np.uint64)
    for i in np.nonzero(signs > 0)[0]:
        out[i // 64] |= np.uint64(1) << np.uint64(i % 64)
    return out


def _bits_to_signs(bits, count):
    bits = np.asarray(bits, dtype=np.uint64)
    i = np.arange(count, dtype=np.uint64)
    return 2.0 * ((bits[i // np.uint64(64)] >> (i % np.uint64(64))) & np.uint64(1)).astype(np.float64) - 1.0


def load_reference_common():
    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda fn: fn

    numba = types.ModuleType("numba")
    numba.njit = njit
    numba.prange = range

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    loguru = types.ModuleType("loguru")
    loguru.logger = _Logger()
    h5py = types.ModuleType("h5py")
    networkx = types.ModuleType("networkx")
    ls = types.ModuleType("lattice_symmetries")
    ls.Operator = object
    ls.SpinBasis = object
    ls.batched_index = lambda basis, spins: basis.batched_index(spins)

    class Hamiltonian:
        def __init__(self, exchange, field):
            # CSR: the reference slices it (common.py:674), which COO does not support
            self.exchange = scipy.sparse.csr_matrix(exchange)
            self.field = field

    sa = types.ModuleType("ising_glass_annealer")
    sa.Hamiltonian = Hamiltonian
    sa.signs_to_bits = _signs_to_bits
    sa.bits_to_signs = _bits_to_signs
    for name, mod in [("numba", numba), ("loguru", loguru), ("h5py", h5py), ("networkx", networkx),
                      ("lattice_symmetries", ls), ("ising_glass_annealer", sa)]:
        sys.modules.setdefault(name, mod)
    spec = importlib.util.spec_from_file_location(
        "reference_common", os.path.join(REFERENCE, "annealing_sign_problem", "common.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


# ---------------------------------------------------------------------------
# models.json
# ---------------------------------------------------------------------------

def export_models():
    import yaml

    models = {}
    for name in ["heisenberg_kagome_16", "j1j2_square_4x4", "sk_16_1", "sk_16_2", "sk_16_3",
                 "heisenberg_kagome_18", "heisenberg_kagome_36", "heisenberg_pyrochlore_2x2x2",
                 "sk_32_1"]:
        with open(os.path.join(REFERENCE, "physical_systems", name + ".yaml")) as f:
            cfg = yaml.load(f, Loader=yaml.SafeLoader)
        basis = {"number_spins": cfg["basis"]["number_spins"],
                 "hamming_weight": cfg["basis"]["hamming_weight"],
                 "symmetries": [{"permutation": s["permutation"], "sector": s["sector"]}
                                for s in cfg["basis"].get("symmetries") or []]}
        if cfg["basis"].get("spin_inversion") is not None:
            basis["spin_inversion"] = cfg["basis"]["spin_inversion"]
        models[name] = {
            "basis": basis,
            "hamiltonian": {"terms": [{"matrix": t["matrix"], "sites": t["sites"]}
                                      for t in cfg["hamiltonian"]["terms"]]},
        }
    path = os.path.join(ROOT, "annealing_sign_problem_amd", "models.json")
    with open(path, "w") as f:
        json.dump(models, f, separators=(",", ":"))
    return models


# ---------------------------------------------------------------------------
# build_matrix / extract_signs vectors from the reference C
# ---------------------------------------------------------------------------

def export_build_vectors():
    rng = np.random.default_rng(20261003)
    cases = {
        "hand": dict(spins=oracle.as_keys512([3, 5, 6, 9]), counts=np.array([1, 2, 1, 1]),
                     psi=np.array([.5, -.5, .5, -.5]), other_spins=oracle.as_keys512([5, 10, 3, 6, 3, 12]),
                     other_coeffs=np.array([2., 2., 2., -1., 2., 2.]), other_counts=np.array([2, 1, 1, 2]),
                     other_psi=np.array([-.5, .1, .5, .5, .5, -.2])),
        "word0": random_build_case(rng, 700, 9.0, False, 0.4, False),
        "multiword": random_build_case(rng, 400, 6.0, True, 0.5, True),
        "allmiss": random_build_case(rng, 50, 5.0, False, 1.0, False),
    }
    for name, c in cases.items():
        nnz, row, col, elements, field = oracle.ref_build_matrix(
            c["spins"], c["counts"], c["psi"], c["other_spins"], c["other_coeffs"],
            c["other_counts"], c["other_psi"])
        np.savez_compressed(os.path.join(HERE, "build_matrix_%s.npz" % name), **c, nnz=nnz,
                            row=row, col=col, elements=elements, field=field)
    psi = rng.normal(size=333)
    psi[::7] = 0.0
    psi[5] = np.nan
    psi[6] = -0.0
    np.savez_compressed(os.path.join(HERE, "extract_signs.npz"), psi=psi,
                        signs=oracle.ref_extract_signs(psi))


# ---------------------------------------------------------------------------
# vectors from the reference's common.py
# ---------------------------------------------------------------------------

def grow_cluster(op, start, size, keep_probability, rng):
    """Breadth-first cluster around `start` (a connected set of basis states)."""
    members = {start}
    frontier = [start]
    while len(members) < size and frontier:
        nxt = []
        for s0 in frontier:
            other, _ = op.apply(s0)
            for x in other[:, 0]:
                x = int(x)
                if x in members or rng.random() > keep_probability:
                    continue
                members.add(x)
                nxt.append(x)
                if len(members) >= size:
                    break
            if len(members) >= size:
                break
        frontier = nxt
    return np.array(sorted(members), dtype=np.uint64)


def export_common_vectors(ref, models):
    rng = np.random.default_rng(435834)
    cases = {}
    # (a) 4-site Heisenberg ring, full sector (SURVEY A.2 check: s^T J s = -8)
    ring = operators.Operator(operators.SpinBasis(4, 2), [operators.Term(
        operators.SIGMA_DOT_SIGMA, [(0, 1), (1, 2), (2, 3), (3, 0)])])
    ring.basis.build()
    e0, psi = ring.ground_state()
    cases["ring4"] = (ring, ring.basis.states, psi)
    # (b) kagome_16: a 400-state cluster sampled ~ |psi|^0.5 from the full ground state
    kag = operators.Operator.from_config(models["heisenberg_kagome_16"])
    kag.basis.build()
    e0, psi = kag.ground_state()
    cluster = grow_cluster(kag, int(kag.basis.states[np.argmax(np.abs(psi))]), 400, 0.5, rng)
    cases["kagome16_cluster"] = (kag, cluster, psi)
    # (c) sk_16_1 cluster: dense rows
    sk = operators.Operator.from_config(models["sk_16_1"])
    sk.basis.build()
    e0, psi_sk = sk.ground_state()
    cluster = np.sort(rng.choice(sk.basis.states, size=150, replace=False))
    cases["sk16_cluster"] = (sk, cluster, psi_sk)

    for name, (op, spins, psi) in cases.items():
        log_fn = ref.ground_state_to_log_coeff_fn(psi, op.basis)
        model = ref.make_ising_model(spins, op, log_psi_fn=log_fn)
        other_spins, other_coeffs, other_counts = ref._batched_apply(op, model.spins)
        m = scipy.sparse.coo_matrix(model.ising_hamiltonian.exchange)
        out = dict(spins=model.spins, log_psi=log_fn(model.spins), other_spins=other_spins,
                   other_coeffs=other_coeffs, other_counts=other_counts, row=m.row, col=m.col,
                   data=m.data, x0=model.initial_signs, ground_state=psi,
                   basis_states=op.basis.states)
        if name == "kagome16_cluster":
            # extension + sparsify, reference semantics (common.py:516-522, 647-692)
            ext = ref.make_hamiltonian_extension(model, log_fn)
            sp = ref.sparsify_using_global_cutoff(ext, 2e-3, model.spins)
            e = scipy.sparse.coo_matrix(ext.ising_hamiltonian.exchange)
            s = scipy.sparse.coo_matrix(sp.ising_hamiltonian.exchange)
            out.update(ext_spins=ext.spins, ext_row=e.row, ext_col=e.col, ext_data=e.data,
                       sp_spins=sp.spins, sp_row=s.row, sp_col=s.col, sp_data=s.data,
                       sp_x0=sp.initial_signs, sp_reltol=2e-3)
        np.savez_compressed(os.path.join(HERE, "make_ising_%s.npz" % name), **out)

    # accuracy / overlap (common.py:211-229)
    n = 333
    exact = _signs_to_bits(rng.choice([-1.0, 1.0], size=n))
    rows = []
    preds, accs, ovs = [], [], []
    w = rng.random(n)
    for flip in [0.0, 0.02, 0.5, 0.97, 1.0]:
        signs = _bits_to_signs(exact, n) * np.where(rng.random(n) < flip, -1.0, 1.0)
        pred = _signs_to_bits(signs)
        a, o = ref.compute_accuracy_and_overlap(pred, exact, w)
        a1, o1 = ref.compute_accuracy_and_overlap(pred, exact, number_spins=n)
        preds.append(pred); accs.append([a, a1]); ovs.append([o, o1])
    np.savez_compressed(os.path.join(HERE, "accuracy_overlap.npz"), exact=exact, weights=w,
                        predicted=np.array(preds), accuracy=np.array(accs), overlap=np.array(ovs),
                        number_spins=n)


if __name__ == "__main__":
    if not os.path.isdir(REFERENCE):
        raise SystemExit("reference checkout not present; golden vectors are committed")
    models = export_models()
    export_build_vectors()
    export_common_vectors(load_reference_common(), models)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))

"""Entry points of ``annealing_sign_problem.common`` on the MI355X path.

Same names, arguments and return values as the reference functions cited in
each docstring, so a caller of the reference (``process_cluster`` in
experiments/sampled_connected_components.py:726-751, the CLIs at
annealing_sign_problem/common.py:838-1002) can switch the import.  The two
numba kernels of the coupling build and the annealer run on the GPU through
libasp_hip.so; orchestration stays numpy/scipy exactly where the reference's is.
"""
from __future__ import annotations

import ctypes
import warnings
from dataclasses import dataclass
from typing import Any, Callable, Optional, Tuple

import numpy as np
import scipy.sparse

from . import _lib
from . import annealer as sa

__all__ = [
    "IsingModel",
    "make_ising_model",
    "solve_ising_model",
    "solve_ising_models",
    "dump_ising_model_to_hdf5",
    "load_ising_model_from_hdf5",
    "load_ground_state",
    "save_ground_state",
    "compute_accuracy_and_overlap",
    "make_hamiltonian_extension",
    "sparsify_using_global_cutoff",
    "get_strongest_off_diag",
    "binary_search",
    "monte_carlo_sampling",
    "add_noise_to_amplitudes",
    "ground_state_to_log_coeff_fn",
    "norm2",
    "SamplingResult",
]

APPLY_CHUNK = 10000  # rows per batched_apply call (common.py:85)


@dataclass
class IsingModel:
    """common.py:46-55."""

    spins: np.ndarray
    quantum_hamiltonian: Any
    ising_hamiltonian: sa.Hamiltonian
    initial_signs: np.ndarray
    # (not in the reference) positions of `spins` in the basis, when the amplitude closure that
    # built the model looked them up: later stages of the pipeline reuse them instead of asking
    # the basis again
    basis_index: Optional[np.ndarray] = None

    @property
    def size(self) -> int:
        return self.spins.shape[0]


BLAS_THREADING_THRESHOLD = 10000  # OpenBLAS splits dot products longer than this over its threads


def dot(a: np.ndarray, b: np.ndarray) -> float:
    """``np.dot`` of two real vectors, kept off the BLAS thread pool.  Up to 10 000 elements it
    IS ``np.dot`` (one BLAS kernel on the calling thread: the reference's arithmetic).  Longer
    vectors OpenBLAS hands to its thread pool: the summation order then depends on the number of
    its threads, the call takes the library's global lock, so that the pipeline's --jobs threads
    queue up behind it, and the pool's workers keep spinning on the cores afterwards — measured
    on the sampled-cluster pipeline: 5 to 30 ms per call and every OTHER stage 2.5x slower
    (profiles/r03_pipeline_stages.txt).  Those go through numpy's pairwise summation: the same
    value to the last bit or two, on every machine, in 0.03 ms."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape[0] <= BLAS_THREADING_THRESHOLD:
        return float(np.dot(a, b))
    return float(np.add.reduce(a * b))


def norm2(x: np.ndarray) -> float:
    """Euclidean norm of a real vector: ``np.linalg.norm`` (= sqrt(dot(x, x)), common.py:166)
    through :func:`dot`."""
    x = np.asarray(x, dtype=np.float64)
    if x.shape[0] <= BLAS_THREADING_THRESHOLD:
        return float(np.linalg.norm(x))
    return float(np.sqrt(np.add.reduce(x * x)))


def _normalize_spins(spins) -> np.ndarray:
    """1-D keys -> C-contiguous (n, 8) zero-padded uint64 (common.py:58-68)."""
    from ._build_matrix import as_bits512

    return as_bits512(spins)


def _on_device(hamiltonian) -> bool:
    """True for this package's symmetry-free operators with real matrices (the stand-in for
    ``ls.Operator``): their action has a HIP implementation.  Foreign operator objects keep
    the reference's route through their own ``batched_apply``."""
    from . import operators

    return isinstance(hamiltonian, operators.Operator) and hamiltonian.is_real


def _batched_apply(hamiltonian, spins, chunk_size: int = APPLY_CHUNK):
    """Chunked ``hamiltonian.batched_apply`` returning flat
    ``(other_spins u64[N], other_coeffs f64[N], other_counts i64[K])`` (common.py:85-106)."""
    if hamiltonian.basis.number_spins > 64:
        raise AssertionError("TODO: only works with up to 64 bits")
    if _on_device(hamiltonian):
        # this package's own operator: the action runs on the GPU in one call
        flat = np.ascontiguousarray(np.asarray(spins, dtype=np.uint64).reshape(len(spins), -1)[:, 0])
        return hamiltonian.device().apply(flat)
    keys, coeffs, counts = [], [], []
    total = spins.shape[0]
    for start in range(0, total, chunk_size):
        block = _normalize_spins(spins[start:start + chunk_size])
        other, c, n = hamiltonian.batched_apply(block)
        c = np.asarray(c)
        if not np.allclose(c.imag, 0, atol=1e-6):
            raise ValueError("expected all Hamiltonian matrix elements to be real")
        keys.append(np.ascontiguousarray(np.asarray(other)[:, 0], dtype=np.uint64))
        coeffs.append(np.ascontiguousarray(c.real, dtype=np.float64))
        counts.append(np.asarray(n, dtype=np.int64))
    if not keys:
        return (np.zeros(0, np.uint64), np.zeros(0, np.float64), np.zeros(0, np.int64))
    return np.hstack(keys), np.hstack(coeffs), np.hstack(counts)


def ising_elements(keys, psi, other_keys, other_coeffs, other_counts):
    """GPU pass over the connections: ``(other_indices i64[N], member bool[N],
    elements f64[N], offsets i64[K+1])`` — the fused equivalent of
    ``_clipped_search_sorted`` (common.py:116-128), the membership test (:173)
    and ``_make_ising_model_compute_elements`` (:71-82)."""
    lib = _lib.load()
    _lib.require_gpu()
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    psi = np.ascontiguousarray(psi, dtype=np.float64)
    other_keys = np.ascontiguousarray(other_keys, dtype=np.uint64)
    other_coeffs = np.ascontiguousarray(other_coeffs, dtype=np.float64)
    other_counts = np.ascontiguousarray(other_counts, dtype=np.int64)
    k, n = keys.shape[0], other_keys.shape[0]
    indices = np.zeros(max(n, 1), dtype=np.int64)
    member = np.zeros(max(n, 1), dtype=np.uint8)
    elements = np.zeros(max(n, 1), dtype=np.float64)
    offsets = np.zeros(k + 1, dtype=np.int64)
    _lib.check(lib.asp_ising_elements(
        ctypes.c_uint64(k), _lib.ptr(keys), _lib.ptr(psi), ctypes.c_uint64(n), _lib.ptr(other_keys),
        _lib.ptr(other_coeffs), _lib.ptr(other_counts), _lib.ptr(indices), _lib.ptr(member),
        _lib.ptr(elements), _lib.ptr(offsets)))
    return indices[:n], member[:n].astype(bool), elements[:n], offsets


def make_ising_model(
    spins,
    quantum_hamiltonian,
    log_psi: Optional[np.ndarray] = None,
    log_psi_fn: Optional[Callable[[np.ndarray], np.ndarray]] = None,
    external_field: bool = False,
    debug: bool = False,
) -> IsingModel:
    """Auxiliary Ising model of a set of basis states (common.py:131-208).

    ``J = (M + M^T) / 2`` with ``M_ij = H_ij |psi_i| |psi_j|`` over the states of
    the cluster (psi L2-normalised over the cluster), zero field, and the signs of
    psi as the initial configuration.  One deliberate difference: for NON-unique
    input the reference indexes the already-deduplicated array with first-occurrence
    positions of the original (common.py:149), which is an indexing slip; here the
    sorted unique states are kept and ``log_psi`` is taken at first occurrences.
    """
    if log_psi is None and log_psi_fn is None:
        raise ValueError("at least one of log_psi or log_psi_fn should be specified")
    if external_field and log_psi_fn is None:
        raise ValueError("log_psi_fn should be specified when external_field=True")
    if external_field:
        # the reference's branch is `assert False` (common.py:199-202)
        raise NotImplementedError("external_field=True is not implemented by the reference")

    spins = np.asarray(spins, dtype=np.uint64)
    if not (spins.ndim == 1 and (spins.shape[0] < 2 or bool(np.all(spins[1:] > spins[:-1])))):
        # (the pipeline's clusters and extensions arrive sorted and unique: no sort for them)
        spins, first, multiplicity = np.unique(spins, return_index=True, return_counts=True, axis=0)
        if np.any(multiplicity != 1):
            warnings.warn("'spins' were not unique, are you sure this is what you want?")
            if log_psi is not None:
                log_psi = np.asarray(log_psi)[first]
    basis_index = None
    if log_psi is None:
        if hasattr(log_psi_fn, "index_of"):  # (ground_state_to_log_coeff_fn's closure)
            basis_index = log_psi_fn.index_of(spins)
            log_psi = log_psi_fn.at_index(basis_index)
        else:
            log_psi = log_psi_fn(spins)
    if spins.ndim > 1:
        spins = np.ascontiguousarray(spins[:, 0])
    n = spins.shape[0]

    psi = np.exp(log_psi, dtype=np.complex128)
    if not np.allclose(psi.imag, 0, atol=1e-6):
        raise ValueError("expected all wavefunction coefficients to be real")
    psi = np.ascontiguousarray(psi.real)
    psi /= norm2(psi)

    matrix = None
    ising_hamiltonian = None
    if _on_device(quantum_hamiltonian):
        # action, search, elements and (M + M^T)/2 on the GPU, nothing materialised on the host
        # (csrc/operator_apply.hip): the pair-fused pass for operators whose rows reach distinct
        # states, the duplicate-keeping passes for symmetry-adapted bases.  Rows with duplicates
        # AND one-directional matrix elements are refused: host route below.
        try:
            # row-major, columns ascending, no duplicates (csrc/operator_apply.hip): canonical CSR
            indptr, col, val = quantum_hamiltonian.device().ising_csr(spins, psi)
            ising_hamiltonian = sa.Hamiltonian.from_canonical_csr(indptr, col, val, np.zeros(n))
        except _lib.AspError as error:
            if error.code != -3:
                raise
    if ising_hamiltonian is None:
        other_spins, other_coeffs, other_counts = _batched_apply(quantum_hamiltonian, spins)
        other_indices, _member, elements, offsets = ising_elements(
            spins, psi, other_spins, other_coeffs, other_counts)
        matrix = scipy.sparse.csr_matrix((elements, other_indices, offsets), shape=(n, n))
        matrix = 0.5 * (matrix + matrix.T)
        matrix.sort_indices()
        matrix = matrix.tocoo()
        ising_hamiltonian = sa.Hamiltonian(matrix, np.zeros(n, dtype=np.float64))
    x0 = sa.signs_to_bits(np.sign(psi))
    return IsingModel(spins, quantum_hamiltonian, ising_hamiltonian, x0, basis_index)


def compute_accuracy_and_overlap(
    predicted: np.ndarray,
    exact: np.ndarray,
    weights: Optional[np.ndarray] = None,
    number_spins: Optional[int] = None,
) -> Tuple[float, float]:
    """Sign accuracy and weighted overlap, both invariant under a global flip
    (common.py:211-229)."""
    if weights is None and number_spins is None:
        raise ValueError("'weights' and 'number_spins' cannot be both None")
    if number_spins is None:
        number_spins = len(weights)
    if weights is None:
        weights = np.ones(number_spins, dtype=np.float64)
    guess = sa.bits_to_signs(predicted, number_spins)
    truth = sa.bits_to_signs(exact, number_spins)
    accuracy = np.mean(truth == guess)
    accuracy = max(accuracy, 1 - accuracy)
    overlap = abs(np.dot(truth * guess, weights / np.sum(weights)))
    return accuracy, overlap


def binary_search(haystack, needles) -> np.ndarray:
    """Positions of ``needles`` in sorted ``haystack``; all must be present
    (common.py:544-548)."""
    haystack = np.asarray(haystack)
    needles = np.asarray(needles)
    assert haystack.shape[0] < 2 or np.all(haystack[1:] >= haystack[:-1])  # (sorted)
    indices = np.searchsorted(haystack, needles)
    assert np.all(haystack[np.minimum(indices, haystack.shape[0] - 1)] == needles)
    return indices


def solve_ising_model(
    model: IsingModel,
    mode: str = "sa",
    frozen_spins: Optional[np.ndarray] = None,
    seed: int = 12345,
    number_sweeps: int = 5120,
    repetitions: int = 64,
    only_best: bool = True,
    sweep_order: Optional[str] = None,
) -> np.ndarray:
    """Optimise the signs of ``model`` and project them onto ``frozen_spins``
    (common.py:232-261).  ``sweep_order`` (not in the reference): ``"shuffled"``, a fresh random
    order every sweep as in the reference's annealer — the default, so that the drop-in call has
    the reference's law —, or ``"colour"``, this package's fixed colour order (faster, a different
    chain; ``sa.anneal``); ``None`` = ``$ASP_SWEEP_ORDER`` or shuffled."""
    if mode == "sa":
        x, _ = sa.anneal(model.ising_hamiltonian, seed=seed, number_sweeps=number_sweeps,
                         repetitions=repetitions, only_best=only_best, sweep_order=sweep_order)
    elif mode == "greedy":
        x, _ = sa.greedy_solve(model.ising_hamiltonian)
    else:
        raise ValueError(
            "invalid mode specified: '{}'; expected either 'sa' or 'greedy'".format(mode))
    return _project_on_frozen(model, x, frozen_spins)


def _project_on_frozen(model: IsingModel, x: np.ndarray, frozen_spins) -> np.ndarray:
    """The signs of ``frozen_spins`` out of a solution of ``model`` (common.py:256-260)."""
    if frozen_spins is None:
        return x
    where = binary_search(model.spins, frozen_spins)
    signs = sa.bits_to_signs(x, count=model.spins.size)
    return sa.signs_to_bits(signs[where])


def solve_ising_models(models, frozen_spins=None, seed: int = 12345, number_sweeps: int = 5120,
                       repetitions: int = 64, sweep_order: Optional[str] = None):
    """``[solve_ising_model(m, "sa", f, seed, number_sweeps, repetitions) for m, f in
    zip(models, frozen_spins)]`` with all annealing chains of all models in ONE device call
    (``sa.anneal_batch``): the same result for every model, at the throughput of a full chip
    instead of one small launch per model."""
    models = list(models)
    frozen = [None] * len(models) if frozen_spins is None else list(frozen_spins)
    best = sa.anneal_batch([m.ising_hamiltonian for m in models], seed=seed,
                           number_sweeps=number_sweeps, repetitions=repetitions, only_best=True,
                           sweep_order=sweep_order)
    return [_project_on_frozen(m, x, f) for m, (x, _), f in zip(models, best, frozen)]


def make_hamiltonian_extension(model: IsingModel, log_psi_fn) -> IsingModel:
    """One-hop extension of a cluster: every state connected to it (common.py:516-522)."""
    if _on_device(model.quantum_hamiltonian):
        spins = model.quantum_hamiltonian.device().extend(model.spins)  # sorted and unique
    else:
        spins, _, _ = _batched_apply(model.quantum_hamiltonian, model.spins)
        spins = np.unique(spins, axis=0)
    return make_ising_model(spins, model.quantum_hamiltonian, log_psi_fn=log_psi_fn)


def get_strongest_off_diag(matrix) -> np.ndarray:
    """Per row, the largest |J_ij| with j != i (common.py:525-541)."""
    m = scipy.sparse.csr_matrix(matrix)
    rows = np.repeat(np.arange(m.shape[0]), np.diff(m.indptr))
    magnitude = np.where(rows != m.indices, np.abs(m.data), 0.0)
    out = np.zeros(m.shape[0], dtype=m.data.dtype)
    np.maximum.at(out, rows, magnitude)
    return out


def sparsify_component(exchange, is_frozen, reltol: float, anchor: int):
    """``(keep bool[K], block csr_matrix)`` through ``asp_sparsify_component``: cutoff, component
    of ``anchor`` and the un-pruned block on it, all on the GPU (csrc/sparsify.hip)."""
    lib = _lib.load()
    _lib.require_gpu()
    full = scipy.sparse.csr_matrix(exchange)
    if not full.has_sorted_indices:
        full = full.sorted_indices()
    full.sum_duplicates()
    k = full.shape[0]
    indptr = np.ascontiguousarray(full.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(full.indices, dtype=np.int32)
    data = np.ascontiguousarray(full.data, dtype=np.float64)
    frozen = np.ascontiguousarray(is_frozen, dtype=np.uint8)
    keep = np.zeros(max(k, 1), dtype=np.uint8)
    kept, nnz = ctypes.c_uint64(0), ctypes.c_uint64(0)
    capacity = int(data.shape[0])
    out_indptr = np.zeros(k + 1, dtype=np.int64)
    out_indices = np.empty(max(capacity, 1), dtype=np.int32)
    out_data = np.empty(max(capacity, 1), dtype=np.float64)
    _lib.check(lib.asp_sparsify_component(
        ctypes.c_uint64(k), _lib.ptr(indptr), _lib.ptr(indices), _lib.ptr(data), _lib.ptr(frozen),
        ctypes.c_double(float(reltol)), ctypes.c_uint64(int(anchor)), _lib.ptr(keep),
        ctypes.byref(kept), ctypes.c_uint64(capacity), _lib.ptr(out_indptr), _lib.ptr(out_indices),
        _lib.ptr(out_data), ctypes.byref(nnz)))
    n, z = int(kept.value), int(nnz.value)
    block = scipy.sparse.csr_matrix((out_data[:z].copy(), out_indices[:z].copy(),
                                     out_indptr[:n + 1].astype(np.int32)), shape=(n, n))
    block.has_sorted_indices = True  # (rows of a canonical matrix restricted to a subset)
    block.has_canonical_format = True
    return keep[:k].astype(bool), block


def sparsify_using_global_cutoff(model: IsingModel, reltol: float, frozen_spins) -> IsingModel:
    """Drop couplings below ``reltol * max|J|`` (unless both ends are frozen) and keep
    the connected component that holds the frozen spins (common.py:634-692)."""
    frozen = binary_search(model.spins, frozen_spins)
    is_frozen = np.zeros(model.spins.shape[0], dtype=bool)
    is_frozen[frozen] = True
    # NB: like the reference (common.py:674) the kept block comes from the UN-pruned matrix:
    # the cutoff only decides which spins survive.
    keep, exchange = sparsify_component(model.ising_hamiltonian.exchange, is_frozen, reltol,
                                        int(frozen[0]))
    spins = model.spins[keep]
    signs = sa.bits_to_signs(model.initial_signs, model.size)[keep]
    field = model.ising_hamiltonian.field[keep]
    hamiltonian = sa.Hamiltonian.from_canonical_csr(exchange.indptr, exchange.indices, exchange.data, field)
    basis_index = None if model.basis_index is None else model.basis_index[keep]
    return IsingModel(spins, model.quantum_hamiltonian, hamiltonian, sa.signs_to_bits(signs), basis_index)


def invert_permutation(p) -> np.ndarray:
    """``s`` with ``s[p[i]] = i`` (common.py:110-113)."""
    p = np.asarray(p)
    s = np.empty_like(p)
    s[p] = np.arange(p.size)
    return s


def _h5py_or_none():
    try:
        import h5py  # optional: not part of this image's main interpreter

        return h5py
    except Exception:
        return None


def dump_ising_model_to_hdf5(model: IsingModel, ground_state, filename: str) -> None:
    """The reference's dump of an Ising model (common.py:750-769), same dataset names, dtypes and
    layout: ``elements f64[nnz]``, ``indices i32[nnz]``, ``indptr i32[K+1]`` (CSR of J),
    ``field f64[K]``, ``energy`` (<psi|H|psi>, scalar f64) and ``signs u64[ceil(K/64)]`` at the
    root.  Written with h5py when it is importable, otherwise with :mod:`.hdf5_lite`."""
    matrix = scipy.sparse.csr_matrix(model.ising_hamiltonian.exchange)
    ground_state = np.asarray(ground_state, dtype=np.float64)
    energy = float(np.real(model.quantum_hamiltonian.expectation(ground_state)))
    content = {
        "elements": np.asarray(matrix.data, dtype=np.float64),
        "indices": np.asarray(matrix.indices, dtype=np.int32),
        "indptr": np.asarray(matrix.indptr, dtype=np.int32),
        "field": np.asarray(model.ising_hamiltonian.field, dtype=np.float64),
        "energy": np.float64(energy),
        "signs": sa.signs_to_bits(np.sign(ground_state)),
    }
    h5py = _h5py_or_none()
    if h5py is not None:
        with h5py.File(filename, "w") as out:
            for key, value in content.items():
                out[key] = value
        return
    from . import hdf5_lite

    hdf5_lite.write(filename, content)


def load_ising_model_from_hdf5(filename: str):
    """Inverse of :func:`dump_ising_model_to_hdf5`: ``(Hamiltonian, energy, signs)``."""
    h5py = _h5py_or_none()
    if h5py is not None:
        with h5py.File(filename, "r") as f:
            c = {k: np.asarray(f[k]) for k in ("elements", "indices", "indptr", "field", "energy", "signs")}
    else:
        from . import hdf5_lite

        c = hdf5_lite.read(filename)
    n = c["field"].shape[0]
    exchange = scipy.sparse.csr_matrix(
        (np.asarray(c["elements"], dtype=np.float64), np.asarray(c["indices"], dtype=np.int32),
         np.asarray(c["indptr"], dtype=np.int32)), shape=(n, n))
    return (sa.Hamiltonian(exchange, np.asarray(c["field"], dtype=np.float64)), float(c["energy"]),
            np.asarray(c["signs"], dtype=np.uint64))


def load_ground_state(filename: str):
    """``(ground_state f64[N], energy, basis_representatives u64[N])`` of a SpinED output file
    (common.py:772-780): ``/hamiltonian/eigenvectors`` (first vector), ``/hamiltonian/eigenvalues``
    and ``/basis/representatives``."""
    h5py = _h5py_or_none()
    if h5py is not None:
        with h5py.File(filename, "r") as f:
            vectors = np.asarray(f["/hamiltonian/eigenvectors"], dtype=np.float64)
            values = np.asarray(f["/hamiltonian/eigenvalues"], dtype=np.float64)
            representatives = np.asarray(f["/basis/representatives"], dtype=np.uint64)
    else:
        from . import hdf5_lite

        tree = hdf5_lite.read(filename)
        vectors = np.asarray(hdf5_lite.lookup(tree, "/hamiltonian/eigenvectors"), dtype=np.float64)
        values = np.asarray(hdf5_lite.lookup(tree, "/hamiltonian/eigenvalues"), dtype=np.float64)
        representatives = np.asarray(hdf5_lite.lookup(tree, "/basis/representatives"), dtype=np.uint64)
    ground_state = vectors.squeeze()
    if ground_state.ndim > 1:
        ground_state = ground_state[0, :]
    return np.ascontiguousarray(ground_state), float(values.reshape(-1)[0]), representatives


def save_ground_state(filename: str, ground_state, energy: float, representatives) -> None:
    """A file in the layout :func:`load_ground_state` (and the reference) reads."""
    content = {
        "hamiltonian": {"eigenvectors": np.asarray(ground_state, dtype=np.float64).reshape(1, -1),
                        "eigenvalues": np.asarray([energy], dtype=np.float64)},
        "basis": {"representatives": np.asarray(representatives, dtype=np.uint64)},
    }
    h5py = _h5py_or_none()
    if h5py is not None:
        with h5py.File(filename, "w") as out:
            for group, members in content.items():
                g = out.create_group(group)
                for key, value in members.items():
                    g[key] = value
        return
    from . import hdf5_lite

    hdf5_lite.write(filename, content)


def load_hamiltonian(filename: str):
    """The operator of a ``physical_systems/*.yaml`` file (common.py:783-788): plain bases and
    symmetry-adapted ones whose generators are in sector 0, with or without spin inversion (every
    shipped file); the result runs its action on the GPU.  Other sectors raise ``ValueError``."""
    import yaml

    from . import operators

    with open(filename, "r") as f:
        config = yaml.safe_load(f)
    return operators.Operator.from_config(config)


@dataclass
class SamplingResult:
    """common.py:264-267."""

    spins: np.ndarray
    weights: Optional[np.ndarray]


def monte_carlo_sampling(states, ground_state, number_samples: int,
                         sampled_power: float = 2) -> SamplingResult:
    """Seed states drawn with probability ~ |psi|^p from numpy's global legacy stream
    (common.py:270-279)."""
    p = np.abs(ground_state) ** sampled_power
    p /= np.sum(p)
    picks = np.random.choice(len(states), size=number_samples, replace=True, p=p)
    return SamplingResult(spins=np.asarray(states)[picks], weights=None)


def add_noise_to_amplitudes(ground_state, eps: float) -> np.ndarray:
    """psi -> sign(psi) |psi| exp(eps U(-1, 1)), renormalised (common.py:825-835)."""
    ground_state = np.asarray(ground_state, dtype=np.float64, order="C")
    assert ground_state.ndim == 1
    log_amplitude = np.log(np.abs(ground_state))
    noise = eps * 2 * (np.random.rand(log_amplitude.size) - 0.5)
    noisy = np.sign(ground_state) * np.exp(log_amplitude + noise)
    return noisy / np.linalg.norm(noisy)


def ground_state_to_log_coeff_fn(ground_state, basis):
    """Closure ``spins -> log|psi| + i pi [psi < 0]`` (common.py:806-822)."""
    ground_state = np.asarray(ground_state, dtype=np.float64, order="C")
    assert ground_state.ndim == 1
    with np.errstate(divide="ignore"):
        log_amplitude = np.log(np.abs(ground_state))
    phase = np.where(ground_state >= 0, 0, np.pi)

    def index_of(spins: np.ndarray) -> np.ndarray:
        spins = np.asarray(spins, dtype=np.uint64, order="C")
        if spins.ndim > 1:
            spins = spins[:, 0]
        return np.asarray(basis.batched_index(spins), dtype=np.int64)

    def at_index(where: np.ndarray) -> np.ndarray:
        return log_amplitude[where] + 1j * phase[where]

    def log_coeff_fn(spins: np.ndarray) -> np.ndarray:
        return at_index(index_of(spins))

    # the two halves of the closure, for callers that keep the positions (make_ising_model)
    log_coeff_fn.index_of = index_of
    log_coeff_fn.at_index = at_index
    return log_coeff_fn

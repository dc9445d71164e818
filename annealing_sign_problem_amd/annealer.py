"""The ``ising_glass_annealer`` surface the reference uses (``import
ising_glass_annealer as sa``; annealing_sign_problem/common.py:8), served by the
gfx950 sweep kernel through the C ABI:

    sa.Hamiltonian(exchange, field)      common.py:204,681
    sa.anneal(h, seed=, number_sweeps=, repetitions=, only_best=)
                                         common.py:242-248,
                                         experiments/full_hilbert_space.py:212-218
    sa.anneal(h, x0, seed=, number_sweeps=, beta0=, beta1=)   (legacy keywords)
                                         annealing_sign_problem/train.py:238-245,297
    sa.signs_to_bits / sa.bits_to_signs  common.py:205,224-225,258-260
    sa.greedy_solve(h)                   common.py:250

Conventions pinned by the reference: the solver MINIMISES
``E(s) = sum_ij J_ij s_i s_j + sum_i h_i s_i`` (full double sum, diagonal
included, no 1/2: common.py:757-760, experiments/full_hilbert_space.py:142-145);
bit ``i`` of word ``i // 64`` is set iff ``s_i = +1`` (cbits/build_matrix.c:72-74).
The annealing schedule, RNG and sweep order are this package's own
specification: the reference's annealer is a third-party library
whose internals are not available.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np
import scipy.sparse

from . import _lib

__all__ = [
    "Hamiltonian",
    "anneal",
    "anneal_batch",
    "greedy_solve",
    "signs_to_bits",
    "bits_to_signs",
    "make_schedule",
]


def signs_to_bits(signs) -> np.ndarray:
    """Pack a ±1 array: bit i of word i // 64 is set iff ``signs[i] > 0``."""
    signs = np.asarray(signs)
    n = signs.shape[0]
    positive = (signs > 0).astype(np.uint8)
    padded = np.zeros(((n + 63) // 64) * 64, dtype=np.uint8)
    padded[:n] = positive
    return np.packbits(padded.reshape(-1, 64), axis=1, bitorder="little").view("<u8").reshape(-1)


def bits_to_signs(bits, count: int) -> np.ndarray:
    """Unpack ``count`` spins to a float64 array of ±1."""
    bits = np.ascontiguousarray(bits, dtype="<u8").reshape(-1)
    count = int(count)
    if count > bits.shape[0] * 64:
        raise ValueError("'bits' holds fewer than {} spins".format(count))
    unpacked = np.unpackbits(bits.view(np.uint8), bitorder="little")[:count]
    return 2.0 * unpacked.astype(np.float64) - 1.0


def make_schedule(beta0: float, beta1: float, number_sweeps: int) -> np.ndarray:
    """Geometric inverse-temperature ladder beta0 -> beta1, one value per sweep."""
    number_sweeps = int(number_sweeps)
    if number_sweeps <= 0:
        return np.zeros(0, dtype=np.float64)
    if not (beta0 > 0 and beta1 > 0 and np.isfinite(beta0) and np.isfinite(beta1)):
        raise ValueError("beta0 and beta1 must be positive and finite")
    if number_sweeps == 1:
        return np.array([beta1], dtype=np.float64)
    return np.geomspace(beta0, beta1, number_sweeps).astype(np.float64)


class Hamiltonian:
    """Classical Ising Hamiltonian ``E(s) = s^T J s + h^T s`` resident on the GPU.

    ``exchange`` may be any scipy sparse matrix (the reference hands over COO,
    common.py:196,204); it is stored as canonical CSR so that the reference's
    later ``.tocoo()``, ``.tocsr()`` and ``[mask][:, mask]`` uses work
    (common.py:444,654,674).
    """

    def __init__(self, exchange, field):
        matrix = scipy.sparse.csr_matrix(exchange, dtype=np.float64, copy=True)  # (frozen below)
        if matrix.shape[0] != matrix.shape[1]:
            raise ValueError("'exchange' must be square, got {}".format(matrix.shape))
        matrix.sum_duplicates()
        matrix.sort_indices()
        field = np.array(field, dtype=np.float64, order="C", copy=True)
        if field.shape != (matrix.shape[0],):
            raise ValueError("'field' must have shape ({},)".format(matrix.shape[0]))
        # The device plan is built once from these arrays and cached (the key is the identity of
        # the two objects): they are frozen, so an in-place edit raises instead of leaving a stale
        # plan behind.  Assigning a NEW matrix or field to the attributes rebuilds the plan.
        for array in (matrix.data, matrix.indices, matrix.indptr, field):
            array.flags.writeable = False
        self.exchange = matrix
        self.field = field
        self._plan = None
        self._plan_key = None

    @classmethod
    def from_canonical_csr(cls, indptr, indices, data, field) -> "Hamiltonian":
        """A Hamiltonian from CSR arrays that ARE canonical already — rows in order, columns
        sorted and duplicate-free, as the device builders deliver them (asp_operator_ising,
        asp_sparsify_component) — without the copy, ``sum_duplicates`` and ``sort_indices`` of the
        general constructor (a quarter of the host time of a sampled-cluster run went there).  The
        arrays are adopted and frozen; ``asp_sa_plan_create`` still validates them."""
        indptr = np.ascontiguousarray(indptr)
        n = indptr.shape[0] - 1
        matrix = scipy.sparse.csr_matrix((np.ascontiguousarray(data, dtype=np.float64),
                                          np.ascontiguousarray(indices), indptr), shape=(n, n), copy=False)
        matrix.has_sorted_indices = True
        matrix.has_canonical_format = True
        field = np.ascontiguousarray(field, dtype=np.float64)
        if field.shape != (n,):
            raise ValueError("'field' must have shape ({},)".format(n))
        self = cls.__new__(cls)
        for array in (matrix.data, matrix.indices, matrix.indptr, field):
            array.flags.writeable = False
        self.exchange = matrix
        self.field = field
        self._plan = None
        self._plan_key = None
        return self

    @property
    def shape(self) -> Tuple[int, int]:
        return self.exchange.shape

    @property
    def size(self) -> int:
        return self.exchange.shape[0]

    # -- device plan -----------------------------------------------------------
    def _arrays(self):
        m = self.exchange
        return (
            np.ascontiguousarray(m.indptr, dtype=np.int64),
            np.ascontiguousarray(m.indices, dtype=np.int32),
            np.ascontiguousarray(m.data, dtype=np.float64),
            np.ascontiguousarray(self.field, dtype=np.float64),
        )

    def plan(self):
        """The (cached) device-resident sweep plan handle."""
        key = (id(self.exchange), id(self.field), self.exchange.nnz)
        if self._plan is not None and self._plan_key == key:
            return self._plan
        self.release()
        lib = _lib.load()
        _lib.require_gpu()
        indptr, indices, data, field = self._arrays()
        handle = lib.asp_sa_plan_create(ctypes.c_uint64(self.size), _lib.ptr(indptr),
                                        _lib.ptr(indices), _lib.ptr(data), _lib.ptr(field))
        if not handle:
            raise _lib.AspError(lib.asp_last_error_code(), _lib.last_error())
        self._plan = ctypes.c_void_p(handle)
        self._plan_key = key
        _lib.track(self)
        return self._plan

    def info(self) -> _lib.SaInfo:
        info = _lib.SaInfo()
        _lib.check(_lib.load().asp_sa_plan_info(self.plan(), ctypes.byref(info)))
        return info

    def release(self) -> None:
        """Destroy the device plan now (it is rebuilt on the next use)."""
        if self._plan is not None:
            try:
                _lib.load().asp_sa_plan_destroy(self._plan)
            finally:
                self._plan = None
                self._plan_key = None

    def __del__(self):
        # never reach into HIP from interpreter teardown: _lib.shutdown (atexit) has released
        # every live plan before that point
        if getattr(self, "_plan", None) is None or _lib.closed():
            return
        try:
            self.release()
        except Exception:
            pass

    # -- energy ----------------------------------------------------------------
    def energy(self, x) -> float:
        """E(x) of one packed configuration (experiments/full_hilbert_space.py:144)."""
        return float(self.energies(np.asarray(x).reshape(1, -1))[0])

    def energies(self, xs) -> np.ndarray:
        words = (self.size + 63) // 64
        xs = np.ascontiguousarray(xs, dtype=np.uint64).reshape(-1, max(words, 1))
        if xs.shape[1] != max(words, 1):
            raise ValueError("configurations must have {} words".format(words))
        out = np.zeros(xs.shape[0], dtype=np.float64)
        _lib.check(_lib.load().asp_sa_energy(self.plan(), ctypes.c_uint32(xs.shape[0]),
                                             _lib.ptr(xs), _lib.ptr(out)))
        return out


def resolve_sweep_order(sweep_order: Optional[str]) -> str:
    """``"shuffled"`` or ``"colour"``; ``None`` reads ``$ASP_SWEEP_ORDER`` and falls back to
    ``"shuffled"`` — the visiting order of the reference's annealer, so
    that every drop-in entry point (``sa.anneal``, ``common.solve_ising_model``, `make small`,
    ``sampled_components --annealing``) runs the chain with the reference's law unless told
    otherwise.  ``"colour"`` (ASP-SA-1's fixed order: faster here, a different Markov chain) is
    one keyword / flag / environment variable away."""
    if sweep_order is None:
        sweep_order = os.environ.get("ASP_SWEEP_ORDER") or "shuffled"
    if sweep_order not in ("colour", "shuffled"):
        raise ValueError("'sweep_order' must be 'colour' or 'shuffled'")
    return sweep_order


def _resolve_seed(seed) -> int:
    if seed is None:
        return int.from_bytes(os.urandom(8), "little")
    return int(seed) & (2**64 - 1)


def anneal_raw(hamiltonian: Hamiltonian, seed: int, betas: np.ndarray, repetitions: int,
               replica_offset: int = 0, x0=None, shuffled: bool = False):
    """All chains, no reduction: (xs[R, words] uint64, es[R] float64).  ``shuffled``: a fresh
    visiting order every sweep (``asp_sa_anneal_shuffled``) instead of the colour order."""
    lib = _lib.load()
    plan = hamiltonian.plan()
    words = (hamiltonian.size + 63) // 64
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    xs = np.zeros((repetitions, max(words, 1)), dtype=np.uint64)
    es = np.zeros(repetitions, dtype=np.float64)
    if x0 is not None:
        x0 = np.ascontiguousarray(x0, dtype=np.uint64).reshape(-1)
        if x0.shape[0] != words:
            raise ValueError("'x0' must have {} words".format(words))
    entry = lib.asp_sa_anneal_shuffled if shuffled else lib.asp_sa_anneal
    _lib.check(entry(plan, ctypes.c_uint64(seed), _lib.ptr(betas),
                     ctypes.c_uint32(betas.shape[0]), ctypes.c_uint32(repetitions),
                     ctypes.c_uint32(replica_offset), _lib.ptr(x0), _lib.ptr(xs), _lib.ptr(es)))
    return xs[:, :words], es


def anneal_raw_into(hamiltonian: Hamiltonian, seed: int, betas: np.ndarray, repetitions: int,
                    replica_offset: int, x0, out_x_ptr: int, out_e_ptr: int,
                    shuffled: bool = False) -> None:
    """``anneal_raw`` writing into caller-owned memory given as raw addresses — host or DEVICE
    (e.g. ``tensor.data_ptr()`` of torch tensors on this library's GPU): ``out_x`` receives
    ``repetitions * ceil(K/64)`` words, ``out_e`` ``repetitions`` doubles."""
    lib = _lib.load()
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    if x0 is not None:
        x0 = np.ascontiguousarray(x0, dtype=np.uint64).reshape(-1)
        if x0.shape[0] != (hamiltonian.size + 63) // 64:
            raise ValueError("'x0' must have {} words".format((hamiltonian.size + 63) // 64))
    entry = lib.asp_sa_anneal_shuffled if shuffled else lib.asp_sa_anneal
    _lib.check(entry(hamiltonian.plan(), ctypes.c_uint64(seed), _lib.ptr(betas),
                     ctypes.c_uint32(betas.shape[0]), ctypes.c_uint32(repetitions),
                     ctypes.c_uint32(replica_offset), _lib.ptr(x0),
                     ctypes.c_void_p(out_x_ptr), ctypes.c_void_p(out_e_ptr)))


def anneal(hamiltonian: Hamiltonian, x0=None, seed=None, number_sweeps: int = 5120,
           beta0: Optional[float] = None, beta1: Optional[float] = None, repetitions: int = 1,
           only_best: bool = True, distributed: bool = True, sweep_order: Optional[str] = None):
    """Simulated annealing of ``hamiltonian``.

    Returns ``(x, e)``: with ``only_best=True`` the best packed configuration
    (``uint64[ceil(K/64)]``) over all repetitions and its energy; with
    ``only_best=False`` the per-repetition arrays ``(xs[R, words], es[R])``
    (zip-able, experiments/full_hilbert_space.py:176).

    ``sweep_order="shuffled"`` (default): a fresh random visiting order every sweep — what the
    reference's ``ising_glass_annealer`` does as far as its published success probabilities can
    tell: statistically the library's behaviour.
    ``sweep_order="colour"``: the fixed colour order of specification ASP-SA-1 — a different
    Markov chain (a HIGHER success probability per sweep than the published curves) and several
    times the rate on this hardware; the headline kernel of ``bench.py``.  ``None``: the value of
    ``$ASP_SWEEP_ORDER`` if set, else ``"shuffled"``.

    When ``torch.distributed`` is initialised with more than one rank (and
    ``distributed`` is true) the repetitions are sharded over the ranks and
    gathered at the end (see :mod:`.distributed`); every rank returns the full
    result, identical to a single-GPU run with the same seed.
    """
    if not isinstance(hamiltonian, Hamiltonian):
        raise TypeError("'hamiltonian' must be a Hamiltonian")
    repetitions = int(repetitions)
    if repetitions < 1:
        raise ValueError("'repetitions' must be positive")
    from . import distributed as _dist  # late import: torch is optional plumbing

    shuffled = resolve_sweep_order(sweep_order) == "shuffled"
    sharded = distributed and _dist.shards_chains()
    seed = _dist.agree_on_seed(seed) if sharded else _resolve_seed(seed)
    if beta0 is None or beta1 is None:
        info = hamiltonian.info()
        beta0 = info.beta0_auto if beta0 is None else beta0
        beta1 = info.beta1_auto if beta1 is None else beta1
    betas = make_schedule(float(beta0), float(beta1), number_sweeps)

    if sharded and only_best:
        return _dist.anneal_sharded_best(hamiltonian, seed, betas, repetitions, x0, shuffled=shuffled)
    if sharded:
        xs, es = _dist.anneal_sharded(hamiltonian, seed, betas, repetitions, x0, shuffled=shuffled)
    else:
        xs, es = anneal_raw(hamiltonian, seed, betas, repetitions, 0, x0, shuffled=shuffled)
    if only_best:
        best = int(np.argmin(es))  # first minimum: deterministic tie-break
        return xs[best].copy(), float(es[best])
    return xs, es


def anneal_batch_raw(hamiltonians, seeds, schedules, repetitions, replica_offsets=None,
                     shuffled: bool = False):
    """Many independent problems in ONE device call (``asp_sa_anneal_batch``): problem ``i`` is
    ``anneal_raw(hamiltonians[i], seeds[i], schedules[i], repetitions[i], replica_offsets[i])``,
    chain for chain, but the groups of all problems share a few launches, so a batch of small
    clusters fills the chip.  ``shuffled``: every problem is an ``asp_sa_anneal_shuffled`` call
    (a fresh visiting order every sweep); their kernels overlap on the problems' own streams.
    Returns ``[(xs, es), ...]`` in order."""
    lib = _lib.load()
    n = len(hamiltonians)
    repetitions = [int(r) for r in (repetitions if np.ndim(repetitions) else [repetitions] * n)]
    offsets = [0] * n if replica_offsets is None else [int(o) for o in replica_offsets]
    if not (len(seeds) == len(schedules) == len(repetitions) == len(offsets) == n):
        raise ValueError("anneal_batch_raw: argument lengths differ")
    if len({id(h) for h in hamiltonians}) != n:
        raise ValueError("anneal_batch_raw: every problem needs its own Hamiltonian object")
    items = (_lib.SaBatchItem * max(n, 1))()
    keep, out = [], []
    for i, h in enumerate(hamiltonians):
        words = (h.size + 63) // 64
        betas = np.ascontiguousarray(schedules[i], dtype=np.float64)
        xs = np.zeros((repetitions[i], max(words, 1)), dtype=np.uint64)
        es = np.zeros(repetitions[i], dtype=np.float64)
        keep.append(betas)
        out.append((xs, es, words))
        items[i].plan = h.plan()
        items[i].seed = int(seeds[i]) & (2**64 - 1)
        items[i].betas = betas.ctypes.data
        items[i].num_sweeps = betas.shape[0]
        items[i].repetitions = repetitions[i]
        items[i].replica_offset = offsets[i]
        items[i].flags = _lib.SA_BATCH_SHUFFLED if shuffled else 0
        items[i].out_x = xs.ctypes.data
        items[i].out_e = es.ctypes.data
    _lib.check(lib.asp_sa_anneal_batch(items, ctypes.c_uint32(n)))
    return [(xs[:, :words], es) for xs, es, words in out]


def anneal_batch(hamiltonians, seed=None, number_sweeps: int = 5120, repetitions: int = 64,
                 only_best: bool = True, beta0: Optional[float] = None,
                 beta1: Optional[float] = None, sweep_order: Optional[str] = None):
    """``[anneal(h, seed=seed, number_sweeps=..., repetitions=..., only_best=...) for h in
    hamiltonians]`` in one device call — identical results (each problem keeps its own automatic
    ladder and the same chains), a fraction of the time for many small problems.  ``seed`` may be
    one value for all problems (what the reference's per-cluster loop passes, common.py:236) or
    a sequence.  Chains stay on this rank (cluster instances are what shards over ranks)."""
    hamiltonians = list(hamiltonians)
    n = len(hamiltonians)
    for h in hamiltonians:
        if not isinstance(h, Hamiltonian):
            raise TypeError("'hamiltonians' must hold Hamiltonian objects")
    repetitions = int(repetitions)
    if repetitions < 1:
        raise ValueError("'repetitions' must be positive")
    if seed is None or np.ndim(seed) == 0:
        seeds = [_resolve_seed(seed) for _ in range(n)] if seed is None else [_resolve_seed(seed)] * n
    else:
        seeds = [_resolve_seed(x) for x in seed]
    schedules = []
    for h in hamiltonians:
        b0, b1 = beta0, beta1
        if b0 is None or b1 is None:
            info = h.info()
            b0 = info.beta0_auto if b0 is None else b0
            b1 = info.beta1_auto if b1 is None else b1
        schedules.append(make_schedule(float(b0), float(b1), number_sweeps))
    results = anneal_batch_raw(hamiltonians, seeds, schedules, [repetitions] * n,
                               shuffled=resolve_sweep_order(sweep_order) == "shuffled")
    if not only_best:
        return results
    best = []
    for xs, es in results:
        k = int(np.argmin(es))  # first minimum, as anneal()
        best.append((xs[k].copy(), float(es[k])))
    return best


def anneal_trace_raw(hamiltonian: Hamiltonian, seed: int, betas: np.ndarray, repetitions: int,
                     replica_offset: int = 0, x0=None):
    """``anneal_raw`` plus ``trace int64[R, T+1]``: tracked energy of every chain after each
    sweep in units of ``2**-energy_scale_exp``, relative to its initial configuration."""
    lib = _lib.load()
    plan = hamiltonian.plan()
    words = (hamiltonian.size + 63) // 64
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    xs = np.zeros((repetitions, max(words, 1)), dtype=np.uint64)
    es = np.zeros(repetitions, dtype=np.float64)
    trace = np.zeros((repetitions, betas.shape[0] + 1), dtype=np.int64)
    if x0 is not None:
        x0 = np.ascontiguousarray(x0, dtype=np.uint64).reshape(-1)
        if x0.shape[0] != words:
            raise ValueError("'x0' must have {} words".format(words))
    _lib.check(lib.asp_sa_anneal_trace(plan, ctypes.c_uint64(seed), _lib.ptr(betas),
                                       ctypes.c_uint32(betas.shape[0]),
                                       ctypes.c_uint32(repetitions),
                                       ctypes.c_uint32(replica_offset), _lib.ptr(x0),
                                       _lib.ptr(xs), _lib.ptr(es), _lib.ptr(trace)))
    return xs[:, :words], es, trace


def anneal_with_traces(hamiltonian: Hamiltonian, x0=None, seed=None, number_sweeps: int = 5120,
                       beta0: Optional[float] = None, beta1: Optional[float] = None):
    """The older annealer API (annealing_sign_problem/train.py:238-245, square_deep.py:181-183):
    one chain, ``(x, e_current, e_best)`` with the energy after every sweep and the best energy
    so far (``e_best[0]`` before the first sweep, ``e_best[-1]`` the returned configuration's).

    The traces come from the kernel's exact integer bookkeeping of the accepted ``dE`` and are
    anchored at the returned configuration's energy, which is recomputed in full precision."""
    if not isinstance(hamiltonian, Hamiltonian):
        raise TypeError("'hamiltonian' must be a Hamiltonian")
    info = hamiltonian.info()
    beta0 = info.beta0_auto if beta0 is None else beta0
    beta1 = info.beta1_auto if beta1 is None else beta1
    betas = make_schedule(float(beta0), float(beta1), number_sweeps)
    xs, es, trace = anneal_trace_raw(hamiltonian, _resolve_seed(seed), betas, 1, 0, x0)
    unit = 2.0 ** -info.energy_scale_exp
    best = np.minimum.accumulate(trace[0])
    e_best = es[0] + (best - best[-1]).astype(np.float64) * unit
    e_current = es[0] + (trace[0] - best[-1]).astype(np.float64) * unit
    return xs[0].copy(), e_current, e_best


def greedy_solve(hamiltonian: Hamiltonian):
    """Strongest-coupling-first greedy sign assignment (common.py:250)."""
    from .greedy import greedy_solve as _greedy

    return _greedy(hamiltonian)

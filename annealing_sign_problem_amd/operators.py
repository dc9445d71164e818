"""Spin-1/2 bases (plain and sector-0 symmetry-adapted) and two-site operators in numpy.

The reference gets these from ``lattice_symmetries`` (annealing_sign_problem/
common.py:9; ``ls.SpinBasis``, ``ls.Operator``), a third-party C library that is
not part of the reference tree.  This module restates the subset the coupling
build consumes — the call surface at common.py:29,38,86,96,283,786-787,817 — for
bases without lattice symmetries (``symmetries: []``, e.g.
physical_systems/heisenberg_kagome_16.yaml:1-4) and for symmetry-adapted bases whose
generators are all in sector 0, with or without spin inversion (heisenberg_kagome_18.yaml:4,
heisenberg_kagome_36.yaml:7-29, heisenberg_pyrochlore_2x2x2.yaml:1-17; :mod:`.symmetry`).
The numpy code here is the host reference (exact diagonalisation, tests); the same action
runs on the GPU through :class:`DeviceOperator` (csrc/operator_apply.hip).

Conventions (unpinned against lattice_symmetries, which is unavailable): bit
``i`` of a basis state is site ``i``, 1 = up; a two-site matrix acts on
``|b_i b_j>`` with row/column index ``2*b_i + b_j``.  All shipped matrices are
``c * sigma.sigma`` (symmetric under both site exchange and spin inversion)
apart from the (3,3) entry of the J2 term in j1j2_square_4x4.yaml:22-25.
"""
from __future__ import annotations

import threading
from dataclasses import dataclass
from itertools import combinations
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse

SIGMA_DOT_SIGMA = np.array(
    [[1.0, 0.0, 0.0, 0.0], [0.0, -1.0, 2.0, 0.0], [0.0, 2.0, -1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]
)


class SpinBasis:
    """All bit strings of ``number_spins`` sites, optionally at fixed hamming weight; with a
    :class:`~.symmetry.SymmetryGroup` the basis states are the orbit REPRESENTATIVES of
    non-zero norm (the smallest state of each orbit)."""

    #: candidate states up to which :meth:`build` enumerates on the host
    HOST_LIMIT = 5_000_000
    #: basis size from which :meth:`batched_index` bisects on the GPU
    DEVICE_INDEX_LIMIT = 2_000_000

    def __init__(self, number_spins: int, hamming_weight: Optional[int] = None, group=None):
        if not 0 < number_spins <= 64:
            raise ValueError("number_spins must be in 1..64")
        self.number_spins = int(number_spins)
        self.hamming_weight = None if hamming_weight is None else int(hamming_weight)
        self.group = None if (group is None or group.is_trivial) else group
        self._states: Optional[np.ndarray] = None
        # the HBM copy of the state list behind batched_index (csrc/key_table.hip): created once,
        # under a lock — the first calls come from the --jobs worker threads of the pipeline
        self._table = None
        self._table_of = None
        self._lib_module = None
        self._table_lock = threading.Lock()
        self._table_users = 0

    def build(self, representatives: Optional[np.ndarray] = None) -> None:
        if representatives is not None:
            states = np.asarray(representatives, dtype=np.uint64)
            ascending = states.shape[0] < 2 or bool(np.all(states[1:] > states[:-1]))
            self._states = states if ascending else np.sort(states)
            return
        n, w = self.number_spins, self.hamming_weight
        count = 1 << n
        if w is not None:
            count = 1
            for k in range(w):
                count = count * (n - k) // (k + 1)
        if count > self.HOST_LIMIT:
            # beyond numpy: the representatives are listed on the GPU (csrc/sector_basis.hip;
            # no CPU fallback — 9e9 states of the 36-site kagome model take half a second there)
            from . import sector_ed

            lattice_only = Operator(self, [])
            reps, _ = sector_ed.enumerate_sector(lattice_only)
            self._states = reps.cpu().numpy().view(np.uint64)
            lattice_only.release_device()
            return
        if w is None:
            states = np.arange(1 << n, dtype=np.uint64)
            self._states = states if self.group is None else self.group.representatives(states)
            return
        states = np.fromiter(
            (sum(1 << b for b in bits) for bits in combinations(range(n), w)),
            dtype=np.uint64, count=count)
        states.sort()
        if self.group is not None:
            states = self.group.representatives(states)
        self._states = states

    @property
    def states(self) -> np.ndarray:
        if self._states is None:
            raise RuntimeError("basis has not been built")
        return self._states

    @property
    def number_states(self) -> int:
        return int(self.states.shape[0])

    def batched_index(self, spins) -> np.ndarray:
        spins = np.asarray(spins, dtype=np.uint64)
        if spins.ndim > 1:
            spins = spins[:, 0]
        if self.number_states > self.DEVICE_INDEX_LIMIT:
            return self._device_index(np.ascontiguousarray(spins))
        idx = np.searchsorted(self.states, spins)
        clipped = np.minimum(idx, self.number_states - 1)
        if not np.array_equal(self.states[clipped], spins):
            raise ValueError("state does not belong to the basis")
        return idx.astype(np.uint64)

    def _device_index(self, spins: np.ndarray) -> np.ndarray:
        """The same through the list kept in HBM (csrc/key_table.hip): numpy's searchsorted over
        tens of millions of representatives is all cache misses."""
        import ctypes

        from . import _lib

        lib = _lib.load()
        with self._table_lock:
            if self._table is None or self._table_of is not self._states:
                _lib.require_gpu()
                if self._table_users:
                    raise RuntimeError("the basis was rebuilt while batched_index calls were running")
                self._release_table_locked()
                table = ctypes.c_void_p()
                _lib.check(lib.asp_table_create(self.number_states, _lib.ptr(self._states), ctypes.byref(table)))
                self._table, self._table_of, self._lib_module = table, self._states, _lib
                _lib.track(self)
            table = self._table
            self._table_users += 1  # release_table() waits for no one: it refuses while in use
        try:
            idx = np.empty(max(spins.shape[0], 1), dtype=np.int64)
            _lib.check(lib.asp_table_index(table, spins.shape[0], _lib.ptr(spins), _lib.ptr(idx)))
        finally:
            with self._table_lock:
                self._table_users -= 1
        idx = idx[: spins.shape[0]]
        if np.any(idx < 0):
            raise ValueError("state does not belong to the basis")
        return idx.astype(np.uint64)

    def _release_table_locked(self) -> None:
        table, self._table = self._table, None
        self._table_of = None
        if table:
            self._lib_module.load().asp_table_destroy(table)

    def release_table(self) -> None:
        """Frees the HBM copy (it comes back on the next ``batched_index``).  A table that a
        running ``batched_index`` call of another thread is using stays."""
        lock = getattr(self, "_table_lock", None)
        if lock is None:
            return
        with lock:
            if self._table_users == 0:
                self._release_table_locked()

    release = release_table  # (the library's shutdown hook calls `release` on tracked objects)

    def __del__(self):
        # (no imports here: at interpreter teardown the import machinery is already gone)
        module = getattr(self, "_lib_module", None)
        if module is not None and getattr(self, "_table", None) and not module.closed():
            try:
                self.release_table()
            except Exception:
                pass

    def index(self, spin) -> int:
        return int(self.batched_index(np.array([spin], dtype=np.uint64))[0])


@dataclass
class Term:
    matrix: np.ndarray  # (4, 4)
    sites: List[Tuple[int, int]]


class Operator:
    """Sum of two-site terms on a :class:`SpinBasis`."""

    #: basis size up to which :meth:`ground_state` diagonalises on the host (scipy)
    HOST_ED_LIMIT = 200_000

    def __init__(self, basis: SpinBasis, terms: Sequence[Term]):
        self.basis = basis
        self.terms = [Term(np.asarray(t.matrix, dtype=np.complex128).reshape(4, 4),
                           [(int(a), int(b)) for a, b in t.sites]) for t in terms]
        for t in self.terms:
            for a, b in t.sites:
                if not (0 <= a < basis.number_spins and 0 <= b < basis.number_spins and a != b):
                    raise ValueError("invalid bond ({}, {})".format(a, b))

    @classmethod
    def from_config(cls, config: dict) -> "Operator":
        """``{"basis": {...}, "hamiltonian": {"terms": [{"matrix", "sites"}]}}``: the
        schema of physical_systems/*.yaml (lattice symmetries in sector 0, spin inversion +-1)."""
        from . import symmetry

        b = config["basis"]
        basis = SpinBasis(b["number_spins"], b.get("hamming_weight"), symmetry.group_from_config(b))
        terms = [Term(np.asarray(t["matrix"]), [tuple(s) for s in t["sites"]])
                 for t in config["hamiltonian"]["terms"]]
        return cls(basis, terms)

    # -- the same action on the GPU --------------------------------------------
    def bond_table(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """``(site_a u8[B], site_b u8[B], matrices f64[B, 16])`` in term order, then site-pair
        order: the arguments of ``asp_operator_create`` (include/asp.h)."""
        a = [x for t in self.terms for x, _ in t.sites]
        b = [y for t in self.terms for _, y in t.sites]
        m = [t.matrix.real.reshape(16) for t in self.terms for _ in t.sites]
        return (np.asarray(a, dtype=np.uint8), np.asarray(b, dtype=np.uint8),
                np.ascontiguousarray(np.asarray(m, dtype=np.float64).reshape(len(a), 16)))

    @property
    def is_real(self) -> bool:
        return all(not np.any(t.matrix.imag) for t in self.terms)

    def device(self) -> "DeviceOperator":
        """HIP-side twin (created on first use; needs a GPU and real matrices)."""
        if getattr(self, "_device", None) is None:
            if not self.is_real:
                raise ValueError("the HIP operator needs real matrices")
            self._device = DeviceOperator(self.basis.number_spins, *self.bond_table(),
                                          group=self.basis.group)
        return self._device

    def release_device(self) -> None:
        device, self._device = getattr(self, "_device", None), None
        if device is not None:
            device.release()

    # -- action on basis states ------------------------------------------------
    def batched_apply(self, spins):
        """``(other_spins (m, 8) u64, coeffs c128[m], counts i64[n])``: per input state one
        diagonal entry followed by its non-zero off-diagonal connections in term
        order (what the reference unpacks at common.py:96-103)."""
        spins = np.asarray(spins, dtype=np.uint64)
        if spins.ndim > 1:
            spins = spins[:, 0]
        n = spins.shape[0]
        rows = [np.arange(n, dtype=np.int64)]
        targets = [spins.copy()]
        diagonal = np.zeros(n, dtype=np.complex128)
        coeffs: List[np.ndarray] = [diagonal]
        one = np.uint64(1)
        for term in self.terms:
            m = term.matrix
            for a, b in term.sites:
                ba = (spins >> np.uint64(a)) & one
                bb = (spins >> np.uint64(b)) & one
                k = (2 * ba + bb).astype(np.int64)
                diagonal += m[k, k]
                for src in range(4):
                    for dst in range(4):
                        if src == dst or m[dst, src] == 0:
                            continue
                        sel = np.nonzero(k == src)[0]
                        if sel.size == 0:
                            continue
                        flip = np.uint64((((src ^ dst) >> 1) & 1) << a | ((src ^ dst) & 1) << b)
                        rows.append(sel)
                        targets.append(spins[sel] ^ flip)
                        coeffs.append(np.full(sel.size, m[dst, src], dtype=np.complex128))
        row = np.concatenate(rows)
        order = np.argsort(row, kind="stable")
        out = np.zeros((row.shape[0], 8), dtype=np.uint64)
        out[:, 0] = np.concatenate(targets)[order]
        counts = np.bincount(row, minlength=n).astype(np.int64)
        values = np.concatenate(coeffs)[order]
        if self.basis.group is not None:
            # symmetry-adapted basis: every target is replaced by its representative and the
            # coefficient by c * chi(g) * norm(target) / norm(source)   (symmetry.py)
            group = self.basis.group
            _, _, source_norm = group.state_info(spins)
            if np.any(source_norm == 0):
                raise ValueError("a state outside the symmetry sector was passed to batched_apply")
            rep, character, norm = group.state_info(out[:, 0])
            out[:, 0] = rep
            scale, source = character * norm, np.repeat(source_norm, counts)
            if not np.any(values.imag):
                # real arithmetic, (c * (chi * norm)) / norm(source) — the device kernel's
                # sequence (numpy's complex division multiplies by a reciprocal instead)
                values = ((values.real * scale) / source).astype(np.complex128)
            else:
                values = values * scale / source
        return out, values, counts

    def apply(self, spin):
        other, coeffs, _ = self.batched_apply(np.array([spin], dtype=np.uint64))
        return other, coeffs

    # -- dense-vector view (exact diagonalisation of 16-site systems) -------------
    def to_sparse(self) -> scipy.sparse.csr_matrix:
        """Matrix in the built basis; rows = output states."""
        states = self.basis.states
        other, coeffs, counts = self.batched_apply(states)
        cols = np.repeat(np.arange(states.shape[0]), counts)
        rows = np.searchsorted(states, other[:, 0])
        ok = (rows < states.shape[0])
        ok[ok] &= states[rows[ok]] == other[ok, 0]
        # (in a symmetry sector with a character -1 a connection can point at an orbit of zero
        # norm: its coefficient is exactly 0 and it is simply absent from the basis)
        if not (ok | (coeffs == 0)).all():
            raise ValueError("operator leaves the basis")
        rows, cols, coeffs = rows[ok], cols[ok], coeffs[ok]
        matrix = scipy.sparse.coo_matrix((coeffs, (rows, cols)), shape=(states.shape[0],) * 2)
        return matrix.tocsr()

    def expectation(self, vector) -> complex:
        vector = np.asarray(vector)
        return complex(np.vdot(vector, self.to_sparse() @ vector) / np.vdot(vector, vector))

    def ground_state(self, seed: int = 0) -> Tuple[float, np.ndarray]:
        """Lowest eigenpair (real symmetric operators), sign-fixed so that the
        largest-magnitude amplitude is positive."""
        import scipy.sparse.linalg

        if self.basis._states is None:
            self.basis.build()
        if self.basis.number_states > self.HOST_ED_LIMIT:
            # whole sectors of the 32- and 36-site models: enumeration, matrix and Lanczos on the
            # GPU (sector_ed.py; no CPU fallback)
            from . import sector_ed

            energy, psi, _, _ = sector_ed.ground_state(self, seed=seed, representatives=self.basis.states)
            return energy, psi
        h = self.to_sparse()
        if abs(h.imag).max() > 1e-12:
            raise ValueError("ground_state expects a real operator")
        h = h.real.tocsr()
        rng = np.random.default_rng(seed)
        v0 = rng.standard_normal(h.shape[0])
        values, vectors = scipy.sparse.linalg.eigsh(h, k=1, which="SA", v0=v0, tol=1e-13)
        psi = vectors[:, 0]
        if self.basis.group is not None and h.shape[0] > 8:
            # A symmetry sector can have a DEGENERATE lowest level (heisenberg_kagome_18.yaml: three
            # times -31.0548143836).  Which vector of that eigenspace an eigensolver returns
            # depends on its arithmetic — another BLAS, another vector, another sign problem
            # (DESIGN.md §6.1).  The vector is therefore fixed by construction: the projection of
            # the start vector v0 onto the eigenspace.
            k = min(6, h.shape[0] - 2)
            several, basis = scipy.sparse.linalg.eigsh(h, k=k, which="SA", v0=v0, tol=1e-13)
            order = np.argsort(several)
            several, basis = several[order], basis[:, order]
            level = np.abs(several - several[0]) <= 1e-8 * max(1.0, abs(several[0]))
            if level.sum() > 1:
                if level.all():
                    raise ValueError("the lowest level is at least {}-fold degenerate".format(k))
                space = np.linalg.qr(basis[:, level])[0]
                psi = space @ (space.T @ v0)
                values = several[:1]
        psi = psi * np.sign(psi[np.argmax(np.abs(psi))])
        return float(values[0]), np.ascontiguousarray(psi / np.linalg.norm(psi))


class DeviceOperator:
    """``asp_operator`` handle: Hamiltonian action, fused coupling build and one-hop extension
    on the GPU (csrc/operator_apply.hip).  No CPU fallback: construction fails without a GPU."""

    def __init__(self, number_spins: int, site_a: np.ndarray, site_b: np.ndarray,
                 matrices: np.ndarray, group=None):
        import ctypes

        from . import _lib

        self._lib_module = _lib
        self._lib = _lib.load()
        _lib.require_gpu()
        handle = ctypes.c_void_p()
        _lib.check(self._lib.asp_operator_create(
            int(number_spins), int(site_a.shape[0]), _lib.ptr(site_a), _lib.ptr(site_b),
            _lib.ptr(matrices), ctypes.byref(handle)))
        self._handle = handle
        if group is not None:
            table = group.device_table()
            _lib.check(self._lib.asp_operator_set_symmetry(
                handle, int(table.shape[0]), _lib.ptr(table), int(group.spin_inversion)))
        self.number_spins = int(number_spins)
        self.unique_targets = bool(self._lib.asp_operator_unique_targets(handle))
        self.max_connections = int(self._lib.asp_operator_max_connections(handle))
        _lib.track(self)

    def release(self) -> None:
        handle, self._handle = getattr(self, "_handle", None), None
        if handle:
            self._lib.asp_operator_destroy(handle)

    def __del__(self):
        if getattr(self, "_handle", None) and not self._lib_module.closed():
            try:
                self.release()
            except Exception:
                pass

    @property
    def last_ms(self) -> float:
        return float(self._lib.asp_operator_last_ms())

    def apply(self, keys) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Flat ``(other_keys u64[N], other_coeffs f64[N], other_counts i64[n])``."""
        import ctypes

        _lib = self._lib_module
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        n = keys.shape[0]
        capacity = n * self.max_connections
        other = np.empty(max(capacity, 1), dtype=np.uint64)
        coeffs = np.empty(max(capacity, 1), dtype=np.float64)
        counts = np.zeros(max(n, 1), dtype=np.int64)
        total = ctypes.c_uint64(0)
        _lib.check(self._lib.asp_operator_apply(self._handle, n, _lib.ptr(keys), capacity,
                                                _lib.ptr(other), _lib.ptr(coeffs),
                                                _lib.ptr(counts), ctypes.byref(total)))
        t = int(total.value)
        return other[:t].copy(), coeffs[:t].copy(), counts[:n]

    def state_info(self, keys) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """``(representative, character, norm)`` of every key (symmetry-adapted bases only)."""
        _lib = self._lib_module
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1)
        n = keys.shape[0]
        rep = np.zeros(max(n, 1), dtype=np.uint64)
        character = np.zeros(max(n, 1), dtype=np.float64)
        norm = np.zeros(max(n, 1), dtype=np.float64)
        _lib.check(self._lib.asp_operator_state_info(self._handle, n, _lib.ptr(keys), _lib.ptr(rep),
                                                     _lib.ptr(character), _lib.ptr(norm)))
        return rep[:n], character[:n], norm[:n]

    def ising(self, keys, psi) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """COO ``(row i32, col i32, val f64)`` of ``0.5 * (M + M^T)`` sorted by (row, col)."""
        import ctypes

        _lib = self._lib_module
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        psi = np.ascontiguousarray(psi, dtype=np.float64)
        if psi.shape != keys.shape:
            raise ValueError("psi and keys differ in length")
        k = keys.shape[0]
        # rows hold at most max_connections entries unless a matrix has one-directional
        # elements; then the library reports the size and the call is repeated
        capacity = k * self.max_connections
        nnz = ctypes.c_uint64(0)
        while True:
            row = np.empty(max(capacity, 1), dtype=np.int32)
            col = np.empty(max(capacity, 1), dtype=np.int32)
            val = np.empty(max(capacity, 1), dtype=np.float64)
            rc = self._lib.asp_operator_ising(self._handle, k, _lib.ptr(keys), _lib.ptr(psi),
                                              capacity, _lib.ptr(row), _lib.ptr(col),
                                              _lib.ptr(val), ctypes.byref(nnz))
            if rc != 0 and int(nnz.value) > capacity:
                capacity = int(nnz.value)
                continue
            _lib.check(rc)
            break
        z = int(nnz.value)
        # views: only the pages the library wrote are resident, and a copy of 10 MB per model
        # is time under the interpreter lock
        return row[:z], col[:z], val[:z]

    def ising_csr(self, keys, psi) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """The matrix of :meth:`ising` as canonical CSR ``(indptr i64[K + 1], col i32, val f64)``."""
        import ctypes

        _lib = self._lib_module
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        psi = np.ascontiguousarray(psi, dtype=np.float64)
        if psi.shape != keys.shape:
            raise ValueError("psi and keys differ in length")
        k = keys.shape[0]
        capacity = k * self.max_connections
        nnz = ctypes.c_uint64(0)
        indptr = np.zeros(k + 1, dtype=np.int64)
        while True:
            col = np.empty(max(capacity, 1), dtype=np.int32)
            val = np.empty(max(capacity, 1), dtype=np.float64)
            rc = self._lib.asp_operator_ising_csr(self._handle, k, _lib.ptr(keys), _lib.ptr(psi),
                                                  capacity, _lib.ptr(indptr), _lib.ptr(col),
                                                  _lib.ptr(val), ctypes.byref(nnz))
            if rc != 0 and int(nnz.value) > capacity:
                capacity = int(nnz.value)
                continue
            _lib.check(rc)
            break
        z = int(nnz.value)
        return indptr, col[:z], val[:z]  # (views: see ising)

    def extend(self, keys) -> np.ndarray:
        """Sorted unique union of the targets of ``keys`` (their own states included)."""
        import ctypes

        _lib = self._lib_module
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        n = keys.shape[0]
        capacity = n * self.max_connections
        out = np.empty(max(capacity, 1), dtype=np.uint64)
        count = ctypes.c_uint64(0)
        _lib.check(self._lib.asp_operator_extend(self._handle, n, _lib.ptr(keys), capacity,
                                                 _lib.ptr(out), ctypes.byref(count)))
        return out[:int(count.value)].copy()

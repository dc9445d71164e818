"""Ground states of whole symmetry sectors, diagonalised on the GPU.

The reference reads its ground states from files written by SpinED (``common.py:783-803``:
``/basis/representatives``, ``/hamiltonian/eigenvectors``; ``experiments/*.py`` take them through
``--hdf5``).  Those files are not part of the reference tree.  For the 16- and 18-site models the
host route (:meth:`.operators.Operator.ground_state`: numpy enumeration + scipy) regenerates them
in seconds; for ``heisenberg_pyrochlore_2x2x2.yaml`` (32 sites, 1.6 million representatives)
and ``heisenberg_kagome_36.yaml`` (36 sites, 31.5 million) it cannot.  This module can, on
one MI355X:

  1. ``asp_sector_enumerate``  (csrc/sector_basis.hip) lists the representatives of the sector;
  2. ``asp_sector_rows``       builds the sector's Hamiltonian ONCE as an ELL matrix that stays in
     HBM (kagome_36: 72 slots x 31.5 M rows x 12 B = 27 GB of the 288 GB);
  3. Lanczos with full reorthogonalisation: one ``asp_sector_matvec`` per step, the Krylov basis
     kept on the device too (250 MB per vector); BLAS-1/2 glue through torch.

``sk_32_1.yaml`` — 6.0e8 states without lattice symmetries, 496 bonds — takes another route: its
matrix does not fit, but its basis has a closed-form index, so ``asp_plain_matvec``
(csrc/plain_basis.hip) regenerates the product on the fly and :func:`lanczos_two_pass` keeps
three vectors of 4.8 GB (about three minutes in all).

Output in the layout the reference's loaders expect (:func:`write_spined_hdf5`), so that the
reference's own command lines (``--yaml ... --hdf5 ...``) run on it.

Device memory and streams come from torch (plumbing); the kernels are the library's.  No CPU
fallback: without a GPU these functions raise.
"""
from __future__ import annotations

import ctypes
import time
from typing import Optional, Tuple

import numpy as np

from . import _lib


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("sector_ed needs a GPU (no CPU fallback; small models: Operator.ground_state)")
    return torch


def _ptr(tensor) -> ctypes.c_void_p:
    return ctypes.c_void_p(tensor.data_ptr())


def binomial(n: int, k: int) -> int:
    out = 1
    for j in range(k):
        out = out * (n - j) // (j + 1)
    return out


def enumerate_sector(operator, log=None) -> Tuple["torch.Tensor", "torch.Tensor"]:
    """``(representatives int64[K] (the u64 states), norms f64[K])`` on the device, ascending:
    the basis of ``operator`` (an :class:`.operators.Operator`; its magnetisation, lattice
    symmetries and spin inversion)."""
    torch = _torch()
    lib = _lib.load()
    device_operator = operator.device()
    basis = operator.basis
    weight = -1 if basis.hamming_weight is None else int(basis.hamming_weight)
    total = (1 << basis.number_spins) if weight < 0 else binomial(basis.number_spins, weight)
    order = basis.group.order if basis.group is not None else 1
    capacity = int(total / order * 1.25) + 4096
    count = ctypes.c_uint64(0)
    tick = time.time()
    for _ in range(2):
        reps = torch.empty(capacity, dtype=torch.int64, device="cuda")
        norms = torch.empty(capacity, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        rc = lib.asp_sector_enumerate(device_operator._handle, weight, capacity, _ptr(reps), _ptr(norms),
                                      ctypes.byref(count))
        if rc == 0:
            break
        if count.value <= capacity:
            _lib.check(rc)
        capacity = int(count.value)  # states with large stabilisers: more orbits than total / |G|
    else:
        _lib.check(rc)
    if log:
        log("sector of %d spins: %d representatives of %d states (|G| = %d) in %.2f s" % (
            basis.number_spins, count.value, total, order, time.time() - tick))
    return reps[: count.value], norms[: count.value]


class SectorMatrix:
    """The operator in its sector basis, resident in HBM (ELL, slot-major)."""

    def __init__(self, operator, representatives, norms, log=None):
        torch = _torch()
        lib = _lib.load()
        device_operator = operator.device()
        self.n = int(representatives.shape[0])
        self.width = int(lib.asp_sector_width(device_operator._handle))
        self.idx = torch.empty((self.width, self.n), dtype=torch.int32, device="cuda")
        self.val = torch.empty((self.width, self.n), dtype=torch.float64, device="cuda")
        self.diag = torch.empty(self.n, dtype=torch.float64, device="cuda")
        tick = time.time()
        torch.cuda.synchronize()
        _lib.check(lib.asp_sector_rows(device_operator._handle, self.n, _ptr(representatives), _ptr(norms),
                                       self.width, _ptr(self.idx), _ptr(self.val), _ptr(self.diag)))
        if log:
            log("sector matrix: %d rows x %d slots = %.2f GB in HBM, built in %.2f s" % (
                self.n, self.width, self.n * (12.0 * self.width + 8.0) / 1e9, time.time() - tick))

    def matvec(self, x, out=None):
        torch = _torch()
        if out is None:
            out = torch.empty_like(x)
        if x.dtype != torch.float64 or not x.is_contiguous() or x.shape[0] != self.n:
            raise ValueError("x must be a contiguous f64 vector of the sector's dimension")
        torch.cuda.synchronize()  # the library runs on a stream of its own
        _lib.check(_lib.load().asp_sector_matvec(self.n, self.width, _ptr(self.idx), _ptr(self.val),
                                                 _ptr(self.diag), _ptr(x), _ptr(out)))
        return out


def lanczos_ground_state(matrix: SectorMatrix, tol: float = 1e-9, max_iterations: int = 400,
                         seed: int = 0, log=None):
    """Lowest eigenpair by Lanczos with full reorthogonalisation.  ``(energy, vector (device),
    info)``; stops when the Ritz residual ``|beta_m s_m|`` falls below ``tol * max(1, |E|)``.
    The Krylov vectors stay on the device (n * 8 bytes each)."""
    torch = _torch()
    import scipy.linalg

    n = matrix.n
    free, _ = torch.cuda.mem_get_info()
    room = int((free - (2 << 30)) // (8 * n)) - 3
    if room < 8:
        raise RuntimeError("not enough device memory for a Krylov basis of %d-vectors" % n)
    steps = min(max_iterations, room, n)
    basis = torch.empty((steps, n), dtype=torch.float64, device="cuda")
    start = np.random.default_rng(seed).standard_normal(n)
    v = torch.from_numpy(start).cuda()
    v /= torch.linalg.vector_norm(v)
    w = torch.empty_like(v)
    alphas, betas = [], []
    energy, ritz, residual = None, None, None
    tick = time.time()
    used = 0
    for j in range(steps):
        basis[j].copy_(v)
        matrix.matvec(v, out=w)
        alphas.append(float(torch.dot(v, w)))
        # full reorthogonalisation against everything so far (covers alpha and beta terms), twice
        for _ in range(2):
            w -= torch.mv(basis[: j + 1].t(), torch.mv(basis[: j + 1], w))
        beta = float(torch.linalg.vector_norm(w))
        used = j + 1
        done = beta < 1e-13 or used == steps
        if used % 5 == 0 or done:
            if used == 1:  # a one-dimensional Krylov space (n = 1, or the start vector is an eigenvector)
                theta, s = np.asarray(alphas), np.ones((1, 1))
            else:
                theta, s = scipy.linalg.eigh_tridiagonal(np.asarray(alphas), np.asarray(betas), select="i",
                                                         select_range=(0, 0))
            energy, ritz = float(theta[0]), s[:, 0]
            residual = abs(beta * ritz[-1])
            if log and (used % 25 == 0 or done or residual < tol * max(1.0, abs(energy))):
                log("  Lanczos step %d: E = %.12f, residual %.2e  [%.1f s]" % (
                    used, energy, residual, time.time() - tick))
            if residual < tol * max(1.0, abs(energy)) or done:
                break
        betas.append(beta)
        v, w = w, v
        v /= beta
    vector = torch.mv(basis[:used].t(), torch.from_numpy(np.ascontiguousarray(ritz)).cuda())
    del basis
    vector /= torch.linalg.vector_norm(vector)
    # the residual proper, with the matrix
    hv = matrix.matvec(vector)
    energy = float(torch.dot(vector, hv))
    true_residual = float(torch.linalg.vector_norm(hv - energy * vector))
    # sign convention of Operator.ground_state: the largest amplitude is positive
    top = int(torch.argmax(torch.abs(vector)))
    if float(vector[top]) < 0:
        vector = -vector
    info = {"iterations": used, "ritz_residual": float(residual), "residual": true_residual,
            "seconds": time.time() - tick}
    if log:
        log("ground state: E = %.12f, |H psi - E psi| = %.2e, %d steps, %.1f s" % (
            energy, true_residual, used, info["seconds"]))
    return energy, vector, info


class PlainBasisMatrix:
    """Matrix-free ``y = Hx`` in a fixed-magnetisation basis without lattice symmetries
    (csrc/plain_basis.hip): sk_32_1.yaml's 6.0e8 states x 496 bonds."""

    def __init__(self, operator, log=None):
        _torch()
        basis = operator.basis
        if basis.group is not None or basis.hamming_weight is None:
            raise ValueError("the matrix-free product needs a magnetisation sector without symmetries")
        lib = _lib.load()
        self._lib_module = _lib
        handle = ctypes.c_void_p()
        _lib.check(lib.asp_plain_basis_create(operator.device()._handle, int(basis.hamming_weight),
                                              ctypes.byref(handle)))
        self._handle = handle
        self.n = int(lib.asp_plain_basis_dimension(handle))
        _lib.track(self)
        if log:
            log("plain basis of %d spins at hamming weight %d: %d states, matrix-free" % (
                basis.number_spins, basis.hamming_weight, self.n))

    def states(self):
        """The basis states, ascending (device int64 tensor holding the u64 values)."""
        torch = _torch()
        out = torch.empty(self.n, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        _lib.check(_lib.load().asp_plain_basis_states(self._handle, _ptr(out)))
        return out

    def matvec(self, x, out=None):
        torch = _torch()
        if out is None:
            out = torch.empty_like(x)
        if x.dtype != torch.float64 or not x.is_contiguous() or x.shape[0] != self.n:
            raise ValueError("x must be a contiguous f64 vector of the basis dimension")
        torch.cuda.synchronize()
        _lib.check(_lib.load().asp_plain_matvec(self._handle, _ptr(x), _ptr(out)))
        return out

    def release(self) -> None:
        handle, self._handle = getattr(self, "_handle", None), None
        if handle:
            self._lib_module.load().asp_plain_basis_destroy(handle)

    def __del__(self):
        module = getattr(self, "_lib_module", None)
        if module is not None and getattr(self, "_handle", None) and not module.closed():
            try:
                self.release()
            except Exception:
                pass


def lanczos_two_pass(matrix, tol: float = 1e-9, max_iterations: int = 1000, seed: int = 0, log=None):
    """Lowest eigenpair with THREE vectors of device memory: the plain three-term recurrence
    (no reorthogonalisation — copies of converged Ritz values appear, the lowest one and its
    vector are not affected), then the same recurrence again to assemble the eigenvector from
    the tridiagonal matrix's.  For bases whose Krylov vectors cannot all be kept (sk_32_1:
    4.8 GB each).  Same return value as :func:`lanczos_ground_state`."""
    torch = _torch()
    import scipy.linalg

    n = matrix.n
    start = torch.from_numpy(np.random.default_rng(seed).standard_normal(n)).cuda()
    start /= torch.linalg.vector_norm(start)
    tick = time.time()

    def recurrence(steps, combine=None):
        """alphas, betas of `steps` steps; with `combine` (Ritz coefficients) also sum_j c_j v_j."""
        v = start.clone()
        previous = torch.zeros_like(v)
        w = torch.empty_like(v)
        total = torch.zeros_like(v) if combine is not None else None
        alphas, betas = [], []
        beta = 0.0
        for j in range(steps):
            if total is not None:
                total.add_(v, alpha=float(combine[j]))
                if j + 1 == steps:
                    break
            matrix.matvec(v, out=w)
            alpha = float(torch.dot(v, w))
            w.add_(v, alpha=-alpha)
            if j > 0:
                w.add_(previous, alpha=-beta)
            alphas.append(alpha)
            beta = float(torch.linalg.vector_norm(w))
            if combine is None:
                done = beta < 1e-13 or j + 1 == steps
                if (j + 1) % 10 == 0 or done:
                    if len(alphas) == 1:  # a one-dimensional Krylov space (see lanczos_ground_state)
                        theta, s = np.array([alphas[0]]), np.ones((1, 1))
                    else:
                        theta, s = scipy.linalg.eigh_tridiagonal(np.asarray(alphas), np.asarray(betas),
                                                                 select="i", select_range=(0, 0))
                    residual = abs(beta * s[-1, 0])
                    if log and ((j + 1) % 50 == 0 or done or residual < tol * max(1.0, abs(theta[0]))):
                        log("  Lanczos step %d: E = %.12f, residual %.2e  [%.1f s]" % (
                            j + 1, theta[0], residual, time.time() - tick))
                    if residual < tol * max(1.0, abs(theta[0])) or done:
                        return alphas, betas, float(theta[0]), s[:, 0], float(residual)
            betas.append(beta)
            previous, v, w = v, w, previous
            v /= beta
        return total

    alphas, betas, energy, ritz, residual = recurrence(min(max_iterations, n))
    vector = recurrence(len(alphas), combine=ritz)
    vector /= torch.linalg.vector_norm(vector)
    hv = matrix.matvec(vector)
    energy = float(torch.dot(vector, hv))
    hv.add_(vector, alpha=-energy)
    true_residual = float(torch.linalg.vector_norm(hv))
    del hv
    top = int(torch.argmax(torch.abs(vector)))
    if float(vector[top]) < 0:
        vector.neg_()
    info = {"iterations": len(alphas), "ritz_residual": residual, "residual": true_residual,
            "seconds": time.time() - tick}
    if log:
        log("ground state: E = %.12f, |H psi - E psi| = %.2e, %d steps (two passes), %.1f s" % (
            energy, true_residual, len(alphas), info["seconds"]))
    return energy, vector, info


def sector_norms(operator, representatives) -> "torch.Tensor":
    """Norms of given representatives (device f64[K]) — ``asp_operator_state_info``."""
    torch = _torch()
    if operator.basis.group is None:
        return torch.ones(len(representatives), dtype=torch.float64, device="cuda")
    _, _, norms = operator.device().state_info(np.ascontiguousarray(representatives, dtype=np.uint64))
    return torch.from_numpy(norms).cuda()


def ground_state(operator, tol: float = 1e-9, max_iterations: int = 400, seed: int = 0, log=None,
                 representatives: Optional[np.ndarray] = None):
    """``(energy, psi f64[K], representatives u64[K], info)`` on the host for the basis of
    ``operator`` — enumeration (unless the sorted ``representatives`` are given), matrix,
    Lanczos, all on the device."""
    torch = _torch()
    basis = operator.basis
    if basis.group is None and basis.hamming_weight is not None:
        # without symmetries the states have a closed-form index: when the resident matrix would
        # not fit in half of the free memory, the product is regenerated on the fly instead
        dimension = binomial(basis.number_spins, int(basis.hamming_weight))
        width = int(_lib.load().asp_sector_width(operator.device()._handle))
        free, _ = torch.cuda.mem_get_info()
        if dimension * (12.0 * width + 24.0) > 0.5 * free:
            matrix = PlainBasisMatrix(operator, log=log)
            energy, vector, info = lanczos_two_pass(matrix, tol=tol, max_iterations=max(max_iterations, 1000),
                                                    seed=seed, log=log)
            info["dimension"] = matrix.n
            psi = vector.cpu().numpy()
            del vector
            states = matrix.states().cpu().numpy().view(np.uint64)
            matrix.release()
            torch.cuda.empty_cache()
            return energy, psi, states, info
    if representatives is None:
        reps, norms = enumerate_sector(operator, log=log)
    else:
        host = np.ascontiguousarray(representatives, dtype=np.uint64)
        reps = torch.from_numpy(host.view(np.int64)).cuda()
        norms = sector_norms(operator, host)
    matrix = SectorMatrix(operator, reps, norms, log=log)
    energy, vector, info = lanczos_ground_state(matrix, tol=tol, max_iterations=max_iterations,
                                                seed=seed, log=log)
    info["dimension"] = matrix.n
    del matrix
    psi = vector.cpu().numpy()
    representatives = reps.cpu().numpy().view(np.uint64)
    torch.cuda.empty_cache()
    return energy, psi, representatives, info


def write_spined_hdf5(filename: str, energy: float, psi: np.ndarray, representatives: np.ndarray) -> None:
    """The three datasets the reference's loaders read (common.py:772-780,
    experiments/full_hilbert_space.py:24-29): ``/basis/representatives`` u64[K],
    ``/hamiltonian/eigenvalues`` f64[1], ``/hamiltonian/eigenvectors`` f64[1, K]
    (:func:`.common.load_ground_state` reads it back)."""
    from . import common

    common.save_ground_state(filename, psi, energy, representatives)


def main(argv=None):
    import argparse

    from . import common, operators, synthetic

    parser = argparse.ArgumentParser(description="Ground state of a model's symmetry sector on the GPU "
                                                 "-> SpinED-layout HDF5")
    parser.add_argument("--model", help="a bundled model (models.json)")
    parser.add_argument("--yaml", help="a physical_systems/*.yaml file of the reference")
    parser.add_argument("--output", required=True, help="HDF5 file to write")
    parser.add_argument("--tol", type=float, default=1e-9)
    parser.add_argument("--max-iterations", type=int, default=400)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--allow-unconverged", action="store_true",
                        help="write the file even when the eigen-residual misses --tol (the step budget "
                             "ran out); without it that is an error: the sampled-cluster pipeline would "
                             "silently consume a wrong ground state")
    args = parser.parse_args(argv)
    if (args.model is None) == (args.yaml is None):
        raise SystemExit("give exactly one of --model and --yaml")
    if args.yaml:
        operator = common.load_hamiltonian(args.yaml)
    else:
        operator = operators.Operator.from_config(synthetic.load_models()[args.model])
    energy, psi, representatives, info = ground_state(operator, args.tol, args.max_iterations, args.seed,
                                                      log=lambda s: print(s, flush=True))
    # (the Ritz estimate; the residual with the matrix is bounded below by the f64 product's
    # own rounding, ~1e-8 * |E| for these sectors, so the criterion is the one the iteration used)
    if info["ritz_residual"] > args.tol * max(1.0, abs(energy)) and not args.allow_unconverged:
        raise SystemExit("Lanczos did not converge: residual %.2e after %d steps against --tol %.1e "
                         "(raise --max-iterations, or pass --allow-unconverged)"
                         % (info["ritz_residual"], info["iterations"], args.tol))
    write_spined_hdf5(args.output, energy, psi, representatives)
    print("wrote %s: K = %d, E0 = %.12f (%.10f per spin)" % (
        args.output, info["dimension"], energy, energy / operator.basis.number_spins), flush=True)


if __name__ == "__main__":
    main()

"""The sampled-cluster pipeline of ``make kagome_36 / pyrochlore_32 / sk_32_1``
(Makefile:101-141 → experiments/sampled_connected_components.py) on the MI355X path.

Same call sequence and CSV schema as the reference driver:

    sample seeds ~ |psi|^p                          common.py:270-279
    grow a cluster around each seed                 common.py:481-513
    order 0: make_ising_model(cluster)              common.py:131-208        (GPU)
    order i: make_hamiltonian_extension + sparsify  common.py:516-522, 647-692
    solve: greedy, optionally SA                    common.py:232-261        (GPU)
    score: accuracy / overlap / amplitude overlap   common.py:211-229; driver :719-723
    one CSV line of 6 * (order + 1) numbers         driver :681-693, 804-830

Inputs: ``--yaml`` + ``--hdf5`` as in the reference (the operator of a physical_systems/*.yaml
file, the ground state and basis representatives of the SpinED output its Makefile downloads),
or ``--model``: a model bundled in ``models.json`` whose ground state is computed on the spot
(16- and 18-site systems on the host, the 32- and 36-site ones on the GPU: :mod:`.sector_ed`).  The operator is :mod:`.operators` instead of ``lattice_symmetries``.  Like the reference, all host-side randomness is numpy's global legacy
stream seeded once with ``--seed`` (driver :776).
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import annealer as sa
from . import common, operators, synthetic


def _in_sorted(values: np.ndarray, haystack: np.ndarray) -> np.ndarray:
    """Membership of ``values`` in the ascending array ``haystack``."""
    if haystack.shape[0] == 0:
        return np.zeros(values.shape[0], dtype=bool)
    at = np.minimum(np.searchsorted(haystack, values), haystack.shape[0] - 1)
    return haystack[at] == values


def create_small_cluster_around_point(s0: int, hamiltonian, required_size: int = 20,
                                      keep_probability: float = 0.5) -> List[int]:
    """Breadth-first growth: each new neighbour is kept with ``keep_probability``
    (common.py:481-513).

    The reference walks a frontier state by state: add the state to the cluster, stop when the
    cluster is large enough, otherwise apply the Hamiltonian to it and draw one random number per
    target that is not in the cluster yet.  Here a whole pass over the frontier is one step: the
    state at which the pass stops follows from the frontier alone, so the states before it are
    applied in ONE call (the GPU action for this package's operators), membership of every
    target at the moment the reference would test it is ``in the cluster before the pass, or
    earlier in the frontier``, and the random numbers of the pass are one draw of as many
    doubles from the same legacy stream.  Same clusters, same stream position afterwards
    (tests/test_host_logic.py checks it against the state-by-state loop); 4096 clusters of
    kagome_36 take seconds instead of 35 s of interpreter loops."""
    assert hamiltonian.basis.number_spins <= 64
    s0 = int(s0)
    members = {s0}

    def connections(states):
        """Targets of every state (own state first), flat, with the number per state."""
        flat, _, counts = common._batched_apply(hamiltonian, np.asarray(states, dtype=np.uint64))
        return np.asarray(flat, dtype=np.uint64).reshape(-1), np.asarray(counts, dtype=np.int64)

    def draw(targets, is_member):
        """The targets that survive: not a member, and one uniform draw each, in order."""
        candidates = np.flatnonzero(~is_member)
        kept = candidates[np.random.rand(candidates.shape[0]) <= keep_probability]
        return kept  # positions into `targets`

    targets, _ = connections([s0])
    frontier = targets[draw(targets, targets == np.uint64(s0))].tolist()
    while len(members) < required_size and len(frontier) > 0:
        order = list(frontier)
        order_arr = np.asarray(order, dtype=np.uint64)
        before = np.fromiter(members, dtype=np.uint64, count=len(members))
        before.sort()
        # the pass adds order[0], order[1], ... and stops at the state that fills the cluster
        unique, first = np.unique(order_arr, return_index=True)
        grows = np.zeros(len(order), dtype=bool)
        grows[first[~_in_sorted(unique, before)]] = True
        full = np.flatnonzero(len(members) + np.cumsum(grows) >= required_size)
        if full.shape[0]:
            stop = int(full[0])  # added, then the pass (and the growth) ends
            members.update(order[:stop + 1])
        else:
            stop = len(order)
            members.update(order)
        upcoming = set()
        if stop > 0:
            targets, counts = connections(order_arr[:stop])
            child = np.repeat(np.arange(stop), counts)
            # position of a target in the frontier (first occurrence), len(order) = not there
            at = np.minimum(np.searchsorted(unique, targets), unique.shape[0] - 1)
            position = np.where(unique[at] == targets, first[at], len(order))
            kept = draw(targets, _in_sorted(targets, before) | (position <= child))
            # the sets are built exactly as the reference builds them (one union per state, in
            # order): the next pass iterates over the result
            bounds = np.searchsorted(child[kept], np.arange(1, stop))
            for part in np.split(targets[kept], bounds):
                if part.shape[0]:
                    upcoming |= set(part.tolist())
        frontier = upcoming
    return sorted(members)


def random_cluster_size(min_size: float, max_size: float) -> int:
    """Log-uniform in [min_size, max_size] (driver :645-649)."""
    u = np.random.random_sample()
    return int(round(np.exp(np.log(min_size) + (np.log(max_size) - np.log(min_size)) * u)))


def iter_clusters(hamiltonian, ground_state, number_samples: int, sampled_power: float,
                  min_cluster_size: int, max_cluster_size: int, keep_probability: float):
    """driver :652-669, one cluster at a time (all draws from numpy's global stream, in the
    reference's order: the seeds first, then size and growth of one cluster after the other)."""
    seeds = common.monte_carlo_sampling(hamiltonian.basis.states, ground_state,
                                        number_samples=number_samples, sampled_power=sampled_power)
    for s in seeds.spins:
        size = random_cluster_size(min_cluster_size, max_cluster_size)
        cluster = create_small_cluster_around_point(s, hamiltonian, required_size=size,
                                                    keep_probability=keep_probability)
        yield np.asarray(cluster, dtype=np.uint64)


def generate_clusters(hamiltonian, ground_state, number_samples: int, sampled_power: float,
                      min_cluster_size: int, max_cluster_size: int,
                      keep_probability: float) -> List[np.ndarray]:
    """driver :652-669."""
    return list(iter_clusters(hamiltonian, ground_state, number_samples, sampled_power,
                              min_cluster_size, max_cluster_size, keep_probability))


@dataclass
class OptimizationResult:
    """driver :672-693."""

    size: int
    greedy_accuracy: float
    greedy_overlap: float
    sa_accuracy: float
    sa_overlap: float
    amplitude_overlap: float

    def to_csv_str(self) -> str:
        return "{},{:.8e},{:.8e},{:.8e},{:.8e},{:.8e}".format(
            self.size, self.greedy_accuracy, self.greedy_overlap, self.sa_accuracy,
            self.sa_overlap, self.amplitude_overlap)

    @staticmethod
    def csv_header() -> str:
        return "size,greedy_accuracy,greedy_overlap,sa_accuracy,sa_overlap,amplitude_overlap"


def solve_and_test_model(h: common.IsingModel, frozen_spins, exact_signs, weights,
                         annealing: bool, sweep_order: Optional[str] = None) -> OptimizationResult:
    """Greedy always, SA when requested, both projected on the original cluster
    (driver :696-716; NB the driver's --number-sweeps/--repetitions are not forwarded, the
    defaults of solve_ising_model apply, common.py:236-239)."""
    x = common.solve_ising_model(h, mode="greedy", frozen_spins=frozen_spins)
    greedy_accuracy, greedy_overlap = common.compute_accuracy_and_overlap(x, exact_signs, weights)
    if annealing:
        x = common.solve_ising_model(h, mode="sa", frozen_spins=frozen_spins, sweep_order=sweep_order)
        sa_accuracy, sa_overlap = common.compute_accuracy_and_overlap(x, exact_signs, weights)
    else:
        sa_accuracy = sa_overlap = float("nan")
    return OptimizationResult(h.size, greedy_accuracy, greedy_overlap, sa_accuracy, sa_overlap,
                              float("nan"))


def amplitude_overlap(cluster, ground_state, noisy_ground_state, basis, where=None) -> float:
    """driver :719-723.  ``where``: the positions of ``cluster`` in the basis when the caller has
    them already (IsingModel.basis_index)."""
    if where is None:
        where = np.asarray(basis.batched_index(cluster), dtype=np.int64)
    a = np.abs(ground_state[where])
    b = np.abs(noisy_ground_state[where])
    # (common.dot / norm2: np.dot and np.linalg.norm kept off the BLAS thread pool)
    return float(common.dot(a, b) / common.norm2(a) / common.norm2(b))


def exact_signs_and_weights(cluster, model, ground_state, basis):
    """Signs and normalised weights psi^2 of the exact ground state on the cluster (driver
    :733-737); ``model`` is the order-0 model of the cluster, whose basis positions are reused."""
    where = model.basis_index
    if where is None or model.size != len(cluster):
        where = np.asarray(basis.batched_index(cluster), dtype=np.int64)
    exact_psi = ground_state[where]
    weights = exact_psi ** 2
    weights /= np.sum(weights)
    return sa.signs_to_bits(np.sign(exact_psi)), weights


def process_cluster(cluster, hamiltonian, ground_state, noisy_ground_state, noisy_log_coeff_fn,
                    order: int, global_cutoff: float, annealing: bool,
                    sweep_order: Optional[str] = None) -> List[OptimizationResult]:
    """driver :726-751."""
    basis = hamiltonian.basis
    results = []
    h = None
    for i in range(order + 1):
        if i == 0:
            h = common.make_ising_model(cluster, hamiltonian, log_psi_fn=noisy_log_coeff_fn)
            exact_signs, weights = exact_signs_and_weights(cluster, h, ground_state, basis)
        else:
            h = common.make_hamiltonian_extension(h, noisy_log_coeff_fn)
            h = common.sparsify_using_global_cutoff(h, global_cutoff, cluster)
        r = solve_and_test_model(h, cluster, exact_signs, weights, annealing, sweep_order)
        r.amplitude_overlap = amplitude_overlap(h.spins, ground_state, noisy_ground_state, basis,
                                                h.basis_index)
        results.append(r)
    return results


def stage_clusters(clusters: Sequence[np.ndarray], hamiltonian, ground_state, noisy_ground_state,
                   noisy_log_coeff_fn, order: int, global_cutoff: float, jobs: int = 1):
    """First half of :func:`process_clusters_batched`: the models of every order of every cluster,
    each with its greedy result — a list of ``(cluster index, model, exact_signs, weights, result
    so far)`` in cluster order.  The models of a cluster do not depend on its solutions (the
    extension of order i grows from the model of order i-1, common.py:516), so they can all be
    built before anything is annealed."""
    basis = hamiltonian.basis

    def stage(item):
        """Models of every order of one cluster, each with its greedy result."""
        index, cluster = item
        h = None
        out = []
        for i in range(order + 1):
            if i == 0:
                h = common.make_ising_model(cluster, hamiltonian, log_psi_fn=noisy_log_coeff_fn)
                exact_signs, weights = exact_signs_and_weights(cluster, h, ground_state, basis)
            else:
                h = common.make_hamiltonian_extension(h, noisy_log_coeff_fn)
                h = common.sparsify_using_global_cutoff(h, global_cutoff, cluster)
            r = solve_and_test_model(h, cluster, exact_signs, weights, annealing=False)
            r.amplitude_overlap = amplitude_overlap(h.spins, ground_state, noisy_ground_state, basis,
                                                    h.basis_index)
            out.append((index, h, exact_signs, weights, r))
        return out

    # the builds of different clusters are independent and their C calls release the GIL: --jobs
    # host threads keep several in flight on the GPU (own streams); order of the list = cluster
    # order either way
    if jobs > 1 and len(clusters) > 1:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=jobs) as pool:
            return [entry for part in pool.map(stage, enumerate(clusters)) for entry in part]
    return [entry for item in enumerate(clusters) for entry in stage(item)]


def anneal_staged(staged, clusters: Sequence[np.ndarray], annealing: bool,
                  sweep_order: Optional[str] = None) -> List[List[OptimizationResult]]:
    """Second half: the annealing of ALL staged models — every cluster at every order — in one
    batched device call, scored and sorted back into one list of results per cluster."""
    if annealing and staged:
        started = time.perf_counter()
        solutions = common.solve_ising_models([m for _, m, _, _, _ in staged],
                                              [clusters[c] for c, _, _, _, _ in staged],
                                              sweep_order=sweep_order)
        if os.environ.get("ASP_PIPELINE_TIMING"):  # development aid
            spins = sum(m.size for _, m, _, _, _ in staged)
            seconds = time.perf_counter() - started
            sys.stderr.write("[pipeline] round of %d clusters: %d models, %d spins (largest %d) annealed "
                             "in %.2f s = %.1f G flips/s\n" % (
                                 len(clusters), len(staged), spins, max(m.size for _, m, _, _, _ in staged),
                                 seconds, spins * 64 * 5120 / seconds * 1e-9))
        for (_, _, exact_signs, weights, r), x in zip(staged, solutions):
            r.sa_accuracy, r.sa_overlap = common.compute_accuracy_and_overlap(x, exact_signs, weights)
    results: List[List[OptimizationResult]] = [[] for _ in clusters]
    for index, h, _, _, r in staged:
        results[index].append(r)
        h.ising_hamiltonian.release()  # the device plan of a finished model
    return results


def process_clusters_batched(clusters: Sequence[np.ndarray], hamiltonian, ground_state,
                             noisy_ground_state, noisy_log_coeff_fn, order: int,
                             global_cutoff: float, annealing: bool, jobs: int = 1,
                             sweep_order: Optional[str] = None) -> List[List[OptimizationResult]]:
    """``[process_cluster(c, ...) for c in clusters]`` with the annealing of ALL models — every
    cluster at every order — in one batched device call (:func:`stage_clusters`, then
    :func:`anneal_staged`); the results are identical to the per-cluster loop."""
    staged = stage_clusters(clusters, hamiltonian, ground_state, noisy_ground_state, noisy_log_coeff_fn,
                            order, global_cutoff, jobs)
    return anneal_staged(staged, clusters, annealing, sweep_order)


def parse_command_line(argv=None):
    parser = argparse.ArgumentParser(description="Test Simulated Annealing on sampled clusters.")
    parser.add_argument("--model", type=str,
                        help="name in models.json; its ground state is computed by exact "
                             "diagonalisation (16- and 18-site models)")
    parser.add_argument("--yaml", type=str,
                        help="physical_systems/<model>.yaml of the reference (driver :755)")
    parser.add_argument("--hdf5", type=str,
                        help="SpinED output with the ground state and the basis representatives "
                             "(driver :757; default <yaml>.h5 when --yaml is given)")
    parser.add_argument("--output", type=str, required=True)
    parser.add_argument("--order", type=int, required=True)
    parser.add_argument("--noise", type=float, default=0)
    parser.add_argument("--annealing", default=True, action=argparse.BooleanOptionalAction)
    parser.add_argument("--global-cutoff", type=float, default=1e-4)
    parser.add_argument("--number-samples", type=int, default=5)
    parser.add_argument("--number-sweeps", type=int, default=5000)
    parser.add_argument("--repetitions", type=int, default=64)
    parser.add_argument("--min-cluster-size", type=int, default=50)
    parser.add_argument("--max-cluster-size", type=int, default=1000)
    parser.add_argument("--sampled-power", type=float, default=0.1)
    parser.add_argument("--keep-probability", type=float, default=0.5)
    parser.add_argument("--seed", type=int, default=12345)
    parser.add_argument("--batch", type=int, default=256,
                        help="clusters whose annealing chains share one batched device call "
                             "(asp_sa_anneal_batch); 1 = one call per model, as the reference's "
                             "loop.  The output does not depend on it")
    parser.add_argument("--sweep-order", type=str, default="shuffled", choices=["colour", "shuffled"],
                        help="visiting order of the annealing sweeps: 'shuffled' (default: a fresh "
                             "random order every sweep, the reference annealer's statistics) or "
                             "'colour' (this package's fixed colour order: faster, a different chain)")
    parser.add_argument("--workers", type=int, default=0,
                        help="worker PROCESSES (forked after the inputs are loaded, so the ground state "
                             "is shared, not read once per rank); worker k computes on GPU k mod "
                             "(number of GPUs) and runs --jobs threads: one command for a multi-GPU "
                             "node.  On ONE GPU it does not pay (the device multiplexes processes at "
                             "a higher cost than the interpreter lock: profiles/r03_pipeline_workers.txt)."
                             "  The output does not depend on it")
    parser.add_argument("--jobs", type=int, default=1,
                        help="host threads building / solving clusters concurrently (independent "
                             "plans and HIP streams on one GPU; the output does not depend on it)")
    return parser.parse_args(argv)


def load_input(args):
    """``(hamiltonian, ground_state)``: either a bundled model diagonalised on the spot
    (``--model``), or the reference's inputs (``--yaml`` + ``--hdf5``; common.py:791-803): the
    operator of the YAML file, the first eigenvector of the HDF5 file and its list of basis
    representatives, which becomes the basis."""
    if (args.model is None) == (args.yaml is None):
        raise SystemExit("give exactly one of --model and --yaml")
    if args.model is not None:
        models = synthetic.load_models()
        if args.model not in models:
            raise SystemExit("unknown model '{}'; available: {}".format(args.model, sorted(models)))
        hamiltonian = operators.Operator.from_config(models[args.model])
        if args.hdf5 is None:
            hamiltonian.basis.build()
            _, ground_state = hamiltonian.ground_state()
            return hamiltonian, ground_state
    else:
        hamiltonian = common.load_hamiltonian(args.yaml)
    hdf5 = args.hdf5 if args.hdf5 is not None else args.yaml.replace(".yaml", ".h5")
    ground_state, _, representatives = common.load_ground_state(hdf5)
    if representatives.shape[0] < 2 or np.all(representatives[1:] > representatives[:-1]):
        # already ascending (SpinED and sector_ed write them so): no 6e8-element argsort for sk_32_1
        hamiltonian.basis.build(representatives)
        return hamiltonian, np.ascontiguousarray(ground_state)
    order = np.argsort(representatives, kind="stable")  # the basis keeps its states sorted
    hamiltonian.basis.build(representatives[order])
    return hamiltonian, np.ascontiguousarray(ground_state[order])


_WORKER = {}


def gpu_maybe_initialised() -> bool:
    """True when this process may already hold HIP state, which a forked child must not inherit.
    ``asp_device_touched()`` only knows about calls made THROUGH libasp_hip; the GPU may also have
    been initialised by a profiler's preloaded tool library (rocprofv3 sets ``ROCP_TOOL_LIBRARIES``
    / ``HSA_TOOLS_LIB`` / ``LD_PRELOAD`` and initialises HSA before the program starts) or by a host
    that used ``torch.cuda``."""
    from . import _lib

    if _lib.gpu_touched():
        return True
    if os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("HSA_TOOLS_LIB"):
        return True
    preload = os.environ.get("LD_PRELOAD", "")
    if any(name in preload for name in ("rocprof", "roctracer", "roctx")):
        return True
    torch = sys.modules.get("torch")
    if torch is not None:
        try:
            if torch.cuda.is_initialized():
                return True
        except Exception:  # noqa: BLE001 - a torch without a usable cuda module has not touched a GPU
            pass
    return False


def _generate_in_child(conn, hamiltonian, ground_state, args):
    """Cluster growth uses the GPU action; done in a child so that the parent stays free of
    any HIP state and can fork its workers afterwards."""
    try:
        clusters = generate_clusters(hamiltonian, ground_state, args.number_samples, args.sampled_power,
                                     args.min_cluster_size, args.max_cluster_size, args.keep_probability)
        conn.send(("ok", clusters))
    except BaseException as error:  # noqa: BLE001 - reported to the parent
        conn.send(("error", "%s: %s" % (type(error).__name__, error)))
    finally:
        conn.close()
        os._exit(0)  # no interpreter teardown in a forked child that used the GPU


def _stream_from_child(conn, hamiltonian, ground_state, args):
    """Child of :func:`clusters_from_child`: grows the clusters and sends them as they come."""
    try:
        pending = []
        for cluster in iter_clusters(hamiltonian, ground_state, args.number_samples, args.sampled_power,
                                     args.min_cluster_size, args.max_cluster_size, args.keep_probability):
            pending.append(cluster)
            if len(pending) >= 16:
                conn.send(("clusters", pending))
                pending = []
        if pending:
            conn.send(("clusters", pending))
        conn.send(("done", None))
    except BaseException as error:  # noqa: BLE001 - reported to the parent
        conn.send(("error", "%s: %s" % (type(error).__name__, error)))
    finally:
        conn.close()
        os._exit(0)  # no interpreter teardown in a forked child that used the GPU


def clusters_from_child(hamiltonian, ground_state, args):
    """The clusters of :func:`iter_clusters`, grown in a FORKED child while the caller solves the
    ones it has received.  Growth is one sequential chain of random draws (a fifth of the run
    for 4096 greedy clusters of kagome_36) and needs the interpreter; the child has its own.  It
    inherits the seeded random stream at the fork, so the clusters are the ones the caller would
    have grown itself.  Only for a caller that has not used the GPU yet (a process that has must
    not fork): otherwise, and where there is no fork, the clusters are grown in place."""
    import multiprocessing

    # (under torch.distributed.run the rank has bound its GPU through torch already; under
    # rocprofv3 the profiler's tool library has initialised it before main() runs)
    in_place = (gpu_maybe_initialised() or "RANK" in os.environ
                or "fork" not in multiprocessing.get_all_start_methods()
                or os.environ.get("ASP_GROW_IN_PLACE") == "1")
    if in_place:
        yield from iter_clusters(hamiltonian, ground_state, args.number_samples, args.sampled_power,
                                 args.min_cluster_size, args.max_cluster_size, args.keep_probability)
        return
    ctx = multiprocessing.get_context("fork")
    receiver, sender = ctx.Pipe(False)
    child = ctx.Process(target=_stream_from_child, args=(sender, hamiltonian, ground_state, args))
    child.start()
    sender.close()
    try:
        while True:
            try:
                kind, payload = receiver.recv()
            except EOFError:
                raise SystemExit("cluster generation ended without a result (child exit code %s)"
                                 % child.exitcode)
            if kind == "clusters":
                yield from payload
            elif kind == "done":
                return
            else:
                raise SystemExit("cluster generation failed: " + payload)
    finally:
        receiver.close()
        child.join(timeout=30)
        if child.is_alive():
            child.terminate()


def _worker_init():
    """Worker k computes on GPU k mod (number of GPUs): on a multi-GPU node one command fills all
    of them from one copy of the inputs; on a one-GPU box the workers share the device."""
    import multiprocessing

    from . import _lib

    identity = multiprocessing.current_process()._identity
    index = (identity[0] - 1) if identity else 0
    devices = _lib.device_count()
    if devices > 1:
        _lib.check(_lib.load().asp_set_device(index % devices))
    if devices <= 1 or _WORKER["args"].workers > devices:
        os.environ.setdefault("ASP_SA_TEAM", "0")  # several processes share a device


def _worker_chunk(indices):
    """CSV lines of clusters[indices], computed in a worker process (inputs inherited by fork)."""
    w = _WORKER
    started = time.perf_counter()
    if "log_coeff_fn" not in w:
        w["log_coeff_fn"] = common.ground_state_to_log_coeff_fn(w["noisy_ground_state"], w["hamiltonian"].basis)
    args = w["args"]
    if os.environ.get("ASP_WORKER_TIMING"):  # development aid
        def report(lines, t0=started, first=indices[0]):
            sys.stderr.write("worker %d: clusters %d..%d in %.2f s\n" % (
                os.getpid(), first, indices[-1], time.perf_counter() - t0))
            return lines
    else:
        report = lambda lines: lines  # noqa: E731
    some = [w["clusters"][c] for c in indices]
    if args.batch > 1 and args.annealing:
        chunk = process_clusters_batched(some, w["hamiltonian"], w["ground_state"], w["noisy_ground_state"],
                                         w["log_coeff_fn"], args.order, args.global_cutoff, args.annealing,
                                         jobs=args.jobs, sweep_order=args.sweep_order)
        return report([",".join(r.to_csv_str() for r in columns) for columns in chunk])

    def work(cluster):
        return process_cluster(cluster, w["hamiltonian"], w["ground_state"], w["noisy_ground_state"],
                               w["log_coeff_fn"], args.order, args.global_cutoff, args.annealing,
                               args.sweep_order)

    if args.jobs > 1 and len(some) > 1:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=args.jobs) as pool:
            return report([",".join(r.to_csv_str() for r in columns) for columns in pool.map(work, some)])
    return report([",".join(r.to_csv_str() for r in work(c)) for c in some])


def _write_header(args):
    with open(args.output, "w") as f:
        f.write("# Generated by annealing_sign_problem_amd.sampled_components\n")
        for key in ["seed", "order", "noise", "global_cutoff", "sampled_power",
                    "min_cluster_size", "max_cluster_size", "keep_probability",
                    "number_sweeps", "repetitions"]:
            f.write("# {} = {}\n".format(key, getattr(args, key)))
        f.write("# {}\n".format(OptimizationResult.csv_header()))


def _main_with_workers(args):
    """One command, several worker processes on the GPU.  The parent loads the inputs (no GPU
    call), a first child grows the clusters (GPU), then ``--workers`` children are FORKED: the
    ground state (0.5 GB for kagome_36, 10 GB for sk_32_1) is shared copy-on-write instead of
    read once per rank, every worker has its own interpreter (no lock shared between them), its
    own HIP context and streams, and takes rounds of ``--batch`` clusters from a common queue;
    the parent appends the lines in cluster order as the rounds come back.  Same file as the
    single process: all random draws happen before the first cluster is solved."""
    import multiprocessing

    if os.path.exists(args.output):
        raise SystemExit("Output file '{}' already exists: refusing to overwrite".format(args.output))
    np.random.seed(args.seed)
    hamiltonian, ground_state = load_input(args)
    if gpu_maybe_initialised():
        # (a ground state computed on the GPU on the spot: --model of a 32-/36-site system without --hdf5)
        raise SystemExit("--workers forks after the inputs are loaded and needs them loaded WITHOUT the "
                         "GPU: write the ground state first (python -m annealing_sign_problem_amd.sector_ed) "
                         "and pass it with --hdf5")
    noisy_ground_state = (common.add_noise_to_amplitudes(ground_state, args.noise) if args.noise > 0
                          else ground_state)
    ctx = multiprocessing.get_context("fork")
    receiver, sender = ctx.Pipe(False)
    child = ctx.Process(target=_generate_in_child, args=(sender, hamiltonian, ground_state, args))
    child.start()
    sender.close()
    status, clusters = receiver.recv()
    child.join()
    if status != "ok":
        raise SystemExit("cluster generation failed: " + clusters)
    _write_header(args)
    _WORKER.update(hamiltonian=hamiltonian, ground_state=ground_state,
                   noisy_ground_state=noisy_ground_state, args=args, clusters=clusters)
    step = max(args.batch, 1) if args.annealing else max(1, min(16, args.batch))
    rounds = [list(range(start, min(start + step, len(clusters)))) for start in range(0, len(clusters), step)]
    # (an executor, not multiprocessing.Pool: a Pool silently replaces a worker that died — a GPU
    # fault, an out-of-memory kill — and never completes its task, so the parent would wait forever)
    from concurrent.futures import ProcessPoolExecutor
    from concurrent.futures.process import BrokenProcessPool

    with ProcessPoolExecutor(max_workers=args.workers, mp_context=ctx, initializer=_worker_init) as pool:
        futures = [pool.submit(_worker_chunk, indices) for indices in rounds]
        for indices, future in zip(rounds, futures):  # (in order: the file grows round by round)
            try:
                lines = future.result()
            except BrokenProcessPool:
                for other in futures:
                    other.cancel()
                raise SystemExit("a worker process died while clusters %d..%d were being solved (a GPU fault "
                                 "or an out-of-memory kill?); the output holds the rounds before them"
                                 % (indices[0], indices[-1]))
            with open(args.output, "a") as f:
                for line in lines:
                    f.write(line + "\n")


def main(argv=None):
    from . import distributed as asp_dist

    args = parse_command_line(argv)
    started = time.perf_counter()

    def phase(name):  # development aid: ASP_PIPELINE_TIMING=1 prints the wall time of the phases
        if os.environ.get("ASP_PIPELINE_TIMING"):
            sys.stderr.write("[pipeline] %-28s at %7.2f s\n" % (name, time.perf_counter() - started))

    if args.workers > 1 and "RANK" not in os.environ:
        if gpu_maybe_initialised():
            import warnings

            warnings.warn("--workers needs a process that has not used the GPU yet (this one has, or "
                          "runs under a profiler that has): running with --jobs threads instead")
        else:
            return _main_with_workers(args)
    # under `python -m torch.distributed.run -m ...sampled_components`: bind this rank's GPU for
    # torch and for libasp_hip and join the process group before anything touches a device
    created_group = asp_dist.init_from_env()
    # Whatever way the rest of main() ends — normally, a SystemExit from the inputs, an exception
    # of a stage, of a worker thread or of a collective whose peer has died — the group this
    # call created is destroyed before the interpreter exits: torch's background threads of a
    # group that is still up have been seen to abort the process at exit ("terminate called
    # without an active exception", commit 1f46a71).  Only the normal end waits for the other
    # ranks (a barrier); a failing rank must not wait for ranks that may be dead or blocked.
    failed = True
    try:
        _main_in_group(args, asp_dist, created_group, phase)
        failed = False
    finally:
        if created_group:
            import torch.distributed as dist

            try:
                if not failed:
                    dist.barrier()
            finally:
                try:
                    dist.destroy_process_group()
                except Exception:  # noqa: BLE001 - nothing more to do for a group that is half gone
                    pass


def _main_in_group(args, asp_dist, created_group, phase):
    np.random.seed(args.seed)
    writer = asp_dist.rank() == 0  # under torch.distributed only rank 0 touches the file
    # rank 0 looks at the file and tells the others, so that all ranks stop together
    refuse = asp_dist.broadcast_object(writer and os.path.exists(args.output))
    if refuse:
        if created_group:
            # (every rank is here: wait for the others, then main() tears the group down)
            import torch.distributed as dist

            dist.barrier()
        raise SystemExit("Output file '{}' already exists: refusing to overwrite".format(args.output))
    hamiltonian, ground_state = load_input(args)
    phase("inputs loaded")
    if args.noise > 0:
        noisy_ground_state = common.add_noise_to_amplitudes(ground_state, args.noise)
    else:
        noisy_ground_state = ground_state
    noisy_log_coeff_fn = common.ground_state_to_log_coeff_fn(noisy_ground_state, hamiltonian.basis)
    phase("amplitude tables")
    # the clusters are grown once, by rank 0, and handed to the others (same file for any world
    # size: the draws come from rank 0's seeded stream either way)
    if asp_dist.world_size() > 1:
        clusters = None
        if writer:
            clusters = generate_clusters(hamiltonian, ground_state, args.number_samples,
                                         args.sampled_power, args.min_cluster_size,
                                         args.max_cluster_size, args.keep_probability)
        clusters = asp_dist.broadcast_object(clusters)
        phase("clusters grown")
    else:
        # one process: a forked child grows the clusters while this one solves them.  The first
        # cluster is pulled here, so that the fork happens in this thread, before any other exists
        import itertools

        clusters = clusters_from_child(hamiltonian, ground_state, args)
        clusters = itertools.chain(list(itertools.islice(clusters, 1)), clusters)
    if args.jobs > 1:
        # several plans are built at once: one host thread each (csrc/sa_plan.cpp starts four for
        # a large model, which pays for one model at a time only)
        os.environ.setdefault("ASP_HOST_THREADS", "1")
    if writer:
        _write_header(args)
    def work(cluster):
        return process_cluster(cluster, hamiltonian, ground_state, noisy_ground_state,
                               noisy_log_coeff_fn, args.order, args.global_cutoff, args.annealing,
                               args.sweep_order)

    def work_many(some):
        """CSV lines of a list of clusters, their annealing batched --batch clusters at a time."""
        if args.batch <= 1 or not args.annealing:
            if args.jobs > 1 and len(some) > 1:
                from concurrent.futures import ThreadPoolExecutor

                with ThreadPoolExecutor(max_workers=args.jobs) as pool:  # (map keeps the order)
                    return [",".join(r.to_csv_str() for r in columns) for columns in pool.map(work, some)]
            return [",".join(r.to_csv_str() for r in work(c)) for c in some]
        lines = []
        for start in range(0, len(some), args.batch):
            chunk = process_clusters_batched(some[start:start + args.batch], hamiltonian,
                                             ground_state, noisy_ground_state, noisy_log_coeff_fn,
                                             args.order, args.global_cutoff, args.annealing,
                                             jobs=args.jobs, sweep_order=args.sweep_order)
            lines += [",".join(r.to_csv_str() for r in columns) for columns in chunk]
        return lines

    def append(lines):
        with open(args.output, "a") as f:
            for line in lines:
                f.write(line + "\n")

    # Clusters are independent problems.  Under torch.distributed (one process per
    # GPU) cluster c is solved by rank c mod world and rank 0 writes the gathered lines; every
    # rank generated the same clusters above (same seed), so the file does not depend on the
    # world size.
    if asp_dist.world_size() > 1:
        # a round of --batch clusters per rank at a time; rank 0 appends every round's lines, so a
        # job that is killed keeps what it has finished (driver :828-830 appends per cluster).
        # The round size is a multiple of the world size: cluster c stays on rank c mod world.
        world, me = asp_dist.world_size(), asp_dist.rank()
        step = max(args.batch, 1) * world
        rounds = [clusters[start:start + step] for start in range(0, len(clusters), step)]
        if args.batch > 1 and args.annealing:
            # as in the single process: while a rank's chains of one round anneal, a second
            # thread builds the models of its share of the next round
            from concurrent.futures import ThreadPoolExecutor

            def my_share(r):
                return [rounds[r][c] for c in range(len(rounds[r])) if c % world == me]

            def stage(some):
                return stage_clusters(some, hamiltonian, ground_state, noisy_ground_state,
                                      noisy_log_coeff_fn, args.order, args.global_cutoff, args.jobs)

            with ThreadPoolExecutor(max_workers=1) as builder:
                upcoming = builder.submit(stage, my_share(0)) if rounds else None
                for r in range(len(rounds)):
                    staged = upcoming.result()
                    upcoming = builder.submit(stage, my_share(r + 1)) if r + 1 < len(rounds) else None

                    def solve(some, staged=staged):  # (`some` is this rank's share of the round)
                        return [",".join(x.to_csv_str() for x in columns)
                                for columns in anneal_staged(staged, some, True, args.sweep_order)]

                    lines = asp_dist.map_sharded_many(rounds[r], solve)
                    if writer:
                        append(lines)
        else:
            for some in rounds:
                lines = asp_dist.map_sharded_many(some, work_many)
                if writer:
                    append(lines)
        return
    # With --jobs > 1 several clusters are in flight on one GPU at once — the C calls release
    # the GIL and every Hamiltonian owns its stream.  All randomness was consumed above, so the
    # lines written are identical for any --jobs and any --batch.
    if args.jobs > 1 and (args.batch <= 1 or not args.annealing):
        from concurrent.futures import ThreadPoolExecutor

        import collections

        with ThreadPoolExecutor(max_workers=args.jobs) as pool:
            # submitted as the child delivers them; written in cluster order as they finish
            pending = collections.deque()
            for cluster in clusters:
                pending.append(pool.submit(work, cluster))
                while pending and pending[0].done():
                    append([",".join(r.to_csv_str() for r in pending.popleft().result())])
            while pending:
                append([",".join(r.to_csv_str() for r in pending.popleft().result())])
    else:
        import itertools
        from concurrent.futures import ThreadPoolExecutor

        # Rounds of --batch clusters, written round by round (a long job's output grows as it
        # runs).  While the chains of one round anneal on the GPU, a second thread builds the
        # models of the next one (host work mostly, small kernels on streams of their own).
        step = max(args.batch, 1)
        lines_of = lambda columns_of_clusters: [",".join(r.to_csv_str() for r in columns)  # noqa: E731
                                                for columns in columns_of_clusters]
        if step > 1 and args.annealing:
            def next_round():
                """(clusters, their staged models) of the next round; only this thread pulls clusters."""
                some = list(itertools.islice(clusters, step))
                return some, stage_clusters(some, hamiltonian, ground_state, noisy_ground_state,
                                            noisy_log_coeff_fn, args.order, args.global_cutoff, args.jobs)

            with ThreadPoolExecutor(max_workers=1) as builder:
                upcoming = builder.submit(next_round)
                while True:
                    some, staged = upcoming.result()
                    if not some:
                        break
                    upcoming = builder.submit(next_round)
                    append(lines_of(anneal_staged(staged, some, args.annealing, args.sweep_order)))
        else:
            while True:
                some = list(itertools.islice(clusters, step))
                if not some:
                    break
                append(work_many(some))
    phase("clusters solved")


if __name__ == "__main__":
    main()

"""Multi-GPU sharding of annealing chains: one process per GPU, RCCL over xGMI.

The reference has no distributed runtime; its only scale-out is independent
SLURM jobs keyed by ``JOBID`` (Makefile:11-15, README.md:171-190).  The chains
of one ``anneal`` call are independent Markov chains over a read-only J, so they
shard with NO collective on the data path: rank ``k`` runs a contiguous block of
global replica ids against its own copy of the plan, and a single
``all_gather`` of (packed spins, energy) at the end gives every rank the full
result.  The counter RNG is keyed by the GLOBAL replica id, so the gathered
result is bit-identical for every world size.

``torch`` is plumbing only (process group + the gather); with backend ``nccl``
(= RCCL on ROCm) the payload is staged through HBM tensors, with ``gloo`` (CPU
tests) through host tensors.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except Exception:  # torch absent: single process
        return None
    if not dist.is_available() or not dist.is_initialized():
        return None
    return dist


def init_from_env(backend: Optional[str] = None) -> bool:
    """One process per GPU under ``python -m torch.distributed.run``: when RANK / WORLD_SIZE are
    in the environment and no process group exists yet, bind this process to GPU LOCAL_RANK —
    for torch (RCCL follows ``torch.cuda.current_device()``) AND for libasp_hip
    (``asp_set_device``; the library otherwise computes on HIP device 0 in every rank) — and
    create the process group, before any other GPU call.  Returns True when a group was created.

    ``backend`` defaults to ``$ASP_DIST_BACKEND``, else ``nccl`` (= RCCL) when GPUs are visible
    and ``gloo`` otherwise.  ``ASP_SINGLE_DEVICE=1`` puts every rank on device 0 (rehearsal on a
    one-GPU box; only meaningful with gloo)."""
    import os

    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return False
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        return False
    from . import _lib

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("ASP_SINGLE_DEVICE") == "1":
        local_rank = 0
        if int(os.environ["WORLD_SIZE"]) > 1:
            # several ranks on one GPU: a team sweep needs all its workgroups resident together
            os.environ.setdefault("ASP_SA_TEAM", "0")
    gpus = _lib.device_count()
    backend = backend or os.environ.get("ASP_DIST_BACKEND") or ("nccl" if gpus > 0 else "gloo")
    if gpus > 0:
        if local_rank >= gpus:
            raise _lib.AspError(-1, "LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, gpus))
        torch.cuda.set_device(local_rank)
        _lib.check(_lib.load().asp_set_device(local_rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend)
    return True


def check_device_binding(group=None) -> None:
    """With RCCL the gathered tensors live on ``torch.cuda.current_device()``; the chains must
    have run on the same GPU, and no two ranks of a node may share one (RCCL refuses duplicate
    devices).  Raises when libasp_hip is bound elsewhere — the mistake :func:`init_from_env`
    exists to prevent."""
    import torch
    import torch.distributed as dist

    if dist.get_backend(group) != "nccl":
        return
    from . import _lib

    mine = int(_lib.load().asp_get_device())
    theirs = int(torch.cuda.current_device())
    if mine != theirs:
        raise _lib.AspError(-3, "libasp_hip computes on device %d but torch.distributed/RCCL uses "
                                "device %d: call distributed.init_from_env() (or asp_set_device and "
                                "torch.cuda.set_device with LOCAL_RANK) before the first GPU call"
                                % (mine, theirs))


def world_size() -> int:
    d = _dist()
    return d.get_world_size() if d is not None else 1


def rank() -> int:
    d = _dist()
    return d.get_rank() if d is not None else 0


def broadcast_object(value, src: int = 0, group=None):
    """``value`` of rank ``src`` on every rank (picklable control data; single process: as is)."""
    d = _dist()
    if d is None or d.get_world_size(group) == 1:
        return value
    box = [value]
    d.broadcast_object_list(box, src=src, group=group)
    return box[0]


def shard_range(total: int, world: int, index: int) -> Tuple[int, int]:
    """Contiguous block of ``total`` items owned by ``index`` of ``world``: (offset, count)."""
    base, extra = divmod(int(total), int(world))
    count = base + (1 if index < extra else 0)
    offset = index * base + min(index, extra)
    return offset, count


def all_gather_rows(local: np.ndarray, counts, group=None) -> np.ndarray:
    """Concatenate per-rank row blocks (``counts[k]`` rows from rank k) on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    width = local.shape[1]
    most = max(counts) if len(counts) else 0
    padded = np.zeros((max(most, 1), width), dtype=np.int64)
    padded[: local.shape[0]] = local.view(np.int64) if local.dtype != np.int64 else local
    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(padded).to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    rows = [parts[k][: counts[k]].cpu().numpy() for k in range(world)]
    return np.concatenate(rows, axis=0) if rows else padded[:0]


def anneal_sharded(hamiltonian, seed: int, betas: np.ndarray, repetitions: int, x0=None,
                   group=None, shuffled: bool = False):
    """Run this rank's block of chains, then gather: (xs[R, words], es[R]) on every rank.

    With RCCL (backend ``nccl``) the kernel's results are copied device-to-device into the
    tensors that are gathered — no host round trip before the collective; with gloo (CPU tests)
    they go through host arrays."""
    from . import annealer

    import torch
    import torch.distributed as dist

    check_device_binding(group)
    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    offset, count = shard_range(repetitions, world, me)
    words = (hamiltonian.size + 63) // 64
    counts = [shard_range(repetitions, world, k)[1] for k in range(world)]
    most = max(max(counts), 1)
    if dist.get_backend(group) == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
        # one payload row per chain: [packed spins | energy bits]
        xs_t = torch.zeros((most, max(words, 1)), dtype=torch.int64, device=device)
        es_t = torch.zeros((most,), dtype=torch.float64, device=device)
        # the zero fill runs on torch's stream, the library writes from its own: order them
        torch.cuda.current_stream().synchronize()
        if count > 0:
            annealer.anneal_raw_into(hamiltonian, seed, betas, count, offset, x0,
                                     xs_t.data_ptr(), es_t.data_ptr(), shuffled=shuffled)
        mine = torch.cat([xs_t[:, :words], es_t.view(torch.int64).unsqueeze(1)], dim=1).contiguous()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        full = torch.cat([parts[k][: counts[k]] for k in range(world)], dim=0).cpu().numpy()
    else:
        if count > 0:
            xs, es = annealer.anneal_raw(hamiltonian, seed, betas, count, offset, x0, shuffled=shuffled)
        else:
            xs = np.zeros((0, words), dtype=np.uint64)
            es = np.zeros(0, dtype=np.float64)
        payload = np.concatenate(
            [np.ascontiguousarray(xs, dtype=np.uint64).view(np.int64).reshape(count, words),
             np.ascontiguousarray(es, dtype=np.float64).view(np.int64).reshape(count, 1)], axis=1)
        full = all_gather_rows(payload, counts, group)
    xs_all = np.ascontiguousarray(full[:, :words]).view(np.uint64)
    es_all = np.ascontiguousarray(full[:, words]).view(np.float64)
    return xs_all, es_all


def _device_for(group=None):
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")


def agree_on_seed(seed: Optional[int], group=None) -> int:
    """``seed=None`` means "draw one": rank 0 draws, everybody uses it (otherwise the ranks would
    run chains of different random streams and the gathered result would depend on the world
    size).  A given seed passes through."""
    import torch
    import torch.distributed as dist

    from .annealer import _resolve_seed

    value = _resolve_seed(seed)
    if seed is not None:
        return value
    box = torch.tensor([value - (1 << 64) if value >= (1 << 63) else value], dtype=torch.int64,
                       device=_device_for(group))
    dist.broadcast(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(box.item()) & (2**64 - 1)


def anneal_sharded_best(hamiltonian, seed: int, betas: np.ndarray, repetitions: int, x0=None,
                        group=None, shuffled: bool = False):
    """``only_best=True`` without moving every chain: each rank reduces its own block, one
    all_gather of (energy, global replica id) per rank names the winner — lowest energy, lowest
    id among equals, i.e. the first minimum of the single-GPU result — and the winner's rank
    broadcasts its configuration.  16 B per rank + one configuration instead of R of them."""
    import torch
    import torch.distributed as dist

    from .annealer import anneal_raw

    check_device_binding(group)
    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    offset, count = shard_range(repetitions, world, me)
    words = (hamiltonian.size + 63) // 64
    device = _device_for(group)
    mine = np.array([np.iinfo(np.int64).max, np.iinfo(np.int64).max], dtype=np.int64)
    local_x = np.zeros(max(words, 1), dtype=np.uint64)
    local_e = np.inf
    if count > 0:
        xs, es = anneal_raw(hamiltonian, seed, betas, count, offset, x0, shuffled=shuffled)
        best = int(np.argmin(es))
        local_x[:words] = xs[best]
        local_e = float(es[best])
        mine[0] = np.float64(local_e).view(np.int64)
        mine[1] = offset + best
    parts = [torch.empty(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(mine).to(device), group=group)
    table = np.stack([p.cpu().numpy() for p in parts])
    energies = table[:, 0].copy().view(np.float64)
    energies[table[:, 1] == np.iinfo(np.int64).max] = np.inf  # ranks without chains
    lowest = energies.min()
    candidates = np.nonzero(energies == lowest)[0]
    winner = int(candidates[np.argmin(table[candidates, 1])])
    box = torch.from_numpy(local_x.view(np.int64).copy()).to(device)
    src = dist.get_global_rank(group, winner) if group is not None else winner
    dist.broadcast(box, src=src, group=group)
    x = box.cpu().numpy().view(np.uint64)[:words].copy()
    return x, float(lowest)


_chains_local = 0


class chains_local:
    """Context manager: inside it ``annealer.anneal`` keeps all chains of a call on this rank
    (no collectives) — needed when the ranks work on DIFFERENT problems, as in
    :func:`map_sharded`."""

    def __enter__(self):
        global _chains_local
        _chains_local += 1
        return self

    def __exit__(self, *exc):
        global _chains_local
        _chains_local -= 1
        return False


def shards_chains() -> bool:
    """True when ``annealer.anneal`` should split its repetitions over the ranks."""
    return _chains_local == 0 and world_size() > 1


def map_sharded(items, work, group=None):
    """``[work(x) for x in items]`` with item ``c`` computed on rank ``c mod world`` (SURVEY §8e:
    cluster instances are independent problems) and the picklable results gathered on every
    rank, in item order.  Single process: a plain loop."""
    d = _dist()
    if d is None or d.get_world_size(group) == 1:
        return [work(x) for x in items]
    world, me = d.get_world_size(group), d.get_rank(group)
    with chains_local():
        mine = [(c, work(x)) for c, x in enumerate(items) if c % world == me]
    parts = [None] * world
    d.all_gather_object(parts, mine, group=group)
    out = [None] * len(items)
    for part in parts:
        for c, result in part:
            out[c] = result
    return out


def map_sharded_many(items, work_many, group=None):
    """:func:`map_sharded` for a worker that takes a LIST of items and returns one result per
    item (so that a rank can batch its share of the problems on its GPU): rank k gets the items
    ``c`` with ``c mod world == k``, in order; results come back in item order on every rank."""
    d = _dist()
    if d is None or d.get_world_size(group) == 1:
        return list(work_many(list(items)))
    world, me = d.get_world_size(group), d.get_rank(group)
    share = [c for c in range(len(items)) if c % world == me]
    with chains_local():
        done = list(work_many([items[c] for c in share]))
    if len(done) != len(share):
        raise ValueError("work_many returned %d results for %d items" % (len(done), len(share)))
    parts = [None] * world
    d.all_gather_object(parts, list(zip(share, done)), group=group)
    out = [None] * len(items)
    for part in parts:
        for c, result in part:
            out[c] = result
    return out

"""CPU, world_size 2, gloo: the replica-sharding + final-gather layer
(annealing_sign_problem_amd/distributed.py).  No GPU here, so each rank's local
block of chains is produced by the oracle standing in for the device call; what
is under test is that sharding by GLOBAL replica id and gathering reproduces the
single-process result bit for bit on every rank."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, repetitions, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import oracle
    from annealing_sign_problem_amd import annealer, distributed, synthetic

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    J, h, _ = synthetic.planted_cluster(300, seed=3, mean_degree=8.0)
    betas = np.geomspace(0.5, 100.0, 15)

    def fake_anneal_raw(hamiltonian, seed, betas, count, offset=0, x0=None, shuffled=False):
        run = oracle.sa_anneal_shuffled if shuffled else oracle.sa_anneal
        xs, es, _, _ = run(hamiltonian.exchange, hamiltonian.field, seed, betas, count, offset, x0, 40)
        return xs, es

    annealer.anneal_raw = fake_anneal_raw
    ham = annealer.Hamiltonian(J, h)
    assert distributed.world_size() == world and distributed.rank() == rank
    xs, es = distributed.anneal_sharded(ham, 999, betas, repetitions)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), xs=xs, es=es)
    # the public entry point picks the sharded path by itself
    annealer_info = type("I", (), {"beta0_auto": 0.5, "beta1_auto": 100.0})()
    ham.info = lambda: annealer_info
    x, e = annealer.anneal(ham, seed=999, number_sweeps=15, repetitions=repetitions, sweep_order="colour")
    np.savez(os.path.join(out_dir, "best%d.npz" % rank), x=x, e=e)
    xs_all, es_all = annealer.anneal(ham, seed=999, number_sweeps=15, repetitions=repetitions,
                                     only_best=False, sweep_order="colour")
    assert np.array_equal(xs_all, xs) and es_all.tobytes() == es.tobytes()
    # seed=None: rank 0 draws, every rank must run the same stream
    xr, er = annealer.anneal(ham, seed=None, number_sweeps=15, repetitions=repetitions, sweep_order="colour")
    np.savez(os.path.join(out_dir, "drawn%d.npz" % rank), x=xr, e=er)
    # the shuffled visiting order shards the same way (global replica ids key its random words too)
    xs_s, es_s = annealer.anneal(ham, seed=999, number_sweeps=15, repetitions=repetitions,
                                 only_best=False, sweep_order="shuffled")
    xb, eb = annealer.anneal(ham, seed=999, number_sweeps=15, repetitions=repetitions,
                             sweep_order="shuffled")
    xd, ed = annealer.anneal(ham, seed=None, number_sweeps=15, repetitions=repetitions,
                             sweep_order="shuffled")
    np.savez(os.path.join(out_dir, "shuffled%d.npz" % rank), xs=xs_s, es=es_s, x=xb, e=eb, xd=xd, ed=ed)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("repetitions", [7, 2, 1])
def test_sharded_anneal_equals_single_process(tmp_path, repetitions):
    import torch.multiprocessing as mp

    import oracle
    from annealing_sign_problem_amd import synthetic

    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, repetitions, str(tmp_path)), nprocs=world, join=True)
    J, h, _ = synthetic.planted_cluster(300, seed=3, mean_degree=8.0)
    betas = np.geomspace(0.5, 100.0, 15)
    xs, es, _, _ = oracle.sa_anneal(J, h, 999, betas, repetitions, 0, None, 40)
    for rank in range(world):
        got = np.load(tmp_path / ("rank%d.npz" % rank))
        assert np.array_equal(got["xs"], xs) and got["es"].tobytes() == es.tobytes()
        best = np.load(tmp_path / ("best%d.npz" % rank))
        k = int(np.argmin(es))
        assert np.array_equal(best["x"], xs[k]) and float(best["e"]) == es[k]
    drawn = [np.load(tmp_path / ("drawn%d.npz" % rank)) for rank in range(world)]
    assert np.array_equal(drawn[0]["x"], drawn[1]["x"]) and float(drawn[0]["e"]) == float(drawn[1]["e"])
    sxs, ses, _, _ = oracle.sa_anneal_shuffled(J, h, 999, betas, repetitions, 0, None, 40)
    shuffled = [np.load(tmp_path / ("shuffled%d.npz" % rank)) for rank in range(world)]
    for got in shuffled:
        assert np.array_equal(got["xs"], sxs) and got["es"].tobytes() == ses.tobytes()
        k = int(np.argmin(ses))
        assert np.array_equal(got["x"], sxs[k]) and float(got["e"]) == ses[k]
    assert np.array_equal(shuffled[0]["xd"], shuffled[1]["xd"])  # seed=None: one draw for all ranks


def _cluster_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import oracle
    from annealing_sign_problem_amd import annealer, distributed, synthetic

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)

    def fake_anneal_raw(hamiltonian, seed, betas, count, offset=0, x0=None, shuffled=False):
        xs, es, _, _ = oracle.sa_anneal(hamiltonian.exchange, hamiltonian.field, seed, betas, count,
                                        offset, x0, 40)
        return xs, es

    annealer.anneal_raw = fake_anneal_raw
    info = type("I", (), {"beta0_auto": 0.5, "beta1_auto": 50.0})()

    def solve(k):
        # a different problem per item; anneal must not try to split its chains over the ranks
        J, h, _ = synthetic.planted_cluster(100 + 10 * k, seed=k, mean_degree=6.0)
        ham = annealer.Hamiltonian(J, h)
        ham.info = lambda: info
        x, e = annealer.anneal(ham, seed=7 + k, number_sweeps=10, repetitions=3, sweep_order="colour")
        return (k, float(e), x.tobytes())

    results = distributed.map_sharded(list(range(7)), solve)   # 4 items on rank 0, 3 on rank 1
    assert distributed.shards_chains()                          # back to normal outside
    import pickle
    with open(os.path.join(out_dir, "clusters%d.pkl" % rank), "wb") as f:
        pickle.dump(results, f)
    dist.barrier()
    dist.destroy_process_group()


def test_cluster_instances_shard_over_ranks(tmp_path):
    """cluster c -> rank c mod world, results gathered in order on every rank, and
    inside a sharded item the annealer keeps its chains on the rank (no collective)."""
    import pickle

    import torch.multiprocessing as mp

    import oracle
    from annealing_sign_problem_amd import synthetic

    world = 2
    mp.spawn(_cluster_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [pickle.load(open(tmp_path / ("clusters%d.pkl" % r), "rb")) for r in range(world)]
    assert got[0] == got[1] and [g[0] for g in got[0]] == list(range(7))
    for k, e, xbytes in got[0]:
        J, h, _ = synthetic.planted_cluster(100 + 10 * k, seed=k, mean_degree=6.0)
        betas = np.geomspace(0.5, 50.0, 10)
        xs, es, _, _ = oracle.sa_anneal(J, h, 7 + k, betas, 3, 0, None, 40)
        best = int(np.argmin(es))
        assert e == es[best] and xbytes == xs[best].tobytes()


def _main_worker(rank, world, port, out_path, expect_refusal):
    """sampled_components.main() as `python -m torch.distributed.run` would start it: nothing
    but RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment.  The device stages are
    replaced by a deterministic stand-in (no GPU here); what is under test is the rank
    discovery, the single writer and the gathered output."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from annealing_sign_problem_amd import distributed, sampled_components as sc

    def fake_clusters(hamiltonian, ground_state, number_samples, *args):
        states = hamiltonian.basis.states
        return [states[7 * c: 7 * c + 5 + c].copy() for c in range(number_samples)]

    def fake_process(cluster, hamiltonian, ground_state, noisy, fn, order, cutoff, annealing,
                     sweep_order=None):
        assert not distributed.shards_chains()  # inside a sharded item chains stay on the rank
        assert sweep_order == "shuffled"       # --sweep-order reaches the solver
        if distributed.rank() == 0 and os.path.exists(out_path):
            # how many result lines the file held when this cluster was being solved
            done = sum(1 for l in open(out_path) if not l.startswith("#"))
            with open(out_path + ".growth", "a") as f:
                f.write("%d %d\n" % (int(cluster.size) - 5, done))
        return [sc.OptimizationResult(int(cluster.size) + i, float(cluster[0] % 97) / 97.0, 0.5, 0.25,
                                      0.125, float(distributed.world_size())) for i in range(order + 1)]

    def fake_generate(*args):
        assert distributed.rank() == 0  # rank 0 grows the clusters, the others receive them
        return fake_clusters(*args)

    sc.generate_clusters = fake_generate
    sc.iter_clusters = lambda *args: iter(fake_generate(*args))  # (a single process grows them one by one)
    sc.process_cluster = fake_process
    sc.process_clusters_batched = lambda clusters, *rest, jobs=1, sweep_order=None: [
        fake_process(c, *rest, sweep_order=sweep_order) for c in clusters]
    argv = ["--model", "heisenberg_kagome_16", "--output", out_path, "--order", "1",
            "--number-samples", "5", "--seed", "3", "--batch", "1", "--sweep-order", "shuffled"]
    if expect_refusal:
        with pytest.raises(SystemExit):
            sc.main(argv)
        return
    sc.main(argv)
    assert distributed.world_size() == 1  # main() tore its own group down


def test_pipeline_main_initialises_ranks_from_env(tmp_path):
    """under torch.distributed.run every rank used to see world_size 1 and raced on
    the output file.  Two ranks, env-var rendezvous: one header, each cluster's line once, in
    cluster order; a second launch onto the same file stops on EVERY rank."""
    import torch.multiprocessing as mp

    out = str(tmp_path / "clusters.csv")
    mp.spawn(_main_worker, args=(2, _free_port(), out, False), nprocs=2, join=True)
    lines = [l for l in open(out).read().splitlines() if not l.startswith("#")]
    assert len(lines) == 5
    sizes = [int(l.split(",")[0]) for l in lines]
    assert sizes == [5 + c for c in range(5)]
    assert all(l.split(",")[5] == "2.00000000e+00" for l in lines)  # computed under world size 2
    # rank 0 appends round by round (--batch 1 x 2 ranks = 2 clusters a round), so a killed job
    # keeps its finished rounds: its clusters 0, 2, 4 saw 0, 2 and 4 lines already in the file
    growth = dict(tuple(map(int, l.split())) for l in open(out + ".growth"))
    assert growth == {0: 0, 2: 2, 4: 4}
    # the same file as a single process writes (up to the column the stand-in fills with the
    # world size)
    single = str(tmp_path / "single.csv")
    mp.spawn(_main_worker, args=(1, _free_port(), single, False), nprocs=1, join=True)
    strip = lambda path: [",".join(c for i, c in enumerate(l.split(",")) if i % 6 != 5)
                          for l in open(path).read().splitlines()]
    assert strip(single) == strip(out)
    mp.spawn(_main_worker, args=(2, _free_port(), out, True), nprocs=2, join=True)


def _pipelined_worker(rank, world, port, out_path):
    """sampled_components.main() with --batch 2 and annealing under two ranks: the branch that
    builds a rank's share of the next round while the current one anneals.  Stand-in stages."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import threading

    from annealing_sign_problem_amd import distributed, sampled_components as sc

    def fake_clusters(hamiltonian, ground_state, number_samples, *args):
        states = hamiltonian.basis.states
        return [states[11 * c: 11 * c + 4 + c].copy() for c in range(number_samples)]

    main_thread = threading.get_ident()
    log = []

    def fake_stage(clusters, hamiltonian, ground_state, noisy, fn, order, cutoff, jobs=1):
        log.append(("stage", [int(c.size) - 4 for c in clusters], threading.get_ident() != main_thread))
        return [(i, ("model", int(c.size), o), None, None,
                 sc.OptimizationResult(int(c.size) + o, 0.5, 0.25, float("nan"), float("nan"), 1.0))
                for i, c in enumerate(clusters) for o in range(order + 1)]

    def fake_anneal(staged, clusters, annealing, sweep_order=None):
        assert annealing and sweep_order == "shuffled" and not distributed.shards_chains()  # (the default)
        log.append(("anneal", [int(c.size) - 4 for c in clusters]))
        results = [[] for _ in clusters]
        for index, model, _, _, r in staged:
            assert model[1] == int(clusters[index].size)  # the models ARE those of these clusters
            r.sa_accuracy, r.sa_overlap = float(model[1] % 7) / 7.0, float(distributed.rank())
            results[index].append(r)
        return results

    sc.generate_clusters = lambda *args: fake_clusters(*args)
    sc.iter_clusters = lambda *args: iter(fake_clusters(*args))
    sc.stage_clusters = fake_stage
    sc.anneal_staged = fake_anneal
    sc.main(["--model", "heisenberg_kagome_16", "--output", out_path, "--order", "1", "--number-samples", "9",
             "--seed", "3", "--batch", "2", "--annealing"])
    if world > 1:
        # rounds of 2 x 2 clusters: this rank's shares, each staged on the builder thread (the
        # next one is submitted before the current one is annealed) and annealed in order
        shares = [[c for c in range(start, min(start + 4, 9)) if (c - start) % 2 == rank] for start in (0, 4, 8)]
        staged = [e[1] for e in log if e[0] == "stage"]
        annealed = [e[1] for e in log if e[0] == "anneal"]
        assert staged == shares and annealed == shares
        assert all(e[2] for e in log if e[0] == "stage")


def test_pipeline_rounds_are_pipelined_under_ranks(tmp_path):
    """--batch > 1 with annealing under two ranks: every rank stages its share of the next round
    on a second thread while it anneals the current one; the file is the single process's."""
    import torch.multiprocessing as mp

    out, single = str(tmp_path / "two.csv"), str(tmp_path / "one.csv")
    mp.spawn(_pipelined_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    mp.spawn(_pipelined_worker, args=(1, _free_port(), single), nprocs=1, join=True)
    rows = [l for l in open(out).read().splitlines() if not l.startswith("#")]
    assert len(rows) == 9 and [int(l.split(",")[0]) for l in rows] == [4 + c for c in range(9)]
    strip = lambda path: [",".join(c for i, c in enumerate(l.split(",")) if i % 6 != 4)  # (sa_overlap = the rank)
                          for l in open(path).read().splitlines()]
    assert strip(out) == strip(single)


def test_growth_child_failure_stops_the_parent_with_its_message(tmp_path, monkeypatch):
    """The forked child that grows the clusters (sampled_components.clusters_from_child) fails:
    the parent stops with the child's error instead of waiting or writing a partial file body;
    with ASP_GROW_IN_PLACE=1 the same error surfaces directly."""
    sys.path.insert(0, ROOT)
    from annealing_sign_problem_amd import sampled_components as sc

    def broken(hamiltonian, ground_state, number_samples, *args):
        yield hamiltonian.basis.states[:5].copy()
        raise RuntimeError("no such lattice")

    monkeypatch.setattr(sc, "iter_clusters", broken)
    monkeypatch.setattr(sc, "process_cluster", lambda cluster, *rest: [
        sc.OptimizationResult(int(cluster.size), 0.5, 0.5, float("nan"), float("nan"), 1.0)])
    argv = ["--model", "heisenberg_kagome_16", "--order", "0", "--number-samples", "3", "--seed", "1",
            "--no-annealing"]
    with pytest.raises(SystemExit) as stop:
        sc.main(argv + ["--output", str(tmp_path / "child.csv")])
    assert "cluster generation failed" in str(stop.value) and "no such lattice" in str(stop.value)
    monkeypatch.setenv("ASP_GROW_IN_PLACE", "1")
    with pytest.raises(RuntimeError, match="no such lattice"):
        sc.main(argv + ["--output", str(tmp_path / "in_place.csv")])


def test_pipeline_worker_processes_write_the_single_process_file(tmp_path, monkeypatch):
    """--workers N: the parent loads the inputs without touching the GPU, a child grows the
    clusters, N forked workers share the inputs and take rounds of clusters from a queue, the
    parent appends the lines in cluster order.  Device stages replaced by a deterministic
    stand-in (no GPU here); the file must be the one a single process writes."""
    sys.path.insert(0, ROOT)
    from annealing_sign_problem_amd import sampled_components as sc

    def fake_clusters(hamiltonian, ground_state, number_samples, *args):
        states = hamiltonian.basis.states
        draws = np.random.randint(0, 50, size=number_samples)  # consumes the seeded stream like the real one
        return [states[int(d): int(d) + 4 + c].copy() for c, d in enumerate(draws)]

    def fake_process(cluster, hamiltonian, ground_state, noisy, fn, order, cutoff, annealing,
                     sweep_order=None):
        assert fn(cluster).shape == (cluster.size,)  # the workers' own log-amplitude closure works
        return [sc.OptimizationResult(int(cluster.size) + i, float(cluster[0] % 97) / 97.0,
                                      float(np.abs(noisy[3])), 0.25, 0.125, float(os.getpid() != 0))
                for i in range(order + 1)]

    monkeypatch.setattr(sc, "generate_clusters", fake_clusters)
    # (the single process streams them from a forked child, one by one: sc.clusters_from_child)
    monkeypatch.setattr(sc, "iter_clusters", lambda *args: iter(fake_clusters(*args)))
    monkeypatch.setattr(sc, "process_cluster", fake_process)
    monkeypatch.setattr(sc, "process_clusters_batched",
                        lambda clusters, *rest, jobs=1, sweep_order=None: [
                            fake_process(c, *rest, sweep_order=sweep_order) for c in clusters])
    common = ["--model", "heisenberg_kagome_16", "--order", "1", "--number-samples", "9", "--seed", "11",
              "--noise", "0.3", "--no-annealing"]
    single, forked = str(tmp_path / "single.csv"), str(tmp_path / "forked.csv")
    sc.main(common + ["--output", single])
    sc.main(common + ["--output", forked, "--workers", "3", "--batch", "2"])
    assert open(single).read() == open(forked).read()
    assert len([l for l in open(forked) if not l.startswith("#")]) == 9
    with pytest.raises(SystemExit):
        sc.main(common + ["--output", forked, "--workers", "3"])  # refuses to overwrite


def test_fork_guard_sees_gpu_state_made_outside_the_library(tmp_path, monkeypatch):
    """asp_device_touched() only knows HIP calls made through libasp_hip.  Under a
    profiler's preloaded tool library (or a host that used torch.cuda) the GPU is initialised
    before main() runs: the clusters must then be grown in place and --workers refused, not
    forked into children that would call HIP in a copy of the runtime."""
    sys.path.insert(0, ROOT)
    import multiprocessing

    from annealing_sign_problem_amd import sampled_components as sc

    for name in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "LD_PRELOAD", "RANK", "ASP_GROW_IN_PLACE"):
        monkeypatch.delenv(name, raising=False)
    assert not sc.gpu_maybe_initialised()
    for name, value in (("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"),
                        ("HSA_TOOLS_LIB", "librocprofiler64.so.1"),
                        ("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")):
        monkeypatch.setenv(name, value)
        assert sc.gpu_maybe_initialised()
        monkeypatch.delenv(name)
    monkeypatch.setenv("LD_PRELOAD", "/usr/lib/libjemalloc.so")
    assert not sc.gpu_maybe_initialised()
    monkeypatch.delenv("LD_PRELOAD")

    forks = []
    real_context = multiprocessing.get_context

    def watched(method=None):
        forks.append(method)
        return real_context(method)

    monkeypatch.setattr(multiprocessing, "get_context", watched)
    monkeypatch.setattr(sc, "iter_clusters", lambda hamiltonian, *rest: iter(
        [hamiltonian.basis.states[:5].copy(), hamiltonian.basis.states[9:16].copy()]))
    monkeypatch.setattr(sc, "process_cluster", lambda cluster, *rest: [
        sc.OptimizationResult(int(cluster.size), 0.5, 0.5, float("nan"), float("nan"), 1.0)])
    argv = ["--model", "heisenberg_kagome_16", "--order", "0", "--number-samples", "2", "--seed", "1",
            "--no-annealing"]
    sc.main(argv + ["--output", str(tmp_path / "child.csv")])
    assert forks == ["fork"]  # an untouched process grows its clusters in a forked child
    del forks[:]
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "librocprofiler-sdk-tool.so")
    sc.main(argv + ["--output", str(tmp_path / "in_place.csv")])
    with pytest.warns(UserWarning, match="--workers"):
        sc.main(argv + ["--output", str(tmp_path / "workers.csv"), "--workers", "2"])
    assert forks == []  # under the profiler: in place, and threads instead of worker processes
    assert (open(tmp_path / "child.csv").read() == open(tmp_path / "in_place.csv").read()
            == open(tmp_path / "workers.csv").read())


def _dying_chunk(indices):
    if 2 in indices:
        os._exit(9)  # what a GPU fault or an out-of-memory kill looks like from the parent
    return ["%d" % i for i in indices]


def test_a_dying_worker_process_stops_the_run(tmp_path, monkeypatch):
    """multiprocessing.Pool replaces a dead worker and never completes its task — the
    parent waited forever.  The executor reports the broken pool; the parent names the round."""
    sys.path.insert(0, ROOT)
    from annealing_sign_problem_amd import sampled_components as sc

    monkeypatch.setattr(sc, "generate_clusters", lambda hamiltonian, *rest: [
        hamiltonian.basis.states[5 * c: 5 * c + 4].copy() for c in range(6)])
    monkeypatch.setattr(sc, "_worker_chunk", _dying_chunk)
    monkeypatch.setattr(sc, "_worker_init", lambda: None)
    out = tmp_path / "dying.csv"
    with pytest.raises(SystemExit) as stop:
        sc.main(["--model", "heisenberg_kagome_16", "--order", "0", "--number-samples", "6", "--seed", "1",
                 "--no-annealing", "--workers", "2", "--batch", "1", "--output", str(out)])
    assert "worker process died" in str(stop.value) and "clusters 2..2" in str(stop.value)
    assert [l for l in open(out).read().splitlines() if not l.startswith("#")] == ["0", "1"]


_FAILING_RANK = r"""
import os, sys
sys.path.insert(0, {root!r})
from annealing_sign_problem_amd import distributed, sampled_components as sc
rank = int(os.environ["RANK"])

def fake_clusters(hamiltonian, ground_state, number_samples, *args):
    states = hamiltonian.basis.states
    return [states[11 * c: 11 * c + 4 + c].copy() for c in range(number_samples)]

def fake_stage(clusters, *rest, **kw):
    if rank == 1 and {where!r} == "stage":
        raise RuntimeError("stage failed on rank 1")
    return [(i, ("model", int(c.size), 0), None, None,
             sc.OptimizationResult(int(c.size), 0.5, 0.25, float("nan"), float("nan"), 1.0))
            for i, c in enumerate(clusters)]

def fake_anneal(staged, clusters, annealing, sweep_order=None):
    results = [[] for _ in clusters]
    for index, model, _, _, r in staged:
        results[index].append(r)
    return results

def fake_load(args):
    if rank == 1 and {where!r} == "load":
        raise SystemExit("rank 1 cannot read its input")
    return real_load(args)

real_load = sc.load_input
sc.load_input = fake_load
sc.generate_clusters = fake_clusters
sc.stage_clusters = fake_stage
sc.anneal_staged = fake_anneal
sc.main(["--model", "heisenberg_kagome_16", "--output", {out!r}, "--order", "0", "--number-samples", "6",
         "--seed", "3", "--batch", "2", "--annealing"])
"""


@pytest.mark.parametrize("where", ["stage", "load"])
def test_a_failing_rank_ends_every_rank_without_abort(tmp_path, where):
    """main() tears its process group down on EVERY way out.  Rank 1 fails — an
    exception inside stage_clusters on the builder thread, or a SystemExit while loading the
    inputs —: both ranks end with a non-zero status, rank 1 with its message, neither with
    torch's 'terminate called without an active exception' (the abort of commit 1f46a71)."""
    import subprocess

    port = _free_port()
    script = _FAILING_RANK.format(root=ROOT, where=where, out=str(tmp_path / "out.csv"))
    ranks = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ASP_DIST_BACKEND="gloo", ASP_SINGLE_DEVICE="1")
        ranks.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outputs = []
    for process in ranks:
        try:
            _, err = process.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for other in ranks:
                other.kill()
            pytest.fail("a rank kept waiting for a rank that had failed")
        outputs.append((process.returncode, err))
    (rc0, err0), (rc1, err1) = outputs
    assert rc0 != 0 and rc1 != 0, outputs
    assert ("stage failed on rank 1" if where == "stage" else "rank 1 cannot read its input") in err1
    assert "terminate called" not in err0 and "terminate called" not in err1
    assert rc0 > 0 and rc1 > 0, "a rank was killed by a signal: %r" % ((rc0, rc1),)

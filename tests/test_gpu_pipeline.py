"""GPU: the whole sampled-cluster pipeline (the call sequence of `make kagome_36`,
experiments/sampled_connected_components.py:726-751) on a real 16-site system."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sampled_components_end_to_end(tmp_path):
    from annealing_sign_problem_amd import sampled_components

    out = tmp_path / "kagome16.csv"
    sampled_components.main(["--model", "heisenberg_kagome_16", "--output", str(out), "--order", "2",
                             "--number-samples", "3", "--seed", "435834", "--global-cutoff", "1e-6",
                             "--min-cluster-size", "50", "--max-cluster-size", "400"])
    lines = out.read_text().splitlines()
    header = [l for l in lines if l.startswith("#")]
    rows = [l for l in lines if not l.startswith("#")]
    assert header[-1] == "# size,greedy_accuracy,greedy_overlap,sa_accuracy,sa_overlap,amplitude_overlap"
    assert len(rows) == 3
    for row in rows:
        v = np.array([float(t) for t in row.split(",")]).reshape(3, 6)   # order 0, 1, 2
        assert np.all(np.diff(v[:, 0]) > 0)                # extensions grow the cluster
        assert np.all((v[:, 1:5] >= 0) & (v[:, 1:5] <= 1 + 1e-12))
        assert np.allclose(v[:, 5], 1.0)                   # no noise: amplitudes identical
        assert v[2, 3] >= v[0, 3] - 0.05                   # SA accuracy does not degrade with order
        assert v[2, 3] > 0.9 and v[2, 4] > 0.9             # order 2 recovers the signs well
    with pytest.raises(SystemExit):
        sampled_components.main(["--model", "heisenberg_kagome_16", "--output", str(out), "--order", "0"])


def test_noise_lowers_amplitude_overlap_only_slightly(tmp_path):
    from annealing_sign_problem_amd import sampled_components

    out = tmp_path / "noisy.csv"
    sampled_components.main(["--model", "sk_16_1", "--output", str(out), "--order", "1", "--noise", "0.79",
                             "--number-samples", "2", "--no-annealing", "--seed", "7",
                             "--max-cluster-size", "200"])
    rows = [l for l in out.read_text().splitlines() if not l.startswith("#")]
    v = np.array([float(t) for t in rows[0].split(",")]).reshape(2, 6)
    assert np.all(np.isnan(v[:, 3:5]))                     # --no-annealing: SA columns are NaN
    assert np.all((v[:, 5] > 0.5) & (v[:, 5] < 1.0))


def test_concurrent_clusters_give_identical_output(tmp_path):
    from annealing_sign_problem_amd import sampled_components

    common_args = ["--model", "heisenberg_kagome_16", "--order", "1", "--number-samples", "6",
                   "--seed", "99", "--max-cluster-size", "300"]
    a, b = tmp_path / "serial.csv", tmp_path / "jobs4.csv"
    sampled_components.main(common_args + ["--output", str(a)])
    sampled_components.main(common_args + ["--output", str(b), "--jobs", "4"])
    assert a.read_text().replace("serial", "") == b.read_text().replace("jobs4", "")


def test_cluster_growth_on_gpu_action_equals_host_action(models):
    """create_small_cluster_around_point applies whole frontiers through asp_operator_apply;
    with the same numpy seed it must grow the very cluster the host action grows."""
    from annealing_sign_problem_amd import operators, sampled_components

    op = operators.Operator.from_config(models["j1j2_square_4x4"])
    op.basis.build()

    class Foreign:
        basis = op.basis

        def batched_apply(self, x):
            return op.batched_apply(x)

    for seed, size in [(1, 40), (2, 700), (3, 2500)]:
        start = int(op.basis.states[seed * 1000])
        np.random.seed(seed)
        on_gpu = sampled_components.create_small_cluster_around_point(start, op, required_size=size)
        np.random.seed(seed)
        on_host = sampled_components.create_small_cluster_around_point(start, Foreign(),
                                                                       required_size=size)
        assert on_gpu == on_host and len(on_gpu) == size


def test_sparsify_on_gpu_matches_reference_golden_and_oracle():
    """asp_sparsify_component: cutoff + connected component + un-pruned block, against the
    reference's own sparsify output and the scipy oracle on random non-symmetric matrices."""
    import scipy.sparse

    import oracle
    from conftest import golden
    from annealing_sign_problem_amd import _lib, common
    from annealing_sign_problem_amd import annealer as sa

    g = golden("make_ising_kagome16_cluster.npz")
    n = g["ext_spins"].shape[0]
    ext = scipy.sparse.coo_matrix((g["ext_data"], (g["ext_row"], g["ext_col"])), shape=(n, n))
    idx = np.searchsorted(g["basis_states"], g["ext_spins"])
    psi = g["ground_state"][idx]
    model = common.IsingModel(g["ext_spins"], None, sa.Hamiltonian(ext, np.zeros(n)),
                              sa.signs_to_bits(np.sign(psi)))
    out = common.sparsify_using_global_cutoff(model, float(g["sp_reltol"]), g["spins"])
    assert np.array_equal(out.spins, g["sp_spins"])
    m = scipy.sparse.coo_matrix(out.ising_hamiltonian.exchange)
    assert np.array_equal(m.row, g["sp_row"]) and np.array_equal(m.col, g["sp_col"])
    assert m.data.tobytes() == g["sp_data"].tobytes()
    assert np.array_equal(out.initial_signs, g["sp_x0"])

    rng = np.random.default_rng(8)
    for trial, (size, density, symmetric) in enumerate([(300, 0.02, True), (800, 0.006, False),
                                                        (2000, 0.002, False), (50, 0.3, True)]):
        a = scipy.sparse.random(size, size, density=density, random_state=trial, format="csr",
                                data_rvs=lambda k: rng.normal(size=k) * np.exp(rng.normal(size=k) * 3))
        a = (a + a.T).tocsr() if symmetric else a.tocsr()
        a.sort_indices()
        frozen = np.zeros(size, dtype=bool)
        anchor = int(rng.integers(size))
        frozen[anchor] = True
        for reltol in (0.0, 1e-3, 0.3):
            keep_o, block_o = oracle.sparsify_component(a, frozen, reltol, anchor)
            keep, block = common.sparsify_component(a, frozen, reltol, anchor)
            assert np.array_equal(keep, keep_o)
            bo = scipy.sparse.csr_matrix(block_o)
            bo.sort_indices()
            assert np.array_equal(block.indptr, bo.indptr)
            assert np.array_equal(block.indices, bo.indices)
            assert block.data.tobytes() == bo.data.tobytes()
        # freezing a second spin of the anchor's component keeps pairs of frozen spins coupled
        keep_o, _ = oracle.sparsify_component(a, frozen, 0.0, anchor)
        others = np.nonzero(keep_o)[0]
        if others.size > 3:
            frozen2 = frozen.copy()
            frozen2[others[-1]] = True
            k1, _ = common.sparsify_component(a, frozen2, 0.0, anchor)
            assert np.array_equal(k1, keep_o)
    # a frozen spin cut off by the cutoff: the reference asserts, the library reports
    lonely = scipy.sparse.csr_matrix(np.array([[0, 1.0, 0], [1.0, 0, 1e-9], [0, 1e-9, 0]]))
    with pytest.raises(_lib.AspError, match="frozen"):
        common.sparsify_component(lonely, np.array([True, False, True]), 1e-3, 0)
    keep, block = common.sparsify_component(lonely, np.array([True, False, False]), 1e-3, 0)
    assert keep.tolist() == [True, True, False] and block.shape == (2, 2)


def test_components_on_long_chains_match_scipy():
    """Path-like graphs in random vertex order give the concurrent union-find long parent chains
    (the case where an in-place flatten would race); every trial must equal scipy."""
    import scipy.sparse

    import oracle
    from annealing_sign_problem_amd import common

    rng = np.random.default_rng(1)
    for trial in range(40):
        n = int(rng.integers(50, 30000))
        perm = rng.permutation(n)
        cut = rng.random(n - 1) < 0.98
        rows = np.concatenate([perm[:-1][cut], rng.integers(0, n, n // 20)])
        cols = np.concatenate([perm[1:][cut], rng.integers(0, n, n // 20)])
        vals = rng.normal(size=rows.size) * np.exp(rng.normal(size=rows.size) * 3)
        a = scipy.sparse.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
        a = (a + a.T).tocsr() if trial % 2 == 0 else a
        a.sum_duplicates()
        a.sort_indices()
        frozen = np.zeros(n, dtype=bool)
        anchor = int(rng.integers(n))
        frozen[anchor] = True
        reltol = [0.0, 1e-4, 1e-2][trial % 3]
        keep_o, block_o = oracle.sparsify_component(a, frozen, reltol, anchor)
        keep, block = common.sparsify_component(a, frozen, reltol, anchor)
        bo = scipy.sparse.csr_matrix(block_o)
        bo.sort_indices()
        assert np.array_equal(keep, keep_o), trial
        assert np.array_equal(block.indices, bo.indices) and block.data.tobytes() == bo.data.tobytes()


def test_batched_annealing_gives_identical_output(tmp_path):
    """The pipeline's --batch (all models' annealing chains in one asp_sa_anneal_batch call) must
    write the file the per-model loop writes, SA columns included."""
    from annealing_sign_problem_amd import sampled_components

    common_args = ["--model", "heisenberg_kagome_16", "--order", "1", "--number-samples", "7",
                   "--seed", "5", "--max-cluster-size", "400"]
    a, b, c = tmp_path / "loop.csv", tmp_path / "batch.csv", tmp_path / "batch3.csv"
    sampled_components.main(common_args + ["--output", str(a), "--batch", "1"])
    sampled_components.main(common_args + ["--output", str(b)])
    sampled_components.main(common_args + ["--output", str(c), "--batch", "3"])
    d = tmp_path / "batch_jobs.csv"
    sampled_components.main(common_args + ["--output", str(d), "--jobs", "4"])  # threaded staging
    assert a.read_text() == b.read_text() == c.read_text() == d.read_text()
    rows = [l for l in a.read_text().splitlines() if not l.startswith("#")]
    v = np.array([float(t) for t in rows[0].split(",")]).reshape(2, 6)
    assert np.all(np.isfinite(v[:, 3:5]))  # the SA columns are filled


def test_pipeline_with_the_shuffled_visiting_order(tmp_path):
    """--sweep-order shuffled: the annealing of the pipeline with the reference annealer's visiting
    order, batched (shared launches) and per model: the same file; other SA columns than the
    colour order's, the same greedy columns."""
    from annealing_sign_problem_amd import sampled_components

    common_args = ["--model", "heisenberg_kagome_16", "--order", "1", "--number-samples", "6",
                   "--seed", "5", "--max-cluster-size", "300"]
    a, b, c = tmp_path / "loop.csv", tmp_path / "batch.csv", tmp_path / "colour.csv"
    sampled_components.main(common_args + ["--output", str(a), "--batch", "1", "--sweep-order", "shuffled"])
    sampled_components.main(common_args + ["--output", str(b), "--sweep-order", "shuffled"])
    sampled_components.main(common_args + ["--output", str(c), "--sweep-order", "colour"])
    assert a.read_text() == b.read_text()
    # the reference's law is the DEFAULT of --annealing: no flag = shuffled
    d = tmp_path / "default.csv"
    sampled_components.main(common_args + ["--output", str(d)])
    data = lambda path: [l for l in path.read_text().splitlines() if not l.startswith("#")]
    assert data(d) == data(b)
    rows = lambda path: np.array([[float(t) for t in l.split(",")] for l in path.read_text().splitlines()
                                  if not l.startswith("#")])
    shuffled, colour = rows(a), rows(c)
    sa_columns = [3, 4, 9, 10]
    greedy_and_sizes = [0, 1, 2, 5, 6, 7, 8, 11]
    assert np.array_equal(shuffled[:, greedy_and_sizes], colour[:, greedy_and_sizes])
    assert np.all(np.isfinite(shuffled[:, sa_columns]))
    assert np.all(shuffled[:, [3, 9]] >= 0.5 - 1e-12) and np.all(shuffled[:, [4, 10]] <= 1.0 + 1e-12)


def test_pipeline_from_yaml_and_hdf5_inputs(tmp_path, models):
    """The reference's own inputs (driver :755-757, common.py:791-803): --yaml names the operator,
    --hdf5 holds the ground state and the basis representatives (here in shuffled order, which
    the loader must undo).  Same output as the bundled-model route."""
    import yaml

    from annealing_sign_problem_amd import common, operators, sampled_components

    name = "heisenberg_kagome_16"
    op = operators.Operator.from_config(models[name])
    op.basis.build()
    energy, psi = op.ground_state()
    shuffle = np.random.default_rng(0).permutation(op.basis.number_states)
    h5 = tmp_path / (name + ".h5")
    common.save_ground_state(str(h5), psi[shuffle], energy, op.basis.states[shuffle])
    yml = tmp_path / (name + ".yaml")
    yml.write_text(yaml.safe_dump(models[name]))
    args = ["--order", "1", "--number-samples", "5", "--seed", "21", "--max-cluster-size", "200",
            "--no-annealing"]
    a, b, c = tmp_path / "model.csv", tmp_path / "yaml.csv", tmp_path / "model_h5.csv"
    sampled_components.main(["--model", name, "--output", str(a)] + args)
    sampled_components.main(["--yaml", str(yml), "--output", str(b)] + args)  # <yaml>.h5 by default
    sampled_components.main(["--model", name, "--hdf5", str(h5), "--output", str(c)] + args)
    assert a.read_text() == b.read_text() == c.read_text()
    with pytest.raises(SystemExit):
        sampled_components.main(["--output", str(tmp_path / "none.csv")] + args)


def test_fresh_process_grows_clusters_in_a_child_and_writes_the_same_file(tmp_path):
    """A fresh process (no GPU call made yet) forks a child that grows the clusters while the
    parent solves them (sampled_components.clusters_from_child): the same file as with the growth
    kept in the parent (ASP_GROW_IN_PLACE=1), greedy with host threads and annealed in pipelined
    rounds, also with the blocking wait policy (ASP_HIP_WAIT=block)."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    def run(name, extra_args, **env):
        out = tmp_path / name
        command = [sys.executable, "-m", "annealing_sign_problem_amd.sampled_components", "--model",
                   "heisenberg_kagome_16", "--output", str(out), "--order", "1", "--number-samples", "24",
                   "--seed", "91", "--noise", "0.2", "--max-cluster-size", "300"] + extra_args
        done = subprocess.run(command, cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True,
                              timeout=600)
        assert done.returncode == 0, done.stderr[-2000:]
        return out.read_text()

    greedy = ["--no-annealing", "--jobs", "4"]
    reference = run("in_place.csv", greedy, ASP_GROW_IN_PLACE="1")
    assert len([l for l in reference.splitlines() if not l.startswith("#")]) == 24
    assert run("child.csv", greedy) == reference
    assert run("child_block.csv", greedy, ASP_HIP_WAIT="block") == reference
    annealed = ["--annealing", "--jobs", "2", "--batch", "7"]  # four rounds, the next built while one anneals
    reference = run("annealed_in_place.csv", annealed + ["--batch", "1"], ASP_GROW_IN_PLACE="1")
    assert run("annealed_child.csv", annealed) == reference

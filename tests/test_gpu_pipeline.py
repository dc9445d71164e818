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

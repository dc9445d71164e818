"""The RCCL branch of the multi-GPU layer, executed for real: backend "nccl" (= RCCL on ROCm),
world size 1 on the box's one GPU, initialised the way `python -m torch.distributed.run` does
(RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment, distributed.init_from_env()).
Device tensors, all_gather, broadcast and all_gather_object all run through RCCL; results
must equal the single-process calls bit for bit.  (The 2-rank logic is covered on CPU by
tests/test_distributed_gloo.py; an 8-GPU curve is the driver's to measure.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from annealing_sign_problem_amd import _lib, annealer as sa, distributed, synthetic
assert distributed.init_from_env() is True
import torch, torch.distributed as dist
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
assert int(_lib.load().asp_get_device()) == torch.cuda.current_device() == 0
distributed.check_device_binding()
J, h, _ = synthetic.planted_cluster(900, seed=4, mean_degree=8.0)
ham = sa.Hamiltonian(J, h)
info = ham.info()
betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, 40)
xs0, es0 = sa.anneal_raw(ham, 77, betas, 37)
xs, es = distributed.anneal_sharded(ham, 77, betas, 37)          # device tensors -> all_gather
assert np.array_equal(xs, xs0) and es.tobytes() == es0.tobytes()
x, e = distributed.anneal_sharded_best(ham, 77, betas, 37)       # all_gather + broadcast
k = int(np.argmin(es0))
assert np.array_equal(x, xs0[k]) and e == es0[k]
seed = distributed.agree_on_seed(None)                            # broadcast of a drawn seed
assert 0 <= seed < 2**64 and distributed.agree_on_seed(5) == 5
# kernel results straight into device tensors (what the gather uses)
xt = torch.zeros((37, xs0.shape[1]), dtype=torch.int64, device="cuda")
et = torch.zeros(37, dtype=torch.float64, device="cuda")
sa.anneal_raw_into(ham, 77, betas, 37, 0, None, xt.data_ptr(), et.data_ptr())
assert np.array_equal(xt.cpu().numpy().view(np.uint64), xs0) and et.cpu().numpy().tobytes() == es0.tobytes()
out = distributed.map_sharded(list(range(5)), lambda k: k * k)   # world 1: plain loop
many = distributed.map_sharded_many(list(range(5)), lambda ks: [k + 1 for k in ks])
assert out == [0, 1, 4, 9, 16] and many == [1, 2, 3, 4, 5]
assert distributed.broadcast_object({"a": 1}) == {"a": 1}
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
"""


def test_rccl_world_size_one_runs_the_device_tensor_path():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env,
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "rccl ok" in proc.stdout, proc.stdout + proc.stderr


def test_pipeline_two_ranks_on_one_gpu_gloo(tmp_path):
    """sampled_components.main() under a 2-rank launch (gloo, both ranks on device 0 — the
    rehearsal a one-GPU box allows): rank discovery from the environment, clusters sharded
    c mod 2, rank 0 the only writer, output identical to the single-process run."""
    from annealing_sign_problem_amd import sampled_components

    args = ["--model", "heisenberg_kagome_16", "--order", "1", "--number-samples", "6",
            "--seed", "11", "--max-cluster-size", "200", "--no-annealing"]
    single = tmp_path / "single.csv"
    sampled_components.main(args + ["--output", str(single)])
    two = tmp_path / "two.csv"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m",
           "annealing_sign_problem_amd.sampled_components"] + args + ["--output", str(two)]
    env = dict(os.environ, ASP_DIST_BACKEND="gloo", ASP_SINGLE_DEVICE="1",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert two.read_text() == single.read_text()
    # annealed, rounds of 2 clusters per rank: every rank builds its share of the next round while
    # the current one anneals (round 3); the per-model loop of a single process is the reference
    annealed = [a for a in args if a != "--no-annealing"] + ["--annealing", "--number-sweeps", "5000"]
    one = tmp_path / "annealed_single.csv"
    sampled_components.main(annealed + ["--batch", "1", "--output", str(one)])
    both = tmp_path / "annealed_two.csv"
    cmd = cmd[:cmd.index("annealing_sign_problem_amd.sampled_components") + 1] + annealed + [
        "--batch", "2", "--output", str(both)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert both.read_text() == one.read_text()


def test_bench_multi_rank_path_at_world_size_one():
    """bench.py's N > 1 code path (process group, device-tensor results, RCCL gather of each
    rank's best chain, max-over-ranks timing) executed with RCCL at world size 1."""
    import json

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ASP_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--sizes", "3000,10000", "--replicas", "256", "--sweeps", "32", "--no-cpu-baseline",
           "--no-build"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    line = json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 1e9 and line["scaling"] == "weak"


def test_bench_two_ranks_gloo_on_one_gpu():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` as the driver
    launches it, rehearsed on one GPU (both ranks on device 0, gloo): rank discovery, global
    replica ids per rank, barriers, max-over-ranks timing, one JSON line from rank 0."""
    import json

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ASP_BENCH_BACKEND="gloo", ASP_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--sizes", "3000,10000", "--replicas", "256",
           "--sweeps", "32"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 1e9
    assert "cpu_baseline" not in line and "build" not in line  # N = 1 legs only

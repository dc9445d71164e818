"""One workload of the roofline table, for a rocprofv3 pass (development aid; tools/gpu_profile_r3.sh
runs it once per case and counter set, tools/summarise_roofline.py condenses the results into
profiles/sweep_counters.json and profiles/traffic.json).

    rocprofv3 --kernel-trace --pmc ... -- python3 tools/profile_roofline.py --case colour_100000 --out DIR

Every annealing launch of the process belongs to the case (no warm-up launches), so the counters
of all dispatches of the case's kernel, divided by the flips written to DIR/<case>.json, are per
flip.  Cases: colour_<K>, shuffled_<K>, shuffled64_<K> (the reference's default call: 64 chains x
5120 sweeps), batch / batch_shuffled (128 small clusters, 64 chains x 512 sweeps each, colour /
shuffled order), team (64 chains on a 1e5-spin cluster), real_kagome_36 (three real clusters,
colour order).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from annealing_sign_problem_amd import _lib, synthetic  # noqa: E402
from annealing_sign_problem_amd import annealer as sa  # noqa: E402
from annealing_sign_problem_amd import build  # noqa: E402

SEED = 783494  # bench.py's CLUSTER_SEED


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--case", required=True)
    p.add_argument("--out", required=True)
    p.add_argument("--chains", type=int, default=1024)
    p.add_argument("--sweeps", type=int, default=128)
    p.add_argument("--runs", type=int, default=2)
    a = p.parse_args()
    lib = _lib.load()
    kind, _, arg = a.case.partition("_")
    # (what the case's kernel is built from: bench.py keeps a case while that is unchanged)
    source_set = "shuffled" if "shuffled" in a.case else "colour"
    record = {"case": a.case, "library_fingerprint": build.built_fingerprint(),
              "kernel_source_set": source_set,
              "source_set_fingerprint": (build.built_source_set_fingerprints() or {}).get(source_set),
              "chains": a.chains, "sweeps": a.sweeps, "runs": a.runs}
    flips, sweep_ms = 0, []
    if kind in ("colour", "shuffled", "shuffled64"):
        k = int(arg)
        J, h, _ = synthetic.planted_cluster(k, seed=SEED)
        ham = sa.Hamiltonian(J, h)
        info = ham.info()
        chains, sweeps = (64, 5120) if kind == "shuffled64" else (a.chains, a.sweeps)
        betas = sa.make_schedule(info.beta0_auto, info.beta1_auto, sweeps)
        for _ in range(a.runs):
            sa.anneal_raw(ham, 12345, betas, chains, shuffled=kind != "colour")
            sweep_ms.append(lib.asp_sa_last_sweep_ms(ham.plan()))
            flips += k * chains * sweeps
        record.update(K=k, dbar=J.nnz / k, chains=chains, sweeps=sweeps,
                      kernel="k_sa_sweep_shuffled" if kind != "colour" else "k_sa_sweep<",
                      side_kernels=["k_shuffled_orders", "k_order_"] if kind != "colour" else [])
    elif kind == "batch":
        rng = np.random.default_rng(SEED)
        sizes = [int(round(np.exp(rng.uniform(np.log(1e2), np.log(1e4))))) for _ in range(128)]
        hams = []
        for i, k in enumerate(sizes):
            J, h, _ = synthetic.planted_cluster(k, seed=SEED + i)
            hams.append(sa.Hamiltonian(J, h))
        shuffled = arg == "shuffled"  # case "batch" (colour order) or "batch_shuffled"
        for _ in range(a.runs):
            sa.anneal_batch(hams, seed=12345, number_sweeps=512, repetitions=64,
                            sweep_order="shuffled" if shuffled else "colour")
            sweep_ms.append(float(lib.asp_sa_batch_last_ms()))
            flips += sum(sizes) * 64 * 512
        record.update(kernel="k_sa_sweep_shuffled_batch" if shuffled else "k_sa_sweep_batch", problems=len(sizes),
                      chains=64, sweeps=512,
                      side_kernels=["k_shuffled_orders", "k_order_"] if shuffled else [])
    elif kind == "team":
        J, h, _ = synthetic.planted_cluster(100000, seed=1, mean_degree=8.0)
        ham = sa.Hamiltonian(J, h)
        for _ in range(a.runs):
            sa.anneal(ham, seed=12345, number_sweeps=1024, repetitions=64, sweep_order="colour")
            sweep_ms.append(lib.asp_sa_last_sweep_ms(ham.plan()))
            flips += 100000 * 64 * 1024
        record.update(kernel="k_sa_sweep_team", K=100000, chains=64, sweeps=1024)
    elif kind == "real":
        import bench

        out = bench.bench_real_kagome_36(a.chains, a.sweeps, calls=a.runs, warmup=0, pipeline=False)
        flips = int(out["flips_profiled"])
        sweep_ms = [c["sweep_kernel_ms"] for c in out["clusters"]]
        record.update(kernel="k_sa_sweep<", clusters=out["clusters"])
    else:
        raise SystemExit("unknown case " + a.case)
    record.update(flips=flips, sweep_ms=sweep_ms)
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, a.case + ".json"), "w") as f:
        json.dump(record, f, indent=1)
    print(json.dumps(record))


if __name__ == "__main__":
    main()

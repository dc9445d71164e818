"""Where a cluster's time goes in `make kagome_36`'s pipeline (real model, order 2, cutoff 1e-6, greedy),
single-threaded and with --jobs threads: wall time summed over the threads, per stage.  A stage whose sum
grows with the number of threads is where the threads wait for each other (interpreter lock, HIP runtime,
the device).  (Development aid; GPU.)

    python tools/time_pipeline_stages.py <clusters> <jobs> [<jobs> ...]
"""
import collections
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from annealing_sign_problem_amd import annealer as sa  # noqa: E402
from annealing_sign_problem_amd import common, greedy, operators, synthetic  # noqa: E402
from annealing_sign_problem_amd import sampled_components as sc  # noqa: E402

spent = collections.defaultdict(float)
calls = collections.defaultdict(int)
lock = threading.Lock()
depth = threading.local()


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        inner_before = getattr(depth, "inner", 0.0)
        depth.inner = 0.0
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            own = dt - depth.inner  # exclusive of timed callees
            depth.inner = inner_before + dt
            with lock:
                spent[name] += own
                calls[name] += 1
    return wrapper


def main():
    number = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    jobs_list = [int(j) for j in sys.argv[2:]] or [1, 8]
    models = synthetic.load_models()
    hamiltonian = operators.Operator.from_config(models["heisenberg_kagome_36"])
    hamiltonian.basis.build()
    t0 = time.time()
    _, ground_state = hamiltonian.ground_state()
    print("ground state: %.1f s" % (time.time() - t0), flush=True)
    np.random.seed(435834)
    log_fn = common.ground_state_to_log_coeff_fn(ground_state, hamiltonian.basis)
    clusters = sc.generate_clusters(hamiltonian, ground_state, number, 0.1, 50, 1000, 0.5)

    device = hamiltonian.device()
    device.ising = timed("C  fused coupling build (asp_operator_ising)", device.ising)
    device.extend = timed("C  extension (asp_operator_extend)", device.extend)
    common.sparsify_component = timed("C+py sparsify_component", common.sparsify_component)
    hamiltonian.basis.batched_index = timed("C  basis index (asp_table_index)", hamiltonian.basis.batched_index)
    sa.Hamiltonian.plan = timed("C  plan (asp_sa_plan_create)", sa.Hamiltonian.plan)
    greedy.greedy_solve = timed("C+py greedy_solve (excl. plan)", greedy.greedy_solve)
    common.make_ising_model = timed("py make_ising_model glue", common.make_ising_model)
    common.sparsify_using_global_cutoff = timed("py sparsify glue", common.sparsify_using_global_cutoff)
    common.compute_accuracy_and_overlap = timed("py accuracy/overlap", common.compute_accuracy_and_overlap)
    common._project_on_frozen = timed("py project on frozen", common._project_on_frozen)
    sc.amplitude_overlap = timed("py amplitude overlap (excl. index)", sc.amplitude_overlap)

    def work(cluster):
        return sc.process_cluster(cluster, hamiltonian, ground_state, ground_state, log_fn, 2, 1e-6, False)

    work = timed("py process_cluster rest", work)
    work(clusters[0])  # warm-up: tables, streams
    for jobs in jobs_list:
        spent.clear()
        calls.clear()
        t0 = time.perf_counter()
        if jobs > 1:
            with ThreadPoolExecutor(max_workers=jobs) as pool:
                list(pool.map(work, clusters))
        else:
            for c in clusters:
                work(c)
        wall = time.perf_counter() - t0
        print("\n--jobs %d: %d clusters x 3 orders in %.2f s = %.1f ms per cluster; thread time per cluster by stage:"
              % (jobs, number, wall, wall / number * 1e3))
        for name, t in sorted(spent.items(), key=lambda kv: -kv[1]):
            print("  %-48s %7.2f ms   (%d calls)" % (name, t / number * 1e3, calls[name]))
        print("  %-48s %7.2f ms" % ("sum", sum(spent.values()) / number * 1e3), flush=True)


if __name__ == "__main__":
    main()

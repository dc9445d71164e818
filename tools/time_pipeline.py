"""Wall time of the sampled-cluster pipeline WITH annealing (64 chains x 5120 sweeps per model,
common.py:236-239): per-model anneal calls (--batch 1, with 1 or 8 host threads) against the
batched anneal (--batch 64).  Identical output required.  (Development aid.)"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import sampled_components  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "heisenberg_kagome_16"
samples = sys.argv[2] if len(sys.argv) > 2 else "64"
with tempfile.TemporaryDirectory() as d:
    base = ["--model", model, "--order", "2", "--number-samples", samples, "--seed", "435834",
            "--global-cutoff", "1e-6"]
    sampled_components.main(base[:5] + ["2"] + base[6:] + ["--output", os.path.join(d, "warm.csv")])
    outs = {}
    for name, extra in (("per-model loop, 1 thread", ["--batch", "1"]),
                        ("per-model loop, 8 threads", ["--batch", "1", "--jobs", "8"]),
                        ("batched anneal, 64 clusters per call", ["--batch", "64"]),
                        ("greedy only (--no-annealing)", ["--no-annealing"])):
        out = os.path.join(d, "%d.csv" % len(outs))
        t0 = time.time()
        sampled_components.main(base + ["--output", out] + extra)
        dt = time.time() - t0
        outs[name] = open(out).read()
        print("%-40s %.2f s for %s clusters x 3 orders (ED + cluster growth included)" % (name, dt, samples),
              flush=True)
    names = list(outs)
    assert outs[names[0]] == outs[names[1]] == outs[names[2]], "outputs differ"
    sizes = [l.split(",")[0::6] for l in outs[names[0]].splitlines() if not l.startswith("#")]
    print("cluster sizes (order 0,1,2):", sizes[:6])

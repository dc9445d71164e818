"""Wall time of the sampled-cluster pipeline with 1 vs N concurrent clusters (development aid)."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import sampled_components  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "heisenberg_kagome_16"
samples = sys.argv[2] if len(sys.argv) > 2 else "24"
with tempfile.TemporaryDirectory() as d:
    base = ["--model", model, "--order", "2", "--number-samples", samples, "--seed", "435834",
            "--global-cutoff", "1e-6"]
    outs = {}
    for jobs in (1, 4, 8, 16):
        out = os.path.join(d, "j%d.csv" % jobs)
        t0 = time.time()
        sampled_components.main(base + ["--output", out, "--jobs", str(jobs)])
        dt = time.time() - t0
        outs[jobs] = open(out).read()
        print("jobs=%2d: %.2f s for %s clusters x 3 orders (ED + cluster growth included)" % (jobs, dt, samples), flush=True)
    assert all(v == outs[1] for v in outs.values()), "outputs differ between --jobs settings"
    sizes = [l.split(",")[0::6] for l in outs[1].splitlines() if not l.startswith("#")]
    print("cluster sizes (order 0,1,2):", sizes[:6])

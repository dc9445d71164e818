"""`make kagome_36`'s pipeline (real model, order 2, cutoff 1e-6, greedy or annealed) in ONE command with
worker processes: wall time against the threaded single process, identical files required.
(Development aid; GPU.)

    python tools/time_pipeline_workers.py <clusters> <workers> [<jobs>] [--annealing]
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

samples = sys.argv[1] if len(sys.argv) > 1 else "512"
workers = sys.argv[2] if len(sys.argv) > 2 else "8"
jobs = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else "1"
annealing = "--annealing" in sys.argv
reference = "--no-reference" not in sys.argv
with tempfile.TemporaryDirectory() as d:
    h5 = os.path.join(d, "kagome_36.h5")
    t0 = time.time()
    subprocess.run([sys.executable, "-m", "annealing_sign_problem_amd.sector_ed", "--model",
                    "heisenberg_kagome_36", "--output", h5], check=True, cwd=ROOT, stdout=subprocess.DEVNULL)
    print("ground state + file: %.1f s" % (time.time() - t0), flush=True)
    base = [sys.executable, "-m", "annealing_sign_problem_amd.sampled_components", "--model",
            "heisenberg_kagome_36", "--hdf5", h5, "--seed", "435834", "--order", "2",
            "--annealing" if annealing else "--no-annealing", "--global-cutoff", "1e-6",
            "--number-samples", samples]
    runs = [("workers", ["--workers", workers, "--jobs", jobs, "--batch", "16"])]
    if reference:
        runs.append(("threads", ["--jobs", "8"]))
    outputs = {}
    for name, extra in runs:
        out = os.path.join(d, name + ".csv")
        t0 = time.time()
        subprocess.run(base + extra + ["--output", out], check=True, cwd=ROOT)
        print("%s %s: %s clusters x 3 orders, %s: %.1f s (incl. start-up and reading the 504 MB "
              "ground-state file)" % (name, " ".join(extra), samples, "annealed" if annealing else "greedy",
                                      time.time() - t0), flush=True)
        outputs[name] = open(out).read()
    if reference:
        print("outputs identical:", outputs["workers"] == outputs["threads"], flush=True)

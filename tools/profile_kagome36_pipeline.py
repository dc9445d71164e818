"""Host profile of `make kagome_36` (greedy only, order 2, cutoff 1e-6) on the real model: ground
state on the GPU, then cProfile of the sampled-cluster pipeline with one host thread.
(Development aid; GPU.)"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from annealing_sign_problem_amd import sampled_components, sector_ed  # noqa: E402

samples = sys.argv[1] if len(sys.argv) > 1 else "128"
jobs = sys.argv[2] if len(sys.argv) > 2 else "1"
with tempfile.TemporaryDirectory() as d:
    h5 = os.path.join(d, "kagome_36.h5")
    t0 = time.time()
    sector_ed.main(["--model", "heisenberg_kagome_36", "--output", h5])
    print("ground state + file: %.1f s" % (time.time() - t0), flush=True)
    argv = ["--model", "heisenberg_kagome_36", "--hdf5", h5, "--seed", "435834", "--order", "2",
            "--no-annealing", "--global-cutoff", "1e-6", "--number-samples", samples, "--jobs", jobs,
            "--output", os.path.join(d, "out.csv")]
    profile = cProfile.Profile()
    t0 = time.time()
    profile.runcall(sampled_components.main, argv)
    print("pipeline: %.1f s for %s clusters x 3 orders, --jobs %s" % (time.time() - t0, samples, jobs), flush=True)
    pstats.Stats(profile).sort_stats("cumulative").print_stats(32)

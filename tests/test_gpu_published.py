"""The only pin of the annealer the reference holds: its PUBLISHED success probabilities on the
full 16-site Hilbert spaces (experiments/*.csv, `make small`, Makefile:27-35: 1024 repetitions
x 10 trials per number of sweeps).  The annealer library itself (ising_glass_annealer) is
absent, so bit parity is out of reach; this test turns the statistical comparison
into an assertion that goes red when the schedule, the random-number use or the acceptance rule
drift.

Two Markov chains are compared with the published numbers:

  * the SHUFFLED visiting order (ASP-SA-1S: a fresh random permutation every sweep, what the
    library does as far as its statistics can tell) must REPRODUCE them: thirteen points here
    with 4 x 1024 chains, within 0.035 (three standard errors of the difference are 0.03), and all
    46 points of the five symmetry-free models in the `slow` test (ASP_RUN_SLOW=1; its output of
    this round is profiles/r03_published_shuffled_all.txt);
  * the default COLOUR order (ASP-SA-1) is a different chain that reaches the exact sign
    structure as often or MORE often: the assertion is one-sided — not more than three standard
    errors BELOW the published value (published spread over its 10 trials and this test's 8 x 1024
    chains combined) — and two-sided only against this repository's own recorded curve, +-0.03 =
    four standard errors of that difference, the regression pin proper: a change to the
    schedule, the random-number use or the acceptance rule moves these numbers.
    The inversion-symmetric kagome_18 basis is the exception: its ground level is three-fold
    degenerate, its published curve belongs to ANOTHER eigenvector than the one used here
    (Operator.ground_state fixes it by construction) and the success probability moves by up to
    0.09 with that choice (profiles/r02_kagome18_degeneracy_probe.txt): +-0.12 around the
    published value for that model, and the own-curve pin.

As in every row of the published CSVs, P(residual <= 1e-12) == P(accuracy > 0.995).
Numbers: tests/golden/published_sa_curves.json (+ generate_published_curves.py)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

POINTS = [("heisenberg_kagome_16", 100), ("heisenberg_kagome_16", 1600),
          ("heisenberg_kagome_16", 25600), ("sk_16_1", 100), ("sk_16_1", 1600), ("sk_16_1", 25600),
          # the symmetry-adapted basis of heisenberg_kagome_18.yaml:4 (24 310 representatives)
          ("heisenberg_kagome_18", 100), ("heisenberg_kagome_18", 400)]
TRIALS = 8
_sims = {}


def _simulation(name):
    from annealing_sign_problem_amd import full_hilbert_space

    if name not in _sims:
        _sims[name] = full_hilbert_space.Simulation(name)
    return _sims[name]


@pytest.mark.parametrize("name,sweeps", POINTS)
def test_success_probability_matches_published_curve(name, sweeps):
    with open(os.path.join(GOLDEN, "published_sa_curves.json")) as f:
        row = json.load(f)["models"][name][str(sweeps)]
    sim = _simulation(name)
    # (the COLOUR order, explicitly: the drop-in default is the shuffled order, tested below)
    results = np.array([sim.run(sweeps, 1024, seed=435834 + 1000003 * trial + sweeps, sweep_order="colour")
                        for trial in range(TRIALS)])
    acc, residual = results[:, 0].mean(), results[:, 2].mean()
    assert residual == acc, "P(residual <= 1e-12) and P(accuracy > 0.995) differ"
    published, own = row["acc_prob_mean"], row.get("mi355x_acc_prob_mean")
    if name == "heisenberg_kagome_18":  # degenerate ground level, see above
        assert abs(acc - published) <= 0.12, \
            "%s @ %d sweeps: %.4f vs published %.4f (%s)" % (name, sweeps, acc, published,
                                                            row["reference_line"])
    else:
        # the colour order anneals at least as well as the library: never significantly below
        standard_error = np.sqrt(row["acc_prob_std"] ** 2 / 10 + results[:, 0].var(ddof=1) / TRIALS)
        assert acc >= published - 3.0 * max(standard_error, 0.004), \
            "%s @ %d sweeps: %.4f is below the published %.4f (%s) by more than 3 s.e. = %.4f" % (
                name, sweeps, acc, published, row["reference_line"], 3.0 * standard_error)
    if own is not None:
        assert abs(acc - own) <= 0.03, \
            "%s @ %d sweeps: %.4f vs this repository's recorded %.4f" % (name, sweeps, acc, own)


# With a fresh visiting order every sweep (sweep_order="shuffled") the annealer
# reproduces the published probabilities of the symmetry-free models — the evidence that this is
# what the reference's library does, turned into a test: 4 x 1024 chains per point, within 0.035
# (three standard errors of the difference are 0.03) of the published mean.
SHUFFLED_POINTS = [("sk_16_3", 200), ("j1j2_square_4x4", 100), ("heisenberg_kagome_16", 3200),
                   ("sk_16_2", 400), ("sk_16_1", 1600),
                   # (more of the 46 points of DESIGN.md §6.1, the cheap ones)
                   ("heisenberg_kagome_16", 100), ("heisenberg_kagome_16", 800),
                   ("j1j2_square_4x4", 400), ("j1j2_square_4x4", 1600),
                   ("sk_16_1", 100), ("sk_16_1", 400), ("sk_16_2", 100), ("sk_16_3", 800)]


@pytest.mark.parametrize("name,sweeps", SHUFFLED_POINTS)
def test_shuffled_order_reproduces_the_published_probabilities(name, sweeps):
    with open(os.path.join(GOLDEN, "published_sa_curves.json")) as f:
        row = json.load(f)["models"][name][str(sweeps)]
    sim = _simulation(name)
    results = np.array([sim.run(sweeps, 1024, seed=435834 + 1000003 * trial + sweeps,
                                sweep_order="shuffled") for trial in range(4)])
    acc = results[:, 0].mean()
    assert results[:, 2].mean() == acc
    assert abs(acc - row["acc_prob_mean"]) <= 0.035, \
        "%s @ %d sweeps, shuffled order: %.4f vs published %.4f" % (name, sweeps, acc, row["acc_prob_mean"])


def _all_shuffled_points():
    with open(os.path.join(GOLDEN, "published_sa_curves.json")) as f:
        models = json.load(f)["models"]
    points = []
    for name in ("heisenberg_kagome_16", "j1j2_square_4x4", "sk_16_1", "sk_16_2", "sk_16_3"):
        # the 46 points of DESIGN.md §6.1: up to 102 400 sweeps, and per model only as far as the
        # published curve still moves (beyond, both sit on the same plateau or at 1)
        limit = {"heisenberg_kagome_16": 102400, "j1j2_square_4x4": 51200, "sk_16_1": 6400,
                 "sk_16_2": 25600, "sk_16_3": 25600}[name]
        points += [(name, int(s)) for s in sorted(models[name], key=int) if int(s) <= limit]
    return points


@pytest.mark.slow
@pytest.mark.parametrize("name,sweeps", _all_shuffled_points())
def test_shuffled_order_reproduces_every_published_point(name, sweeps):
    """All 46 points, 4 x 1024 chains each (about 3.7e14 proposals in total): within 0.035 of
    the published mean, as the thirteen of the default run."""
    with open(os.path.join(GOLDEN, "published_sa_curves.json")) as f:
        row = json.load(f)["models"][name][str(sweeps)]
    sim = _simulation(name)
    results = np.array([sim.run(sweeps, 1024, seed=435834 + 1000003 * trial + sweeps,
                                sweep_order="shuffled") for trial in range(4)])
    acc = results[:, 0].mean()
    print("shuffled %-22s %7d sweeps: %.4f +- %.4f  published %.4f +- %.4f  difference %+.4f" % (
        name, sweeps, acc, results[:, 0].std(ddof=1), row["acc_prob_mean"], row["acc_prob_std"],
        acc - row["acc_prob_mean"]))
    assert results[:, 2].mean() == acc
    assert abs(acc - row["acc_prob_mean"]) <= 0.035

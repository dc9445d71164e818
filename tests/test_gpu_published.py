"""The only pin of the annealer the reference holds: its PUBLISHED success probabilities on the
full 16-site Hilbert spaces (experiments/*.csv, `make small`, Makefile:27-35: 1024 repetitions
x 10 trials per number of sweeps).  The annealer library itself (ising_glass_annealer) is
absent, so bit parity is out of reach; this test turns the statistical comparison
into an assertion that goes red when the schedule, the random-number use or the acceptance rule
drift.

For each point (model, number of sweeps) it repeats the experiment with 8 x 1024 chains, fixed
seeds, and requires P(accuracy > 0.995)
  * within +-0.08 of the published value: specification ASP-SA-1 is a different Markov chain
    from the library's and statistically distinguishable from it — on the symmetry-free models
    it reaches the exact sign structure as often or more often (up to +0.078, 15 standard errors),
    DESIGN.md §6.1; the ground level of the kagome_18 basis is three-fold degenerate, so its
    published curve belongs to ANOTHER eigenvector than the one used here (Operator.ground_state
    fixes it by construction) and the success probability moves by up to 0.09 with that choice
    (profiles/r02_kagome18_degeneracy_probe.txt): +-0.12 for that model
    — so this band only catches gross changes;
  * within +-0.03 of this repository's own recorded measurement (10 x 1024 chains; 4 standard
    errors of the difference are 0.03), the regression pin proper;
and, as in every row of the published CSVs, P(residual <= 1e-12) == P(accuracy > 0.995).
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
    results = np.array([sim.run(sweeps, 1024, seed=435834 + 1000003 * trial + sweeps)
                        for trial in range(TRIALS)])
    acc, residual = results[:, 0].mean(), results[:, 2].mean()
    assert residual == acc, "P(residual <= 1e-12) and P(accuracy > 0.995) differ"
    published, own = row["acc_prob_mean"], row.get("mi355x_acc_prob_mean")
    band = 0.12 if name == "heisenberg_kagome_18" else 0.08   # degenerate ground level, see above
    assert abs(acc - published) <= band, \
        "%s @ %d sweeps: %.4f vs published %.4f (%s)" % (name, sweeps, acc, published,
                                                        row["reference_line"])
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

"""Success probability of simulated annealing on a full Hilbert space — the experiment behind
``make small`` (Makefile:27-35) and the published ``experiments/*.csv``.

The reference's driver ``experiments/full_hilbert_space.py`` does not parse at the surveyed
commit (empty ``def`` at :295-299) and its ``run()`` calls undefined functions; this module
implements what it is meant to do (commented block :212-218, ``_analyze`` :164-186, CSV writer
:319-369): build the Ising model of the WHOLE basis from the exact ground state, anneal
``repetitions`` chains for each number of sweeps, and record per trial the fraction of chains with
sign accuracy > 0.995, overlap > 0.995 and relative energy residual <= 1e-12.

Ground states are not downloadable here (Makefile:143-153), so they are recomputed by exact
diagonalisation of the symmetry-free 16-site models (``operators.py``).

    python -m annealing_sign_problem_amd.full_hilbert_space --model heisenberg_kagome_16 \\
        --output kagome_16.csv --number-sweeps 100,200,400 --repetitions 1024 --trials 10
"""
from __future__ import annotations

import argparse
import os
import time
from typing import List, Sequence, Tuple

import numpy as np

from . import annealer as sa
from . import common, operators, synthetic

CSV_COLUMNS = [
    "number_sweeps",
    "acc_prob_mean", "acc_prob_std", "acc_prob_median", "acc_prob_min", "acc_prob_max",
    "overlap_prob_mean", "overlap_prob_std", "overlap_prob_median", "overlap_prob_min",
    "overlap_prob_max",
    "residual_prob_mean", "residual_prob_std", "residual_prob_median", "residual_prob_min",
    "residual_prob_max",
]


class Simulation:
    """experiments/full_hilbert_space.py:23-44 (intended form)."""

    def __init__(self, model_name: str = None, yaml_filename: str = None, hdf5_filename: str = None):
        if yaml_filename is not None:
            # the reference's inputs (experiments/full_hilbert_space.py:24-29): operator from the
            # YAML file, ground state and basis representatives from the SpinED output
            self.hamiltonian = common.load_hamiltonian(yaml_filename)
            hdf5_filename = hdf5_filename or yaml_filename.replace(".yaml", ".h5")
        else:
            models = synthetic.load_models()
            if model_name not in models:
                raise ValueError("unknown model '{}'; available: {}".format(model_name, sorted(models)))
            self.hamiltonian = operators.Operator.from_config(models[model_name])
        if hdf5_filename is not None:
            ground_state, self.energy, representatives = common.load_ground_state(hdf5_filename)
            order = np.argsort(representatives, kind="stable")
            self.hamiltonian.basis.build(representatives[order])
            self.ground_state = np.ascontiguousarray(ground_state[order])
        else:
            self.hamiltonian.basis.build()
            self.energy, self.ground_state = self.hamiltonian.ground_state()
        log_psi_fn = common.ground_state_to_log_coeff_fn(self.ground_state, self.hamiltonian.basis)
        self.exact_model = common.make_ising_model(self.hamiltonian.basis.states, self.hamiltonian,
                                                   log_psi_fn=log_psi_fn)
        self.weights = self.ground_state ** 2
        self.weights /= np.sum(self.weights)
        h = self.exact_model.ising_hamiltonian
        # energy identity of the construction (experiments/full_hilbert_space.py:142-145)
        e_signs = h.energy(self.exact_model.initial_signs)
        if abs(e_signs - self.energy) > 1e-9 * abs(self.energy):
            raise RuntimeError("E(sign psi) = {} differs from the ED energy {}".format(e_signs, self.energy))
        # the eigensolver's eigenvalue carries its own tolerance; the classical minimum the
        # chains can reach exactly is E(sign psi)
        self.energy = e_signs

    def analyze(self, xs: np.ndarray, es: np.ndarray, accuracy_threshold: float = 0.995,
                overlap_threshold: float = 0.995, residual_threshold: float = 1e-12):
        """Fractions of chains above the thresholds (experiments/full_hilbert_space.py:164-186)."""
        results = np.zeros((len(xs), 3), dtype=np.float64)
        for i, (x, e) in enumerate(zip(xs, es)):
            accuracy, overlap = common.compute_accuracy_and_overlap(
                predicted=x, exact=self.exact_model.initial_signs, weights=self.weights)
            results[i] = [accuracy, overlap, abs((e - self.energy) / self.energy)]
        return (float(np.mean(results[:, 0] > accuracy_threshold)),
                float(np.mean(results[:, 1] > overlap_threshold)),
                float(np.mean(results[:, 2] <= residual_threshold)))

    def run(self, number_sweeps: int, repetitions: int, seed=None, sweep_order: str = None):
        """One trial (commented block at experiments/full_hilbert_space.py:212-218)."""
        xs, es = sa.anneal(self.exact_model.ising_hamiltonian, seed=seed,
                           number_sweeps=number_sweeps, repetitions=repetitions, only_best=False,
                           sweep_order=sweep_order)
        return self.analyze(xs, es)


def summarise(number_sweeps: int, results: np.ndarray) -> List:
    row = [number_sweeps]
    for column in range(3):
        v = results[:, column]
        row += [np.mean(v), np.std(v), np.median(v), np.min(v), np.max(v)]
    return row


def run_experiment(model: str, sweeps: Sequence[int], repetitions: int, trials: int, seed: int,
                   output: str, log=print, yaml_filename: str = None,
                   hdf5_filename: str = None, sweep_order: str = None) -> List[List]:
    if os.path.exists(output):
        raise ValueError("output file '{}' already exists".format(output))
    simulation = Simulation(model, yaml_filename, hdf5_filename)
    log("model {}: K = {}, E0 = {:.12f}".format(model, simulation.exact_model.size, simulation.energy))
    with open(output, "w") as f:
        f.write(",".join(CSV_COLUMNS) + "\n")
    rows = []
    for number_sweeps in sweeps:
        results = np.zeros((trials, 3), dtype=np.float64)
        tick = time.time()
        for trial in range(trials):
            results[trial] = simulation.run(number_sweeps, repetitions,
                                            seed=seed + 1000003 * trial + number_sweeps,
                                            sweep_order=sweep_order)
        row = summarise(number_sweeps, results)
        rows.append(row)
        with open(output, "a") as f:
            f.write(",".join(map(str, row)) + "\n")
        log("{} sweeps: P(acc>0.995) = {:.4f} +- {:.4f}  P(residual<=1e-12) = {:.4f}  [{:.1f} s]".format(
            number_sweeps, row[1], row[2], row[11], time.time() - tick))
    return rows


def main(argv=None):
    parser = argparse.ArgumentParser(description="Test Simulated Annealing on a small system.")
    parser.add_argument("--model", type=str, help="name in models.json (ground state by ED)")
    parser.add_argument("--yaml", type=str, help="physical_systems/<model>.yaml of the reference")
    parser.add_argument("--hdf5", type=str, help="SpinED output (default <yaml>.h5 with --yaml)")
    parser.add_argument("--output", type=str, required=True)
    parser.add_argument("--number-sweeps", type=str, required=True)
    parser.add_argument("--repetitions", type=int, default=1024)
    parser.add_argument("--trials", type=int, default=10)
    parser.add_argument("--seed", type=int, default=12345)
    parser.add_argument("--sweep-order", type=str, default="shuffled", choices=["colour", "shuffled"],
                        help="'shuffled' (default): a fresh visiting order every sweep, the reference "
                             "annealer's statistics — the published curves (DESIGN.md §6.1); "
                             "'colour': this package's fixed colour order, faster, a different chain")
    args = parser.parse_args(argv)
    sweeps = [int(s) for s in args.number_sweeps.split(",")]
    if (args.model is None) == (args.yaml is None):
        raise SystemExit("give exactly one of --model and --yaml")
    run_experiment(args.model, sweeps, args.repetitions, args.trials, args.seed, args.output,
                   yaml_filename=args.yaml, hdf5_filename=args.hdf5, sweep_order=args.sweep_order)


if __name__ == "__main__":
    main()

"""MI355X-native sign-optimisation hot path of annealing-sign-problem.

``from annealing_sign_problem_amd import *`` mirrors ``from annealing_sign_problem
import *`` (annealing_sign_problem/__init__.py:1-4): it re-exports the entry
points of :mod:`.common`.  ``annealer`` stands in for ``ising_glass_annealer`` and
``_build_matrix`` for the reference's cffi extension of the same name.
"""
__version__ = "0.4.0"  # = asp_version() of libasp_hip.so

from .common import *  # noqa: F401,F403
from . import annealer  # noqa: F401

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of GPU time; runs only with ASP_RUN_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("ASP_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow: set ASP_RUN_SLOW=1 (minutes of GPU time)")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def models():
    from annealing_sign_problem_amd import synthetic

    return synthetic.load_models()

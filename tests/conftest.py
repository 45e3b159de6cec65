import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """No test may hold a (GPU) box indefinitely: pytest-timeout is in the image; a stuck test fails after 10 minutes."""
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def cnn_golden():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "cnn_golden.npz"))


@pytest.fixture(scope="session")
def state_dict():
    from suo_slam_amd import weights
    return weights.make_random_state_dict(seed=0, logit_gain=8.0)

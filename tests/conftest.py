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


@pytest.fixture(autouse=True)
def host_scoring_without_a_gpu(request, monkeypatch):
    """The SLAM rules' inlier counts run on the device in the product (suo_slam_amd/slam_score.py; it raises without a GPU).  The CPU suite
    checks the host logic AROUND them (hypotheses, float32 containers, the 3x rule, whole sequences with the oracle's PnP / LM injected): there
    the counts come from the host restatement in tests/host_scoring.py -- test-side injection, like the PnP / LM backends of
    tests/test_slam_golden.py.  With a GPU nothing is replaced (tests/test_gpu_slam_score.py re-runs these rule tests on the kernel)."""
    if request.node.get_closest_marker("gpu") is None:
        import torch
        if not torch.cuda.is_available():
            from suo_slam_amd import slam_score
            from tests import host_scoring
            monkeypatch.setattr(slam_score, "chi2_counts", host_scoring.chi2_counts)
    yield

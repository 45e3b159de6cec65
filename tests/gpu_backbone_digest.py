"""Child process of tests/test_gpu_cnn.py::test_schedule_options_do_not_change_the_result: runs the HIP backbone on seeded crops
under whatever SUO_* schedule options the parent put in the environment and prints a digest of the logits."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import weights  # noqa: E402
from suo_slam_amd.pkpnet import PkpNet  # noqa: E402
from tests.gpu_backbone import run_backbone_from_staged  # noqa: E402

L = int(sys.argv[1])
graph = bool(int(sys.argv[2]))
sd = weights.make_random_state_dict(seed=0, logit_gain=8.0)
net = PkpNet(state_dict=sd, max_crops=L)
net.set_graph(graph)
rng = np.random.default_rng(11)
xin = np.zeros((L, 256, 256, 48), np.float32)
xin[..., :44] = rng.uniform(0, 1, (L, 256, 256, 44)).astype(np.float32)
logits = run_backbone_from_staged(net, xin)
assert np.isfinite(logits).all()
print("DIGEST", hashlib.sha256(np.ascontiguousarray(logits).tobytes()).hexdigest(), float(np.abs(logits).max()))

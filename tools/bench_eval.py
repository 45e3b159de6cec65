"""Time the ADD-S pair kernel through the C ABI: python tools/bench_eval.py [P] [n_poses]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import eval_meter as EM  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
rng = np.random.default_rng(0)
pts = (rng.standard_normal((P, 3)) * 50).astype(np.float32)
meter = EM.EvalMeter({1: {"points": pts, "is_symmetric": True}})
for n in ([int(sys.argv[2])] if len(sys.argv) > 2 else [1, 8, 64]):
    gt = np.tile(np.eye(4)[None], (n, 1, 1))
    gt[:, 2, 3] = 800
    pr = gt.copy()
    pr[:, :3, 3] += rng.standard_normal((n, 3))
    for _ in range(3):
        meter.pose_errors([1] * n, pr, gt)
    t0 = time.perf_counter()
    it = 10
    for _ in range(it):
        meter.pose_errors([1] * n, pr, gt)
    dt = (time.perf_counter() - t0) / it
    pairs = n * P * P
    print(f"P={P} n={n}: {dt*1e3:8.3f} ms/call (host round trip included)  {pairs/dt/1e12:6.3f} Tpairs/s  "
          f"= {7*pairs/dt/78.6e12:5.3f} of the fp32 VALU rate (7 ops/pair, 78.6 Tlane-op/s)")

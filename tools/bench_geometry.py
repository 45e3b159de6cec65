"""Time the HIP PnP and LM calls per frame: python tools/bench_geometry.py [n_obj] [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from suo_slam_amd import ba, lambdatwist  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
pool = bench.make_pool(np.random.default_rng(0), 16, L)
tp = tb = tl = 0.0
stats = []
for i in range(n + 5):
    fr = pool[i % 16]
    t0 = time.perf_counter()
    T, status, info = lambdatwist.pnp_batch(fr["pnp_xs"], fr["pnp_ys"], 1e-3, seed=i, return_info=True)
    t1 = time.perf_counter()
    B = fr["ba"]
    prob = ba.Problem(B["cam_T"], B["cam_fixed"], T[:, :3, :], B["obj_fixed"], B["edge_cam"], B["edge_obj"], B["edge_camk"], B["edge_p"],
                      B["edge_uv"], B["edge_info"], B["edge_inlier"], its=(10, 10, 40, 40))
    t2 = time.perf_counter()
    ba.optimize_batch([prob])
    t3 = time.perf_counter()
    if i >= 5:
        tp += t1 - t0; tb += t2 - t1; tl += t3 - t2
        stats.append(list(prob.stats) + [int(info["iterations"].max())])
print(f"L={L}: pnp {tp/n*1e3:.3f} ms  build {tb/n*1e3:.3f} ms  lm {tl/n*1e3:.3f} ms per frame")
s = np.array(stats)
print("mean rounds, LM its, LM trials, num_good, max ransac iters:", s.mean(0))

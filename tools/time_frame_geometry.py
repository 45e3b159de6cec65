"""Host-side time of one frame's geometry through the C ABI (8 objects): PnP launch, problem assembly, LM launch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from suo_slam_amd import ba, lambdatwist  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pool = bench.make_pool(np.random.default_rng(0), 16, L)
ts = {"pnp": [], "build": [], "lm": []}
for rep in range(60):
    frames = [pool[(rep * F + k) % 16] for k in range(F)]
    xs = [x for fr in frames for x in fr["pnp_xs"]]
    ys = [y for fr in frames for y in fr["pnp_ys"]]
    t0 = time.perf_counter()
    T, status = lambdatwist.pnp_batch(xs, ys, 1e-3, seed=rep)
    t1 = time.perf_counter()
    probs = []
    for j, fr in enumerate(frames):
        B = fr["ba"]
        probs.append(ba.Problem(B["cam_T"], B["cam_fixed"], T[j * L:(j + 1) * L, :3, :], B["obj_fixed"], B["edge_cam"], B["edge_obj"], B["edge_camk"],
                                B["edge_p"], B["edge_uv"], B["edge_info"], B["edge_inlier"], its=(10, 10, 40, 40)))
    t2 = time.perf_counter()
    ba.optimize_batch(probs)
    t3 = time.perf_counter()
    if rep >= 10:
        ts["pnp"].append(t1 - t0); ts["build"].append(t2 - t1); ts["lm"].append(t3 - t2)
print(f"L={L} F={F}: " + "  ".join(f"{k} {1e3 * np.median(v):.3f} ms" for k, v in ts.items()), " stats of last LM:", list(probs[0].stats))

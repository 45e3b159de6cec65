"""Global adjustments of several sizes and seeds through suo_optimize (one C call each): ms, LM trials, us per trial.  The LM trajectory is sensitive to the last bits of the
solve (a trial accepted or rejected at the boundary), so two builds are compared over a SET of problems:  SUO_HIP_LIB=<variant> python tools/ab_global_ba.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import ba  # noqa: E402
from suo_slam_amd import synthetic as S  # noqa: E402

keys = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")
tot_ms = tot_trials = 0
for n_cam, n_obj in ((32, 16), (60, 8), (60, 16), (120, 8), (24, 12)):
    for seed in (5, 6, 7, 8):
        P = S.make_pose_graph(np.random.default_rng(seed), n_cam, n_obj)
        ts = []
        for _ in range(3):
            one = ba.Problem(*[P[k].copy() for k in keys])
            t0 = time.perf_counter()
            ba.optimize_batch([one])
            ts.append(time.perf_counter() - t0)
        err = float(max(np.linalg.norm(one.obj_T.reshape(-1, 3, 4)[o][:, 3] - P["obj_gt"][o][:, 3]) for o in range(n_obj)))
        ms, tr = 1e3 * min(ts), int(one.stats[2])
        tot_ms += ms; tot_trials += tr
        print(f"{n_cam:4d} x {n_obj:2d} seed {seed}: {ms:7.2f} ms  {tr:4d} trials  {1e3 * ms / max(tr, 1):6.1f} us/trial  good {int(one.stats[3])}  max object error {err:.3f} mm")
print(f"sum: {tot_ms:.1f} ms, {tot_trials} trials, {1e3 * tot_ms / tot_trials:.1f} us/trial")

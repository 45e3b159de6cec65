"""Where a SLAM view's wall clock goes relative to its two network passes (bench.py's `slam` leg configuration): per view, the host time before pass A is enqueued,
the wait for pass A (network + chain), the host time BETWEEN the passes (detections, camera-hypothesis voting, prior projection, staging), the wait for pass B, and
the host time after it (re-initialisation checks, current-view LM, bookkeeping).   python tools/time_slam_gaps.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suo_slam_amd import object_slam as OS  # noqa: E402
from suo_slam_amd import synthetic as S, weights  # noqa: E402
from suo_slam_amd.frame_geom import FrameGeometry  # noqa: E402

marks = []
orig_fetch, orig_launch = FrameGeometry.fetch, FrameGeometry.launch


def fetch(self, *a, **k):
    t0 = time.perf_counter()
    r = orig_fetch(self, *a, **k)
    marks.append(("fetch", t0, time.perf_counter()))
    return r


def launch(self, *a, **k):
    r = orig_launch(self, *a, **k)
    marks.append(("launched", time.perf_counter(), 0.0))
    return r


FrameGeometry.fetch, FrameGeometry.launch = fetch, launch
seq = S.make_slam_sequence(np.random.default_rng(3), 60, 8)
sd = weights.make_random_state_dict(0, 8.0)
rows = []
for rep in range(3):
    slam = OS.ObjectSLAM(None, seq["mesh_db"], debug_gt_kp=True, manual_kp_std=0.01, state_dict=sd, max_crops=16, run_network_in_debug=True, debug_gt_on_device=True)
    rows = []
    for vw in seq["views"]:
        marks.clear()
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        slam.process_view(vw["view_id"], vw["image"], vw["K"], vw["obj_ids"].copy(), vw["bboxes"].copy(), vw["model_kps"], vw["model_kps_masks"], vw["kp_masks"], uv_gt=vw["uv_gt"])
        t1 = time.perf_counter()
        f = [m for m in marks if m[0] == "fetch"]
        ln = [m for m in marks if m[0] == "launched"]
        if len(f) == 2 and len(ln) == 2 and slam.all_time_num_views > 5 and not (len(slam.view_ids) > 1 and len(slam.view_ids) % slam.global_opt_every == 0):
            chained = ln[1][1] < f[0][1]                      # pass B enqueued before pass A's results were waited for (the two-pass device chain)
            if chained:   # host until A enqueued | host until B enqueued (under A's GPU time) | wait for A's block | host under B (A's bookkeeping) | wait for B | host after B
                rows.append((ln[0][1] - t0, ln[1][1] - ln[0][1], f[0][2] - ln[1][1], f[1][1] - f[0][2], f[1][2] - f[1][1], t1 - f[1][2], t1 - t0))
            else:         # host until A enqueued | 0 | wait for A | host between the passes | wait for B | host after B
                rows.append((ln[0][1] - t0, 0.0, f[0][2] - ln[0][1], ln[1][1] - f[0][2], f[1][2] - ln[1][1], t1 - f[1][2], t1 - t0))
r = 1e3 * np.median(np.array(rows), axis=0)
print("views with two passes and no global adjustment: %d   (median per view, ms)" % len(rows))
print("host until pass A is enqueued %.3f | host until pass B is enqueued (two-pass chain: under pass A's GPU time) %.3f | wait for pass A's results %.3f | "
      "host between / under the passes %.3f | wait for pass B %.3f | host after pass B %.3f | view %.3f" % tuple(r))

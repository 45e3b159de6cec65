"""Wall-clock per view of the stages of ObjectSLAM.process_view in SLAM mode (bench.py's `slam` leg: network on the frame's pixels, geometry on
ground-truth keypoints), perf_counter around each method -- no profiler overhead.  python tools/time_slam_stages.py [net=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from suo_slam_amd import ba as BA  # noqa: E402
from suo_slam_amd import lambdatwist as LT  # noqa: E402
from suo_slam_amd import object_slam as OS  # noqa: E402
from suo_slam_amd import synthetic as S, weights  # noqa: E402
from suo_slam_amd.pkpnet import PkpNet  # noqa: E402

net = (sys.argv[1] != "0") if len(sys.argv) > 1 else True
acc = {}


def wrap(owner, name, label=None):
    f = getattr(owner, name)
    label = label or name

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(owner, name, g)


for n in ("_process_objects", "_run_kp_model", "_run_kp_model_chain", "_estimate_camera_pose", "_maybe_reinit_objects", "optimize", "build_problem", "apply_problem", "_cull_after_optimize"):
    wrap(OS.ObjectSLAM, n)
wrap(OS._sc, "chi2_counts", "chi2_counts (device)")
wrap(OS._lt, "pnp_batch", "pnp_batch")
wrap(OS._ba, "optimize_batch", "optimize_batch")
wrap(PkpNet, "forward", "net.forward")
wrap(PkpNet, "stage_block", "stage_block")
wrap(PkpNet, "_to_device", "frame upload (host side)")
from suo_slam_amd.frame_geom import FrameGeometry  # noqa: E402
wrap(FrameGeometry, "launch", "chain launch (host side)")
wrap(FrameGeometry, "fetch", "chain fetch (waits for network + chain)")
PkpNet.__call__ = PkpNet.forward

seq = S.make_slam_sequence(np.random.default_rng(3), 60, 8)
sd = weights.make_random_state_dict(0, 8.0)


def run():
    slam = OS.ObjectSLAM(None, seq["mesh_db"], debug_gt_kp=True, manual_kp_std=0.01, state_dict=sd if net else None, max_crops=16, run_network_in_debug=net, debug_gt_on_device=(net and os.environ.get("SUO_SLAM_HOST_DEBUG", "0") == "0"))
    for vw in seq["views"]:
        slam.process_view(vw["view_id"], vw["image"], vw["K"], vw["obj_ids"].copy(), vw["bboxes"].copy(), vw["model_kps"], vw["model_kps_masks"], vw["kp_masks"], uv_gt=vw["uv_gt"])
    return slam


run()
best = None
for rep in range(3):
    acc.clear()
    slam = run()
    t = 1e3 * slam.track_time_meter.average()
    if best is None or t < best[0]:
        best = (t, dict(acc))
nv = len(seq["views"])
print(f"network={net}: tracking {best[0]:.2f} ms/view (best of 3); per view, nested stages listed under their callers:")
for k, v in sorted(best[1].items(), key=lambda kv: -kv[1]):
    print(f"  {k:26s} {1e3 * v / nv:7.3f} ms")

"""cProfile of bench.py's `slam` leg (60 views x 8 objects, network on the frame's pixels, geometry on ground-truth keypoints): host functions by own time and by
cumulative time per view.   python tools/profile_slam_tracking.py"""
import cProfile
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suo_slam_amd import synthetic as S, weights  # noqa: E402
from suo_slam_amd.object_slam import ObjectSLAM  # noqa: E402

seq = S.make_slam_sequence(np.random.default_rng(3), 60, 8)
sd = weights.make_random_state_dict(0, 8.0)


def run():
    slam = ObjectSLAM(None, seq["mesh_db"], debug_gt_kp=True, manual_kp_std=0.01, state_dict=sd, max_crops=16, run_network_in_debug=True)
    for vw in seq["views"]:
        slam.process_view(vw["view_id"], vw["image"], vw["K"], vw["obj_ids"].copy(), vw["bboxes"].copy(), vw["model_kps"], vw["model_kps_masks"], vw["kp_masks"], uv_gt=vw["uv_gt"])
    return slam


run()
slam = run()
print("tracking %.3f ms/view, global opt %.3f ms" % (1e3 * slam.track_time_meter.average(), 1e3 * slam.opt_time_meter.average()))
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
st.sort_stats("cumulative").print_stats(40)

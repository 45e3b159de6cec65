"""cProfile of the SLAM leg of bench.py (network on the frame's pixels + geometry on ground-truth keypoints): where does a view's 8-9 ms go?
python tools/profile_slam_view.py"""
import cProfile, os, pstats, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
torch.set_num_threads(8)
from suo_slam_amd import synthetic as S, weights
from suo_slam_amd.object_slam import ObjectSLAM
seq = S.make_slam_sequence(np.random.default_rng(3), 60, 8)
sd = weights.make_random_state_dict(0, 8.0)
def run(net=True):
    slam = ObjectSLAM(None, seq["mesh_db"], debug_gt_kp=True, manual_kp_std=0.01, state_dict=sd if net else None, max_crops=16, run_network_in_debug=net)
    for vw in seq["views"]:
        slam.process_view(vw["view_id"], vw["image"], vw["K"], vw["obj_ids"].copy(), vw["bboxes"].copy(), vw["model_kps"], vw["model_kps_masks"], vw["kp_masks"], uv_gt=vw["uv_gt"])
    slam.collect_results(final=True)
    return slam
run()
for net in (True, False):
    pr = cProfile.Profile(); pr.enable(); slam = run(net); pr.disable()
    print(f"==== network={net}: tracking {1e3*slam.track_time_meter.average():.2f} ms/view, global opt {1e3*slam.opt_time_meter.average():.2f} ms")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

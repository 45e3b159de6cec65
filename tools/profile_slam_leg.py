import cProfile, pstats, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from suo_slam_amd import synthetic as S, weights
from suo_slam_amd.object_slam import ObjectSLAM
seq = S.make_slam_sequence(np.random.default_rng(3), 60, 8)
sd = weights.make_random_state_dict(0, 8.0)
def run():
    slam = ObjectSLAM(None, seq["mesh_db"], debug_gt_kp=True, manual_kp_std=0.01, state_dict=sd, max_crops=16, run_network_in_debug=True)
    for vw in seq["views"]:
        slam.process_view(vw["view_id"], vw["image"], vw["K"], vw["obj_ids"].copy(), vw["bboxes"].copy(), vw["model_kps"], vw["model_kps_masks"], vw["kp_masks"], uv_gt=vw["uv_gt"])
    slam.collect_results(no_viz=True, final=True)
    return slam
run()
pr = cProfile.Profile(); pr.enable(); slam = run(); pr.disable()
print("tracking ms", 1e3 * slam.track_time_meter.average(), "opt ms", 1e3 * slam.opt_time_meter.average())
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(35)
st.print_callers("ascontiguousarray")
st.sort_stats("tottime").print_stats(12)

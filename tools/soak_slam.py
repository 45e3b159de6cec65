import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from suo_slam_amd import synthetic as S, weights
from suo_slam_amd.object_slam import ObjectSLAM
seq = S.make_slam_sequence(np.random.default_rng(3), 400, 8)
sd = weights.make_random_state_dict(0, 8.0)
slam = ObjectSLAM(None, seq["mesh_db"], debug_gt_kp=True, manual_kp_std=0.01, state_dict=sd, max_crops=16, run_network_in_debug=True)
ts = []
for vw in seq["views"]:
    t0 = time.perf_counter()
    slam.process_view(vw["view_id"], vw["image"], vw["K"], vw["obj_ids"].copy(), vw["bboxes"].copy(), vw["model_kps"], vw["model_kps_masks"], vw["kp_masks"], uv_gt=vw["uv_gt"])
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("views", len(ts), "wall ms/view: first 100 %.2f, last 100 %.2f (incl. global optimisations every 10 views)" % (ts[:100].mean(), ts[-100:].mean()))
print("tracking meter %.2f ms, store slots %d, device mem %.1f MB" % (1e3 * slam.track_time_meter.average(), slam._score_store.n_slots, torch.cuda.memory_allocated() / 1e6))
res = slam.collect_results(final=True)
print("camera poses", len(res))

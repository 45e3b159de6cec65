"""Where the host's time goes in the batched single-view evaluation (ObjectSLAM.submit_views_single / collect_views_single, two batches in flight):
cProfile of the timed loop of bench.drop_in_leg's second half.   python tools/time_views_single_host.py [views_per_call]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402
from suo_slam_amd.object_slam import ObjectSLAM         # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = 8
pool = bench.make_pool(np.random.default_rng(0), 32, L)
frames = [pool[i % len(pool)] for i in range(B)]
mesh_all = {100 * i + o: {"diameter": float(fr["diameter"][k]), "is_symmetric": False} for i, fr in enumerate(frames) for k, o in enumerate(fr["obj_ids"])}
slam = ObjectSLAM(None, mesh_all, sfm_mode=True, single_view_mode=True, state_dict=bench.confident_state_dict(), max_crops=B * 16, kp_var_thresh=bench.KP_VAR_THRESH,
                  bbox_thresh=bench.BBOX_THRESH)


def views(it):
    return [(it * B + i, fr["image"], fr["K"], 100 * i + np.array(fr["obj_ids"]), fr["boxes"].astype(np.float64), fr["model_kps"], fr["model_kps_masks"],
             fr["model_kps_masks"]) for i, fr in enumerate(frames)]


def loop(n):
    for it in range(n):
        slam.submit_views_single(views(it))
        if slam.views_in_flight() == 2:
            slam.collect_views_single()
    while slam.views_in_flight():
        slam.collect_views_single()


loop(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
loop(8)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{B} views per call, two batches in flight: {1e3 * dt / (8 * B):.3f} ms per frame ({8 * B / dt:.1f} frames/s), {1e3 * dt / 8:.2f} ms per batch")
pr = cProfile.Profile()
pr.enable()
loop(8)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)

"""SLAM-mode (BASELINE configs[2]) timing on a synthetic sequence: python tools/bench_slam.py [n_views] [n_objs]

Random network weights give meaningless keypoints, so the two halves of a SLAM view are timed separately on the same
sequence:
  * geometry + host logic: ObjectSLAM.process_view in the reference's --debug_gt_kp mode (projected model keypoints +
    noise drive PnP, camera-pose hypotheses, re-initialisation checks, the current-view LM and the global BA every 10
    views) -- everything of the view except the network;
  * network: the two passes a SLAM view makes (objects without prior: image-only stem; objects with prior: priors
    rendered on the device from the projected keypoints), L crops each, through PkpNet.forward.
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suo_slam_amd import bop, weights  # noqa: E402
from suo_slam_amd.object_slam import ObjectSLAM  # noqa: E402
from suo_slam_amd.pkpnet import PkpNet  # noqa: E402
from tests import bop_tree  # noqa: E402

n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_objs = int(sys.argv[2]) if len(sys.argv) > 2 else 8

with tempfile.TemporaryDirectory() as root:
    desc = bop_tree.build_sequence(root, seed=3, n_views=n_views, n_objs=n_objs)
    ds = bop.BopDataset(desc["data_root"], desc["split"], bop_dset="ycbv", ignore_symmetry=True)
    mesh_db = bop.load_mesh_db(os.path.join(desc["data_root"], "models_bop-compat_eval"))
    scene = ds.scene_ids()[0]
    samples = [(v, ds.get_all_obj(scene, v), ds.obj_ids(scene, v)) for v in ds.view_ids(scene)]

    for rep in range(2):                                          # the first pass pays library load, first launches, allocator growth
        slam = ObjectSLAM(None, mesh_db, debug_gt_kp=True, manual_kp_std=0.01)
        t0 = time.perf_counter()
        for v, s, ids in samples:
            img = (255 * s["img"].numpy().transpose(1, 2, 0)).astype(np.uint8)
            slam.process_view(v, img, s["K"].numpy(), np.array(ids), s["bboxes"].numpy(), s["model_kps"].numpy(), s["kp_model_masks"].numpy(),
                              s["kp_masks"].numpy(), uv_gt=s["kp_uvs"].numpy())
    res = slam.collect_results(final=True)
    t_geo = (time.perf_counter() - t0) / len(samples)
    err = []
    for v, s, ids in samples:
        for o in ids:
            T = res[v]["poses"].get(o, {}).get("T_OtoC")
            if T is not None:
                gt = ds.get_obj_pose(scene, v, o)
                err.append(np.linalg.norm(T[:3, 3] - gt[:3, 3]) / gt[2, 3])
    print(f"  of which global adjustments (every {slam.global_opt_every} views, graphs of 10 .. {len(samples)} cameras): {1e3 * slam.opt_time_meter.average():.2f} ms each; "
          f"tracking (PnP, hypotheses, current-view LM) {1e3 * slam.track_time_meter.average():.2f} ms per view")
    print(f"geometry + host logic (debug_gt_kp): {t_geo * 1e3:.2f} ms/view over {len(samples)} views x {n_objs} objects; "
          f"{len(err)} poses, median rel. translation error {np.median(err):.4f}")

    net = PkpNet(state_dict=weights.make_random_state_dict(0, 8.0), max_crops=n_objs)
    v, s, ids = samples[0]
    img = torch.from_numpy((255 * s["img"].numpy().transpose(1, 2, 0)).astype(np.uint8)).cuda()
    boxes = [s["bboxes"]]
    puv, pm = s["kp_uvs"].numpy(), s["kp_model_masks"].numpy()
    for _ in range(3):
        net(img, boxes, None)
        net(img, boxes, None, prior_uv=puv, prior_mask=pm)
    torch.cuda.synchronize()
    for name, kw in (("no priors (image-only stem)", {}), ("priors rendered on device", {"prior_uv": puv, "prior_mask": pm})):
        t0 = time.perf_counter()
        for _ in range(20):
            net(img, boxes, None, **kw)
        torch.cuda.synchronize()
        print(f"network pass, {n_objs} crops, {name}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms")
    dense = torch.zeros((n_objs, 41, 256, 256))
    t0 = time.perf_counter()
    for _ in range(5):
        net(img, boxes, [dense])
    torch.cuda.synchronize()
    print(f"network pass, {n_objs} crops, dense host priors uploaded (the reference's data flow): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")

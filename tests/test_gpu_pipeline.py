"""End-to-end GPU tests of the ObjectSLAM mirror (suo_slam_amd/object_slam.py): the per-frame pipeline in the
reference's own debug mode (--debug_gt_kp: projected GT keypoints + N(0,0.01) noise, object_slam.py:1129-1131),
compared with the same steps driven through the CPU oracle, plus a SLAM-mode sequence and the network path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import geometry as G  # noqa: E402
from suo_slam_amd import geometry as geo  # noqa: E402
from suo_slam_amd import synthetic as S  # noqa: E402

STRIDE = 0x9E3779B97F4A7C15


def _mesh_db(fr, sym=()):
    return {o: {"diameter": float(fr["diameter"][i]), "is_symmetric": o in sym} for i, o in enumerate(fr["obj_ids"])}


def _gt_uv(fr):
    uv = np.zeros_like(fr["uv"])
    for o in range(len(fr["boxes"])):
        uv[o] = geo.project_ndc(fr["K_bbox"][o], fr["T_OtoC"][o], fr["model_kps"][o].astype(np.float64))[0]
    return uv


def test_single_view_debug_gt_pipeline_matches_oracle():
    from suo_slam_amd import _lib
    from suo_slam_amd.object_slam import ObjectSLAM
    _lib.require_gpu()
    rng = np.random.default_rng(0)
    slam = ObjectSLAM(None, None, debug_gt_kp=True, sfm_mode=True, single_view_mode=True, seed=77)
    orc_rng = np.random.default_rng(77)
    pnp_seed = 77
    for f in range(4):
        fr = S.make_frame(rng, 8, noise=0.0, with_image=False)
        slam.mesh_db = _mesh_db(fr)
        slam.reset()
        uv_gt = _gt_uv(fr)
        boxes = fr["boxes"].astype(np.float64)
        slam.process_view(f, None, fr["K"], np.array(fr["obj_ids"]), boxes.copy(), fr["model_kps"], fr["model_kps_masks"],
                          fr["model_kps_masks"], uv_gt=uv_gt)
        res = slam.collect_results(last_only=False, no_viz=True)[f]["poses"]
        # ---- the same steps through the oracle -------------------------------------------------------
        L = 8
        K_bbox = np.stack([geo.fix_K_for_bbox_ndc(fr["K"], boxes[k]).astype(np.float32).astype(np.float64) for k in range(L)])
        uv_meas, init, ok = [], [], []
        j = 0
        for k in range(L):
            m = fr["model_kps_masks"][k]
            u = uv_gt[k][m].astype(np.float64)
            u = u + orc_rng.normal(scale=0.01, size=u.shape)
            uv_meas.append(u)
            T, best, its = G.pnp(fr["model_kps"][k][m].astype(np.float64), geo.normalize_uv(u, K_bbox[k]), 1e-3,
                                 seed=(pnp_seed + j * STRIDE) % 2 ** 64)
            j += 1
            good = (not np.allclose(T, np.eye(4))) and T[2, 3] > 0.5 * fr["diameter"][k]
            ok.append(good)
            init.append(T)
        pnp_seed += L
        objs = [k for k in range(L) if ok[k]]
        e_cam, e_obj, e_k, e_p, e_uv = [], [], [], [], []
        for jj, k in enumerate(objs):
            m = fr["model_kps_masks"][k]
            for i, p in enumerate(fr["model_kps"][k][m].astype(np.float64)):
                e_cam.append(0); e_obj.append(jj); e_p.append(p); e_uv.append(uv_meas[k][i])
                e_k.append([K_bbox[k][0, 0], K_bbox[k][1, 1], K_bbox[k][0, 2], K_bbox[k][1, 2]])
        E = len(e_cam)
        cam, obj, inl, chi2, stats = G.optimize(np.eye(4)[None, :3], np.array([1], np.uint8), np.stack([init[k][:3] for k in objs]),
                                                np.zeros(len(objs), np.uint8), e_cam, e_obj, np.array(e_k), np.array(e_p),
                                                np.array(e_uv), np.tile([1.0, 0, 1.0], (E, 1)), np.ones(E, np.uint8))
        # culling (object_slam.py:905-930)
        for jj, k in enumerate(objs):
            oid = fr["obj_ids"][k]
            n_inl = int(inl[np.array(e_obj) == jj].sum())
            removed = obj[jj][2, 3] < 0.5 * fr["diameter"][k] or n_inl < 3
            if removed:
                assert res[oid]["T_OtoC"] is None
            else:
                T = res[oid]["T_OtoC"]
                assert np.linalg.norm(T[:3, :3] - obj[jj][:, :3]) < 1e-6 and np.linalg.norm(T[:3, 3] - obj[jj][:, 3]) < 1e-3
                assert res[oid]["score"] == 1 + n_inl
                # and the estimate is near the ground truth for this noise level
                assert np.linalg.norm(T[:3, 3] - fr["T_OtoC"][k][:3, 3]) < 0.05 * fr["T_OtoC"][k][2, 3]
        for k in range(L):
            if not ok[k]:
                assert res[fr["obj_ids"][k]]["T_OtoC"] is None


def test_slam_mode_sequence_tracks_camera():
    """SLAM mode over a synthetic sequence: static objects, moving camera; exercises the camera-pose
    hypotheses (a22), re-initialisation check (a23), curr-only LM and the periodic global BA."""
    from suo_slam_amd.object_slam import ObjectSLAM
    rng = np.random.default_rng(3)
    K = S.K_YCBV
    n_obj = 6
    model_kps = np.zeros((n_obj, 41, 3), np.float32)
    masks = np.zeros((n_obj, 41), bool)
    T_OtoG = np.zeros((n_obj, 4, 4))
    diam = np.zeros(n_obj)
    for o in range(n_obj):
        masks[o] = S.class_mask(o + 1)
        ext = rng.uniform(30, 60, 3)
        model_kps[o] = (rng.uniform(-1, 1, (41, 3)) * ext).astype(np.float32)
        diam[o] = 2 * np.linalg.norm(ext)
        T_OtoG[o] = np.eye(4)
        T_OtoG[o, :3, :3] = S.random_rotation(rng)
        T_OtoG[o, :3, 3] = [(-250 + 100 * o), rng.uniform(-80, 80), rng.uniform(900, 1100)]
    obj_ids = np.arange(1, n_obj + 1)
    mesh_db = {int(o): {"diameter": float(diam[i]), "is_symmetric": bool(i % 3 == 2)} for i, o in enumerate(obj_ids)}
    slam = ObjectSLAM(None, mesh_db, debug_gt_kp=True, no_prior_det=False, seed=5, global_opt_every=5)
    n_views = 12
    errs = []
    for v in range(n_views):
        ang = 0.02 * v
        T_GtoC = np.eye(4)
        T_GtoC[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        T_GtoC[:3, 3] = [-15.0 * v, 2.0 * v, 3.0 * v]
        boxes = np.zeros((n_obj, 4))
        uv_gt = np.zeros((n_obj, 41, 2), np.float32)
        for o in range(n_obj):
            T = T_GtoC @ T_OtoG[o]
            pc = model_kps[o].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
            px = pc @ K.T
            px = px[:, :2] / px[:, 2:3]
            boxes[o] = [px[:, 0].min() - 8, px[:, 1].min() - 8, px[:, 0].max() + 8, px[:, 1].max() + 8]
            uv_gt[o] = geo.project_ndc(geo.fix_K_for_bbox_ndc(K, boxes[o]), T, model_kps[o].astype(np.float64))[0]
        slam.process_view(v, None, K, obj_ids.copy(), boxes.copy(), model_kps, masks, masks, uv_gt=uv_gt)
        assert v in slam.cam_poses
        est = geo.to4x4(slam.cam_poses[v])
        errs.append(np.linalg.norm(est[:3, 3] - T_GtoC[:3, 3]))
    res = slam.collect_results(last_only=True, final=True)
    assert len(slam.view_ids) == n_views and len(slam.obj_poses) >= n_obj - 1
    assert max(errs) < 40.0 and np.median(errs) < 15.0, errs        # mm, with 0.01 NDC keypoint noise at ~1 m
    last = res[n_views - 1]["poses"]
    n_ok = sum(1 for o in obj_ids if last[int(o)]["T_OtoC"] is not None)
    assert n_ok >= n_obj - 1


def test_network_path_runs_and_masks_match_oracle(state_dict):
    """Non-debug path: uint8 frame -> HIP network -> device masks -> PnP/LM.  Random weights give no valid
    keypoints to speak of, so this checks plumbing and the boolean masks against the oracle."""
    import torch
    from oracle import cnn_oracle as O
    from suo_slam_amd.object_slam import ObjectSLAM
    rng = np.random.default_rng(8)
    fr = S.make_frame(rng, 3, noise=0.0)
    slam = ObjectSLAM(None, _mesh_db(fr), sfm_mode=True, single_view_mode=True, state_dict=state_dict, max_crops=4,
                      kp_var_thresh=0.5, bbox_thresh=1.0)
    boxes = fr["boxes"].astype(np.float64)
    slam.process_view(0, fr["image"], fr["K"], np.array(fr["obj_ids"]), boxes.copy(), fr["model_kps"], fr["model_kps_masks"],
                      fr["model_kps_masks"])
    ref = O.pkpnet_forward(fr["image"], boxes.astype(np.float32), None, state_dict)
    ref_mask = O.keypoint_masks(ref["uv"].numpy(), ref["cov"].numpy(), ref["kp_mask"].numpy(), fr["model_kps_masks"], 1.0, 0.5)
    got = np.stack([slam.detections[0][o]["kp_mask"] for o in fr["obj_ids"]])
    uv, cov, kp = ref["uv"].numpy(), ref["cov"].numpy(), ref["kp_mask"].numpy()
    near = (np.abs(kp - 0.3) < 1e-3) | (np.abs(np.abs(uv).max(-1) - 1.0) < 1e-3) | (np.abs(np.sqrt(cov[..., [0, 1], [0, 1]]) - 1.0).min(-1) < 1e-3)
    assert np.array_equal(got[~near], ref_mask[~near])
    assert set(slam.collect_results()[0]["poses"].keys()) == set(fr["obj_ids"])


def test_symmetric_objects_get_prior_heatmaps_through_the_network(state_dict):
    """SLAM pass B (object_slam.py:486-519): a symmetric object with a map pose gets a rendered prior that is fed
    to the network as the 41 extra input channels.  Random weights: only the plumbing and the effect of the prior
    on the logits are checked (prior given != prior omitted; zero prior == omitted)."""
    import torch
    from suo_slam_amd.object_slam import ObjectSLAM, make_prior_kp_input
    from suo_slam_amd.pkpnet import PkpNet
    rng = np.random.default_rng(12)
    fr = S.make_frame(rng, 2, noise=0.0)
    mesh_db = {o: {"diameter": float(fr["diameter"][i]), "is_symmetric": True} for i, o in enumerate(fr["obj_ids"])}
    slam = ObjectSLAM(None, mesh_db, state_dict=state_dict, max_crops=4, kp_var_thresh=0.5, bbox_thresh=1.0)
    # seed the map as if a first view had initialised the objects
    slam.cam_poses[0] = np.eye(4)[:3]
    slam.view_ids.append(0)
    slam.detections[0] = {}
    for k, o in enumerate(fr["obj_ids"]):
        slam.obj_poses[o] = fr["T_OtoC"][k].copy()
    slam.process_view(1, fr["image"], fr["K"], np.array(fr["obj_ids"]), fr["boxes"].astype(np.float64), fr["model_kps"],
                      fr["model_kps_masks"], fr["model_kps_masks"], cam_pose=np.eye(4)[:3])
    for k, o in enumerate(fr["obj_ids"]):
        pu = slam.detections[1][o]["prior_uv"]
        assert pu is not None
        gt_uv = geo.project_ndc(fr["K_bbox"][k], fr["T_OtoC"][k], fr["model_kps"][k].astype(np.float64))[0]
        m = fr["model_kps_masks"][k]
        np.testing.assert_allclose(pu[m], gt_uv[m], atol=1e-5)        # projected map pose == GT projection here
    net = PkpNet(state_dict=state_dict, max_crops=2)
    pri = np.stack([make_prior_kp_input(slam.detections[1][o]["prior_uv"], fr["model_kps_masks"][k], (256, 256)) for k, o in enumerate(fr["obj_ids"])])
    bx = [torch.from_numpy(fr["boxes"])]
    a = net(fr["image"], bx, [torch.from_numpy(pri)])["prob_logits"]
    b = net(fr["image"], bx, None)["prob_logits"]
    c = net(fr["image"], bx, [torch.zeros(2, 41, 256, 256)])["prob_logits"]
    # (zero prior vs omitted: the 44-channel fp32-pipe stem vs the fused 3-channel stem on the bf16 pipe -- equal to rounding, tests/test_gpu_stem.py)
    assert float((b - c).abs().max()) < 3e-6 * float(b.abs().max()) and float((a - b).abs().max()) > 1e-3


def test_debug_gt_kp_with_the_network_running_still_uses_the_ground_truth_keypoints():
    """debug_gt_kp + run_network_in_debug in single-view mode has a model, but the keypoints must still be the projected ground
    truth (lib/object_slam.py:1129-1131), not the (random-weight) network's -- the device chain, which consumes the network's own output,
    must not take that frame.  Same seed with and without the network pass => identical poses; and they recover the ground truth."""
    from suo_slam_amd import _lib, weights
    from suo_slam_amd.object_slam import ObjectSLAM
    _lib.require_gpu()
    rng = np.random.default_rng(4)
    fr = S.make_frame(rng, 8, noise=0.0)
    uv_gt = _gt_uv(fr)
    sd = weights.make_random_state_dict(0, 8.0)
    poses = []
    for net in (False, True):
        slam = ObjectSLAM(None, _mesh_db(fr), debug_gt_kp=True, sfm_mode=True, single_view_mode=True, seed=9,
                          state_dict=sd if net else None, max_crops=8, run_network_in_debug=net)
        slam.process_view(0, fr["image"], fr["K"], np.array(fr["obj_ids"]), fr["boxes"].astype(np.float64), fr["model_kps"], fr["model_kps_masks"],
                          fr["model_kps_masks"], uv_gt=uv_gt)
        res = slam.collect_results(last_only=False, no_viz=True)[0]["poses"]
        poses.append({o: r["T_OtoC"] for o, r in res.items() if r["T_OtoC"] is not None})
    assert len(poses[0]) >= 7 and poses[0].keys() == poses[1].keys()
    for o in poses[0]:
        assert np.array_equal(poses[0][o], poses[1][o])
        k = list(fr["obj_ids"]).index(o)
        assert np.linalg.norm(poses[0][o][:3, 3] - fr["T_OtoC"][k][:3, 3]) < 0.03 * fr["T_OtoC"][k][2, 3]

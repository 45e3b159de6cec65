"""Round 6: a SLAM pass (ObjectSLAM._run_kp_model, lib/object_slam.py:1077-1167) on the device chain -- network -> masks -> compaction -> PnP -> acceptance in one
stream-ordered chain with ONE read-back (suo_frame_geom_launch, do_lm = 0) -- against the host route that restates the reference's data flow (three read-backs,
Python compaction, suo_pnp_batch on host arrays).  Same kernels on the same numbers: everything a view leaves behind must be equal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _confident(sd):
    sd = dict(sd)
    sd["classifier.2.bias"] = (np.asarray(sd["classifier.2.bias"]) + 4.0).astype(np.float32)
    return sd


def _run(seq, sd, n_views, **kw):
    from suo_slam_amd.object_slam import ObjectSLAM
    slam = ObjectSLAM(None, seq["mesh_db"], state_dict=sd, max_crops=16, **kw)
    for vw in seq["views"][:n_views]:
        slam.process_view(vw["view_id"], vw["image"], vw["K"], vw["obj_ids"].copy(), vw["bboxes"].copy(), vw["model_kps"], vw["model_kps_masks"], vw["kp_masks"],
                          uv_gt=vw["uv_gt"])
    return slam


def _same_detections(a, b, pose_tol):
    assert list(a.detections.keys()) == list(b.detections.keys())
    n = 0
    for v in a.detections:
        assert list(a.detections[v].keys()) == list(b.detections[v].keys()), v
        for o, da in a.detections[v].items():
            db = b.detections[v][o]
            np.testing.assert_array_equal(da["kp_mask"], db["kp_mask"])
            np.testing.assert_array_equal(da["uv_pred"], db["uv_pred"])
            np.testing.assert_array_equal(da["inliers"], db["inliers"])
            assert (da["cov_pred"] is None) == (db["cov_pred"] is None)
            if da["cov_pred"] is not None:
                np.testing.assert_array_equal(da["cov_pred"], db["cov_pred"])
            assert (da["pose"] is None) == (db["pose"] is None), (v, o)
            if da["pose"] is not None:
                np.testing.assert_allclose(da["pose"], db["pose"], rtol=0, atol=pose_tol * max(1.0, np.abs(db["pose"]).max()))
                n += 1
    return n


def test_slam_pass_on_the_device_chain_equals_the_host_route(state_dict, monkeypatch):
    """Network keypoints (seeded random weights whose validity head says yes, T-LESS thresholds so the masks pass): 12 views of a SLAM sequence through both routes --
    masks, keypoints, covariances bit-equal; PnP poses bit-equal (same kernel, same sampler keys); the maps and camera poses that follow from them equal."""
    from suo_slam_amd import synthetic as S
    monkeypatch.setenv("SUO_SLAM_VOTE_CHAIN", "0")           # (the passes one at a time, the host voting between them; the two-pass chain has its own test below)
    seq = S.make_slam_sequence(np.random.default_rng(11), 12, 6)
    sd = _confident(state_dict)
    kw = dict(kp_var_thresh=0.5, bbox_thresh=1.0, manual_kp_std=0.1)
    chain, host = _run(seq, sd, 12, device_chain=True, **kw), _run(seq, sd, 12, device_chain=False, **kw)
    assert _same_detections(chain, host, 0.0) >= 10
    assert list(chain.cam_poses.keys()) == list(host.cam_poses.keys()) and list(chain.obj_poses.keys()) == list(host.obj_poses.keys())
    for v in chain.cam_poses:
        np.testing.assert_allclose(np.asarray(chain.cam_poses[v])[:3], np.asarray(host.cam_poses[v])[:3], rtol=0, atol=1e-9 * max(1.0, np.abs(host.cam_poses[v]).max()))
    for o in chain.obj_poses:
        np.testing.assert_allclose(np.asarray(chain.obj_poses[o])[:3], np.asarray(host.obj_poses[o])[:3], rtol=0, atol=1e-9 * max(1.0, np.abs(host.obj_poses[o]).max()))
    assert chain._pnp_seed == host._pnp_seed


def test_ground_truth_keypoints_injected_on_the_device_track_like_the_host_debug_mode(state_dict):
    """bench.py's `slam` leg: --debug_gt_kp with the substitution made on the device tensors (float32, what the network emits) and the view continued on the
    product route, against the reference's host-side debug mode (float64 keypoints, lib/object_slam.py:1129-1131) on the same noise draws: same keypoint sets, poses
    equal to float32 keypoint rounding, the same tracking error against the ground truth."""
    from suo_slam_amd import synthetic as S
    seq = S.make_slam_sequence(np.random.default_rng(3), 20, 8)
    kw = dict(debug_gt_kp=True, manual_kp_std=0.01, run_network_in_debug=True)
    dev, host = _run(seq, state_dict, 20, debug_gt_on_device=True, **kw), _run(seq, state_dict, 20, **kw)
    assert dev.debug_gt_on_device and not host.debug_gt_on_device
    assert list(dev.cam_poses.keys()) == list(host.cam_poses.keys()) and list(dev.obj_poses.keys()) == list(host.obj_poses.keys())
    for v in dev.detections:
        for o, da in dev.detections[v].items():
            db = host.detections[v][o]
            np.testing.assert_array_equal(da["kp_mask"], db["kp_mask"])
            assert np.abs(da["uv_pred"] - db["uv_pred"]).max() < 2e-7                      # float32 rounding of NDC keypoints
            assert (da["pose"] is None) == (db["pose"] is None)
    for v in dev.cam_poses:
        np.testing.assert_allclose(np.asarray(dev.cam_poses[v])[:3], np.asarray(host.cam_poses[v])[:3], rtol=0, atol=2e-4 * max(1.0, np.abs(host.cam_poses[v]).max()))

    def err(s):
        e = []
        for vw in seq["views"][:20]:
            for o in vw["obj_ids"]:
                if int(o) in s.obj_poses and vw["view_id"] in s.cam_poses:
                    T = np.vstack([s.cam_poses[vw["view_id"]][:3], [0, 0, 0, 1]]) @ np.vstack([s.obj_poses[int(o)][:3], [0, 0, 0, 1]])
                    gt = vw["T_GtoC_gt"] @ seq["T_OtoG_gt"][int(o)]
                    e.append(np.linalg.norm(T[:3, 3] - gt[:3, 3]) / gt[2, 3])
        return float(np.median(e))
    assert abs(err(dev) - err(host)) < 1e-4 and err(dev) < 0.05


def _states_equal(a, b, pose_tol, cam_tol):
    n = _same_detections(a, b, pose_tol)
    assert list(a.cam_poses.keys()) == list(b.cam_poses.keys()) and list(a.obj_poses.keys()) == list(b.obj_poses.keys()) and a.view_ids == b.view_ids
    for v in a.cam_poses:
        np.testing.assert_allclose(np.asarray(a.cam_poses[v])[:3], np.asarray(b.cam_poses[v])[:3], rtol=0, atol=cam_tol * max(1.0, np.abs(b.cam_poses[v]).max()))
    for o in a.obj_poses:
        np.testing.assert_allclose(np.asarray(a.obj_poses[o])[:3], np.asarray(b.obj_poses[o])[:3], rtol=0, atol=cam_tol * max(1.0, np.abs(b.obj_poses[o]).max()))
    for v in a.detections:
        for o, da in a.detections[v].items():
            pa, pb = da["prior_uv"], b.detections[v][o]["prior_uv"]
            assert (pa is None) == (pb is None), (v, o)
            if pa is not None:
                assert np.abs(pa - pb).max() < 1e-6, (v, o)          # float32 NDC of a double projection: equal unless the products' last bit crosses a float32 tie
    assert a._pnp_seed == b._pnp_seed and dict(a.obj_num_dets) == dict(b.obj_num_dets) and dict(a.obj_num_det_kps) == dict(b.obj_num_det_kps)
    return n


def test_both_passes_of_a_slam_view_as_one_device_chain(state_dict, monkeypatch):
    """Round 6: pass A -> PnP -> camera-hypothesis vote -> prior projection -> pass B enqueued back to back (csrc/slam_vote.hip; ObjectSLAM._process_view_slam_chain)
    against the same views with the host voting between the passes (SUO_SLAM_VOTE_CHAIN=0): detections, votes, priors, camera poses, maps, sampler keys -- with
    ground-truth keypoints injected on the device (a sequence that tracks: every view takes the chain) and with network keypoints (seeded random weights: most PnP
    poses fail the vote, the views fall back to the bbox-centroid pose and issue pass B again)."""
    from suo_slam_amd import synthetic as S
    seq = S.make_slam_sequence(np.random.default_rng(3), 30, 8)
    kw = dict(debug_gt_kp=True, manual_kp_std=0.01, run_network_in_debug=True, debug_gt_on_device=True)
    monkeypatch.setenv("SUO_SLAM_VOTE_CHAIN", "1")
    calls = []
    from suo_slam_amd.object_slam import ObjectSLAM
    orig = ObjectSLAM._process_view_slam_chain

    def spy(self, *a, **k):
        r = orig(self, *a, **k)
        calls.append(r)
        return r
    monkeypatch.setattr(ObjectSLAM, "_process_view_slam_chain", spy)
    chain = _run(seq, state_dict, 30, **kw)
    assert len(calls) >= 25 and all(calls), "the tracking views of this sequence take the chain and find a camera pose"
    hyp_chain = chain.last_cam_hypotheses
    monkeypatch.setenv("SUO_SLAM_VOTE_CHAIN", "0")
    n0 = len(calls)
    host = _run(seq, state_dict, 30, **kw)
    assert len(calls) == n0
    assert _states_equal(chain, host, 0.0, 1e-9) >= 100
    assert hyp_chain == host.last_cam_hypotheses
    # network keypoints: random weights, T-LESS thresholds
    sd = _confident(state_dict)
    kw = dict(kp_var_thresh=0.5, bbox_thresh=1.0, manual_kp_std=0.1)
    seq2 = S.make_slam_sequence(np.random.default_rng(11), 12, 6)
    monkeypatch.setenv("SUO_SLAM_VOTE_CHAIN", "1")
    n0 = len(calls)
    chain2 = _run(seq2, sd, 12, **kw)
    assert len(calls) > n0
    monkeypatch.setenv("SUO_SLAM_VOTE_CHAIN", "0")
    host2 = _run(seq2, sd, 12, **kw)
    _states_equal(chain2, host2, 0.0, 1e-9)

"""SLAM host rules on the GPU path: ``process_view`` is driven into each branch of
/root/reference/lib/object_slam.py:933-973 (backup camera pose: bbox-centroid PnP -> constant velocity -> copy),
:595-697 (re-initialisation fires / does not fire) and :975-1072 (no usable hypothesis -> backup), and the outcome is
compared with the loop-per-detection restatement in oracle/slam_rules.py (PnP there = the C oracle of lambdatwist)."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import slam_rules as R  # noqa: E402
from suo_slam_amd import geometry as geo  # noqa: E402
from suo_slam_amd import synthetic as S  # noqa: E402


class Scene:
    """Static objects in a world frame, a camera on a smooth path; views in the reference's debug_gt_kp form."""

    def __init__(self, seed, n_obj, sym=()):
        rng = np.random.default_rng(seed)
        self.K = S.K_YCBV
        self.n_obj = n_obj
        self.obj_ids = np.arange(1, n_obj + 1)
        self.model_kps = np.zeros((n_obj, 41, 3), np.float32)
        self.masks = np.zeros((n_obj, 41), bool)
        self.T_OtoG = np.zeros((n_obj, 4, 4))
        diam = np.zeros(n_obj)
        for o in range(n_obj):
            self.masks[o] = S.class_mask(o + 1)
            ext = rng.uniform(30, 60, 3)
            self.model_kps[o] = (rng.uniform(-1, 1, (41, 3)) * ext).astype(np.float32)
            diam[o] = 2 * np.linalg.norm(ext)
            self.T_OtoG[o] = np.eye(4)
            self.T_OtoG[o, :3, :3] = S.random_rotation(rng)
            self.T_OtoG[o, :3, 3] = [-250 + 500 * (o + 0.5) / n_obj, rng.uniform(-80, 80), rng.uniform(900, 1100)]
        self.mesh_db = {int(o): {"diameter": float(diam[i]), "is_symmetric": int(o) in sym} for i, o in enumerate(self.obj_ids)}

    def cam(self, v):
        ang = 0.02 * v
        T = np.eye(4)
        T[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        T[:3, 3] = [-15.0 * v, 2.0 * v, 3.0 * v]
        return T

    def view(self, v):
        T_GtoC = self.cam(v)
        boxes = np.zeros((self.n_obj, 4))
        uv_gt = np.zeros((self.n_obj, 41, 2), np.float32)
        for o in range(self.n_obj):
            T = T_GtoC @ self.T_OtoG[o]
            pc = self.model_kps[o].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
            px = pc @ self.K.T
            px = px[:, :2] / px[:, 2:3]
            boxes[o] = [px[:, 0].min() - 8, px[:, 1].min() - 8, px[:, 0].max() + 8, px[:, 1].max() + 8]
            uv_gt[o] = geo.project_ndc(geo.fix_K_for_bbox_ndc(self.K, boxes[o]), T, self.model_kps[o].astype(np.float64))[0]
        return boxes, uv_gt

    def feed(self, slam, v, kp_masks=None):
        boxes, uv_gt = self.view(v)
        slam.process_view(v, None, self.K, self.obj_ids.copy(), boxes.copy(), self.model_kps, self.masks,
                          self.masks if kp_masks is None else kp_masks, uv_gt=uv_gt)
        return boxes


def _slam(scene, seed=5, **kw):
    from suo_slam_amd import _lib
    from suo_slam_amd.object_slam import ObjectSLAM
    _lib.require_gpu()
    return ObjectSLAM(None, scene.mesh_db, debug_gt_kp=True, seed=seed, global_opt_every=1000, **kw)


def _oracle_backup(slam, scene, boxes):
    """The reference's backup rule on a snapshot of the state, BEFORE the product runs it."""
    return R.backup_estimate_camera_pose(copy.deepcopy(slam.cam_poses), copy.deepcopy(slam.obj_poses), list(slam.view_ids), scene.K,
                                         scene.obj_ids, boxes, lambda p3, p2, K: R.pnp(p3, p2, K, seed=0))


def test_no_non_symmetric_objects_takes_the_bbox_centroid_pnp():
    """All objects symmetric => n_non_sym == 0 on the second view => backup rule, first branch (:944-956)."""
    scene = Scene(1, 6, sym=set(range(1, 7)))
    slam = _slam(scene)
    scene.feed(slam, 0)                                       # first view: symmetric objects are treated as new (no map yet)
    assert len(slam.obj_poses) >= 5 and 0 in slam.cam_poses
    boxes, _ = scene.view(1)
    want, which = _oracle_backup(slam, scene, boxes)
    assert which == "centroid_pnp"
    calls = []
    orig = slam._backup_estimate_camera_pose
    slam._backup_estimate_camera_pose = lambda *a: (calls.append(1), orig(*a))[1]
    # stop after the backup rule: the pose it stored is what the rest of process_view starts from
    snap = {}
    orig_po = slam._process_objects
    slam._process_objects = lambda *a, **k: (snap.setdefault("cam", np.array(slam.cam_poses[1])), orig_po(*a, **k))[1]
    scene.feed(slam, 1)
    assert calls == [1]
    np.testing.assert_allclose(snap["cam"], want, rtol=0, atol=1e-8)
    assert slam.view_ids == [0, 1]
    # the centroid pose is crude (box centres are not projected object centres) but in the right place
    assert np.linalg.norm(geo.to4x4(snap["cam"])[:3, 3] - scene.cam(1)[:3, 3]) < 150.0


def test_centroid_pnp_failure_falls_back_to_constant_velocity_then_copy():
    """Fewer than 4 map objects => pnp() returns None (:31) => constant velocity with two earlier views (:961-968),
    copy of the last pose with one (:969-971)."""
    scene = Scene(2, 3, sym={1, 2, 3})
    slam = _slam(scene)
    scene.feed(slam, 0)
    boxes, _ = scene.view(1)
    want, which = _oracle_backup(slam, scene, boxes)
    assert which == "copy"
    slam.cam_K[1] = slam.cam_K[2] = scene.K                  # process_view stores K before it calls the rule (:333)
    slam._backup_estimate_camera_pose(1, scene.obj_ids, boxes)
    assert np.array_equal(np.asarray(slam.cam_poses[1]), np.asarray(want)) and slam.view_ids == [0, 1]
    # give view 1 a real motion, then ask for view 2
    slam.cam_poses[1] = scene.cam(1)[:3]
    slam.detections[1] = {}
    boxes, _ = scene.view(2)
    want, which = _oracle_backup(slam, scene, boxes)
    assert which == "const_velocity"
    slam._backup_estimate_camera_pose(2, scene.obj_ids, boxes)
    np.testing.assert_allclose(slam.cam_poses[2], want, rtol=0, atol=1e-12)
    T1, T2 = geo.to4x4(slam.cam_poses[0]), geo.to4x4(slam.cam_poses[1])
    np.testing.assert_allclose(slam.cam_poses[2], T2 @ np.linalg.inv(T1) @ T2, atol=1e-9)


def test_no_usable_hypothesis_reaches_the_backup_rule_through_process_view():
    """Non-symmetric objects exist but none yields a PnP pose in the new view (< 4 visible keypoints, :31) =>
    __estimate_camera_pose returns None (:998-1000) => "Non-symmetric camera pose estimation failed" (:398-401)."""
    scene = Scene(3, 5)
    slam = _slam(scene)
    scene.feed(slam, 0)
    scene.feed(slam, 1)
    few = scene.masks.copy()
    for o in range(scene.n_obj):
        few[o, np.nonzero(few[o])[0][3:]] = False            # three keypoints per object
    boxes, _ = scene.view(2)
    want, which = _oracle_backup(slam, scene, boxes)
    assert which == "centroid_pnp"
    calls = []
    orig = slam._backup_estimate_camera_pose
    slam._backup_estimate_camera_pose = lambda *a: (calls.append(np.array(a[2])), orig(*a), calls.append(np.array(slam.cam_poses[2])))[1]
    scene.feed(slam, 2, kp_masks=few)
    assert len(calls) == 2 and slam.last_cam_hypotheses is None
    np.testing.assert_allclose(calls[1], want, rtol=0, atol=1e-8)
    assert slam.view_ids == [0, 1, 2]


def test_reinit_fires_for_a_corrupted_map_pose_and_only_for_it():
    """A badly initialised object (:595-602): its PnP pose in the new view explains every keypoint of the last views,
    the map pose none => re-initialised from the PnP pose; the healthy objects keep their map poses."""
    scene = Scene(4, 5)
    slam = _slam(scene)
    scene.feed(slam, 0)
    scene.feed(slam, 1)
    bad = 3
    good_before = {o: np.array(T) for o, T in slam.obj_poses.items() if o != bad}
    T_bad = geo.to4x4(slam.obj_poses[bad]).copy()
    T_bad[:3, 3] += [60.0, -40.0, 80.0]
    slam.obj_poses[bad] = T_bad
    reports = []
    orig = slam._maybe_reinit_objects

    def spy(view_id, n):
        state = (copy.deepcopy(slam.detections), copy.deepcopy(slam.cam_poses), copy.deepcopy(slam.obj_poses), list(slam.view_ids))
        want = R.maybe_reinit_objects(state[0], state[1], state[2], state[3], view_id, slam.manual_kp_std, n)
        got = orig(view_id, n)
        reports.append((want, got, n))
        return got
    slam._maybe_reinit_objects = spy
    scene.feed(slam, 2)
    want, got, n = reports[0]
    assert n == 15 and set(want) == set(got) == set(slam.obj_poses)
    for o in want:
        assert (got[o]["pnp"], got[o]["estim"], got[o]["reinit"]) == (want[o]["pnp"], want[o]["estim"], want[o]["reinit"])
        assert got[o]["reinit"] == (o == bad)
    assert got[bad]["estim"] == 0 and got[bad]["pnp"] >= 10         # manual sigma 0.005 against 0.01 noise: a fraction of 3 x 22 keypoints
    np.testing.assert_allclose(geo.to4x4(slam.obj_poses[bad]), want[bad]["T_OtoG_pnp"], atol=1e-9)
    truth = scene.T_OtoG[bad - 1]                                            # world frame == first camera frame here
    assert np.linalg.norm(geo.to4x4(slam.obj_poses[bad])[:3, 3] - truth[:3, 3]) < 25.0
    for o, T in good_before.items():
        assert np.array_equal(np.asarray(slam.obj_poses[o]), T)             # curr_only optimisation moves the camera only

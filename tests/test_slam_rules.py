"""SLAM host rules: the product's vectorised ``_estimate_camera_pose`` /
``_maybe_reinit_objects`` (suo_slam_amd/object_slam.py) against the loop-per-detection restatement of
/root/reference/lib/object_slam.py:975-1072 and :595-697 in oracle/slam_rules.py, on random SLAM states and on states
built to sit on each branch: the 3x re-initialisation boundary (both sides), the >= 3 floor, the >= 4 hypothesis
inliers floor, the covariance clamp at 1e-4, the manual-sigma path, the 15-view window, objects without a PnP pose or
without a map pose.  Host numpy only (no GPU): ObjectSLAM in debug_gt_kp mode owns no network.
(Row a24 needs PnP and lives in tests/test_gpu_slam_rules.py.)"""
import numpy as np
import pytest

from oracle import slam_rules as R
from suo_slam_amd import geometry as geo
from suo_slam_amd import synthetic as S
from suo_slam_amd.object_slam import ObjectSLAM


def _pose(rng, z=(700, 1100)):
    T = np.eye(4)
    T[:3, :3] = S.random_rotation(rng)
    T[:3, 3] = [rng.uniform(-150, 150), rng.uniform(-100, 100), rng.uniform(*z)]
    return T


def _small(rng, rot, trans):
    return np.vstack([S._perturb_pose(np.eye(4)[:3], rng, rot, trans), [0, 0, 0, 1]])


def make_state(rng, n_obj=5, n_views=4, use_cov=True, noise=0.01, pnp_rot=5e-4, pnp_trans=0.5, map_rot=5e-4, map_trans=0.5,
               miss=0.2, drop_pose=0.15):
    """A SLAM state as process_view leaves it before the rules run: n_views views with poses, the last one current;
    detections with float32 covariances, a PnP pose per detection (ground truth perturbed), inlier flags."""
    K = S.K_YCBV
    T_OtoG = {o: _pose(rng) for o in range(1, n_obj + 1)}
    n_kp = {o: int(rng.integers(6, 14)) for o in T_OtoG}
    kps = {o: rng.uniform(-60, 60, (n_kp[o], 3)) for o in T_OtoG}
    slam = ObjectSLAM(None, {o: {"diameter": 120.0, "is_symmetric": False} for o in T_OtoG}, debug_gt_kp=True, manual_kp_std=0.01)
    if use_cov:
        slam.no_network_cov = False
    for v in range(n_views):
        T_GtoC = _small(rng, 0.05, 30.0)
        slam.cam_poses[v] = (T_GtoC @ _small(rng, 1e-3, 1.0))[:3] if v % 2 else T_GtoC @ _small(rng, 1e-3, 1.0)     # [3,4] and [4,4] both occur
        slam.view_ids.append(v)
        slam.detections[v] = {}
        for o in T_OtoG:
            if rng.random() < miss and v != n_views - 1:
                continue
            T_OtoC = T_GtoC @ T_OtoG[o]
            pc = kps[o] @ T_OtoC[:3, :3].T + T_OtoC[:3, 3]
            px = pc @ K.T
            px = px[:, :2] / px[:, 2:3]
            bbox = np.array([px[:, 0].min() - 8, px[:, 1].min() - 8, px[:, 0].max() + 8, px[:, 1].max() + 8])
            Kb = geo.fix_K_for_bbox_ndc(K, bbox)
            uv = geo.project_ndc(Kb, T_OtoC, kps[o])[0] + rng.normal(0, noise, (n_kp[o], 2))
            out = rng.random(n_kp[o]) < 0.15
            uv[out] += rng.uniform(-0.5, 0.5, (int(out.sum()), 2))
            cov = None
            if use_cov:
                A = rng.normal(0, 0.3, (n_kp[o], 2, 2)) + np.eye(2)
                cov = ((A @ A.transpose(0, 2, 1)) * noise * noise).astype(np.float32)
            pose = None if rng.random() < drop_pose else _small(rng, pnp_rot, pnp_trans) @ T_OtoC
            slam.detections[v][o] = {"pose": pose, "inliers": rng.random(n_kp[o]) < 0.85, "model_kp": kps[o].copy(), "uv_pred": uv,
                                     "cov_pred": cov, "K": Kb.astype(np.float32).astype(np.float64), "bbox": bbox}
    for o in T_OtoG:
        if rng.random() < 0.9:
            slam.obj_poses[o] = (_small(rng, map_rot, map_trans) @ T_OtoG[o])[:3] if o % 2 else _small(rng, map_rot, map_trans) @ T_OtoG[o]
    return slam


@pytest.mark.parametrize("use_cov", [True, False])
def test_estimate_camera_pose_matches_the_restatement_on_random_states(use_cov):
    rng = np.random.default_rng(11 + use_cov)
    n_found = 0
    for trial in range(40):
        slam = make_state(rng, n_obj=int(rng.integers(1, 7)), n_views=3, use_cov=use_cov)
        view = slam.view_ids.pop()                       # the rule runs before the view has a pose (:446, :997)
        slam.cam_poses.pop(view)
        got = slam._estimate_camera_pose(view)
        want, want_n, counts = R.estimate_camera_pose(slam.detections, slam.obj_poses, view, slam.manual_kp_std)
        if want is None:
            assert got is None
            continue
        n_found += 1
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
        assert slam.last_cam_hypotheses["counts"] == counts and slam.last_cam_hypotheses["best_num_inliers"] == want_n
    assert n_found >= 25


@pytest.mark.parametrize("use_cov", [True, False])
def test_maybe_reinit_matches_the_restatement_on_random_states(use_cov):
    rng = np.random.default_rng(21 + use_cov)
    fired = quiet = 0
    for trial in range(40):
        bad_map = trial % 2 == 0                         # half of the states carry badly initialised map poses
        slam = make_state(rng, n_obj=int(rng.integers(2, 7)), n_views=int(rng.integers(2, 22)), use_cov=use_cov,
                          map_rot=0.2 if bad_map else 1e-3, map_trans=60.0 if bad_map else 1.0)
        view = slam.view_ids[-1]
        before = {o: np.array(T) for o, T in slam.obj_poses.items()}
        want = R.maybe_reinit_objects(slam.detections, slam.cam_poses, before, slam.view_ids, view, slam.manual_kp_std, 15)
        got = slam._maybe_reinit_objects(view, 15)
        assert set(got) == set(want)
        for o in want:
            assert (got[o]["pnp"], got[o]["estim"], got[o]["reinit"]) == (want[o]["pnp"], want[o]["estim"], want[o]["reinit"]), (trial, o)
            if want[o]["reinit"]:
                fired += 1
                np.testing.assert_allclose(slam.obj_poses[o], want[o]["T_OtoG_pnp"], rtol=0, atol=1e-9)
            else:
                quiet += 1
                assert np.array_equal(slam.obj_poses[o], before[o])
    assert fired >= 20 and quiet >= 20, (fired, quiet)


def _two_pose_state(n_a, n_b, n_views=2, in_old_views=False):
    """One object whose current detection has n_a keypoints that agree with its PnP pose A, n_b that agree with the map
    pose B and 4 that agree with neither; manual sigma, so a keypoint is an inlier iff its residual is < 0.0245 NDC."""
    rng = np.random.default_rng(100 * n_a + n_b)
    K = S.K_YCBV
    n = n_a + n_b + 4
    kps = rng.uniform(-60, 60, (n, 3))
    T_A = _pose(rng)
    T_B = _small(rng, 0.3, 40.0) @ T_A
    slam = ObjectSLAM(None, {1: {"diameter": 120.0, "is_symmetric": False}}, debug_gt_kp=True, manual_kp_std=0.01)
    for v in range(n_views):
        slam.cam_poses[v] = np.eye(4)[:3]
        slam.view_ids.append(v)
        slam.detections[v] = {}
    pc = kps @ T_A[:3, :3].T + T_A[:3, 3]
    px = pc @ K.T
    px = px[:, :2] / px[:, 2:3]
    bbox = np.array([px[:, 0].min() - 40, px[:, 1].min() - 40, px[:, 0].max() + 40, px[:, 1].max() + 40])
    Kb = geo.fix_K_for_bbox_ndc(K, bbox)
    uv_a, uv_b = geo.project_ndc(Kb, T_A, kps)[0], geo.project_ndc(Kb, T_B, kps)[0]
    assert np.linalg.norm(uv_a - uv_b, axis=1).min() > 0.1            # the two poses never explain the same keypoint
    uv = uv_a + 3.0
    uv[:n_a] = uv_a[:n_a]
    uv[n_a:n_a + n_b] = uv_b[n_a:n_a + n_b]
    det = {"pose": T_A, "inliers": np.ones(n, bool), "model_kp": kps, "uv_pred": uv, "cov_pred": None, "K": Kb, "bbox": bbox}
    for v in (range(n_views) if in_old_views else [n_views - 1]):
        slam.detections[v][1] = dict(det)
    slam.obj_poses[1] = T_B.copy()
    return slam, T_A, T_B


@pytest.mark.parametrize("n_a,n_b,fires", [(9, 3, False), (10, 3, True), (2, 0, False), (3, 0, True), (3, 1, False), (4, 1, True), (0, 5, False)])
def test_reinit_rule_at_its_boundaries(n_a, n_b, fires):
    """lib/object_slam.py:683-687: re-initialise iff pnp >= 3 and pnp > 3 * estim."""
    slam, T_A, T_B = _two_pose_state(n_a, n_b)
    want = R.maybe_reinit_objects(slam.detections, slam.cam_poses, dict(slam.obj_poses), slam.view_ids, 1, 0.01, 15)
    got = slam._maybe_reinit_objects(1, 15)
    assert (want[1]["pnp"], want[1]["estim"], want[1]["reinit"]) == (n_a, n_b, fires)
    assert (got[1]["pnp"], got[1]["estim"], got[1]["reinit"]) == (n_a, n_b, fires)
    np.testing.assert_allclose(slam.obj_poses[1], T_A if fires else T_B, atol=1e-9)


def test_reinit_counts_only_the_last_fifteen_views_and_needs_two():
    slam, T_A, T_B = _two_pose_state(2, 1, n_views=20, in_old_views=True)
    got = slam._maybe_reinit_objects(19, 15)
    want = R.maybe_reinit_objects(slam.detections, slam.cam_poses, dict(slam.obj_poses), slam.view_ids, 19, 0.01, 15)
    assert got[1]["pnp"] == want[1]["pnp"] == 30 and got[1]["estim"] == want[1]["estim"] == 15 and not got[1]["reinit"]
    got = ObjectSLAM._maybe_reinit_objects(slam, 19, len(slam.view_ids))        # sfm_mode passes the whole history (:417)
    assert got[1]["pnp"] == 40 and got[1]["estim"] == 20
    one, _, _ = _two_pose_state(10, 0, n_views=1)
    assert one._maybe_reinit_objects(0, 15) == {} and R.maybe_reinit_objects(one.detections, one.cam_poses, one.obj_poses, one.view_ids, 0, 0.01) == {}
    assert np.array_equal(one.obj_poses[1], _two_pose_state(10, 0, n_views=1)[2])


def test_covariance_clamp_decides_the_inlier():
    """lib/object_slam.py:1054 / :669: the covariance diagonal is raised to 1e-4 before inversion.  A residual of 0.02
    under cov = 1e-6 I has chi2 = 400 without the clamp and 4 < 5.991 with it."""
    rng = np.random.default_rng(5)
    slam = make_state(rng, n_obj=1, n_views=2, use_cov=True, noise=0.0, miss=0.0, drop_pose=0.0, pnp_rot=0.0, pnp_trans=0.0,
                      map_rot=0.0, map_trans=0.0)
    if not slam.obj_poses:
        slam.obj_poses[1] = np.linalg.inv(geo.to4x4(slam.cam_poses[1])) @ slam.detections[1][1]["pose"]
    view = slam.view_ids.pop()
    slam.cam_poses.pop(view)
    d = slam.detections[view][1]
    n = len(d["uv_pred"])
    d["inliers"] = np.ones(n, bool)
    exact = geo.project_ndc(d["K"], d["pose"], d["model_kp"])[0]
    d["uv_pred"] = exact + np.array([0.02, 0.0])
    d["cov_pred"] = np.tile(np.eye(2, dtype=np.float32) * 1e-6, (n, 1, 1))
    slam.obj_poses[1] = np.linalg.inv(np.eye(4)) @ d["pose"]                    # hypothesis == the PnP pose itself
    got = slam._estimate_camera_pose(view)
    want, want_n, _ = R.estimate_camera_pose(slam.detections, slam.obj_poses, view, 0.01)
    assert want_n == n and slam.last_cam_hypotheses["best_num_inliers"] == n and got is not None
    d["uv_pred"] = exact + np.array([0.03, 0.0])                                # 9 > 5.991 even when clamped
    assert slam._estimate_camera_pose(view) is None and R.estimate_camera_pose(slam.detections, slam.obj_poses, view, 0.01)[0] is None


def test_camera_pose_needs_four_hypothesis_inliers_and_first_best_wins():
    slam, T_A, T_B = _two_pose_state(3, 0)
    slam.view_ids.pop()
    slam.cam_poses.pop(1)
    slam.obj_poses[1] = np.eye(4)                                               # hypothesis = T_A
    assert slam._estimate_camera_pose(1) is None and slam.last_cam_hypotheses["counts"] == [3]
    assert R.estimate_camera_pose(slam.detections, slam.obj_poses, 1, 0.01)[0] is None
    slam, T_A, T_B = _two_pose_state(4, 0)
    slam.view_ids.pop()
    slam.cam_poses.pop(1)
    slam.obj_poses[1] = np.eye(4)
    slam.detections[1][2] = dict(slam.detections[1][1])                         # a second object with the same evidence: a tie
    slam.obj_poses[2] = np.eye(4)
    got = slam._estimate_camera_pose(1)
    want = R.estimate_camera_pose(slam.detections, slam.obj_poses, 1, 0.01)
    np.testing.assert_allclose(got, T_A, atol=1e-9)
    np.testing.assert_allclose(want[0], T_A, atol=1e-9)
    assert slam.last_cam_hypotheses["counts"] == want[2] == [8, 8] and want[1] == 8
    # objects without a PnP pose, or without a map pose, propose and score nothing (:992-995)
    slam.detections[1][2]["pose"] = None
    slam.obj_poses.pop(1)
    assert slam._estimate_camera_pose(1) is None and R.estimate_camera_pose(slam.detections, slam.obj_poses, 1, 0.01)[0] is None

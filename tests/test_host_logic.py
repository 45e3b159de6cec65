"""CPU tests of the host-side mirror (suo_slam_amd/object_slam.py): prior rendering, chi2 scoring, graph
flattening and read-back/culling rules -- everything that does not need the HIP extension to execute."""
import numpy as np
import pytest

from suo_slam_amd import geometry as geo
from suo_slam_amd import object_slam as OS
from suo_slam_amd import synthetic as S
from tests import host_scoring as HS


def test_prior_patch_properties():
    """utils.py:356-385 semantics: 91x91 max-normalised Gaussian (OpenCV sigma rule -> 14.0), pasted by
    assignment into the window [pt-45, pt+45) clipped to the image."""
    g = OS._gaussian_patch(91)
    assert g.shape == (91, 91) and g.dtype == np.float32 and g.max() == 1.0 and g[45, 45] == 1.0
    np.testing.assert_allclose(g[45, 45 + 14], np.exp(-0.5), rtol=1e-6)            # one sigma = 14 px
    np.testing.assert_allclose(g[45, 0], 2 * np.exp(-45 ** 2 / (2 * 14.0 ** 2)), rtol=1e-5)   # reflected border ring
    img = np.zeros((256, 256), np.float32)
    OS.draw_gaussian_2d(img, (100, 120))
    assert img[120, 100] == 1.0
    ys, xs = np.nonzero(img)
    assert xs.min() == 55 and xs.max() == 144 and ys.min() == 75 and ys.max() == 164          # 90-px window, exclusive end
    img2 = np.ones((256, 256), np.float32) * 7
    OS.draw_gaussian_2d(img2, (5, 250))                                                      # clipped at two borders
    assert img2[250, 5] == 1.0 and img2[0, 0] == 7 and img2[255, 49] < 1.0 and img2[255, 50] == 7
    img3 = np.zeros((256, 256), np.float32)
    OS.draw_gaussian_2d(img3, (-100, 40))                                                    # fully outside: untouched
    assert not img3.any()


def test_make_prior_kp_input_ndc_mapping_and_masking():
    uv = np.zeros((41, 2), np.float32)
    mask = np.zeros(41, bool)
    uv[3] = [0.0, 0.0]; mask[3] = True            # centre -> pixel (127.5 -> round-half-even 128, 127.5 -> 128)
    uv[7] = [-1.0, 1.0]; mask[7] = True           # NDC (-1,+1) = top-left: u=-0.5 -> 0 (banker's), v = -0.5 -> 0
    uv[9] = [np.nan, 0.1]; mask[9] = True         # non-finite: skipped
    uv[11] = [0.5, 0.5]                           # masked out
    x = OS.make_prior_kp_input(uv, mask, (256, 256), ndc=True)
    assert x.shape == (41, 256, 256)
    assert x[3, 128, 128] == 1.0 and x[7, 0, 0] == 1.0
    assert not x[9].any() and not x[11].any() and not x[0].any()


def _slam_with_state(rng, noise=0.0):
    fr = S.make_frame(rng, 3, noise=noise, with_image=False)
    mesh_db = {o: {"diameter": float(fr["diameter"][i]), "is_symmetric": False} for i, o in enumerate(fr["obj_ids"])}
    slam = OS.ObjectSLAM(None, mesh_db, debug_gt_kp=True, sfm_mode=True, single_view_mode=True)
    det = {}
    for k, o in enumerate(fr["obj_ids"]):
        m = fr["model_kps_masks"][k]
        det[o] = {"pose": fr["T_OtoC"][k], "inliers": np.ones(m.sum(), bool), "kp_mask": m, "model_kp": fr["model_kps"][k][m].astype(np.float64),
                  "uv_pred": fr["uv"][k][m].astype(np.float64), "cov_pred": fr["cov"][k][m], "K": fr["K_bbox"][k], "bbox": fr["boxes"][k],
                  "model_kp_mask": m, "prior_uv": None}
        slam.obj_poses[o] = fr["T_OtoC"][k].copy()
        slam.obj_num_dets[o] = 1
    slam.detections[0] = det
    slam.cam_poses[0] = np.eye(4)[:3]
    slam.view_ids.append(0)
    return slam, fr


def test_k_bbox_of_all_boxes_at_once_is_bit_identical():
    """geometry.fix_K_for_bbox_ndc_many (what process_view calls per frame) == fix_K_for_bbox_ndc per box (pinned to the reference's
    utils.fix_K_for_bbox_ndc, tests/test_host_golden.py), bit for bit, also through the float32 container of object_slam.py:1082."""
    rng = np.random.default_rng(0)
    K = S.K_YCBV
    for _ in range(100):
        n = int(rng.integers(1, 17))
        c, hw = rng.uniform([50, 50], [590, 430], (n, 2)), rng.uniform(3, 250, (n, 2))
        b = np.concatenate([c - hw, c + hw], 1)
        many = geo.fix_K_for_bbox_ndc_many(K, b)
        one = np.stack([geo.fix_K_for_bbox_ndc(K, b[i]) for i in range(n)])
        assert np.array_equal(many, one) and many.dtype == np.float64
    assert geo.fix_K_for_bbox_ndc_many(K, np.zeros((0, 4))).shape == (0, 3, 3)


def test_chi2_scoring_counts_all_keypoints_at_the_true_pose():
    slam, fr = _slam_with_state(np.random.default_rng(0))
    for k, o in enumerate(fr["obj_ids"]):
        d = slam.detections[0][o]
        n = len(d["model_kp"])
        assert HS._chi2_inliers(fr["T_OtoC"][k], d, True, 0.005) == n
        T_bad = fr["T_OtoC"][k].copy()
        T_bad[:3, 3] += [80.0, 0, 0]
        assert HS._chi2_inliers(T_bad, d, True, 0.005) < n // 2
        d2 = dict(d, cov_pred=None)                                   # manual sigma path (object_slam.py:1059-1061)
        assert HS._chi2_inliers(fr["T_OtoC"][k], d2, False, 0.005) == n


@pytest.mark.parametrize("with_cov", [True, False])
def test_vectorised_chi2_scoring_equals_the_per_detection_rule(with_cov):
    """ObjectSLAM scores all (pose, detection) pairs of a view in one pass (_chi2_inliers_many); it must count exactly
    what the per-detection rule counts: good and bad poses, points behind the camera, inlier subsets, empty detections."""
    rng = np.random.default_rng(5)
    slam, fr = _slam_with_state(rng, noise=0.004)
    Ts, dets = [], []
    for rep in range(6):
        for k, o in enumerate(fr["obj_ids"]):
            d = dict(slam.detections[0][o])
            d.pop("_cache", None)
            if not with_cov:
                d["cov_pred"] = None
            n = len(d["model_kp"])
            d["inliers"] = rng.uniform(size=n) < 0.7
            T = fr["T_OtoC"][k].copy()
            if rep == 1:
                T[:3, 3] += rng.normal(scale=3.0, size=3)            # a few keypoints drop out
            elif rep == 2:
                T[2, 3] = -T[2, 3]                                   # everything behind the camera
            elif rep == 3:
                T[2, 3] = 0.02                                       # some behind, some in front
            elif rep == 4:
                d["inliers"] = np.zeros(n, bool)
            elif rep == 5:                                           # empty detection
                d.update(model_kp=d["model_kp"][:0], uv_pred=d["uv_pred"][:0], inliers=np.zeros(0, bool),
                         cov_pred=None if d["cov_pred"] is None else d["cov_pred"][:0])
            Ts.append(T)
            dets.append(d)
    for subset in (True, False):
        many = HS._chi2_inliers_many(Ts, dets, subset, 0.005)
        one = [HS._chi2_inliers(T, d, subset, 0.005) for T, d in zip(Ts, dets)]
        assert list(many) == one
    assert max(one) > 0 and min(one) == 0
    assert len(HS._chi2_inliers_many([], [], True, 0.005)) == 0


def test_build_problem_flattens_the_reference_graph():
    slam, fr = _slam_with_state(np.random.default_rng(1))
    prob, book = slam.build_problem(curr_only=False)
    cam_index, obj_index, e_ref, curr_only, view_curr = book
    n_edges = sum(int(fr["model_kps_masks"][k].sum()) for k in range(3))
    assert len(prob.edge_cam) == n_edges == len(e_ref)
    assert list(prob.cam_fixed) == [1] and list(prob.obj_fixed) == [0, 0, 0]          # gauge: first camera fixed (:774)
    assert prob.its == (10, 10, 40, 40) and not prob.init_with_outliers                # sfm_mode => long rounds (:843)
    k0 = fr["K_bbox"][0]
    np.testing.assert_allclose(prob.edge_camk[0], [k0[0, 0], k0[1, 1], k0[0, 2], k0[1, 2]])
    Om = np.linalg.inv(fr["cov"][0][fr["model_kps_masks"][0]][0].astype(np.float64))
    np.testing.assert_allclose(prob.edge_info[0], [Om[0, 0], Om[0, 1], Om[1, 1]])
    # curr_only: objects fixed, camera free, short rounds (slam mode only selects them; here sfm_mode keeps the long ones)
    prob2, book2 = slam.build_problem(curr_only=True)
    assert list(prob2.cam_fixed) == [0] and list(prob2.obj_fixed) == [1, 1, 1]
    slam.sfm_mode, slam.slam_mode = False, True
    assert slam.build_problem(curr_only=True)[0].its == (10, 10, 10, 10)
    slam.opt_init_with_outliers = True
    assert slam.build_problem(curr_only=True)[0].init_with_outliers and not slam.build_problem(curr_only=False)[0].init_with_outliers


def test_apply_problem_culls_like_the_reference():
    slam, fr = _slam_with_state(np.random.default_rng(2))
    prob, book = slam.build_problem(curr_only=False)
    o0, o1, o2 = fr["obj_ids"]
    # object 0: behind the camera after "optimisation"; object 1: only 2 inliers left; object 2: fine
    prob.obj_T.reshape(-1, 3, 4)[book[1][o0], 2, 3] = 0.1 * fr["diameter"][0]
    inl = prob.inlier
    idx1 = [i for i, (v, o, k) in enumerate(book[2]) if o == o1]
    inl[idx1[2:]] = 0
    slam.apply_problem(prob, book)
    assert o0 not in slam.obj_poses and o1 not in slam.obj_poses and o2 in slam.obj_poses
    assert slam.detections[0][o1]["inliers"].sum() == 2
    res = slam.collect_results()
    assert res[0]["poses"][o0]["T_OtoC"] is None and res[0]["poses"][o2]["score"] == 1 + slam.obj_num_inliers(o2)


def test_too_few_camera_edges_skips_curr_only_optimisation():
    slam, fr = _slam_with_state(np.random.default_rng(3))
    for o in fr["obj_ids"]:
        slam.detections[0][o]["inliers"][:] = False
    assert slam.build_problem(curr_only=True) is None            # < 3 camera edges (:730)
    slam.reset()
    assert slam.build_problem() is None and slam.num_views_processed() == 0


def test_pnp_wrapper_contract_without_gpu():
    # fewer than 4 points never reaches the native call (object_slam.py:31-32)
    assert OS.pnp(np.zeros((3, 3)), np.zeros((3, 2)), np.eye(3)) is None
    uv = np.array([[0.1, -0.2], [0.3, 0.4]])
    K = geo.fix_K_for_bbox_ndc(S.K_YCBV, [10.0, 20.0, 110.0, 220.0])
    np.testing.assert_allclose(geo.normalize_uv(uv, K), (np.c_[uv, np.ones(2)] @ np.linalg.inv(K).T)[:, :2], atol=1e-12)

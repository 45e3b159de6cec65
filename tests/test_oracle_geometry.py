"""Pin the geometry oracle (oracle/pnp_oracle.c, oracle/lm_oracle.c).  CPU only.

P3P/P4P: bit-exact against golden vectors produced by the reference's own p4p.cpp
(tests/golden/make_pnp_golden.py) and, when oracle/_ref is present, against that library live;
the reference's known-answer vector (thirdparty/lambdatwist/test_pnp.py:5-14) to its printed
precision.  RANSAC/refine and the g2o-style LM cannot be pinned by execution (Ceres / Eigen /
CHOLMOD are absent): property tests follow thirdparty/lambdatwist/test_pnp.cpp:68-147 and
thirdparty/g2opy/python/examples/object_slam_demo.py:49-178."""
import os

import numpy as np
import pytest

from oracle import geometry as G
from suo_slam_amd import geometry as geo
from suo_slam_amd import synthetic as S

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pnp_golden.npz"))


def test_p4p_p3p_bit_exact_vs_reference_vectors():
    L = G.lib()
    for i in range(len(GOLD["xs"])):
        x, y = GOLD["xs"][i], GOLD["ys"][i]
        T = G.p4p(x, y, [0, 1, 2, 3])
        assert np.array_equal(T, GOLD["p4p_T"][i]), i
        yh = np.ascontiguousarray(np.c_[y, np.ones(4)])
        r = np.zeros(36)
        t = np.zeros(12)
        v = L.orc_p3p(yh[0].copy(), yh[1].copy(), yh[2].copy(), x[0].copy(), x[1].copy(), x[2].copy(), r, t)
        assert v == GOLD["p3p_valid"][i]
        assert np.array_equal(r.reshape(4, 9)[:v], GOLD["p3p_R"][i][:v], equal_nan=True) and np.array_equal(t.reshape(4, 3)[:v], GOLD["p3p_t"][i][:v], equal_nan=True)


def test_p4p_live_reference_library():
    if G.ref() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(5)
    for _ in range(300):
        x = rng.uniform(-100, 100, (6, 3))
        Q = S.random_rotation(rng)
        X = x @ Q.T + np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(400, 1200)])
        y = X[:, :2] / X[:, 2:3] + rng.normal(0, 1e-3, (6, 2))
        idx = sorted(rng.choice(6, 4, replace=False).tolist())
        assert np.array_equal(G.p4p(x, y, idx), G.ref_p4p(x, y, idx))


def test_known_answer_vector():
    xs, ys, pose = GOLD["kat_xs"], GOLD["kat_ys"], GOLD["kat_pose"]
    assert np.abs(G.p4p(xs, ys, [0, 1, 2, 3]) - pose).max() < 1e-4       # printed at 6 significant digits
    T, best, its = G.pnp(xs, ys, 1e-3, seed=0)
    assert best == 10 and its == 100
    assert np.abs(T - pose).max() < 1e-4
    # refinement must not move an exact consensus solution by more than the data's print precision
    T2, _, _ = G.pnp(xs, ys, 1e-3, seed=0, refine=False)
    assert np.abs(T - T2).max() < 5e-4


def test_get_iterations_law():
    L = G.lib()
    assert L.orc_get_iterations(0.0) == 1000          # no inliers yet -> cap
    assert L.orc_get_iterations(1.0) == 100           # floor
    assert L.orc_get_iterations(0.5) == int(np.ceil(np.log(1e-5) / np.log(1 - 0.45 ** 4)) + 50)
    vals = [L.orc_get_iterations(b / 41) for b in range(42)]
    assert all(a >= b for a, b in zip(vals, vals[1:]))


def test_sample4_distinct_sorted():
    L = G.lib()
    idx = np.zeros(4, np.int32)
    seen = set()
    for it in range(500):
        L.orc_sample4(123, it, 6, idx)
        assert list(idx) == sorted(set(idx.tolist())) and 0 <= idx[0] and idx[3] < 6
        seen.add(tuple(idx))
    assert len(seen) == 15                             # all C(6,4) subsets occur


def _scene(rng, n, sigma_px=0.0, outlier_frac=0.0, f=1000.0):
    """thirdparty/lambdatwist/simulator.h:46-94 flavoured generator: points in front of a random pose."""
    Q = S.random_rotation(rng)
    t = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(4, 8)])
    xs = rng.uniform(-2, 2, (n, 3))
    X = xs @ Q.T + t
    ys = X[:, :2] / X[:, 2:3] + rng.normal(0, sigma_px / f, (n, 2)) if sigma_px > 0 else X[:, :2] / X[:, 2:3]
    out = rng.random(n) < outlier_frac
    ys[out] = rng.uniform(-0.5, 0.5, (int(out.sum()), 2))
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = Q, t
    return xs, ys, T


def test_pnp_noise_free_exact_recovery():
    rng = np.random.default_rng(1)
    for n in (4, 5, 8, 22, 41):
        xs, ys, T = _scene(rng, n)
        Te, best, _ = G.pnp(xs, ys, 1e-3, seed=n)
        assert best == n
        assert np.linalg.norm(Te[:3, :3] - T[:3, :3]) < 1e-6 and np.linalg.norm(Te[:3, 3] - T[:3, 3]) < 1e-6


def test_pnp_statistics_like_reference_benchmark():
    """test_pnp.cpp:68-147: N=250 points, 50 % outliers, sigma in {0,.25,.5,1} px; failure (angle+|t|
    error > 0.05) rate must stay below 5 %.  200 problems per level here instead of 1000."""
    rng = np.random.default_rng(2)
    for sigma in (0.0, 0.25, 0.5, 1.0):
        fails = 0
        for k in range(200):
            xs, ys, T = _scene(rng, 250, sigma_px=sigma, outlier_frac=0.5)
            Te, _, _ = G.pnp(xs, ys, 1e-3 * max(1.0, 3 * sigma), seed=k)
            dR = Te[:3, :3] @ T[:3, :3].T
            ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
            fails += (ang + np.linalg.norm(Te[:3, 3] - T[:3, 3])) > 0.05
        assert fails / 200 < 0.05, (sigma, fails)


def test_pnp_statistical_contract_of_the_reference_benchmark_exact_generator():
    """thirdparty/lambdatwist/test_pnp.cpp:68-147 with ITS generator (simulator.h:46-94, tests/pnp_simulator.py) and ITS
    settings: default PnpParams (threshold 0.001), sigma in {0, .25, .5, 1} px, 250 points, 50 % outlier draws; a pose
    counts as failed when |angle| + |t| of P * Pcw^-1 exceeds 0.05; every level must stay below 5 % (500 problems per
    level here; the HIP kernel runs the full 1000 in tests/test_gpu_geometry.py)."""
    from tests import pnp_simulator as PS
    rng = np.random.default_rng(12)
    for sigma in (0.0, 0.25, 0.5, 1.0):
        fails, errs = 0, []
        for k in range(500):
            xs, yns, Pcw = PS.point_cloud_with_noisy_measurements(rng, 250, sigma, 0.5)
            T, best, its = G.pnp(xs, yns, 1e-3, seed=k)
            assert np.isfinite(T).all() and 100 <= its <= 1050
            e = PS.pose_error(T, Pcw)
            errs.append(e)
            fails += e > 0.05
        assert fails / 500 < 0.05, (sigma, fails)
        assert np.median(errs) < 0.03


def test_pnp_total_failure_returns_identity():
    rng = np.random.default_rng(3)
    xs = rng.uniform(-1, 1, (6, 3))
    ys = rng.uniform(-0.5, 0.5, (6, 2))
    xs[:] = xs[0]                                        # all points coincide: degenerate
    Te, best, _ = G.pnp(xs, ys, 1e-3, seed=0)
    assert best == 0 and np.array_equal(Te, np.eye(4))


# ---- LM ---------------------------------------------------------------------------------------
def test_fix_K_for_bbox_ndc_maps_box_to_unit_square():
    K = S.K_YCBV
    bbox = [100.0, 50.0, 300.0, 290.0]
    Kb = geo.fix_K_for_bbox_ndc(K, bbox)
    Kinv = np.linalg.inv(K)
    for (px, py), ndc in (((100, 50), (-1, 1)), ((300, 290), (1, -1)), ((200, 170), (0, 0))):
        ray = Kinv @ np.array([px, py, 1.0])
        uvw = Kb @ ray
        np.testing.assert_allclose(uvw[:2] / uvw[2], ndc, atol=1e-12)
    assert np.allclose(Kb[[0, 1, 2, 2, 2], [1, 0, 0, 1, 2]], [0, 0, 0, 0, 1])


def test_edge_jacobians_match_finite_differences():
    L = G.lib()
    rng = np.random.default_rng(4)
    for _ in range(20):
        cam = np.ascontiguousarray(np.eye(4)[:3])
        L.orc_pose_oplus(cam.reshape(-1), np.r_[rng.normal(0, 0.2, 3), rng.uniform(-50, 50, 3)])
        obj = np.c_[S.random_rotation(rng), np.array([rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(600, 1200)])]
        k = np.array([rng.uniform(5, 15), -rng.uniform(5, 15), rng.uniform(-1, 1), rng.uniform(-1, 1)])
        p = rng.uniform(-80, 80, 3)
        uv = rng.uniform(-1, 1, 2)
        Jo, Jc = np.zeros(12), np.zeros(12)
        L.orc_edge_jacobians(np.ascontiguousarray(cam.ravel()), np.ascontiguousarray(obj.ravel()), k, p, Jo, Jc)

        def err(c, o):
            e = np.zeros(2)
            L.orc_edge_error(c, o, k, p, uv, e)
            return e
        for which, J in (("cam", Jc.reshape(2, 6)), ("obj", Jo.reshape(2, 6))):
            for i in range(6):
                u = np.zeros(6)
                u[i] = 1e-4      # above SE3Quat::exp's small-angle branch (theta < 1e-5 uses I + W + W^2)
                a = [cam.ravel().copy(), np.ascontiguousarray(obj.ravel()).copy()]
                b = [a[0].copy(), a[1].copy()]
                L.orc_pose_oplus(a[0 if which == "cam" else 1], u)
                L.orc_pose_oplus(b[0 if which == "cam" else 1], -u)
                fd = (err(a[0], a[1]) - err(b[0], b[1])) / 2e-4
                assert np.abs(fd - J[:, i]).max() <= 1e-6 * max(1.0, np.abs(J).max())


def _perturb(T, rng, rot=0.05, trans=20.0):
    L = G.lib()
    t = np.ascontiguousarray(T[:3, :].ravel().copy())
    L.orc_pose_oplus(t, np.r_[rng.normal(0, rot, 3), rng.normal(0, trans, 3)])
    return t.reshape(3, 4)


def test_single_view_noise_free_recovery_and_block_independence():
    rng = np.random.default_rng(6)
    fr = S.make_frame(rng, 8, noise=0.0, with_image=False)
    init = np.stack([_perturb(T, rng) for T in fr["T_OtoC"]])
    P = S.frame_to_ba_problem(fr, init)
    cam, obj, inl, chi2, stats = G.optimize(**{k: P[k] for k in P}, init_with_outliers=True)
    assert inl.all() and stats[0] == 4
    for o in range(8):
        assert np.linalg.norm(obj[o] - fr["T_OtoC"][o][:3]) < 1e-3      # uv is float32-quantised; t in mm at ~1 m depth
    assert np.array_equal(cam[0], np.eye(4)[:3])          # fixed camera untouched


def test_outliers_are_gated_by_chi2_rounds():
    rng = np.random.default_rng(7)
    fr = S.make_frame(rng, 6, noise=0.002, outlier_frac=0.15, with_image=False)
    init = np.stack([_perturb(T, rng, 1e-4, 0.05) for T in fr["T_OtoC"]])   # PnP-quality initial guess
    P = S.frame_to_ba_problem(fr, init)
    cam, obj, inl, chi2, stats = G.optimize(**{k: P[k] for k in P})
    assert 0.5 < inl.mean() < 1.0 and np.all((chi2 <= G.CHI2_THR) == inl.astype(bool))
    for o in range(6):
        assert np.linalg.norm(obj[o][:, 3] - fr["T_OtoC"][o][:3, 3]) < 25.0


def test_object_slam_demo_scenario_rmse_drops():
    """object_slam_demo.py:49-178: 15 cameras (first 2 fixed), 6 objects x 8 points, f=320, c=(320,240),
    pixel noise; object-pose RMSE must drop after 10 LM iterations with binary edges."""
    rng = np.random.default_rng(8)
    n_cam, n_obj = 15, 6
    k = np.array([320.0, 320.0, 320.0, 240.0])
    cam_T = np.zeros((n_cam, 3, 4))
    for i in range(n_cam):
        cam_T[i, :, :3] = np.eye(3)
        cam_T[i, :, 3] = [-(i * 0.04 - 0.3), 0, 0]
    obj_gt = np.zeros((n_obj, 3, 4))
    pts = rng.uniform(-0.15, 0.15, (n_obj, 8, 3))
    for o in range(n_obj):
        obj_gt[o, :, :3] = S.random_rotation(rng)
        obj_gt[o, :, 3] = [rng.uniform(-1, 1), rng.uniform(-0.6, 0.6), rng.uniform(3, 5)]
    e_cam, e_obj, e_p, e_uv = [], [], [], []
    for c in range(n_cam):
        for o in range(n_obj):
            pw = pts[o] @ obj_gt[o, :, :3].T + obj_gt[o, :, 3]
            pc = pw @ cam_T[c, :, :3].T + cam_T[c, :, 3]
            uv = np.c_[k[0] * pc[:, 0] / pc[:, 2] + k[2], k[1] * pc[:, 1] / pc[:, 2] + k[3]] + rng.normal(0, 1.0, (8, 2))
            for j in range(8):
                e_cam.append(c); e_obj.append(o); e_p.append(pts[o, j]); e_uv.append(uv[j])
    obj_init = np.stack([_perturb(np.vstack([T, [0, 0, 0, 1]]), rng, 0.03, 0.05) for T in obj_gt])
    cam_fixed = np.zeros(n_cam, np.uint8)
    cam_fixed[:2] = 1
    cam_init = cam_T.copy()
    for c in range(2, n_cam):
        cam_init[c] = _perturb(np.vstack([cam_T[c], [0, 0, 0, 1]]), rng, 0.005, 0.01)
    E = len(e_cam)
    rmse0 = np.sqrt(np.mean([(obj_init[o][:, 3] - obj_gt[o][:, 3]) ** 2 for o in range(n_obj)]))
    cam, obj, inl, chi2, stats = G.optimize(cam_init, cam_fixed, obj_init, np.zeros(n_obj, np.uint8), e_cam, e_obj,
                                            np.tile(k, (E, 1)), np.array(e_p), np.array(e_uv), np.tile([1.0, 0, 1.0], (E, 1)),
                                            np.ones(E, np.uint8), its=(10,), init_with_outliers=True)
    rmse1 = np.sqrt(np.mean([(obj[o][:, 3] - obj_gt[o][:, 3]) ** 2 for o in range(n_obj)]))
    assert rmse1 < 0.5 * rmse0


def test_rounds_abort_below_four_edges():
    rng = np.random.default_rng(9)
    fr = S.make_frame(rng, 1, noise=0.0, with_image=False)
    P = S.frame_to_ba_problem(fr, fr["T_OtoC"])
    keep = slice(0, 3)
    cam, obj, inl, chi2, stats = G.optimize(P["cam_T"], P["cam_fixed"], P["obj_T"], P["obj_fixed"], P["edge_cam"][keep],
                                            P["edge_obj"][keep], P["edge_camk"][keep], P["edge_p"][keep], P["edge_uv"][keep],
                                            P["edge_info"][keep], P["edge_inlier"][keep])
    assert stats[0] == 0 and np.allclose(obj[0], fr["T_OtoC"][0][:3])

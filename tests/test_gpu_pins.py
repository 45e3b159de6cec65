"""Independent checks of the HIP geometry kernels that do NOT go through the oracle's transcription of the same formulas
("what's weak" 1): finite differences of the error the KERNEL computes against the Jacobians the KERNEL
linearises; scipy.optimize.least_squares as an unrelated optimiser on the final cost (PnP refine and LM / BA); the g2o demo
scenario of thirdparty/g2opy/python/examples/object_slam_demo.py:49-178 on suo_optimize."""
import ctypes as C

import numpy as np
import pytest
from scipy.optimize import least_squares

pytestmark = pytest.mark.gpu

from suo_slam_amd import synthetic as S  # noqa: E402
from tests.ba_numpy_phases import _exp  # noqa: E402   (numpy SE3 exp map, se3quat.h:220-254 restated for the CPU gloo tests)

KEYS = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")


def _kernel_linearisation(P):
    """(jac [E,29], err [E,2]) as the LM kernels compute them at the poses of P (suo_debug_ba_jacobians)."""
    from suo_slam_amd import _lib, ba
    lib = _lib.lib()
    prob = ba.Problem(*[P[k] for k in KEYS])
    s = _lib.BaProblem()
    prob._fill(s)
    h = C.c_void_p()
    _lib.check(lib.suo_ba_ctx_create(C.byref(s), C.byref(h)), "ctx")
    try:
        good = np.zeros(1)
        _lib.check(lib.suo_ba_classify(h, 1, good.ctypes.data), "classify")          # keep_all: every edge stays active
        out = np.zeros(2 + 27 * len(prob.obj_T))
        _lib.check(lib.suo_ba_linearize(h, 0, out.ctypes.data), "linearize")
        E = len(prob.edge_cam)
        jac, err = np.zeros((E, 29)), np.zeros((E, 2))
        _lib.check(lib.suo_debug_ba_jacobians(h, E, jac.ctypes.data, err.ctypes.data), "jac")
    finally:
        lib.suo_ba_ctx_destroy(h)
    return jac, err


def _oplus(T34, u):
    return (_exp(u) @ np.vstack([T34, [0, 0, 0, 1]]))[:3]


def test_kernel_jacobians_match_finite_differences_of_the_kernel_error():
    """EdgeSE3ProjectFromObject::linearizeOplus (types_object_slam.cpp:70-123) as implemented in csrc/lm_device.h, checked
    against central differences of the error the same kernel computes at poses perturbed by exp(u) T (the update of
    types_six_dof_expmap.h:100-103).  Step 1e-4 stays above SE3Quat::exp's small-angle branch (se3quat.h:234-240)."""
    rng = np.random.default_rng(31)
    P = S.make_pose_graph(rng, 4, 3, kp_per_obj=6, miss=0.0, outlier_frac=0.0)
    P["cam_fixed"][:] = 0
    jac, err0 = _kernel_linearisation(P)
    E = len(P["edge_cam"])
    assert np.abs(jac[:, :24]).max() > 0
    h = 1e-4
    worst = 0.0
    for kind, key, col0, owner in (("cam", "cam_T", 0, P["edge_cam"]), ("obj", "obj_T", 12, P["edge_obj"])):
        for v in range(len(P[key])):
            for i in range(6):
                u = np.zeros(6)
                u[i] = h
                Pp, Pm = dict(P), dict(P)
                Pp[key], Pm[key] = P[key].copy(), P[key].copy()
                Pp[key][v], Pm[key][v] = _oplus(P[key][v], u), _oplus(P[key][v], -u)
                fd = (_kernel_linearisation(Pp)[1] - _kernel_linearisation(Pm)[1]) / (2 * h)
                for e in range(E):
                    J = jac[e, col0:col0 + 12].reshape(2, 6)[:, i]
                    if owner[e] == v:
                        worst = max(worst, np.abs(fd[e] - J).max() / max(1.0, np.abs(jac[e, col0:col0 + 12]).max()))
                    else:
                        assert np.abs(fd[e]).max() == 0.0              # an edge depends on its own two vertices only
    assert worst < 1e-6, worst
    # information / gradient factors: w = 1 without the robust kernel
    info = P["edge_info"]
    np.testing.assert_allclose(jac[:, 24:27], info, rtol=1e-15)
    g = -np.stack([info[:, 0] * err0[:, 0] + info[:, 1] * err0[:, 1], info[:, 1] * err0[:, 0] + info[:, 2] * err0[:, 1]], 1)
    np.testing.assert_allclose(jac[:, 27:29], g, rtol=1e-12, atol=1e-12)


def _residuals(delta, P, cam_T, obj_T, sel, free_cams, free_objs):
    """Whitened reprojection errors of the selected edges at poses exp(delta) T: plain numpy, written from the edge definition
    e = uv - pi(T_cw T_wo p) (types_object_slam.cpp:45-60), chi2 = e^T info e."""
    cam = {c: _oplus(cam_T[c], delta[6 * i:6 * i + 6]) for i, c in enumerate(free_cams)}
    obj = {o: _oplus(obj_T[o], delta[6 * (len(free_cams) + j):6 * (len(free_cams) + j) + 6]) for j, o in enumerate(free_objs)}
    out = []
    for e in sel:
        c, o = int(P["edge_cam"][e]), int(P["edge_obj"][e])
        Tc, To = cam.get(c, cam_T[c]), obj.get(o, obj_T[o])
        pw = To[:, :3] @ P["edge_p"][e] + To[:, 3]
        pc = Tc[:, :3] @ pw + Tc[:, 3]
        k = P["edge_camk"][e]
        r = P["edge_uv"][e] - np.array([k[0] * pc[0] / pc[2] + k[2], k[1] * pc[1] / pc[2] + k[3]])
        i = P["edge_info"][e]
        L = np.linalg.cholesky(np.array([[i[0], i[1]], [i[1], i[2]]]))
        out.append(L.T @ r)
    return np.concatenate(out)


@pytest.mark.parametrize("scene", ["single_view", "global"])
def test_lm_result_is_a_stationary_point_for_an_unrelated_optimiser(scene):
    """The last robust round runs WITHOUT the Huber kernel on the chi2-inliers (object_slam.py:877-896): its result must
    minimise sum e^T info e over those edges.  scipy's trust-region least squares, started at the HIP result in the
    tangent space, must stay there; the cost must not drop and the gradient must vanish."""
    from suo_slam_amd import ba
    rng = np.random.default_rng(41)
    if scene == "single_view":
        fr = S.make_frame(rng, 5, noise=0.004, outlier_frac=0.1, with_image=False)
        init = np.stack([S._perturb_pose(T[:3], rng, 2e-4, 0.1) for T in fr["T_OtoC"]])
        P = S.frame_to_ba_problem(fr, init)
    else:
        P = S.make_pose_graph(rng, 8, 4, kp_per_obj=8)
    got3 = ba.optimize(*[P[k] for k in KEYS], its=(10, 10, 40))
    cam_T, obj_T, inl, chi2, stats = ba.optimize(*[P[k] for k in KEYS], its=(10, 10, 40, 40))
    assert stats[0] == 4 and np.array_equal(inl, got3[2])          # the last round optimised exactly the final inlier set
    sel = np.nonzero(inl)[0]
    free_cams = [c for c in range(len(cam_T)) if not P["cam_fixed"][c]]
    free_objs = [o for o in range(len(obj_T)) if not P["obj_fixed"][o]]
    n = 6 * (len(free_cams) + len(free_objs))
    f = lambda d: _residuals(d, P, cam_T, obj_T, sel, free_cams, free_objs)  # noqa: E731
    r0 = f(np.zeros(n))
    assert abs(r0 @ r0 - chi2[sel].sum()) < 1e-6 * max(1.0, r0 @ r0)            # the kernel's chi2 == this cost
    sol = least_squares(f, np.zeros(n), method="trf", x_scale=np.tile([1e-3] * 3 + [1.0] * 3, n // 6), xtol=1e-15, ftol=1e-15, gtol=1e-15)
    rot = np.abs(sol.x.reshape(-1, 6)[:, :3]).max()
    trans = np.abs(sol.x.reshape(-1, 6)[:, 3:]).max()
    drop = (r0 @ r0 - 2 * sol.cost) / (r0 @ r0)
    print(f"{scene}: scipy moved the HIP optimum by {rot:.2e} rad / {trans:.2e} mm, relative cost drop {drop:.2e}")
    assert rot < 1e-6 and trans < 1e-3 and drop < 1e-8


def test_pnp_refine_reaches_the_optimum_an_unrelated_optimiser_finds():
    """PNP::refine minimises sum |x/z - y|^2 over the inliers of the RANSAC pose (pnp_ransac.cpp:96-156, 240-326) with at most
    5 (+3) LM steps at tolerance 1e-6 (1e-8): a bounded effort, not a converged optimum.  Measured against scipy's converged
    optimum on the SAME inlier set (an optimiser that shares no text with csrc/pnp.hip / oracle/pnp_oracle.c): the refine must
    never increase the cost and must close >= 97 % of the gap between the minimal-solver pose and the optimum (on the inlier set
    of whichever pass ran last -- pass 2 selects at the intermediate pose, which is not returned --, >= 85 % on the other); noise-free data must give the exact pose."""
    from suo_slam_amd import lambdatwist
    from tests import pnp_simulator as PS
    rng = np.random.default_rng(51)
    for sigma, n, out in ((0.25, 40, 0.3), (0.5, 120, 0.3), (0.3, 250, 0.5), (0.1, 30, 0.2), (0.0, 12, 0.3)):
        for rep in range(5):
            xs, yns, Pcw = PS.point_cloud_with_noisy_measurements(rng, n, sigma, out)
            T0 = lambdatwist.pnp_batch([xs], [yns], 1e-3, seed=5 + rep, refine=False)[0][0]
            T1 = lambdatwist.pnp_batch([xs], [yns], 1e-3, seed=5 + rep, refine=True)[0][0]
            if sigma == 0.0:
                assert np.abs(T1 - Pcw).max() < 1e-9                      # noise-free: the exact pose
                continue
            closed = []
            for Tsel in (T0, T1):         # pass 1 selects its inliers at the RANSAC pose (:254-261), pass 2 (only when >= 5 % of the
                pc = xs @ Tsel[:3, :3].T + Tsel[:3, 3]                    # set changed, :303) at the pose pass 1 reached (:292-301)
                inl = (pc[:, 2] >= 0) & (((pc[:, :2] / pc[:, 2:3] - yns) ** 2).sum(1) <= 1e-6)
                assert inl.sum() >= 6

                def cost(T, d=np.zeros(6)):
                    Td = _oplus(T[:3], d)
                    p = xs[inl] @ Td[:, :3].T + Td[:, 3]
                    return (p[:, :2] / p[:, 2:3] - yns[inl]).ravel()
                sol = least_squares(lambda d: cost(T1, d), np.zeros(6), method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)
                f0, f1, fopt = float(cost(T0) @ cost(T0)), float(cost(T1) @ cost(T1)), 2 * sol.cost
                assert f1 <= f0 * (1 + 1e-12)
                closed.append((f0 - f1) / max(f0 - fopt, 1e-300))
            print(f"  sigma {sigma} n {n} #{rep}: gap closed {closed[0]:.4f} on the RANSAC pose's inliers, {closed[1]:.4f} on the refined pose's")
            assert max(closed) >= 0.97 and min(closed) >= 0.85


def test_object_slam_demo_scenario_on_the_hip_solver():
    """thirdparty/g2opy/python/examples/object_slam_demo.py:49-178: 15 cameras on a line of which the first TWO are fixed,
    6 objects x 8 points, focal 320, principal point (320, 240), unit information, 1 px noise, 10 LM iterations, no robust
    kernel classes (all edges level 0).  The demo's claim -- object-pose RMSE drops -- on suo_optimize, and HIP == oracle."""
    from oracle import geometry as G
    from suo_slam_amd import ba
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
            e_cam += [c] * 8
            e_obj += [o] * 8
            e_p.append(pts[o])
            e_uv.append(uv)
    E = len(e_cam)
    obj_init = np.stack([S._perturb_pose(T, rng, 0.03, 0.05) for T in obj_gt])
    cam_init = cam_T.copy()
    for c in range(2, n_cam):
        cam_init[c] = S._perturb_pose(cam_T[c], rng, 0.005, 0.01)
    cam_fixed = np.zeros(n_cam, np.uint8)
    cam_fixed[:2] = 1
    args = (cam_init, cam_fixed, obj_init, np.zeros(n_obj, np.uint8), np.array(e_cam, np.int32), np.array(e_obj, np.int32), np.tile(k, (E, 1)),
            np.concatenate(e_p), np.concatenate(e_uv), np.tile([1.0, 0, 1.0], (E, 1)), np.ones(E, np.uint8))
    got = ba.optimize(*args, its=(10,), init_with_outliers=True)
    ref = G.optimize(*args, its=(10,), init_with_outliers=True)
    rmse = lambda T: np.sqrt(np.mean([(T[o][:, 3] - obj_gt[o][:, 3]) ** 2 for o in range(n_obj)]))  # noqa: E731
    assert rmse(got[1]) < 0.5 * rmse(obj_init)
    assert np.array_equal(got[0][:2], cam_init[:2])                          # both fixed cameras untouched
    for a, b in zip(list(got[0]) + list(got[1]), list(ref[0]) + list(ref[1])):
        assert np.linalg.norm(a[:, :3] - b[:, :3]) < 1e-6 and np.linalg.norm(a[:, 3] - b[:, 3]) < 1e-6 * max(1.0, np.linalg.norm(b[:, 3]))

"""GPU parity tests of the geometry path (HIP PnP + LM/BA through the C ABI) against the C oracle on
identical seeded inputs.  fp64 on both sides; tolerances written per test."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import geometry as G  # noqa: E402
from suo_slam_amd import geometry as geo  # noqa: E402
from suo_slam_amd import synthetic as S  # noqa: E402

SEED_STRIDE = 0x9E3779B97F4A7C15
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pnp_golden.npz"))


@pytest.fixture(scope="module")
def lt():
    from suo_slam_amd import _lib, lambdatwist
    _lib.require_gpu()
    return lambdatwist


@pytest.fixture(scope="module")
def ba():
    from suo_slam_amd import _lib, ba
    _lib.require_gpu()
    return ba


def _pose_close(A, B, rtol_R=1e-9, tol_t=1e-7):
    return np.linalg.norm(A[:3, :3] - B[:3, :3]) < rtol_R and np.linalg.norm(A[:3, 3] - B[:3, 3]) < tol_t * max(1.0, np.linalg.norm(B[:3, 3]))


def test_pnp_known_answer_vector(lt):
    T = lt.pnp(GOLD["kat_xs"], GOLD["kat_ys"])
    assert T.shape == (4, 4) and np.abs(T - GOLD["kat_pose"]).max() < 1e-4
    To, best, its = G.pnp(GOLD["kat_xs"], GOLD["kat_ys"], 1e-3, seed=0)
    assert np.abs(T - To).max() < 1e-10


def test_pnp_four_point_problems_match_reference_p4p_vectors(lt):
    """With exactly 4 points every hypothesis is p4p(0,1,2,3): the no-refine RANSAC result must equal the
    reference's own p4p output (golden vectors from its compiled p4p.cpp) whenever it has >0 inliers."""
    xs = [GOLD["xs"][i] for i in range(100)]
    ys = [GOLD["ys"][i] for i in range(100)]
    T, status, info = lt.pnp_batch(xs, ys, 1e-3, seed=7, refine=False, return_info=True)
    n_checked = 0
    for i in range(100):
        if info["best_inliers"][i] > 0:
            assert np.abs(T[i] - GOLD["p4p_T"][i]).max() < 1e-12, i
            n_checked += 1
        else:
            assert status[i] == 1 and np.array_equal(T[i], np.eye(4))
    assert n_checked > 60


@pytest.mark.parametrize("noise,outliers", [(0.0, 0.0), (0.002, 0.0), (0.002, 0.2), (0.01, 0.1)])
def test_pnp_batch_matches_oracle(lt, noise, outliers):
    rng = np.random.default_rng(int(noise * 1e4) + int(outliers * 100))
    xs, ys = [], []
    for f in range(6):
        fr = S.make_frame(rng, 8, noise=noise, outlier_frac=outliers, with_image=False)
        for o in range(8):
            m = fr["model_kps_masks"][o]
            xs.append(fr["model_kps"][o][m].astype(np.float64))
            ys.append(geo.normalize_uv(fr["uv"][o][m].astype(np.float64), fr["K_bbox"][o]))
    seed = 1234
    T, status, info = lt.pnp_batch(xs, ys, 1e-3, seed=seed, return_info=True)
    for o in range(len(xs)):
        To, best, its = G.pnp(xs[o], ys[o], 1e-3, seed=(seed + o * SEED_STRIDE) % 2**64)
        assert info["best_inliers"][o] == best and info["iterations"][o] == its, o     # integer outputs: exact
        assert _pose_close(T[o], To), (o, np.abs(T[o] - To).max())
        assert status[o] == int(np.array_equal(To, np.eye(4)))


def test_pnp_edge_cases(lt):
    rng = np.random.default_rng(5)
    fr = S.make_frame(rng, 3, noise=0.0, with_image=False)
    xs, ys = [], []
    for o in range(3):
        m = fr["model_kps_masks"][o]
        xs.append(fr["model_kps"][o][m].astype(np.float64))
        ys.append(geo.normalize_uv(fr["uv"][o][m].astype(np.float64), fr["K_bbox"][o]))
    xs.append(xs[0][:3]); ys.append(ys[0][:3])                 # < 4 points -> identity (object_slam.py:31)
    xs.append(np.zeros((0, 3))); ys.append(np.zeros((0, 2)))   # empty
    xs.append(np.tile(xs[0][:1], (6, 1))); ys.append(ys[0][:6])  # degenerate: coincident points
    T, status = lt.pnp_batch(xs, ys)
    assert list(status) == [0, 0, 0, 1, 1, 1]
    for o in (3, 4, 5):
        assert np.array_equal(T[o], np.eye(4))
    for o in range(3):
        assert np.linalg.norm(T[o][:3, 3] - fr["T_OtoC"][o][:3, 3]) < 1e-2
    # large N (strided lanes): 250 points, half outliers (thirdparty/lambdatwist/test_pnp.cpp:68-147)
    Q = S.random_rotation(rng)
    t = np.array([0.3, -0.2, 6.0])
    X = rng.uniform(-2, 2, (250, 3))
    P = X @ Q.T + t
    y = P[:, :2] / P[:, 2:3]
    y[::2] = rng.uniform(-0.5, 0.5, (125, 2))
    T1, st = lt.pnp_batch([X], [y], 1e-3, seed=3)
    To, best, its = G.pnp(X, y, 1e-3, seed=3)
    assert _pose_close(T1[0], To) and np.linalg.norm(T1[0][:3, 3] - t) < 1e-6


def test_pnp_statistical_contract_of_the_reference_benchmark(lt):
    """thirdparty/lambdatwist/test_pnp.cpp:68-147 on suo_pnp_batch: 4 noise levels x 1000 problems x 250 points with 50 %
    outlier draws (the reference's generator, tests/pnp_simulator.py), default threshold 0.001, ONE launch per level
    (1000 wavefronts); < 5 % of the poses may be off by more than 0.05 (angle + translation).  A sample of each level is
    also compared with the oracle pose by pose."""
    import time
    from tests import pnp_simulator as PS
    rng = np.random.default_rng(21)
    for sigma in (0.0, 0.25, 0.5, 1.0):
        data = [PS.point_cloud_with_noisy_measurements(rng, 250, sigma, 0.5) for _ in range(1000)]
        t0 = time.perf_counter()
        T, status, info = lt.pnp_batch([d[0] for d in data], [d[1] for d in data], 1e-3, seed=77, return_info=True)
        dt = time.perf_counter() - t0
        assert np.isfinite(T).all()
        errs = np.array([PS.pose_error(T[i], data[i][2]) for i in range(1000)])
        fails = int((errs > 0.05).sum())
        print(f"sigma {sigma}: {fails} bad poses of 1000, median err {np.median(errs):.2e}, median iterations "
              f"{int(np.median(info['iterations']))}, {1e3 * dt:.1f} ms for the launch ({dt * 1e3:.3f} us per problem)")
        assert fails / 1000 < 0.05, (sigma, fails)
        for i in range(0, 1000, 97):
            To, best, its = G.pnp(data[i][0], data[i][1], 1e-3, seed=(77 + i * SEED_STRIDE) % 2 ** 64)
            assert _pose_close(T[i], To) and info["best_inliers"][i] == best and info["iterations"][i] == its


def _perturb(T, rng, rot, trans):
    t = np.ascontiguousarray(np.asarray(T)[:3, :].ravel().copy())
    G.lib().orc_pose_oplus(t, np.r_[rng.normal(0, rot, 3), rng.normal(0, trans, 3)])
    return t.reshape(3, 4)


def _compare_ba(ba, P, **kw):
    args = [P[k] for k in ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv",
                           "edge_info", "edge_inlier")]
    ref = G.optimize(*args, **kw)
    got = ba.optimize(*args, **kw)
    assert np.array_equal(got[2], ref[2]), "inlier flags differ"            # boolean output: bit-exact
    # rounds and num_good are exact; LM iteration / trial counts are NOT a parity property: once converged the
    # gain ratio is rounding noise, so the number of rejected trials differs between any two summation orders
    assert got[4][0] == ref[4][0] and got[4][3] == ref[4][3], (got[4], ref[4])
    # pose tolerance (stated): |dR|_F < 1e-6 and |dt| < 1e-6 * |t| (~1e-3 mm at 1 m).  Both sides stop on the same
    # iteration caps, not at convergence, so the last accepted steps may differ by LM-tolerance-sized amounts.
    for a, b in zip(got[0], ref[0]):
        assert _pose_close(a, b, 1e-6, 1e-6), np.abs(a - b).max()
    for a, b in zip(got[1], ref[1]):
        assert _pose_close(a, b, 1e-6, 1e-6), np.abs(a - b).max()
    np.testing.assert_allclose(got[3], ref[3], rtol=1e-4, atol=1e-7)
    return got, ref


@pytest.mark.parametrize("noise,outliers,n_obj", [(0.0, 0.0, 8), (0.002, 0.15, 8), (0.01, 0.1, 16), (0.002, 0.0, 1)])
def test_single_view_refinement_matches_oracle(ba, noise, outliers, n_obj):
    rng = np.random.default_rng(100 + n_obj + int(noise * 1e4))
    fr = S.make_frame(rng, n_obj, noise=noise, outlier_frac=outliers, with_image=False)
    init = np.stack([_perturb(T, rng, 2e-4, 0.1) for T in fr["T_OtoC"]])
    P = S.frame_to_ba_problem(fr, init)
    got, ref = _compare_ba(ba, P)
    assert got[4][0] == 4                     # all four robust rounds ran
    assert np.array_equal(got[0][0], np.eye(4)[:3])


def test_curr_only_camera_tracking_matches_oracle(ba):
    """SLAM curr_only mode: objects fixed (unary edges), one free camera, its=[10]*4 (object_slam.py:846)."""
    rng = np.random.default_rng(11)
    fr = S.make_frame(rng, 6, noise=0.003, outlier_frac=0.1, with_image=False)
    P = S.frame_to_ba_problem(fr, fr["T_OtoC"])
    P["cam_T"] = _perturb(np.eye(4), rng, 1e-4, 0.05)[None]
    P["cam_fixed"] = np.array([0], np.uint8)
    P["obj_fixed"] = np.ones(6, np.uint8)
    for iwo in (False, True):
        _compare_ba(ba, P, its=(10, 10, 10, 10), init_with_outliers=iwo)


@pytest.mark.parametrize("n_obj,noise,outliers", [(1, 0.002, 0.0), (8, 0.004, 0.15), (16, 0.01, 0.1)])
def test_camera_tracking_one_wave_kernel(ba, n_obj, noise, outliers):
    """One free camera and only fixed objects takes the one-wave kernel (csrc/lm_cam.hip): against the oracle, as a batch of
    several views at once, and with further (fixed) cameras in the same problem."""
    rng = np.random.default_rng(300 + n_obj)
    keys = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")
    probs, singles = [], []
    for f in range(4):
        fr = S.make_frame(rng, n_obj, noise=noise, outlier_frac=outliers, with_image=False)
        P = S.frame_to_ba_problem(fr, fr["T_OtoC"])
        P["cam_T"] = _perturb(np.eye(4), rng, 2e-4, 0.1)[None]
        P["cam_fixed"] = np.array([0], np.uint8)
        P["obj_fixed"] = np.ones(n_obj, np.uint8)
        got, ref = _compare_ba(ba, P, its=(10, 10, 10, 10))
        probs.append(ba.Problem(*[P[k] for k in keys], its=(10, 10, 10, 10)))
        singles.append(got)
    ba.optimize_batch(probs)                                        # one workgroup (= one wave) per problem
    for p, s in zip(probs, singles):
        assert np.array_equal(p.cam_T.reshape(-1, 3, 4), s[0]) and np.array_equal(p.inlier, s[2])
    # a second, fixed camera with its own edges: they are inactive (camera and objects fixed) and must stay untouched
    P2 = {k: np.array(P[k]) for k in keys}
    P2["cam_T"] = np.concatenate([P["cam_T"], P["cam_T"]])
    P2["cam_fixed"] = np.array([1, 0], np.uint8)
    P2["edge_cam"] = np.concatenate([np.zeros_like(P["edge_cam"]), np.ones_like(P["edge_cam"])])
    for k in ("edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier"):
        P2[k] = np.concatenate([P[k], P[k]])
    got2, ref2 = _compare_ba(ba, P2, its=(10, 10, 10, 10))
    np.testing.assert_allclose(got2[0][0], P["cam_T"][0], rtol=0, atol=1e-12)      # (matrix -> quaternion -> matrix round trip)


def test_both_camera_tracking_kernels_against_the_oracle_and_each_other():
    """SUO_LM_CAM2 (read once per process) selects between lm_cam2_kernel (csrc/lm_cam2.hip: the default, one wave per problem, pose in registers) and
    lm_cam_kernel (csrc/lm_cam.hip: graphs with further fixed cameras or > 1024 edges, and SUO_LM_CAM2=0).  The same tracking problems through BOTH,
    each in its own child process: either agrees with the C oracle to the LM tolerance and flags the same inliers, and the two agree with each other."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r)\n"
            "from suo_slam_amd import ba, synthetic as S\nfrom oracle import geometry as G\nfrom tests.test_gpu_geometry import _perturb\n"
            "rng = np.random.default_rng(77); keys = ('cam_T','cam_fixed','obj_T','obj_fixed','edge_cam','edge_obj','edge_camk','edge_p','edge_uv','edge_info','edge_inlier')\n"
            "out = []\n"
            "for n_obj, noise, outl in ((1, 0.002, 0.0), (8, 0.004, 0.15), (16, 0.01, 0.1)):\n"
            "    fr = S.make_frame(rng, n_obj, noise=noise, outlier_frac=outl, with_image=False)\n"
            "    P = S.frame_to_ba_problem(fr, fr['T_OtoC']); P['cam_T'] = _perturb(np.eye(4), rng, 2e-4, 0.1)[None]\n"
            "    P['cam_fixed'] = np.array([0], np.uint8); P['obj_fixed'] = np.ones(n_obj, np.uint8)\n"
            "    a = [P[k] for k in keys]\n"
            "    got, ref = ba.optimize(*a, its=(10, 10, 10, 10)), G.optimize(*a, its=(10, 10, 10, 10))\n"
            "    out.append({'cam': got[0].ravel().tolist(), 'inl': got[2].tolist(), 'd_oracle': float(np.abs(got[0] - ref[0]).max()), 'inl_eq': bool(np.array_equal(got[2], ref[2]))})\n"
            "print('RESULT ' + json.dumps(out))\n") % root
    res = {}
    for mode in ("1", "0"):
        env = dict(os.environ, SUO_LM_CAM2=mode)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        res[mode] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for a, b in zip(res["1"], res["0"]):
        assert a["d_oracle"] < 1e-6 and b["d_oracle"] < 1e-6 and a["inl_eq"] and b["inl_eq"]
        assert a["inl"] == b["inl"] and np.abs(np.array(a["cam"]) - np.array(b["cam"])).max() < 1e-6


def _multi_view_scene(rng, n_cam, n_obj, noise_px=0.5):
    k = np.array([600.0, 600.0, 320.0, 240.0])
    cam_gt = np.zeros((n_cam, 3, 4))
    for c in range(n_cam):
        ang = 0.5 * (c / max(n_cam - 1, 1) - 0.5)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        cam_gt[c, :, :3] = R
        cam_gt[c, :, 3] = [-300 * (c / max(n_cam - 1, 1) - 0.5), rng.uniform(-20, 20), rng.uniform(-20, 20)]
    obj_gt = np.zeros((n_obj, 3, 4))
    pts = rng.uniform(-60, 60, (n_obj, 10, 3))
    for o in range(n_obj):
        obj_gt[o, :, :3] = S.random_rotation(rng)
        obj_gt[o, :, 3] = [rng.uniform(-250, 250), rng.uniform(-150, 150), rng.uniform(800, 1100)]
    e_cam, e_obj, e_p, e_uv = [], [], [], []
    for c in range(n_cam):
        for o in range(n_obj):
            if rng.random() < 0.15:
                continue                                      # object not detected in this view
            pw = pts[o] @ obj_gt[o, :, :3].T + obj_gt[o, :, 3]
            pc = pw @ cam_gt[c, :, :3].T + cam_gt[c, :, 3]
            uv = np.c_[k[0] * pc[:, 0] / pc[:, 2] + k[2], k[1] * pc[:, 1] / pc[:, 2] + k[3]] + rng.normal(0, noise_px, (10, 2))
            for j in range(10):
                if rng.random() < 0.05:
                    uv[j] = rng.uniform(0, 480, 2)
                e_cam.append(c); e_obj.append(o); e_p.append(pts[o, j]); e_uv.append(uv[j])
    E = len(e_cam)
    info = np.tile([1.0 / noise_px ** 2, 0, 1.0 / noise_px ** 2], (E, 1))
    cam_fixed = np.zeros(n_cam, np.uint8)
    cam_fixed[0] = 1
    cam_init = cam_gt.copy()
    for c in range(1, n_cam):
        cam_init[c] = _perturb(np.vstack([cam_gt[c], [0, 0, 0, 1]]), rng, 5e-4, 0.3)
    obj_init = np.stack([_perturb(np.vstack([T, [0, 0, 0, 1]]), rng, 5e-4, 0.3) for T in obj_gt])
    return {"cam_T": cam_init, "cam_fixed": cam_fixed, "obj_T": obj_init, "obj_fixed": np.zeros(n_obj, np.uint8),
            "edge_cam": np.array(e_cam, np.int32), "edge_obj": np.array(e_obj, np.int32), "edge_camk": np.tile(k, (E, 1)),
            "edge_p": np.array(e_p), "edge_uv": np.array(e_uv), "edge_info": info, "edge_inlier": np.ones(E, np.uint8)}, obj_gt


@pytest.mark.parametrize("n_cam,n_obj", [(3, 2), (12, 6), (30, 8), (64, 10), (25, 16)])
def test_global_bundle_adjustment_matches_oracle(ba, n_cam, n_obj):
    """Global mode: first camera fixed, all other cameras and all objects free (object_slam.py:746-778).
    HIP eliminates cameras by Schur complement; the oracle solves the full dense system.  Below 512 edges one
    workgroup runs the whole adjustment (csrc/lm.hip); the larger graphs (612 - 5400 edges) take the phase kernels
    of csrc/lm_dist.hip under the device-resident schedule, driven from C (geom_api.hip: optimize_phases_one_rank;
    rounds 4-5: the grid-barrier kernel of csrc/lm_grid.hip)."""
    rng = np.random.default_rng(n_cam * 7 + n_obj)
    P, obj_gt = _multi_view_scene(rng, n_cam, n_obj)
    got, ref = _compare_ba(ba, P)
    err = max(np.linalg.norm(got[1][o][:, 3] - obj_gt[o][:, 3]) for o in range(n_obj))
    assert err < 5.0


@pytest.mark.parametrize("n_cam,n_obj", [(30, 20), (4, 18), (12, 33)])
def test_global_bundle_adjustment_beyond_sixteen_free_objects(ba, n_cam, n_obj):
    """The reference's graph has no object limit (lib/object_slam.py:746-778; T-LESS scenes hold ~20 objects).  Beyond 16 free
    objects next to free cameras the reduced system (6 n_obj rows) leaves LDS: suo_optimize runs the phase kernels with the
    system factorised in global memory (csrc/lm_dist.hip: ba_solve_big_kernel) under the host LM schedule -- same results as
    the dense oracle, and a batch may mix such graphs with ordinary frames."""
    rng = np.random.default_rng(n_cam * 13 + n_obj)
    P = S.make_pose_graph(rng, n_cam, n_obj, kp_per_obj=8)
    obj_gt = P.pop("obj_gt")
    P.pop("cam_gt")
    got, ref = _compare_ba(ba, P)
    err = max(np.linalg.norm(got[1][o][:, 3] - obj_gt[o][:, 3]) for o in range(n_obj))
    assert err < 5.0
    # mixed batch: [frame, big graph, frame] == the three alone
    fr = S.make_frame(rng, 8, noise=0.004, outlier_frac=0.1, with_image=False)
    F = S.frame_to_ba_problem(fr, np.stack([_perturb(T, rng, 2e-4, 0.1) for T in fr["T_OtoC"]]))
    keys = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")
    probs = [ba.Problem(*[F[k] for k in keys]), ba.Problem(*[P[k] for k in keys]), ba.Problem(*[F[k] for k in keys])]
    ba.optimize_batch(probs)
    alone = ba.optimize(*[F[k] for k in keys])
    for p in (probs[0], probs[2]):
        assert np.array_equal(p.obj_T.reshape(-1, 3, 4), alone[1]) and np.array_equal(p.inlier, alone[2])
    assert np.array_equal(probs[1].obj_T.reshape(-1, 3, 4), got[1]) and np.array_equal(probs[1].inlier, got[2])


def test_batch_of_frames_equals_individual_calls(ba):
    rng = np.random.default_rng(42)
    probs, singles = [], []
    for f in range(5):
        fr = S.make_frame(rng, 8, noise=0.004, outlier_frac=0.1, with_image=False)
        init = np.stack([_perturb(T, rng, 2e-4, 0.1) for T in fr["T_OtoC"]])
        P = S.frame_to_ba_problem(fr, init)
        args = [P[k] for k in ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv",
                               "edge_info", "edge_inlier")]
        probs.append(ba.Problem(*args))
        singles.append(ba.optimize(*args))
    ba.optimize_batch(probs)
    for p, s in zip(probs, singles):
        assert np.array_equal(p.obj_T.reshape(-1, 3, 4), s[1]) and np.array_equal(p.inlier, s[2])


def test_too_few_edges_is_a_noop(ba):
    rng = np.random.default_rng(9)
    fr = S.make_frame(rng, 1, noise=0.0, with_image=False)
    P = S.frame_to_ba_problem(fr, fr["T_OtoC"])
    for k in ("edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier"):
        P[k] = P[k][:3]
    got, ref = _compare_ba(ba, P)
    assert got[4][0] == 0


@pytest.mark.parametrize("n_cam,n_obj", [(4, 1), (5, 2), (12, 6), (40, 8), (70, 12), (20, 16), (10, 20)])
def test_phase_wise_global_ba_matches_oracle_and_single_kernel(ba, n_cam, n_obj):
    """The multi-GPU path's phase kernels (csrc/lm_dist.hip) under the host LM schedule (suo_slam_amd/ba_dist.py),
    here with one rank: must agree with the dense oracle and with the single-kernel path (csrc/lm.hip).  The
    2-rank exchange itself is covered on CPU with gloo (tests/test_ba_dist_gloo.py)."""
    from suo_slam_amd import ba_dist
    rng = np.random.default_rng(n_cam * 31 + n_obj)
    P, obj_gt = _multi_view_scene(rng, n_cam, n_obj)
    keys = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")
    args = [P[k] for k in keys]
    ref = G.optimize(*args)
    single = ba.optimize(*args)
    full = ba.Problem(*args)
    ba_dist.optimize_distributed(full)
    assert np.array_equal(full.inlier, ref[2]) and np.array_equal(full.inlier, single[2])
    assert full.stats[0] == ref[4][0] and full.stats[3] == ref[4][3]
    for got, want in ((full.cam_T.reshape(-1, 3, 4), ref[0]), (full.obj_T.reshape(-1, 3, 4), ref[1])):
        for a, b in zip(got, want):
            assert _pose_close(a, b, 1e-6, 1e-6), np.abs(a - b).max()
    np.testing.assert_allclose(full.chi2, ref[3], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("n_cam,n_obj,big", [(12, 6, False), (32, 16, False), (10, 20, True)])
def test_device_resident_lm_schedule_equals_the_host_schedule(ba, monkeypatch, n_cam, n_obj, big):
    """suo_slam_amd/ba_dist.py: with the HIP phases g2o's accept / reject arithmetic (optimization_algorithm_levenberg.cpp:88-148) runs in
    one-thread kernels on a device control block and the host looks at it once per batch of units; SUO_BA_HOST_SCHEDULE=1 keeps the
    round-2/3 schedule (the host reads four doubles per trial and decides).  Same kernels, same arithmetic, same order: poses, inlier
    flags, chi2 and the round / iteration / trial counters are bit-identical -- incl. a graph whose reduced system is factorised in global
    memory (> 16 free objects)."""
    from suo_slam_amd import ba_dist
    rng = np.random.default_rng(n_cam * 31 + n_obj)
    if big:
        P = S.make_pose_graph(rng, n_cam, n_obj, kp_per_obj=8)
        P.pop("obj_gt"); P.pop("cam_gt")
    else:
        P, _ = _multi_view_scene(rng, n_cam, n_obj)
    keys = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SUO_BA_HOST_SCHEDULE", mode)
        out[mode] = ba_dist.optimize_distributed(ba.Problem(*[P[k].copy() for k in keys]))
    a, b = out["0"], out["1"]
    assert np.array_equal(a.stats, b.stats), (a.stats, b.stats)
    assert a.stats[2] >= a.stats[1] > 4
    assert np.array_equal(a.cam_T, b.cam_T) and np.array_equal(a.obj_T, b.obj_T)
    assert np.array_equal(a.inlier, b.inlier) and np.array_equal(a.chi2, b.chi2)


@pytest.mark.parametrize("n_cam,n_obj", [(12, 6), (32, 16), (60, 8), (25, 13)])
def test_the_one_c_call_runs_the_schedule_of_the_python_driver(ba, n_cam, n_obj):
    """suo_optimize on ONE large graph with free cameras and objects (what ObjectSLAM.optimize hands over for the global adjustment, lib/object_slam.py:746-778)
    enqueues the phase kernels itself (csrc/geom_api.hip: optimize_phases_one_rank) -- the units, the looks at the control block and the robust rounds of
    suo_slam_amd/ba_dist.py at one rank, without Python between the launches: poses, inlier flags, chi2 and the round / iteration / trial counters are
    bit-identical to the Python-driven run."""
    from suo_slam_amd import ba_dist
    rng = np.random.default_rng(n_cam * 17 + n_obj)
    P, _ = _multi_view_scene(rng, n_cam, n_obj)
    assert len(P["edge_cam"]) >= 512
    keys = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")
    a = ba.optimize_batch([ba.Problem(*[P[k].copy() for k in keys])])[0]
    b = ba_dist.optimize_distributed(ba.Problem(*[P[k].copy() for k in keys]))
    assert np.array_equal(a.stats, b.stats) and a.stats[2] > 4, (a.stats, b.stats)
    assert np.array_equal(a.cam_T, b.cam_T) and np.array_equal(a.obj_T, b.obj_T)
    assert np.array_equal(a.inlier, b.inlier) and np.array_equal(a.chi2[:len(b.chi2)], b.chi2)


def test_one_rank_unit_with_folded_control_steps_equals_the_four_calls(ba, monkeypatch):
    """On one rank nothing is exchanged between the phases of an LM unit and the control steps ride in the tail kernels (suo_ba_lm_unit_one_rank_dev: 12 launches instead
    of 14); SUO_BA_FOLD_CTL=0 keeps the four calls a multi-rank run makes.  Same arithmetic, same order: every output bit-identical."""
    from suo_slam_amd import ba_dist
    rng = np.random.default_rng(404)
    P, _ = _multi_view_scene(rng, 24, 12)
    keys = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SUO_BA_FOLD_CTL", mode)
        out[mode] = ba_dist.optimize_distributed(ba.Problem(*[P[k].copy() for k in keys]))
    a, b = out["1"], out["0"]
    assert np.array_equal(a.stats, b.stats) and a.stats[2] > 4
    assert np.array_equal(a.cam_T, b.cam_T) and np.array_equal(a.obj_T, b.obj_T) and np.array_equal(a.inlier, b.inlier) and np.array_equal(a.chi2, b.chi2)


def test_single_view_frame_with_many_edges_per_object_leaves_the_one_wave_kernel(ba):
    """csrc/lm_frame2.hip keeps a lane's outlier flags in a 32-bit mask: at most 32 edges per lane = 256 per object with 8 lanes.
    A two-object frame with 300 keypoints each fits its edge budget (656) but not that cap -- the dispatcher must hand it to the general
    kernel; the result still matches the oracle."""
    rng = np.random.default_rng(77)
    n_obj, n_kp = 2, 300
    k = np.array([600.0, 600.0, 320.0, 240.0])
    obj_gt = np.zeros((n_obj, 3, 4))
    e_obj, e_p, e_uv = [], [], []
    for o in range(n_obj):
        obj_gt[o, :, :3] = S.random_rotation(rng)
        obj_gt[o, :, 3] = [rng.uniform(-150, 150), rng.uniform(-100, 100), rng.uniform(800, 1000)]
        pts = rng.uniform(-60, 60, (n_kp, 3))
        pc = pts @ obj_gt[o, :, :3].T + obj_gt[o, :, 3]
        uv = np.c_[k[0] * pc[:, 0] / pc[:, 2] + k[2], k[1] * pc[:, 1] / pc[:, 2] + k[3]] + rng.normal(0, 0.5, (n_kp, 2))
        uv[rng.random(n_kp) < 0.05] = rng.uniform(0, 480, 2)
        e_obj += [o] * n_kp; e_p += list(pts); e_uv += list(uv)
    E = len(e_obj)
    P = {"cam_T": np.eye(4)[None, :3], "cam_fixed": np.array([1], np.uint8),
         "obj_T": np.stack([_perturb(np.vstack([T, [0, 0, 0, 1]]), rng, 3e-4, 0.2) for T in obj_gt]), "obj_fixed": np.zeros(n_obj, np.uint8),
         "edge_cam": np.zeros(E, np.int32), "edge_obj": np.array(e_obj, np.int32), "edge_camk": np.tile(k, (E, 1)), "edge_p": np.array(e_p),
         "edge_uv": np.array(e_uv), "edge_info": np.tile([4.0, 0, 4.0], (E, 1)), "edge_inlier": np.ones(E, np.uint8)}
    got, ref = _compare_ba(ba, P)
    assert got[4][0] == 4

"""SLAM host rules and optimize() control flow against outputs of the REFERENCE's own lib/object_slam.py
(tests/golden/slam_golden.npz, recorded by tests/golden/make_slam_golden.py: the reference file imported unmodified, its
missing native modules replaced by a recording g2o stand-in and the PnP oracle).  SURVEY.md 8a rows a16, a17, a21-a26.

Both the product (suo_slam_amd/object_slam.py, vectorised) and the loop-per-detection restatement (oracle/slam_rules.py) are
compared with the recorded outputs on 258 states -- random ones plus states searched to sit ON each rule's boundary (the 3x
re-initialisation rule from both sides, its >= 3 floor, the >= 4 floor of the camera hypotheses) -- and on 35 optimize()
graphs: build_problem() must hand the kernel the graph the reference handed g2o, field for field.

CPU only: ObjectSLAM in debug_gt_kp mode owns no network, and the two native calls the host logic makes (PnP, LM) are routed
to the C oracle HERE, in the test, so that what is checked is the product's host logic.  tests/test_gpu_slam_golden.py
replays the same fixture through the HIP kernels."""
import json
import os

import numpy as np
import pytest

from oracle import geometry as G
from oracle import slam_rules as R
from suo_slam_amd import ba as BA
from suo_slam_amd import object_slam as OS
from tests import slam_states as SS
from tests.golden import treeio

GOLD = treeio.load(os.path.join(os.path.dirname(__file__), "golden", "slam_golden.npz"))
UNITS, GRAPHS, SEQS = GOLD["units"], GOLD["graphs"], GOLD["sequences"]


def product(st, mode="slam", **kw):
    s = OS.ObjectSLAM(None, st["mesh_db"], debug_gt_kp=True, sfm_mode=(mode in ("sfm", "single")), single_view_mode=(mode == "single"),
                      manual_kp_std=0.01, **kw)
    return SS.install(s, st)


def isnone(T):
    return not np.isfinite(T).all()


@pytest.fixture
def oracle_backend(monkeypatch):
    """PnP and LM of the host logic routed to the C oracle (test-side injection; the product itself has no CPU path)."""
    def pnp(xs, ys, threshold=0.001):
        return G.pnp(xs, ys, threshold, seed=0)[0]

    def pnp_batch(xs_list, ys_list, threshold=0.001, seed=0, refine=True, return_info=False):
        T = np.stack([G.pnp(x, y, threshold, seed=(seed + j * OS._lt.SEED_STRIDE) % 2 ** 64)[0] for j, (x, y) in enumerate(zip(xs_list, ys_list))])
        return T, np.array([int(np.allclose(t, np.eye(4))) for t in T], np.int32)

    def optimize_batch(problems):
        for p in problems:
            cam, obj, inl, chi2, stats = G.optimize(p.cam_T, p.cam_fixed, p.obj_T, p.obj_fixed, p.edge_cam, p.edge_obj, p.edge_camk, p.edge_p,
                                                    p.edge_uv, p.edge_info, p.inlier, its=p.its, init_with_outliers=p.init_with_outliers)
            p.cam_T[:], p.obj_T[:], p.inlier[:], p.stats[:] = cam.reshape(-1, 12), obj.reshape(-1, 12), inl, stats
        return problems
    monkeypatch.setattr(OS._lt, "pnp", pnp)
    monkeypatch.setattr(OS._lt, "pnp_batch", pnp_batch)
    monkeypatch.setattr(OS._ba, "optimize_batch", optimize_batch)


def test_fixture_inputs_are_what_the_reference_saw():
    for u in UNITS:
        assert SS.digest(SS.make_state(**json.loads(u["spec"]))) == u["rec"]["digest"]
    for g in GRAPHS:
        assert SS.digest(SS.make_state(**json.loads(g["spec"]))) == g["rec"]["digest"]
    for q in SEQS:
        assert SS.digest(SS.make_sequence(**json.loads(q["spec"]))) == q["rec"]["digest"]
    assert len(UNITS) >= 200


def test_estimate_camera_pose_equals_the_reference():
    """a22, lib/object_slam.py:975-1072."""
    found = none = 0
    for u in UNITS:
        st, rec = SS.make_state(**json.loads(u["spec"])), u["rec"]
        for impl in ("product", "oracle"):
            s = product(st)
            last = s.view_ids.pop()
            s.cam_poses.pop(last)
            if impl == "product":
                T = s._estimate_camera_pose(last)
                best = s.last_cam_hypotheses["best_num_inliers"] if s.last_cam_hypotheses else -2
            else:
                T, best, _ = R.estimate_camera_pose(s.detections, s.obj_poses, last, s.manual_kp_std)
                best = best if any(d.get("pose") is not None and o in s.obj_poses for o, d in s.detections[last].items()) else -2
            assert best == rec["est_best"], (u["spec"], impl)
            if isnone(rec["est_T"]):
                assert T is None
                none += 1
            else:
                np.testing.assert_allclose(T[:3], rec["est_T"], rtol=0, atol=1e-9)
                found += 1
    assert found >= 300 and none >= 40


def test_maybe_reinit_objects_equals_the_reference():
    """a23, :595-697: the (pnp, estim) counts of every object the reference checks, which objects it re-initialises, the map after."""
    fired = 0
    for u in UNITS:
        st, rec = SS.make_state(**json.loads(u["spec"])), u["rec"]
        want = {int(o): (int(a), int(b)) for o, a, b in rec["reinit_counts"]}
        s = product(st)
        last = s.view_ids[-1]
        orc = R.maybe_reinit_objects(s.detections, s.cam_poses, {o: np.array(T) for o, T in s.obj_poses.items()}, s.view_ids, last, s.manual_kp_std, 15)
        assert {o: (r["pnp"], r["estim"]) for o, r in orc.items()} == want, u["spec"]
        assert sorted(o for o, r in orc.items() if r["reinit"]) == rec["reinit_objs"].tolist()
        got = s._maybe_reinit_objects(last, 15)
        assert {o: (r["pnp"], r["estim"]) for o, r in got.items()} == want, u["spec"]
        assert sorted(o for o, r in got.items() if r["reinit"]) == rec["reinit_objs"].tolist()
        fired += len(rec["reinit_objs"])
        assert list(s.obj_poses.keys()) == rec["reinit_obj_poses"]["ids"].tolist()
        for o, T in zip(rec["reinit_obj_poses"]["ids"], rec["reinit_obj_poses"]["T"]):
            np.testing.assert_allclose(np.asarray(s.obj_poses[int(o)])[:3], T, rtol=0, atol=1e-9)
    assert fired >= 100


def test_reinit_rule_boundaries_are_in_the_fixture():
    """pnp == 3 * estim must NOT fire, 3 * estim + 1 must; pnp 2 / 3 with estim 0 sit on the >= 3 floor (:683-687); a best
    hypothesis of exactly 4 inliers is accepted, 3 is not (:1068)."""
    eq = plus1 = f2 = f3 = h4 = h3 = 0
    for u in UNITS:
        rec = u["rec"]
        fired = set(rec["reinit_objs"].tolist())
        for o, a, b in rec["reinit_counts"]:
            if b > 0 and a == 3 * b:
                eq += 1
                assert o not in fired
            if a >= 3 and a == 3 * b + 1:
                plus1 += 1
                assert o in fired
            if (a, b) == (2, 0):
                f2 += 1
                assert o not in fired
            if (a, b) == (3, 0):
                f3 += 1
                assert o in fired
        h4 += rec["est_best"] == 4 and not isnone(rec["est_T"])
        h3 += rec["est_best"] == -1
    assert min(eq, plus1, f2, f3, h4, h3) >= 3, (eq, plus1, f2, f3, h4, h3)


def test_backup_estimate_camera_pose_equals_the_reference(oracle_backend):
    """a24, :933-973: centroid PnP (seed 0 on both sides) / constant velocity / copy."""
    seen = {"centroid_pnp": 0, "const_velocity": 0, "copy": 0}
    for u in UNITS:
        st, rec = SS.make_state(**json.loads(u["spec"])), u["rec"]
        s = product(st)
        last = s.view_ids.pop()
        s.cam_poses.pop(last)
        det = st["detections"][last]
        ids = list(det.keys())
        bboxes = np.stack([det[o]["bbox"] for o in ids])
        pose, which = R.backup_estimate_camera_pose(s.cam_poses, s.obj_poses, s.view_ids, s.cam_K[last], ids, bboxes, R.pnp)
        assert which == rec["backup_branch"]
        np.testing.assert_allclose(np.asarray(pose)[:3], rec["backup_pose"], rtol=0, atol=1e-9)
        s._backup_estimate_camera_pose(last, ids, bboxes)
        np.testing.assert_allclose(np.asarray(s.cam_poses[last])[:3], rec["backup_pose"], rtol=0, atol=1e-9)
        assert s.view_ids[-1] == last
        seen[rec["backup_branch"]] += 1
    assert min(seen.values()) >= 20, seen


def test_collect_results_equals_the_reference():
    """a26, :175-225."""
    n = 0
    for u in UNITS:
        rec = u["rec"]
        if "collect" not in rec:
            continue
        s = product(SS.make_state(**json.loads(u["spec"])))
        s.needs_opt = False
        res = s.collect_results(False, True, False)
        rows = [(v, o, r["score"]) for v in res for o, r in res[v]["poses"].items()]
        want = {(int(v), int(o)): (int(sc), T) for (v, o, sc), T in zip(rec["collect"]["rows"], rec["collect"]["T"])}
        assert len(rows) == len(want)
        for v in res:
            for o, r in res[v]["poses"].items():
                sc, T = want[(v, o)]
                assert r["score"] == sc
                if isnone(T):
                    assert r["T_OtoC"] is None
                else:
                    np.testing.assert_allclose(r["T_OtoC"][:3], T, rtol=0, atol=1e-9)
                n += 1
    assert n >= 500


def _expected_problem(st, s, g, curr_only):
    """The recorded g2o graph re-expressed in the flat layout of suo_ba_problem (include/suo_hip.h)."""
    n_objs_all = len(st["obj_poses"])
    vid, vfix, vT = g["vertex_id"], g["vertex_fixed"], g["vertex_T"]
    if curr_only:
        cams = np.arange(len(vid))
        objs = np.zeros(0, int)
    else:
        objs = np.nonzero(vid < n_objs_all)[0]
        cams = np.nonzero(vid >= n_objs_all)[0]
        assert np.all(objs < cams.min())                                        # objects are added first (:746-778)
    cam_of = {int(vid[i]): k for k, i in enumerate(cams)}
    obj_of = {int(vid[i]): k for k, i in enumerate(objs)}
    return {"cams": cams, "objs": objs, "cam_of": cam_of, "obj_of": obj_of, "cam_T": vT[cams], "cam_fixed": vfix[cams], "obj_T": vT[objs]}


@pytest.mark.parametrize("gi", range(len(GRAPHS)))
def test_build_problem_hands_over_the_graph_the_reference_built(gi, oracle_backend):
    """a16 + a17 + a21: vertices / ids / fixed flags / edges / information / levels / kernels field for field, then the rounds
    (oracle LM on the product's problem) and the product's read-back + culling against the state the reference ended in."""
    gr = GRAPHS[gi]
    st, rec, curr_only = SS.make_state(**json.loads(gr["spec"])), gr["rec"], bool(gr["curr_only"])
    s = product(st, gr["mode"], opt_init_with_outliers=bool(gr["init_with_outliers"]))
    built = s.build_problem(curr_only)
    g = rec["graph"]
    if g is None or g["n_edges"] == 0:
        assert built is None, "the reference returned before building a graph"
    else:
        prob, (cam_index, obj_index, e_ref, _, view_curr) = built
        ex = _expected_problem(st, s, g, curr_only)
        E = g["n_edges"]
        assert len(prob.edge_cam) == E
        # vertices: estimates, fixed flags, and the ids the reference gave them (:754, :765)
        np.testing.assert_array_equal(prob.cam_T.reshape(-1, 3, 4), ex["cam_T"])
        np.testing.assert_array_equal(prob.cam_fixed.astype(bool), ex["cam_fixed"])
        all_cams, all_objs = list(s.cam_poses.keys()), list(s.obj_poses.keys())
        cam_ids = [([view_curr] if curr_only else all_cams).index(v) + (0 if curr_only else len(all_objs)) for v in cam_index]
        assert cam_ids == g["vertex_id"][ex["cams"]].tolist()
        if curr_only:
            assert g["solver"] == "LinearSolverDenseSE3" and np.all(g["fixed_object"]) and np.all(prob.obj_fixed == 1)
            np.testing.assert_array_equal(prob.obj_T.reshape(-1, 3, 4)[prob.edge_obj], g["T_OtoG"])      # object folded into the edge (:816)
            np.testing.assert_array_equal(prob.edge_cam, [ex["cam_of"][int(i)] for i in g["v0"]])
        else:
            assert g["solver"] == "LinearSolverCholmodSE3" and not np.any(g["fixed_object"]) and np.all(prob.obj_fixed == 0)
            np.testing.assert_array_equal(prob.obj_T.reshape(-1, 3, 4), ex["obj_T"])
            assert [all_objs.index(o) for o in obj_index] == g["vertex_id"][ex["objs"]].tolist()
            np.testing.assert_array_equal(prob.edge_obj, [ex["obj_of"][int(i)] for i in g["v0"]])
            np.testing.assert_array_equal(prob.edge_cam, [ex["cam_of"][int(i)] for i in g["v1"]])
        # edges, in the reference's order
        np.testing.assert_array_equal(prob.edge_camk, g["cam_k"])
        np.testing.assert_array_equal(prob.edge_p, g["p"])
        np.testing.assert_array_equal(prob.edge_uv, g["uv"])
        np.testing.assert_array_equal(prob.edge_info[:, 0], g["info"][:, 0, 0])
        np.testing.assert_array_equal(prob.edge_info[:, 2], g["info"][:, 1, 1])
        np.testing.assert_array_equal(prob.edge_info[:, 1], 0.5 * (g["info"][:, 0, 1] + g["info"][:, 1, 0]))
        assert np.all(g["level_at_add"] == 0) and np.allclose(g["delta"], prob.huber_delta, rtol=0, atol=0)
        assert prob.chi2_thr == 5.991 and prob.init_with_outliers == bool(gr["init_with_outliers"] and curr_only)
        # rounds: the optimize(n) arguments are a prefix of its[] (a round loop that breaks early stops calling)
        n_calls = len(g["opt_n"])
        assert list(prob.its[:n_calls]) == g["opt_n"].tolist() and len(prob.its) == 4
        # the oracle, run on the PRODUCT's problem, goes through the same rounds: levels and kernels seen by each optimize(n)
        cam, obj, inl, chi2, stats = G.optimize(prob.cam_T, prob.cam_fixed, prob.obj_T, prob.obj_fixed, prob.edge_cam, prob.edge_obj, prob.edge_camk,
                                                prob.edge_p, prob.edge_uv, prob.edge_info, prob.inlier, its=prob.its,
                                                init_with_outliers=prob.init_with_outliers)
        assert stats[0] == n_calls
        if n_calls:
            drop = 2                                                             # it == max(1, len(its)//2) (:895)
            for k in range(n_calls):
                assert np.all(g["opt_robust"][k] == (1 if k <= drop else 0))
            assert stats[2] == g["opt_lm_trials"].sum() and stats[1] == g["opt_lm_iterations"].clip(0).sum()
    s.optimize(curr_only)
    after = rec["after"]
    assert list(s.cam_poses.keys()) == after["cam_poses"]["ids"].tolist()
    assert list(s.obj_poses.keys()) == after["obj_poses"]["ids"].tolist(), "culling (:898-930) differs"
    for v, T in zip(after["cam_poses"]["ids"], after["cam_poses"]["T"]):
        np.testing.assert_allclose(np.asarray(s.cam_poses[int(v)])[:3], T, rtol=0, atol=1e-9)
    for o, T in zip(after["obj_poses"]["ids"], after["obj_poses"]["T"]):
        np.testing.assert_allclose(np.asarray(s.obj_poses[int(o)])[:3], T, rtol=0, atol=1e-9)
    flags = np.concatenate([np.asarray(d["inliers"], bool) for det in s.detections.values() for d in det.values()])
    np.testing.assert_array_equal(flags, after["inliers"]["flags"])


def replay_sequence(q, slam_factory, pose_tol, check_every_view=True):
    """Drive the product's process_view over a recorded sequence and compare every view with what the reference's own
    process_view left behind.  Returns the number of compared quantities."""
    spec = json.loads(q["spec"])
    seq, rec, mode = SS.make_sequence(**spec), q["rec"], q["mode"]
    s = slam_factory(seq["mesh_db"], mode, q)
    s._rng = np.random.RandomState(spec["seed"])         # the reference draws its keypoint noise from np.random (seeded by the recorder)
    n = 0
    for vw, rv in zip(seq["views"], rec["views"]):
        if mode == "single":
            s.reset()
        seed0 = s._pnp_seed
        s.process_view(vw["view_id"], np.zeros((4, 4, 3), np.uint8), vw["K"], vw["obj_ids"].copy(), vw["bboxes"].copy(), vw["model_kps"],
                       vw["model_kps_masks"], vw["kp_masks"], uv_gt=vw["uv_gt"])
        v = vw["view_id"]
        # the PnP calls of the view: as many problems, same sizes, same seeds (the product batches them per network pass)
        assert s._pnp_seed - seed0 == int(np.count_nonzero(rv["pnp_in_kp_model"]))
        det = s.detections.get(v, {})
        assert list(det.keys()) == rv["det_pose"]["ids"].tolist()
        np.testing.assert_allclose(np.concatenate([d["uv_pred"] for d in det.values()]) if det else np.zeros((0, 2)), rv["det_uv_pred"], rtol=0, atol=1e-12)
        for o, T in zip(rv["det_pose"]["ids"], rv["det_pose"]["T"]):
            P = det[int(o)]["pose"]
            assert (P is None) == isnone(T), (v, o)
            if P is not None:
                np.testing.assert_allclose(P[:3], T, rtol=0, atol=pose_tol * max(1.0, np.abs(T).max()))
                n += 1
        got_flags = np.concatenate([np.asarray(d["inliers"], bool) for d in det.values()]) if det else np.zeros(0, bool)
        np.testing.assert_array_equal(got_flags, rv["det_inliers"]["flags"], err_msg=f"view {v}")
        assert [o for o, d in det.items() if d["prior_uv"] is not None] == rv["det_prior"].tolist()
        for o, puv in zip(rv["det_prior"], rv["det_prior_uv"]):
            np.testing.assert_allclose(det[int(o)]["prior_uv"], puv, rtol=0, atol=2e-5)         # float32 container (:507)
            n += 1
        assert (v in s.cam_poses) == (not isnone(rv["cam_pose"]))
        if v in s.cam_poses:
            np.testing.assert_allclose(np.asarray(s.cam_poses[v])[:3], rv["cam_pose"], rtol=0, atol=pose_tol * max(1.0, np.abs(rv["cam_pose"]).max()))
        assert list(s.obj_poses.keys()) == rv["obj_poses"]["ids"].tolist(), f"map objects after view {v}"
        for o, T in zip(rv["obj_poses"]["ids"], rv["obj_poses"]["T"]):
            np.testing.assert_allclose(np.asarray(s.obj_poses[int(o)])[:3], T, rtol=0, atol=pose_tol * max(1.0, np.abs(T).max()))
            n += 1
        if rv["state"] is not None:
            st = rv["state"]
            assert s.view_ids == st["view_ids"].tolist()
            flags = np.concatenate([np.asarray(d["inliers"], bool) for dd in s.detections.values() for d in dd.values()])
            np.testing.assert_array_equal(flags, st["inliers"]["flags"])
            for vv, T in zip(st["cam_poses"]["ids"], st["cam_poses"]["T"]):
                np.testing.assert_allclose(np.asarray(s.cam_poses[int(vv)])[:3], T, rtol=0, atol=pose_tol * max(1.0, np.abs(T).max()))
    res = s.collect_results(False, True, True)
    want = {(int(v), int(o)): (int(sc), T) for (v, o, sc), T in zip(rec["final"]["rows"], rec["final"]["T"])}
    assert sum(len(r["poses"]) for r in res.values()) == len(want)
    for v in res:
        for o, r in res[v]["poses"].items():
            sc, T = want[(v, o)]
            assert r["score"] == sc and (r["T_OtoC"] is None) == isnone(T)
            if r["T_OtoC"] is not None:
                np.testing.assert_allclose(r["T_OtoC"][:3], T, rtol=0, atol=pose_tol * max(1.0, np.abs(T).max()))
                n += 1
    return n


def _factory(mesh_db, mode, q):
    return OS.ObjectSLAM(None, mesh_db, debug_gt_kp=True, sfm_mode=(mode in ("sfm", "single")), single_view_mode=(mode == "single"),
                         global_opt_every=int(q["global_opt_every"]), manual_kp_std=0.01, opt_init_with_outliers=bool(q["init_with_outliers"]),
                         no_prior_det=bool(q["no_prior_det"]))


@pytest.mark.parametrize("qi", range(len(SEQS)))
def test_process_view_sequences_equal_the_reference(qi, oracle_backend):
    """Whole --debug_gt_kp sequences (SLAM with priors / opt_init_with_outliers / no_prior_det, SfM, single-view) through the
    product's process_view, PnP and LM from the same oracle the recorder's stand-ins used: every view's detections, camera pose,
    map, inlier flags and prior projections (a25, :486-519) must be the reference's."""
    assert replay_sequence(SEQS[qi], _factory, 1e-9) > 50

"""The device-resident frame chain (csrc/frame_geom.hip: masks -> compaction -> normalisation -> PnP -> acceptance -> graph -> LM, one
read-back) against the host route it replaces (the reference's data flow, lib/object_slam.py:1100-1165 + :703-903, restated in
suo_slam_amd/object_slam.py: three read-backs, Python lists, suo_pnp_batch / suo_optimize with host arrays) and against the oracle.

Parity statement: PnP inputs are bit-identical on both routes (same compaction order, fp64 widening, host-inverted K_bbox, same sampler
keys) => PnP poses bit-identical; the graph handed to the LM kernel is bit-identical when the information matrices are (the chain inverts
the float32 covariance in fp64 closed form; fed the same matrices, suo_optimize returns the same bits when every crop is accepted and
agrees to 1e-8 when rejected crops leave the accepted ones in other lanes of the one-wave-per-frame kernel); against the host route's
float32-rounded np.linalg.inv (what the reference hands g2o) refined poses agree to the LM tolerance 1e-6 and every inlier flag whose
chi2 is not within 1e-5 of the gate."""
import numpy as np
import pytest
import torch

from oracle import geometry as G
from suo_slam_amd import ba, lambdatwist
from suo_slam_amd import geometry as geo
from suo_slam_amd import synthetic as S
from suo_slam_amd.frame_geom import FrameGeometry, kbbox_terms

pytestmark = pytest.mark.gpu


def _frame_inputs(rng, n_obj, drop=0.25, few=(), far=()):
    """Device-shaped network outputs for one synthetic frame: uv / cov float32 [L,41,...] and a validity mask with random
    drop-outs; objects in `few` keep < 4 keypoints (PnP impossible), objects in `far` get a diameter that fails the depth test."""
    fr = S.make_frame(rng, n_obj, noise=0.004, outlier_frac=0.08, with_image=False)
    mask = fr["model_kps_masks"] & (rng.random((n_obj, 41)) >= drop)
    for o in few:
        keep = np.nonzero(mask[o])[0][:int(rng.integers(0, 4))]
        mask[o] = False
        mask[o, keep] = True
    diam = fr["diameter"].copy()
    for o in far:
        diam[o] = 4.0 * fr["T_OtoC"][o][2, 3]
    return fr, mask, diam


def _info64(cov32):
    a, b, c, d = (cov32[..., 0, 0].astype(np.float64), cov32[..., 0, 1].astype(np.float64), cov32[..., 1, 0].astype(np.float64),
                  cov32[..., 1, 1].astype(np.float64))
    det = a * d - b * c
    return np.stack([d / det, 0.5 * (-b / det + -c / det), a / det], axis=-1)


def _host_route(fr, mask, diam, seed, its=(10, 10, 40, 40), use_cov=True):
    """The same frame through the host-array entry points, fed exactly what the chain derives on the device."""
    L = len(mask)
    Kb = fr["K_bbox"].astype(np.float32).astype(np.float64)
    idx = [o for o in range(L) if mask[o].sum() >= 4]
    xs = [fr["model_kps"][o][mask[o]].astype(np.float64) for o in idx]
    ys = [geo.normalize_uv(fr["uv"][o][mask[o]].astype(np.float64), Kb[o]) for o in idx]
    T, status = lambdatwist.pnp_batch(xs, ys, 1e-3, seed=seed)
    T_pnp = np.tile(np.eye(4), (L, 1, 1))
    st = np.ones(L, np.int32)
    for j, o in enumerate(idx):
        T_pnp[o], st[o] = T[j], status[j]
    acc = np.array([st[o] == 0 and mask[o].sum() >= 4 and T_pnp[o][2, 3] > 0.5 * diam[o] for o in range(L)])
    objs = [o for o in range(L) if acc[o]]
    res = {"T_pnp": T_pnp, "status": st, "accepted": acc, "T_opt": T_pnp[:, :3, :].copy(), "inlier": {}, "stats": np.zeros(4, np.int32)}
    if objs:
        e_obj = np.concatenate([np.full(int(mask[o].sum()), j, np.int32) for j, o in enumerate(objs)])
        camk = np.concatenate([np.tile([Kb[o][0, 0], Kb[o][1, 1], Kb[o][0, 2], Kb[o][1, 2]], (int(mask[o].sum()), 1)) for o in objs])
        p = np.concatenate([fr["model_kps"][o][mask[o]].astype(np.float64) for o in objs])
        uv = np.concatenate([fr["uv"][o][mask[o]].astype(np.float64) for o in objs])
        info = np.concatenate([_info64(fr["cov"][o][mask[o]]) for o in objs]) if use_cov else np.tile([1.0, 0.0, 1.0], (len(p), 1))
        args = (np.eye(4)[None, :3], np.array([1], np.uint8), T_pnp[objs][:, :3], np.zeros(len(objs), np.uint8), np.zeros(len(p), np.int32), e_obj,
                camk, p, uv, info, np.ones(len(p), np.uint8))
        cam, obj, inl, chi2, stats = ba.optimize(*args, its=its)
        res["oracle"] = G.optimize(*args, its=its)
        k = 0
        for j, o in enumerate(objs):
            n = int(mask[o].sum())
            res["T_opt"][o] = obj[j]
            res["inlier"][o] = inl[k:k + n].astype(bool)
            k += n
        res["stats"] = stats
    return res


def _launch(fg, frames, seed, its=(10, 10, 40, 40), use_cov=True, do_lm=True):
    uv = torch.from_numpy(np.concatenate([f[0]["uv"] for f in frames])).cuda()
    cov = torch.from_numpy(np.concatenate([f[0]["cov"] for f in frames])).cuda()
    mask = torch.from_numpy(np.concatenate([f[1] for f in frames]).astype(np.uint8)).cuda()
    kps = torch.from_numpy(np.concatenate([f[0]["model_kps"] for f in frames]).astype(np.float32)).cuda()
    kinv, camk = kbbox_terms(np.concatenate([f[0]["K_bbox"] for f in frames]).astype(np.float32))
    first = np.concatenate([[0], np.cumsum([len(f[1]) for f in frames])])
    fg.launch(first, uv, cov, mask, kps, kinv, camk, 0.5 * np.concatenate([f[2] for f in frames]), seed=seed, use_cov=use_cov, do_lm=do_lm, its=its)
    return fg.fetch()


@pytest.mark.parametrize("n_obj,use_cov", [(8, True), (8, False), (5, True), (16, True), (1, True)])
def test_chain_equals_the_host_array_entry_points_bit_for_bit(n_obj, use_cov):
    rng = np.random.default_rng(100 + n_obj + use_cov)
    fg = FrameGeometry(16, 1)
    for trial in range(4):
        few = (1,) if n_obj > 2 and trial % 2 else ()
        far = (2,) if n_obj > 3 and trial >= 2 else ()
        fr, mask, diam = _frame_inputs(rng, n_obj, few=few, far=far)
        seed = 1000 * trial + 7
        r = _launch(fg, [(fr, mask, diam)], seed, use_cov=use_cov)
        h = _host_route(fr, mask, diam, seed, use_cov=use_cov)
        assert np.array_equal(r["mask"], mask) and np.array_equal(r["n_kp"], mask.sum(1))
        assert np.array_equal(r["uv"], fr["uv"]) and np.array_equal(r["cov"], fr["cov"])
        assert np.array_equal(r["pnp_status"], h["status"])
        assert np.array_equal(r["T_pnp"], h["T_pnp"]), "PnP poses must be bit-identical (same inputs, same sampler keys)"
        assert np.array_equal(r["accepted"], h["accepted"])
        for o in few:
            assert not r["accepted"][o] and r["pnp_status"][o] == 1
        for o in far:
            assert not r["accepted"][o] and r["pnp_status"][o] == 0
        if r["accepted"].all():
            # same kernel (csrc/lm_frame2.hip), same graph, same lanes: same bits.  With rejected crops the chain keeps their (empty)
            # slots, so the accepted objects sit in other lane groups than in the host route's compacted graph and the frame-wide sums
            # pair up differently: rounding-level differences, which ~100 non-converged LM iterations carry up to ~1e-9 relative
            assert np.array_equal(r["T_opt"], h["T_opt"]), np.abs(r["T_opt"] - h["T_opt"]).max()
            assert np.array_equal(r["lm_stats"][0], h["stats"])
        else:
            np.testing.assert_allclose(r["T_opt"], h["T_opt"], rtol=0, atol=1e-8 * np.abs(h["T_opt"]).max())
            assert r["lm_stats"][0][0] == h["stats"][0] and r["lm_stats"][0][3] == h["stats"][3]
        for o, inl in h["inlier"].items():
            assert np.array_equal(r["inlier"][o, :len(inl)], inl)
        if "oracle" in h:                                                 # and the CPU oracle on the same graph
            objs = [o for o in range(n_obj) if h["accepted"][o]]
            np.testing.assert_allclose(r["T_opt"][objs], h["oracle"][1], rtol=0, atol=1e-6 * np.abs(h["oracle"][1]).max())
            k = 0
            for o in objs:
                n = int(mask[o].sum())
                assert np.array_equal(r["inlier"][o, :n], h["oracle"][2][k:k + n].astype(bool))
                k += n


def test_several_frames_in_one_launch_equal_one_launch_per_frame():
    """Frames are independent problems (one workgroup each); the sampler keys continue across the frames of a launch the way
    ObjectSLAM._pnp_seed advances between process_view calls."""
    rng = np.random.default_rng(5)
    frames = [_frame_inputs(rng, n, few=((0,) if n > 4 else ())) for n in (8, 3, 16, 1, 8)]
    big = FrameGeometry(64, 8)
    one = FrameGeometry(16, 1)
    r = _launch(big, frames, seed=11)
    g0, seed = 0, 11
    for f in frames:
        L = len(f[1])
        r1 = _launch(one, [f], seed=seed)
        for key in ("T_pnp", "T_opt", "accepted", "inlier", "n_kp", "pnp_status", "mask", "pnp_iterations"):
            assert np.array_equal(r[key][g0:g0 + L], r1[key]), key
        seed += int(np.count_nonzero(r1["n_kp"] >= 4))
        g0 += L
    assert r["lm_stats"].shape == (5, 4) and (r["lm_stats"][:, 0] > 0).all()


def test_do_lm_false_stops_after_acceptance():
    rng = np.random.default_rng(6)
    f = _frame_inputs(rng, 6)
    fg = FrameGeometry(16, 1)
    r = _launch(fg, [f], seed=3, do_lm=False)
    assert np.array_equal(r["T_opt"], r["T_pnp"][:, :3, :]) and (r["lm_stats"] == 0).all() and r["inlier"].all()
    r2 = _launch(fg, [f], seed=3, do_lm=True)
    assert np.array_equal(r2["T_pnp"], r["T_pnp"]) and not np.array_equal(r2["T_opt"], r["T_opt"])


def test_limits_fail_loudly():
    from suo_slam_amd._lib import SuoError
    rng = np.random.default_rng(7)
    fg = FrameGeometry(16, 1)
    with pytest.raises(SuoError):
        _launch(fg, [_frame_inputs(rng, 8), _frame_inputs(rng, 4)], seed=0)          # 2 frames, context holds 1
    fg2 = FrameGeometry(32, 1)
    with pytest.raises(SuoError):
        _launch(fg2, [_frame_inputs(rng, 17)], seed=0)                                # > 16 objects in one frame with LM


def test_frames_of_more_than_64_crops_are_refused_even_without_lm():
    """fg_build_kernel is one wave per frame, a lane per crop: beyond 64 crops of one frame nothing would be written and
    fetch() would return stale bytes -- the launch is refused whatever do_lm says; 17 ... 64 crops without LM are fine."""
    from suo_slam_amd._lib import SuoError
    rng = np.random.default_rng(8)
    fg = FrameGeometry(80, 1)
    r = _launch(fg, [_frame_inputs(rng, 40)], seed=1, do_lm=False)
    assert r["T_pnp"].shape[0] == 40 and np.isfinite(r["T_pnp"]).all() and set(np.unique(r["accepted"])) <= {0, 1}
    with pytest.raises(SuoError):
        _launch(fg, [_frame_inputs(rng, 65)], seed=1, do_lm=False)


def _mesh_db(fr, diam=None):
    return {o: {"diameter": float((fr["diameter"] if diam is None else diam)[i]), "is_symmetric": False} for i, o in enumerate(fr["obj_ids"])}


@pytest.mark.parametrize("n_obj,no_cov", [(8, False), (8, True), (13, False)])
def test_object_slam_single_view_both_routes_leave_the_same_state(n_obj, no_cov):
    """ObjectSLAM.process_view on NETWORK OUTPUT (random but confident weights: the masks pass), device chain vs host route:
    same detections (masks, keypoints, covariances, PnP poses bit for bit), same map after culling, refined poses within the LM
    tolerance, same inlier flags away from the chi2 gate, same results dict."""
    from suo_slam_amd import weights
    from suo_slam_amd.object_slam import ObjectSLAM
    sd = weights.make_random_state_dict(seed=0, logit_gain=8.0)
    sd["classifier.2.bias"] = (np.asarray(sd["classifier.2.bias"]) + 4.0).astype(np.float32)
    rng = np.random.default_rng(40 + n_obj)
    states = []
    frames = [S.make_frame(rng, n_obj, noise=0.0) for _ in range(3)]
    for chain in (True, False):
        slam = ObjectSLAM(None, _mesh_db(frames[0]), sfm_mode=True, single_view_mode=True, state_dict=sd, max_crops=16, kp_var_thresh=0.5,
                          bbox_thresh=1.0, no_network_cov=no_cov, device_chain=chain)
        per_frame = []
        for v, fr in enumerate(frames):
            slam.reset()
            slam.mesh_db = _mesh_db(fr)
            slam.process_view(v, fr["image"], fr["K"], np.array(fr["obj_ids"]), fr["boxes"].astype(np.float64), fr["model_kps"], fr["model_kps_masks"],
                              fr["model_kps_masks"])
            per_frame.append((slam.detections[v], dict(slam.obj_poses), slam.cam_poses[v].copy(), slam.collect_results()[v]["poses"]))
        states.append((per_frame, slam._pnp_seed, slam.avg_std_meter.average(), dict(slam.obj_num_det_kps)))
    (a, seed_a, std_a, kps_a), (b, seed_b, std_b, kps_b) = states
    assert seed_a == seed_b and kps_a == kps_b and abs(std_a - std_b) < 1e-12
    n_poses = 0
    for (det_a, map_a, cam_a, res_a), (det_b, map_b, cam_b, res_b) in zip(a, b):
        assert list(det_a.keys()) == list(det_b.keys()) and np.array_equal(cam_a, cam_b)
        for o in det_a:
            da, db = det_a[o], det_b[o]
            assert np.array_equal(da["kp_mask"], db["kp_mask"]) and np.array_equal(da["uv_pred"], db["uv_pred"]) and np.array_equal(da["model_kp"], db["model_kp"])
            assert np.array_equal(da["K"], db["K"]) and (da["cov_pred"] is None) == (db["cov_pred"] is None)
            if da["cov_pred"] is not None:
                assert np.array_equal(da["cov_pred"], db["cov_pred"])
            assert (da["pose"] is None) == (db["pose"] is None)
            if da["pose"] is not None:
                assert np.array_equal(da["pose"], db["pose"])                       # PnP: bit-identical
                n_poses += 1
            assert da["inliers"].shape == db["inliers"].shape
            assert np.count_nonzero(da["inliers"] != db["inliers"]) == 0
        assert list(map_a.keys()) == list(map_b.keys())
        for o in map_a:
            np.testing.assert_allclose(np.asarray(map_a[o])[:3], np.asarray(map_b[o])[:3], rtol=0, atol=1e-6 * np.abs(map_b[o]).max())
        assert {o: r["score"] for o, r in res_a.items()} == {o: r["score"] for o, r in res_b.items()}
    assert n_poses >= n_obj


def test_upload_kernel_copies_pinned_host_memory_bit_for_bit():
    """suo_upload (the frame H2D of the timed region): any size incl. a tail that is not a multiple of 16 bytes; misaligned pointers refused."""
    import ctypes as C
    from suo_slam_amd import _lib
    from suo_slam_amd._lib import SuoError
    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (921600, 32 * 921600, 1000, 16, 33, 4096 + 7):
        h = torch.from_numpy(np.random.default_rng(n).integers(0, 255, n).astype(np.uint8)).pin_memory()
        d = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        _lib.check(lib.suo_upload(C.c_void_p(d.data_ptr()), C.c_void_p(h.data_ptr()), n, st))
        torch.cuda.synchronize()
        got = d.cpu()
        assert torch.equal(got[:n], h) and int(got[n:].sum()) == 0
    with pytest.raises(SuoError):
        _lib.check(lib.suo_upload(C.c_void_p(d.data_ptr() + 4), C.c_void_p(h.data_ptr()), 64, st))


def test_more_than_sixteen_objects_refine_through_suo_optimize():
    """The device chain REFINES frames of <= 16 crops (one wave per frame, csrc/lm_frame2.hip); ObjectSLAM sends larger frames (T-LESS scenes can have more) through
    process_view's general route instead of failing -- since round 6 with network -> masks -> compaction -> PnP -> acceptance still as one device chain
    (do_lm = 0, no crop limit) and the pose graph through suo_optimize: same interface, same kind of result, both halves replayed against the oracles."""
    from suo_slam_amd import weights
    from suo_slam_amd.object_slam import ObjectSLAM
    from tests import replay
    sd = weights.make_random_state_dict(seed=0, logit_gain=8.0)
    sd["classifier.2.bias"] = (np.asarray(sd["classifier.2.bias"]) + 4.0).astype(np.float32)
    fr = S.make_frame(np.random.default_rng(61), 18, noise=0.0)
    slam = ObjectSLAM(None, _mesh_db(fr), sfm_mode=True, single_view_mode=True, state_dict=sd, max_crops=18, kp_var_thresh=0.5, bbox_thresh=1.0)
    with replay.record() as rec:
        slam.process_view(0, fr["image"], fr["K"], np.array(fr["obj_ids"]), fr["boxes"].astype(np.float64), fr["model_kps"], fr["model_kps_masks"],
                          fr["model_kps_masks"])
    assert len(rec.chain) == 1 and not rec.chain[0]["do_lm"] and len(rec.pnp) == 0 and len(rec.ba) == 1
    n_pnp, n_lm = replay.check_chain(rec)
    assert n_pnp >= 12 and n_lm == 0 and replay.check_ba(rec) == 1
    # ... and the host route (device_chain=False: three read-backs, suo_pnp_batch on host arrays) leaves the same frame behind
    host = ObjectSLAM(None, _mesh_db(fr), sfm_mode=True, single_view_mode=True, state_dict=sd, max_crops=18, kp_var_thresh=0.5, bbox_thresh=1.0, device_chain=False)
    with replay.record() as rec_h:
        host.process_view(0, fr["image"], fr["K"], np.array(fr["obj_ids"]), fr["boxes"].astype(np.float64), fr["model_kps"], fr["model_kps_masks"],
                          fr["model_kps_masks"])
    assert len(rec_h.chain) == 0 and len(rec_h.pnp) == 1 and len(rec_h.ba) == 1
    assert list(host.obj_poses.keys()) == list(slam.obj_poses.keys())
    for o in host.obj_poses:
        np.testing.assert_allclose(np.asarray(slam.obj_poses[o])[:3], np.asarray(host.obj_poses[o])[:3], rtol=0, atol=1e-9 * np.abs(host.obj_poses[o]).max())
    assert len(slam.collect_results()[0]["poses"]) == 18

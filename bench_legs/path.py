"""Legs of bench.py that run the product path in other call shapes: PCIe-inclusive frames, the pose check, one frame per call, ObjectSLAM, SLAM, global BA."""
import os
import time

import numpy as np

from .common import BBOX_THRESH, KP_VAR_THRESH, confident_state_dict
from .pipeline import FramePipeline


def frames_from_host_leg(L, pool, F, use_graph, depth, steps, warmup, fps_value):
    """The timed region again with the frames handed over in PINNED HOST memory and uploaded inside it (0.92 MB per frame over PCIe, a copy kernel on the step's
    stream) -- the rate rounds 1-4 reported as `value`.  `value` itself follows the bench contract: inputs resident in HBM when the timed region starts (the C ABI
    takes device pointers for the frames: include/suo_hip.h, suo_net_forward_frames); the boxes / model keypoints (0.2 MB per step) come from the host either way."""
    import torch
    pipe = FramePipeline(L, pool, F, use_graph=use_graph, depth=depth, resident=False)
    for i in range(warmup):
        pipe.step(i)
    pipe.drain(warmup)
    pipe.reset_metrics()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pipe.step(warmup + i)
    pipe.drain(warmup + steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert pipe.n_frames == steps * F
    fps = steps * F / dt
    return {"frames_per_s": round(fps, 2), "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps, "warmup": warmup,
            "over_value": round(fps / fps_value, 4),
            "note": "the PCIe-inclusive rate (frames in pinned host memory, H2D inside the timed region): what `value` was in rounds 1-4; `value` has the frames resident in HBM"}


def pose_check_leg(L, pool, use_graph):
    """Is the geometry RIGHT?  The same device chain on good measurements (projected ground-truth keypoints + N(0, 0.01^2) NDC noise,
    5 % gross outliers, random SPD covariances -- the reference's --debug_gt_kp mode, lib/object_slam.py:1129-1131) against the ground
    truth, and on a sample of frames against the CPU oracle on identical inputs (PnP: same sampler keys)."""
    import torch
    from oracle import geometry as G
    from suo_slam_amd import geometry as geo
    F = 8
    pipe = FramePipeline(L, pool[:F], F, use_graph=use_graph, depth=1, gt_keypoints=True)
    pipe.step(0)
    r = pipe.retire(pipe.slots[0])
    torch.cuda.synchronize()
    out = {"mean_rel_translation_err": round(pipe.pose_err / max(pipe.n_pose_gt, 1), 5), "poses": pipe.n_pose_gt, "of_objects": L * F,
           "inlier_edges": pipe.n_inl, "lm_trials": pipe.n_trials}
    dT = dR = 0.0
    flags_differ = n_cmp = 0
    rank = 0
    for j in range(min(F, 4)):
        fr = pool[j]
        Kb = fr["K_bbox"].astype(np.float32).astype(np.float64)
        init, objs = [], []
        for o in range(L):
            m = fr["model_kps_masks"][o]
            g = j * L + o
            To = G.pnp(fr["model_kps"][o][m].astype(np.float64), geo.normalize_uv(fr["uv"][o][m].astype(np.float64), Kb[o]), 1e-3,
                       seed=(rank + o * 0x9E3779B97F4A7C15) % 2 ** 64)[0]
            dT = max(dT, float(np.abs(r["T_pnp"][g] - To).max()))
            if r["accepted"][g]:
                init.append(To[:3])
                objs.append(o)
        rank += L
        if not objs:
            continue
        e_obj = np.concatenate([np.full(int(fr["model_kps_masks"][o].sum()), k, np.int32) for k, o in enumerate(objs)])
        sel = [fr["model_kps_masks"][o] for o in objs]
        camk = np.concatenate([np.tile([Kb[o][0, 0], Kb[o][1, 1], Kb[o][0, 2], Kb[o][1, 2]], (int(m.sum()), 1)) for o, m in zip(objs, sel)])
        p = np.concatenate([fr["model_kps"][o][m].astype(np.float64) for o, m in zip(objs, sel)])
        uv = np.concatenate([fr["uv"][o][m].astype(np.float64) for o, m in zip(objs, sel)])
        c = np.concatenate([fr["cov"][o][m].astype(np.float64) for o, m in zip(objs, sel)])
        det = c[:, 0, 0] * c[:, 1, 1] - c[:, 0, 1] * c[:, 1, 0]
        info = np.stack([c[:, 1, 1] / det, 0.5 * (-c[:, 0, 1] / det + -c[:, 1, 0] / det), c[:, 0, 0] / det], 1)
        ref = G.optimize(np.eye(4)[None, :3], np.array([1], np.uint8), np.array(init), np.zeros(len(objs), np.uint8), np.zeros(len(p), np.int32), e_obj,
                         camk, p, uv, info, np.ones(len(p), np.uint8))
        k = 0
        for i, o in enumerate(objs):
            g = j * L + o
            n = int(fr["model_kps_masks"][o].sum())
            dR = max(dR, float(np.abs(r["T_opt"][g][:, :3] - ref[1][i][:, :3]).max()))
            dT = max(dT, float(np.abs(r["T_opt"][g][:, 3] - ref[1][i][:, 3]).max() / np.abs(ref[1][i][:, 3]).max()))
            flags_differ += int(np.count_nonzero(r["inlier"][g, :n] != ref[2][k:k + n].astype(bool)))
            k += n
            n_cmp += 1
    out["vs_oracle"] = {"objects": n_cmp, "max_abs_dR_entry": float(f"{dR:.3e}"), "max_rel_dt": float(f"{dT:.3e}"), "inlier_flags_differing": flags_differ}
    return out


def latency_leg(L, pool, use_graph, seconds=0.6):
    """The reference's call shape (evaluate.py:338-395: one frame per network call).  ONE frame in flight: launch -> results on the host,
    nothing overlapped (the latency of a frame); four in flight: the same calls pipelined."""
    import torch
    out = {}
    for depth, only in ((1, "cnn"), (1, "all"), (4, "all")):
        pipe = FramePipeline(L, pool, 1, use_graph=use_graph, depth=depth, only=only)
        for i in range(8):
            pipe.step(i)
        pipe.drain(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while n < 16 or time.perf_counter() - t0 < seconds:
            pipe.step(8 + n)
            n += 1
        pipe.drain(8 + n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if only == "cnn":
            out["network_ms_per_frame"] = round(1e3 * dt / n, 3)          # H2D + network + decode + masks of one 8-crop frame, nothing else
        elif depth == 1:
            out["one_in_flight_ms_per_frame"] = round(1e3 * dt / n, 3)
            out["one_in_flight_fps"] = round(n / dt, 2)
        else:
            out["four_in_flight_fps"] = round(n / dt, 2)
        del pipe
    return out


def drop_in_leg(L, pool, n=40):
    """The call path evaluate.py takes (evaluate.py:338-395): ObjectSLAM(single_view_mode, sfm_mode) -- reset(), process_view(...),
    collect_results() per frame, synchronous, one frame in flight -- on network output (confident random weights)."""
    import torch
    from suo_slam_amd.object_slam import ObjectSLAM
    fr0 = pool[0]
    mesh = lambda fr: {o: {"diameter": float(fr["diameter"][k]), "is_symmetric": False} for k, o in enumerate(fr["obj_ids"])}  # noqa: E731
    slam = ObjectSLAM(None, mesh(fr0), sfm_mode=True, single_view_mode=True, state_dict=confident_state_dict(), max_crops=max(16, L),
                      kp_var_thresh=KP_VAR_THRESH, bbox_thresh=BBOX_THRESH)
    n_pose = 0
    for it in range(n + 6):
        if it == 6:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        fr = pool[it % len(pool)]
        slam.reset()
        slam.mesh_db = mesh(fr)
        slam.process_view(it, fr["image"], fr["K"], np.array(fr["obj_ids"]), fr["boxes"].astype(np.float64), fr["model_kps"], fr["model_kps_masks"],
                          fr["model_kps_masks"])
        res = slam.collect_results(no_viz=True)
        n_pose += sum(r["T_OtoC"] is not None for r in res[it]["poses"].values())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"call": "ObjectSLAM.reset / process_view / collect_results per frame (evaluate.py:338-395), device chain", "frames": n,
           "process_view_ms": round(1e3 * dt / n, 3), "evaluator_fps": round(n / dt, 2), "poses_returned": n_pose,
           "tracking_meter_ms": round(1e3 * slam.track_time_meter.average(), 3)}
    # the same loop with B views per device call (Evaluator(frames_per_call=B) -> ObjectSLAM.process_views_single): the views of a single-view
    # evaluation are independent, so they can share a network call and a geometry launch; object ids made unique per frame (one mesh table)
    B = 16
    del slam
    frames = [pool[i % len(pool)] for i in range(B)]
    mesh_all = {100 * i + o: {"diameter": float(fr["diameter"][k]), "is_symmetric": False} for i, fr in enumerate(frames) for k, o in enumerate(fr["obj_ids"])}
    slam = ObjectSLAM(None, mesh_all, sfm_mode=True, single_view_mode=True, state_dict=confident_state_dict(), max_crops=B * max(16, L),
                      kp_var_thresh=KP_VAR_THRESH, bbox_thresh=BBOX_THRESH)
    # as Evaluator.run drives it: batch i + 1 is submitted before batch i is collected (ObjectSLAM.submit_views_single / collect_views_single), so the
    # host's bookkeeping of one batch runs under the device work of the next; every batch's results are collected inside the timed region
    n_calls, n_pose_b = 6, 0

    def collect():
        res = slam.collect_views_single()
        return sum(r["T_OtoC"] is not None for rv in res for v in rv.values() for r in v["poses"].values())
    for it in range(n_calls + 2):
        if it == 2:
            slam.drain_views_single()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        views = [(it * B + i, fr["image"], fr["K"], 100 * i + np.array(fr["obj_ids"]), fr["boxes"].astype(np.float64), fr["model_kps"], fr["model_kps_masks"],
                  fr["model_kps_masks"]) for i, fr in enumerate(frames)]
        slam.submit_views_single(views)
        if slam.views_in_flight() == 2:
            n_pose_b += collect()
    while slam.views_in_flight():
        n_pose_b += collect()
    torch.cuda.synchronize()
    dtb = time.perf_counter() - t0
    out["views_per_call_%d" % B] = {"call": "ObjectSLAM.submit_views_single / collect_views_single, two batches in flight (Evaluator(frames_per_call=%d))" % B, "frames": n_calls * B,
                                    "ms_per_frame": round(1e3 * dtb / (n_calls * B), 3), "evaluator_fps": round(n_calls * B / dtb, 2),
                                    "poses_returned_per_frame": round(n_pose_b / ((n_calls + 2) * B), 2)}
    return out


def slam_leg(n_views=60, n_obj=8, on_device=True):
    """BASELINE configs[2]: one synthetic sequence through ObjectSLAM.process_view in SLAM mode, the reference's two meters
    (lib/object_slam.py:155-164, 421-427, 444-451): tracking = network pass without priors + camera-pose hypotheses + network pass with
    device-rendered priors for the symmetric objects + PnP + re-initialisation checks + current-view LM; global optimisation = the
    pose-graph adjustment every 10 views.  The network runs on the frame's pixels (both passes, timed), its output is read back and
    then replaced by the projected ground-truth keypoints + noise (--debug_gt_kp, :1129-1131): random weights cannot track."""
    from suo_slam_amd import synthetic as S
    from suo_slam_amd import weights
    from suo_slam_amd.object_slam import ObjectSLAM
    seq = S.make_slam_sequence(np.random.default_rng(3), n_views, n_obj)
    sd = weights.make_random_state_dict(0, 8.0)
    out, reps = None, []
    for rep in range(4):                        # the first pass pays graph captures / first launches; then three measured passes: the MEDIAN is reported (these
        #                                         boxes are shared -- a pass that coincides with a neighbour's burst reads 0.5-1 ms per view high, see tracking_ms_reps)
        slam = ObjectSLAM(None, seq["mesh_db"], debug_gt_kp=True, manual_kp_std=0.01, state_dict=sd, max_crops=max(16, n_obj), run_network_in_debug=True,
                          debug_gt_on_device=on_device)
        t0 = time.perf_counter()
        for vw in seq["views"]:
            slam.process_view(vw["view_id"], vw["image"], vw["K"], vw["obj_ids"].copy(), vw["bboxes"].copy(), vw["model_kps"], vw["model_kps_masks"],
                              vw["kp_masks"], uv_gt=vw["uv_gt"])
        res = slam.collect_results(no_viz=True, final=True)
        dt = time.perf_counter() - t0
        if rep == 0:
            continue
        err = []
        for vw in seq["views"]:
            for o in vw["obj_ids"]:
                T = res.get(vw["view_id"], {}).get("poses", {}).get(int(o), {}).get("T_OtoC")
                if T is not None:
                    gt = vw["T_GtoC_gt"] @ seq["T_OtoG_gt"][int(o)]
                    err.append(np.linalg.norm(T[:3, 3] - gt[:3, 3]) / gt[2, 3])
        reps.append({"views": n_views, "objects": n_obj, "tracking_ms_per_view": round(1e3 * slam.track_time_meter.average(), 3),
                     "global_opt_ms": round(1e3 * slam.opt_time_meter.average(), 3), "global_opts": slam.opt_time_meter.count,
                     "wall_ms_per_view": round(1e3 * dt / n_views, 3), "camera_poses": len(slam.cam_poses), "poses": len(err),
                     "median_rel_translation_err": round(float(np.median(err)), 5) if err else None})
        del slam
    reps.sort(key=lambda r: r["tracking_ms_per_view"])
    out = dict(reps[len(reps) // 2], tracking_ms_reps=[r["tracking_ms_per_view"] for r in reps], global_opt_ms_reps=[r["global_opt_ms"] for r in reps],
               keypoints="network run on the frame's pixels (both passes), output replaced by projected GT + N(0,0.01^2) (debug_gt_kp)" +
               (" ON THE DEVICE (float32, what the network emits): the view continues on the product route -- both passes as ONE device chain: pass A, PnP, "
                "camera-hypothesis vote + prior projection (suo_slam_vote), pass B, PnP, the host reading pass A's block while pass B runs "
                "(ObjectSLAM._process_view_slam_chain)" if on_device else " on the host (rounds 1-5's leg: the reference's host-side debug route)"))
    if on_device:
        # rounds 1-5's figure beside it: the same sequence on the host-side debug route (three read-backs per pass, Python compaction, PnP on host arrays)
        try:
            host = slam_leg(n_views, n_obj, on_device=False)
            out["host_debug_route"] = {k: host[k] for k in ("tracking_ms_per_view", "global_opt_ms", "wall_ms_per_view", "median_rel_translation_err")}
        except Exception as e:
            out["host_debug_route"] = {"error": repr(e)[:200]}
    return out


def global_ba_leg(world, L, n_cam_per_rank=32, reps=3):
    """BASELINE configs[4]'s exchange step: ONE global pose-graph adjustment (first camera fixed, all other cameras and all
    L objects free, lib/object_slam.py:746-778) whose cameras are partitioned over the ranks; each LM trial all-reduces the
    reduced object system over RCCL (suo_slam_amd/ba_dist.py).  Weak scaling: n_cam_per_rank cameras per GPU."""
    from suo_slam_amd import ba, ba_dist
    from suo_slam_amd import synthetic as S
    import torch.distributed as dist
    n_cam = n_cam_per_rank * world
    P = S.make_pose_graph(np.random.default_rng(5), n_cam, L)
    keys = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")

    def run():
        ts = []
        for _ in range(reps):
            full = ba.Problem(*[P[k].copy() for k in keys])
            t0 = time.perf_counter()
            ba_dist.optimize_distributed(full)
            ts.append(time.perf_counter() - t0)
        return full, min(ts)
    extra = {}
    if world == 1 and dist.is_initialized():
        # one rank: the collectives of the schedule are identities and ba_dist skips them -- time that, then the SAME call with every
        # all-reduce really issued on the one-rank RCCL group (what an 8-GPU node executes per trial), and compare the results bit for bit
        os.environ["SUO_FORCE_COLLECTIVES"] = "0"
        plain, t_plain = run()
        os.environ["SUO_FORCE_COLLECTIVES"] = "1"
        full, t = run()
        os.environ["SUO_FORCE_COLLECTIVES"] = "0"
        extra = {"ms_collectives_skipped": round(1e3 * t_plain, 2), "collectives": "every all-reduce issued on a one-rank %s group (SUO_FORCE_COLLECTIVES=1), in place on device buffers" % dist.get_backend(),
                 "identical_to_skipped": bool(np.array_equal(plain.cam_T, full.cam_T) and np.array_equal(plain.obj_T, full.obj_T) and np.array_equal(plain.inlier, full.inlier))}
    else:
        full, t = run()
    ts = [t]
    if world == 1:
        # the same adjustment through the ONE C call ObjectSLAM.optimize makes (suo_optimize: the phase kernels under the device-resident schedule, driven from C --
        # no Python between the launches, no collective)
        tc = []
        for _ in range(reps):
            one = ba.Problem(*[P[k].copy() for k in keys])
            t0 = time.perf_counter()
            ba.optimize_batch([one])
            tc.append(time.perf_counter() - t0)
        extra["suo_optimize_ms"] = round(1e3 * min(tc), 2)
        # (the LM trial count follows the last bits of the reduced system's solve -- +-25 % per problem between two correct builds, profiles/r06_global_ba_ab.txt --
        #  so the per-trial figure is the one to compare across builds)
        extra["suo_optimize_us_per_lm_trial"] = round(1e6 * min(tc) / max(int(one.stats[2]), 1), 1)
        extra["suo_optimize_identical"] = bool(np.array_equal(one.cam_T, full.cam_T) and np.array_equal(one.obj_T, full.obj_T) and np.array_equal(one.inlier, full.inlier))
    err = float(max(np.linalg.norm(full.obj_T.reshape(-1, 3, 4)[o][:, 3] - P["obj_gt"][o][:, 3]) for o in range(L)))
    return {**extra, "ranks": world, "cameras": n_cam, "objects": L, "edges": int(len(P["edge_cam"])), "ms": round(1e3 * min(ts), 2),
            "lm_trials": int(full.stats[2]), "collectives_per_trial": ba_dist.COLLECTIVES_PER_TRIAL,
            "reduce_bytes_per_trial": int(8 * ((6 * L) ** 2 + 6 * L + 4)),
            "max_object_translation_err_mm": round(err, 3), "inlier_edges": int(full.inlier.sum())}


def tless_leg(use_graph, depth, F=32, steps=8, warmup=3):
    """BASELINE configs[3]'s workload shape on one GPU: T-LESS frames (720 x 540), 8 detections per frame whose boxes come from a saved-detection file in Pix2Pose's
    (y1, x1, y2, x2) columns through suo_slam_amd.detections.load_pix2pose_results (lib/utils/utils.py:538-569, evaluate.py:104-125) -- five boxes <= 256 px, two of
    256-512 px and one > 512 px per frame, so the fused RoIAlign + stem launch runs its two- and three-samples-per-bin paths (torchvision's adaptive
    ceil(roi / 256), pkpnet.py:93) -- through the same timed region as `value` (frames resident in HBM), T-LESS thresholds (evaluate.py:66-74).
    Beside it: the stem launch alone at this shape and at the 640 x 480 / <= 240 px shape of the headline (HIP events)."""
    import ctypes as C
    import pickle
    import tempfile
    import torch
    from suo_slam_amd import _lib, detections
    from suo_slam_amd import synthetic as S
    from .common import _timed
    rng = np.random.default_rng(4242)
    pool = [S.make_frame_tless(rng) for _ in range(2 * F)]
    L = len(pool[0]["boxes"])
    # the boxes travel the way the reference reads them: a pickle of rois in (y1, x1, y2, x2), one entry per "scene/view"
    with tempfile.TemporaryDirectory() as root:
        os.makedirs(os.path.join(root, "saved_detections"))
        res = {"%d/%d" % (1, v): {"rois": fr["boxes"][:, [1, 0, 3, 2]].astype(np.float32), "labels_txt": ["obj_%d" % o for o in fr["obj_ids"]],
                                 "poses": [np.concatenate([fr["T_OtoC"][k][:3, :3], fr["T_OtoC"][k][:3, 3:4] / 1000.0], 1) for k in range(L)]} for v, fr in enumerate(pool)}
        with open(os.path.join(root, "saved_detections", "tless_pix2pose_retinanet_siso_top1.pkl"), "wb") as f:
            pickle.dump(res, f)
        det = detections.load_pix2pose_results(root)
    dmap = detections.build_detection_map(det)
    for v, fr in enumerate(pool):
        idx = [dmap[1][v][o] for o in fr["obj_ids"]]
        got = np.stack([det["bboxes"][i] for i in idx]).astype(np.float32)
        assert np.array_equal(got, fr["boxes"]), "the saved-detection loader did not hand back the boxes that were written"
        fr["boxes"] = got
    side = np.concatenate([np.maximum(fr["boxes"][:, 2] - fr["boxes"][:, 0], fr["boxes"][:, 3] - fr["boxes"][:, 1]) for fr in pool])
    pipe = FramePipeline(L, pool, F, use_graph=use_graph, depth=depth, resident=True)
    for i in range(warmup):
        pipe.step(i)
    pipe.drain(warmup)
    pipe.reset_metrics()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pipe.step(warmup + i)
    pipe.drain(warmup + steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": "T-LESS single-view eval shape (BASELINE configs[3], one GPU): 720x540 frames, %d saved detections per frame read through load_pix2pose_results "
                       "((y1,x1,y2,x2) columns), network + masks + PnP + LM as in `value`" % L,
           "frames_per_s": round(steps * F / dt, 2), "crops_per_s": round(steps * F * L / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3), "frames_per_step": F, "steps": steps,
           "box_longer_side_px": {"<=256": int((side <= 256).sum()), "256-512": int(((side > 256) & (side <= 512)).sum()), ">512": int((side > 512).sum()), "max": round(float(side.max()), 1)},
           "poses_accepted": pipe.n_pose, "keypoints_passed_by_the_masks": pipe.n_kp}
    # the stem launch alone (what the box sizes change): this shape against the headline's
    S0 = pipe.slots[0]
    net, st = S0["net"], S0["tstream"]
    lib, P = _lib.lib(), (lambda t: C.c_void_p(t.data_ptr()))
    LF = L * F
    kp, uv, cov = torch.empty((LF, 41), device="cuda"), torch.empty((LF, 41, 2), device="cuda"), torch.empty((LF, 41, 2, 2), device="cuda")
    imgs_t = pipe.d_imgs[:F]
    with torch.cuda.stream(st):
        def fwd(imgs, H, W, boxes):
            _lib.check(lib.suo_net_forward_frames(net._h, P(imgs), 0, H, W, P(boxes), P(S0["box_img"]), LF, None, P(uv), P(cov), P(kp), None, None, C.c_void_p(st.cuda_stream)), "fwd")
        call_t = _timed(lambda: fwd(imgs_t, pipe.H, pipe.W, S0["boxes"]), st, 8)
        rng2 = np.random.default_rng(7)
        ycb = [S.make_frame(rng2, L) for _ in range(F)]
        imgs_y = torch.from_numpy(np.stack([fr["image"] for fr in ycb])).cuda()
        boxes_y = torch.from_numpy(np.concatenate([fr["boxes"] for fr in ycb]).astype(np.float32)).cuda()
        call_y = _timed(lambda: fwd(imgs_y, 480, 640, boxes_y), st, 8)
    out["network_call_us"] = {"tless_720x540": round(call_t, 1), "ycbv_640x480": round(call_y, 1),
                              "note": "one suo_net_forward_frames call of %d crops under HIP events (stem + backbone + decode): the launches after the stem are the same, the difference is the fused RoIAlign + stem launch's sampling" % LF}
    del pipe
    return out

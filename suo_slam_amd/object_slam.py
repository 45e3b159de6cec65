"""Host-side mirror of the reference's ``ObjectSLAM`` driver (/root/reference/lib/object_slam.py:51-1167)
for the hot path: same public methods, argument order and state dictionaries (``detections``,
``cam_poses``, ``obj_poses``), so an ``evaluate.py``-shaped harness can call it unchanged:

    ObjectSLAM(chkpt_path, mesh_db, ...)                 object_slam.py:52-57
    reset()                                              :125-153
    process_view(view_id, img, K, obj_ids, bboxes, model_kps, model_kps_masks, kp_masks, uv_gt, cam_pose)  :327-451
    collect_results(last_only, no_viz, final)            :175-225 (visualisation is out of scope: no_viz only)
    optimize(curr_only)                                  :703-930
    pnp(points_3d, points_2d, camera_matrix)             :25-41

What differs is where the work runs: the network, the keypoint masks, PnP for all objects of the frame
and every LM round execute in libsuo_hip.so (HIP, gfx950); this file only keeps the reference's
bookkeeping (thresholds, acceptance / removal / re-initialisation rules).  There is no CPU fallback.
"""
from __future__ import annotations

from collections import defaultdict
from time import time

import os

import numpy as np

from . import ba as _ba
from . import lambdatwist as _lt
from . import slam_score as _sc
from .geometry import fix_K_for_bbox_ndc, fix_K_for_bbox_ndc_many, invert_SE3, normalize_uv, to4x4
from .weights import NUM_KP

CHI2_2DOF_95 = 5.991


class AverageMeter:
    """lib/utils/eval_meter.py:47-63."""

    def __init__(self):
        self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.sum += float(val) * n
        self.count += n

    def average(self):
        return self.sum / self.count if self.count > 0 else 0.0


def pnp(points_3d, points_2d, camera_matrix):
    """object_slam.py:25-41: PnP pose [3,4] + all-true inlier mask, or None (needs >= 4 points; the
    identity pose is the native failure code)."""
    assert points_3d.shape[0] == points_2d.shape[0], "points 3D and points 2D must have same number of rows"
    assert camera_matrix.shape == (3, 3), "Camera matrix must be of shape (3,3)"
    n = points_3d.shape[0]
    if n < 4:
        return None
    res = _lt.pnp(np.asarray(points_3d, np.float64), normalize_uv(np.asarray(points_2d, np.float64), camera_matrix))
    if np.allclose(res, np.eye(4)):
        return None
    return res[:3, :], np.ones(n, dtype=bool)


# ---- prior heat-maps (lib/utils/utils.py:356-411) ------------------------------------------------
def _gaussian_patch(size=91):
    """gaussian_2d(): a unit impulse blurred by cv2.GaussianBlur(ksize=size, sigma=0) and divided by its
    max.  OpenCV derives sigma = 0.3*((size-1)*0.5-1)+0.8 (= 14.0 for 91) and reflects at the border
    (BORDER_REFLECT_101), which doubles the outermost ring of the impulse response."""
    sigma = 0.3 * ((size - 1) * 0.5 - 1) + 0.8
    i = np.arange(size, dtype=np.float64) - (size - 1) / 2
    k = np.exp(-(i * i) / (2 * sigma * sigma))
    k /= k.sum()
    k[0] *= 2
    k[-1] *= 2
    g = np.outer(k, k)
    return (g / g.max()).astype(np.float32)


_PATCH = None


def draw_gaussian_2d(img, pt, sigma=15):
    """utils.py:364-385: paste (assign) the 91x91 patch around pt=(x,y); window [pt-45, pt+45)."""
    global _PATCH
    tmp = int(np.ceil(3 * sigma))
    ul = [int(np.floor(pt[0] - tmp)), int(np.floor(pt[1] - tmp))]
    br = [int(np.floor(pt[0] + tmp)), int(np.floor(pt[1] + tmp))]
    H, W = img.shape
    if ul[0] > W or ul[1] > H or br[0] < 1 or br[1] < 1:
        return img
    if _PATCH is None or _PATCH.shape[0] != 2 * tmp + 1:
        _PATCH = _gaussian_patch(2 * tmp + 1)
    gx = [max(0, -ul[0]), min(br[0], W) - max(0, ul[0]) + max(0, -ul[0])]
    gy = [max(0, -ul[1]), min(br[1], H) - max(0, ul[1]) + max(0, -ul[1])]
    ix = [max(0, ul[0]), min(br[0], W)]
    iy = [max(0, ul[1]), min(br[1], H)]
    img[iy[0]:iy[1], ix[0]:ix[1]] = _PATCH[gy[0]:gy[1], gx[0]:gx[1]]
    return img


def make_prior_kp_input(kp_uv, kp_uv_mask, img_shape, ndc=True):
    """utils.py:398-411: one Gaussian stamp per valid keypoint channel."""
    n = kp_uv.shape[0]
    vh, vw = img_shape
    x = np.zeros((n, vh, vw), dtype=np.float32)
    for i in range(n):
        if kp_uv_mask[i] and np.all(np.isfinite(kp_uv[i, :2])):
            u, v = float(kp_uv[i, 0]), float(kp_uv[i, 1])
            if ndc:
                u = (min(max(u, -1), 1) * vw / 2 + vw / 2) - 0.5
                v = vh - 0.5 - (min(max(v, -1), 1) * vh / 2 + vh / 2)
            draw_gaussian_2d(x[i], (int(round(u)), int(round(v))))
    return x


def _det_cache(d):
    """Padded copies of a detection's immutable arrays (keypoints, predictions, covariances, intrinsics), built once
    per detection: the scoring rules below run O(objects^2 + 15 objects) times per SLAM view and the graph assembly
    once per keypoint, which as per-detection numpy calls dominated a view's host time."""
    c = d.get("_cache")
    if c is not None and c["src"][0] is d["uv_pred"] and c["src"][1] is d["cov_pred"] and c["src"][2] is d["model_kp"] and c["src"][3] is d["K"]:
        return c
    n = int(d["uv_pred"].shape[0])
    nk = max(NUM_KP, n)
    pts = np.zeros((nk, 3))
    uv = np.zeros((nk, 2))
    pts[:n] = d["model_kp"]
    uv[:n] = d["uv_pred"]
    cov = None
    if d["cov_pred"] is not None:
        cov = np.zeros((nk, 2, 2))
        cov[:, 0, 0] = cov[:, 1, 1] = 1.0
        cov[:n] = np.asarray(d["cov_pred"], dtype=np.float64)
    c = {"src": (d["uv_pred"], d["cov_pred"], d["model_kp"], d["K"]), "n": n, "pts": pts, "uv": uv, "cov": cov,
         "K": np.asarray(d["K"], dtype=np.float64)}
    d["_cache"] = c
    return c


def _det_edges(d):
    """Per-keypoint graph-edge data of a detection (object_slam.py:795-821): information matrices [n,3] as
    (xx, xy, yy) and the pinhole parameters (fx, fy, cx, cy)."""
    c = _det_cache(d)
    if "info" not in c:
        Kd = c["K"]
        # (np.allclose's rule -- |a - b| <= 1e-8 + 1e-5 |b| -- on five scalars: the numpy call costs 15 us per detection on the view's critical path)
        assert (abs(Kd[0, 1]) <= 1e-8 and abs(Kd[1, 0]) <= 1e-8 and abs(Kd[2, 0]) <= 1e-8 and abs(Kd[2, 1]) <= 1e-8 and abs(Kd[2, 2] - 1.0) <= 1e-8 + 1e-5), \
            f"K matrix has off-diagonals!\n\n{Kd}"
        n = c["n"]
        if d["cov_pred"] is not None and n > 0:
            # np.linalg.inv(cov_uv[k]) in the covariance's OWN dtype (float32 from the network, :825-828), then g2o's doubles;
            # the kernels keep Omega as (xx, xy, yy): its symmetric part is all that error^T Omega error sees
            Om = np.linalg.inv(np.asarray(d["cov_pred"]).reshape(n, 2, 2)).astype(np.float64)
            info = np.stack([Om[:, 0, 0], 0.5 * (Om[:, 0, 1] + Om[:, 1, 0]), Om[:, 1, 1]], axis=1)
        else:
            info = np.tile(np.array([1.0, 0.0, 1.0]), (n, 1))
        c["info"] = info
        c["camk"] = np.array([Kd[0, 0], Kd[1, 1], Kd[0, 2], Kd[1, 2]])
    return c


class _EdgeRefs:
    """(view, object, keypoint) of every graph edge, kept as per-detection segments [(view, object, first edge, n)]."""

    def __init__(self, segs, n_edges):
        self.segs, self.n_edges = segs, n_edges

    def __len__(self):
        return self.n_edges

    def __iter__(self):
        for v, o, _, n in self.segs:
            for k in range(n):
                yield (v, o, k)


def _on_stream(fn):
    """Run a method's device work on the object's own (non-NULL) stream.  torch's default stream is the legacy NULL stream, on which libsuo_hip's network
    entries BLOCK (include/suo_hip.h: "NULL = internal stream + blocking"): measured 13.3 ms of host time per 128-crop call that an asynchronous call returns
    from in 0.3 ms.  Everything a method enqueues -- uploads, network, masks, geometry chain, read-backs -- goes to the one stream, so it stays ordered."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        st = getattr(self, "_gpu_stream", None)
        if st is None:
            return fn(self, *a, **kw)
        import torch
        with torch.cuda.stream(st):
            return fn(self, *a, **kw)
    return wrapped


class ObjectSLAM:
    def __init__(self, chkpt_path, mesh_db, no_network_cov=False, no_prior_det=False, pred_res=(256, 256),
                 debug_gt_kp=False, sfm_mode=False, single_view_mode=False, viz_cov=False, do_viz_extra=False,
                 global_opt_every=10, kp_var_thresh=0.2, bbox_thresh=0.9, bbox_inflate=0.0, manual_kp_std=0.005,
                 opt_init_with_outliers=False, give_all_prior=False, state_dict=None, max_crops=16, seed=0, verbose=False,
                 device_chain=True, run_network_in_debug=False, debug_gt_on_device=False):
        """Same keyword surface as the reference.  ``chkpt_path`` is a torch checkpoint whose ``['model']`` is the
        PkpNet state_dict (object_slam.py:92-97); ``state_dict`` may be given directly instead."""
        self.mesh_db = mesh_db
        self.no_network_cov = no_network_cov or debug_gt_kp
        self.no_prior_det = no_prior_det
        self.pred_res = list(pred_res)
        self.debug_gt_kp = debug_gt_kp
        self.sfm_mode = sfm_mode
        self.single_view_mode = single_view_mode
        self.slam_mode = not (sfm_mode or single_view_mode)
        self.global_opt_every = global_opt_every
        self.kp_var_thresh = kp_var_thresh
        self.bbox_thresh = bbox_thresh
        self.bbox_inflate = bbox_inflate
        self.manual_kp_std = manual_kp_std
        self.opt_init_with_outliers = opt_init_with_outliers
        self.give_all_prior = give_all_prior
        self.verbose = verbose
        # single-view frames: masks -> compaction -> PnP -> acceptance -> graph -> LM as ONE device chain behind the network
        # (suo_slam_amd/frame_geom.py); False = the host route that restates the reference's data flow (three read-backs, Python lists)
        self.device_chain = bool(device_chain)
        self._fg = None
        self._rng = np.random.default_rng(seed)
        self._pnp_seed = int(seed)
        self.reset()
        self.model = None
        self.model_epoch = -1
        # benchmark aid: in --debug_gt_kp mode ALSO run the network (both passes, device-rendered priors, masks, read-back) on the
        # frame's pixels and discard what it says in favour of the ground-truth keypoints -- random weights give meaningless
        # keypoints, a SLAM sequence needs meaningful ones, and the view's time must include the network
        self.run_network_in_debug = bool(run_network_in_debug and debug_gt_kp and state_dict is not None)
        # ... and, with debug_gt_on_device, make that substitution WHERE THE NETWORK'S OUTPUT LIES: the ground-truth keypoints (+ the same noise draws) overwrite the
        # device tensors in float32 -- the type the network emits -- and the view continues on the product route (_run_kp_model_chain), not on the host restatement
        self.debug_gt_on_device = bool(debug_gt_on_device and self.run_network_in_debug and device_chain)
        if not debug_gt_kp or self.run_network_in_debug:
            from .pkpnet import PkpNet
            if state_dict is None:
                import torch
                ck = torch.load(chkpt_path, map_location="cpu")
                state_dict = ck["model"]
                self.model_epoch = ck.get("epoch", -1)
            self.model = PkpNet(calc_cov=True, state_dict=state_dict, max_crops=max_crops)
            import torch
            self._gpu_stream = torch.cuda.Stream(device=self.model.device)
            # frames carry a varying number of detections: capture the graph of every crop count now, not inside a timed view
            self.model.prepare(with_priors=(False,) if (single_view_mode or no_prior_det) else (False, True))
        self.fp16_range_reissues = 0          # network calls re-issued on the bf16x3 form because an activation left fp16's range (reported by Evaluator.run / bench.py)
        self.avg_std_meter = AverageMeter()
        self.track_time_meter = AverageMeter()
        self.opt_time_meter = AverageMeter()
        self.all_time_num_views = 0

    def _log(self, *a):
        if self.verbose:
            print(*a)

    def reset(self):
        self.detections = {}
        self.cam_poses = {}
        self.view_ids = []
        self.cam_K = {}
        self.images = {}
        self.obj_poses = {}
        self.obj_num_dets = defaultdict(int)
        self.obj_num_det_kps = defaultdict(int)
        self.remove_penalty = defaultdict(int)
        self.needs_opt = False
        st = getattr(self, "_score_store", None)             # the scene's detections on the device (suo_slam_amd/slam_score.py)
        if st is not None:
            st.clear()

    def num_views_processed(self):
        return len(self.cam_poses)

    def obj_num_inliers(self, obj_id):
        n = 0
        for det in self.detections.values():
            n += int(np.count_nonzero(det.get(obj_id, {}).get("inliers", np.array([]))))
        return n

    def remove_obj(self, obj_id):
        self.obj_poses.pop(obj_id)

    def get_tracking_strtime(self, tt0=np.nan, tt1=np.nan):
        avg = self.track_time_meter.average()
        return f"TIMING: Tracking time: {1000 * (tt1 - tt0):.3f} ms ({1000 * avg:.3f} avg) ({'inf' if avg < 1e-12 else 1 / avg} Hz)"

    def get_global_opt_strtime(self, t0=np.nan, t1=np.nan):
        avg = self.opt_time_meter.average()
        return f"TIMING: Global opt time: {1000 * (t1 - t0)} ms ({1000 * avg} avg) ({'inf' if avg < 1e-12 else 1 / avg} Hz)"

    # ---------------------------------------------------------------------------------------------
    def collect_results(self, last_only=False, no_viz=True, final=False):
        """object_slam.py:175-225: T_OtoC = T_GtoC @ T_OtoG per view; score = 1 + total inliers."""
        if self.slam_mode and self.needs_opt and final:
            t0 = time()
            self.optimize()
            self.opt_time_meter.update(time() - t0)
        results = {}
        n_inl = {}                     # per object, the same for every view: counted once (the reference recounts per view, :199-214)
        assert len(self.view_ids) == len(self.cam_poses)
        for view_id in ([self.view_ids[-1]] if last_only else self.view_ids):
            T_GtoC = to4x4(self.cam_poses[view_id])
            detection = self.detections[view_id]
            results[view_id] = {"poses": {}}
            for obj_id in set(list(self.obj_poses.keys()) + list(detection.keys())):
                T_OtoC = None
                if obj_id in self.obj_poses:
                    T_OtoC = T_GtoC @ to4x4(self.obj_poses[obj_id])
                if obj_id not in n_inl:
                    n_inl[obj_id] = self.obj_num_inliers(obj_id)
                results[view_id]["poses"][obj_id] = {"T_OtoC": T_OtoC, "score": 1 + n_inl[obj_id]}
        return results

    # ---------------------------------------------------------------------------------------------
    @_on_stream
    def process_view(self, view_id, img, K, obj_ids, bboxes, model_kps, model_kps_masks, kp_masks, uv_gt=None, cam_pose=None):
        """object_slam.py:327-451."""
        assert view_id not in self.cam_poses, f"Repeat view_id {view_id}"
        if self.views_in_flight():                          # (their results carry PnP sampler keys that continue from _pnp_seed: a view in between would reuse them)
            raise RuntimeError("process_view: batches of submit_views_single are still in flight -- collect them first")
        import torch
        self._frame_key = self._frame_dev = None            # (the frame is uploaded once per view: _frame_on_device)
        if self.model is not None:
            torch.cuda.synchronize()
        tt0 = time()
        obj_ids = np.asarray(obj_ids)
        bboxes = np.array(bboxes, dtype=np.float64)
        model_kps = np.asarray(model_kps)
        model_kps_masks = np.asarray(model_kps_masks, dtype=bool)
        kp_masks = np.asarray(kp_masks, dtype=bool)
        self.cam_K[view_id] = K
        self.images[view_id] = img
        self.all_time_num_views += 1
        if not self.no_prior_det:
            is_sym = np.array([bool(self.mesh_db[o]["is_symmetric"]) for o in obj_ids], dtype=bool)
        else:
            is_sym = np.zeros(len(obj_ids), dtype=bool)
        if cam_pose is not None:
            self.cam_poses[view_id] = cam_pose
            self.view_ids.append(view_id)
            is_sym = np.ones(len(obj_ids), dtype=bool)
        if self.give_all_prior:
            is_sym = np.ones(len(obj_ids), dtype=bool)
        if self.single_view_mode:
            is_sym = np.zeros(len(obj_ids), dtype=bool)
        is_non_sym = ~is_sym
        n_sym, n_non_sym = int(is_sym.sum()), int(is_non_sym.sum())
        if cam_pose is None and not self.single_view_mode and len(self.view_ids) > 0 and n_non_sym == 0:
            self._backup_estimate_camera_pose(view_id, obj_ids, bboxes)
        self.needs_opt = True
        bboxes[:, [0, 1]] *= 1.0 - self.bbox_inflate
        bboxes[:, [2, 3]] *= 1.0 + self.bbox_inflate
        # (debug_gt_kp replaces the network's keypoints by the ground truth, :1129-1131: that substitution lives on the host route)
        if (self.single_view_mode and self.device_chain and self.model is not None and not self.debug_gt_kp and cam_pose is None and len(self.view_ids) == 0
                and not self.cam_poses and not self.obj_poses and 0 < len(obj_ids) <= 16):
            self._process_view_single_device(view_id, img, K, obj_ids, bboxes, model_kps, model_kps_masks)
            torch.cuda.synchronize()
            tt1 = time()
            if self.all_time_num_views > 5:
                self.track_time_meter.update(tt1 - tt0)        # network + PnP + refinement: one chain, not separable from the host
            self.needs_opt = False
            return

        def sub(mask):
            return (obj_ids[mask], bboxes[mask], model_kps[mask], model_kps_masks[mask], kp_masks[mask],
                    uv_gt[mask] if uv_gt is not None else None)
        chained = False
        if self._slam_view_takes_the_vote_chain(view_id, cam_pose, n_non_sym, n_sym):
            chained = self._process_view_slam_chain(view_id, img, K, sub(is_non_sym), sub(is_sym))
        elif n_non_sym > 0:
            self._process_objects(False, view_id, img, K, *sub(is_non_sym))
        if view_id not in self.cam_poses:
            if len(self.view_ids) == 0:
                self.view_ids.append(view_id)
                self.cam_poses[view_id] = np.eye(4)[:3, :]
            else:
                self._backup_estimate_camera_pose(view_id, obj_ids, bboxes)
        if n_sym > 0 and not chained and ((view_id in self.cam_poses) or self.no_prior_det):
            self._process_objects(True, view_id, img, K, *sub(is_sym))
        if not self.single_view_mode:
            self._maybe_reinit_objects(view_id, len(self.view_ids) if self.sfm_mode else 15)
            self.optimize(curr_only=True)
        if self.model is not None:
            torch.cuda.synchronize()
        tt1 = time()
        if self.all_time_num_views > 5:            # warm-up views are not timed (:425)
            self.track_time_meter.update(tt1 - tt0)
        if self.sfm_mode or self.single_view_mode or (len(self.view_ids) > 1 and len(self.view_ids) % self.global_opt_every == 0):
            t0 = time()
            self.optimize()
            self.opt_time_meter.update(time() - t0)
            self.needs_opt = False

    # ---------------------------------------------------------------------------------------------
    def _process_view_single_device(self, view_id, img, K, obj_ids, bboxes, model_kps, model_kps_masks):
        """A single-view frame (evaluate.py --nviews 1: __process_objects(False, ...) :464-593, __run_kp_model :1077-1167, then
        optimize() :703-930 with the camera fixed at identity) with everything between the network and the poses on the device
        (csrc/frame_geom.hip): one launch chain, one read-back.  Leaves the same state behind as the host route."""
        import torch
        from .frame_geom import FrameGeometry, kbbox_terms
        from .pkpnet import keypoint_masks
        L = len(obj_ids)
        K_bbox = fix_K_for_bbox_ndc_many(K, bboxes).astype(np.float32)                                   # float32 container (:1082)
        kinv, camk = kbbox_terms(K_bbox)
        min_depth = np.array([0.5 * self.mesh_db[o]["diameter"] for o in obj_ids], dtype=np.float64)
        if self._fg is None or self._fg.max_crops < L:
            self._fg = FrameGeometry(max(16, L), 1)
        its = (10, 10, 40, 40) if self.sfm_mode else (10, 10, 10, 10)                                    # (:843-846)
        for _attempt in range(2):
            pred = self.model(np.ascontiguousarray(img), [torch.as_tensor(np.asarray(bboxes, np.float32))], None, check=False)
            vt = 1e30 if self.no_network_cov else self.kp_var_thresh
            masks_dev = keypoint_masks(pred["uv"], pred["cov"], pred["kp_mask"], model_kps_masks, self.bbox_thresh, vt)
            kps_dev = torch.as_tensor(np.ascontiguousarray(model_kps, dtype=np.float32)).to(pred["uv"].device)
            self._fg.launch([0, L], pred["uv"], pred["cov"], masks_dev, kps_dev, kinv, camk, min_depth, seed=self._pnp_seed,
                            use_cov=not self.no_network_cov, do_lm=True, its=its)
            r = self._fg.fetch(copy=True)
            if not self.model.range_exceeded():               # (fp16 form only: an activation left its range -> the network is on bf16x3 now, once more)
                break
            self.fp16_range_reissues += 1
        self._pnp_seed += int(np.count_nonzero(r["n_kp"] >= 4))
        self._ingest_single_view(view_id, obj_ids, bboxes, model_kps, model_kps_masks, K_bbox, r, 0, 0)

    def _ingest_single_view(self, view_id, obj_ids, bboxes, model_kps, model_kps_masks, K_bbox, r, lo, frame):
        """State of a single-view frame from the device chain's read-back (crops [lo, lo + L) of launch result r, frame index `frame`)."""
        detection = {}
        for k, obj_id in enumerate(obj_ids):
            c = lo + k
            m = r["mask"][c]
            n = int(r["n_kp"][c])
            cov_pred = None if self.no_network_cov else r["cov"][c][m]
            pose = r["T_pnp"][c].copy() if r["accepted"][c] else None
            if cov_pred is not None and cov_pred.size > 0:
                std = np.sqrt(cov_pred[..., [0, 1], [0, 1]])
                self.avg_std_meter.update(std.mean(), std.size)
            self.obj_num_dets[obj_id] += 1
            self.obj_num_det_kps[obj_id] += n
            assert obj_id not in self.obj_poses and obj_id not in detection, f"Object {obj_id} is in detections twice! obj_id must be an instance label."
            detection[obj_id] = {"bbox": bboxes[k], "model_kp_mask": model_kps_masks[k], "prior_uv": None, "pose": pose,
                                 "inliers": r["inlier"][c, :n].copy(), "kp_mask": m, "model_kp": model_kps[k][m].astype(np.float64), "uv_gt": None,
                                 "uv_pred": r["uv"][c][m].astype(np.float64), "cov_pred": cov_pred, "K": K_bbox[k].astype(np.float64),
                                 "score": 0.0 if n == 0 else 1.0}
            if pose is not None:
                self.obj_poses[obj_id] = r["T_opt"][c].copy()
        self.detections[view_id] = detection
        self.cam_poses[view_id] = np.eye(4)[:3, :]
        self.view_ids.append(view_id)
        self.last_lm_stats = r["lm_stats"][frame].copy()
        t0 = time()
        self._cull_after_optimize([o for k, o in enumerate(obj_ids) if r["accepted"][lo + k]], False, view_id)
        self.opt_time_meter.update(time() - t0)

    def single_views_take_the_device_chain(self, views):
        """True when process_views_single can run `views` as ONE device call (else it processes them one by one)."""
        if not (self.single_view_mode and self.device_chain and self.model is not None and not self.debug_gt_kp and len(views) > 1):
            return False
        shape = np.asarray(views[0][1]).shape
        if sum(len(v[3]) for v in views) > self.model.max_crops:      # more crops than the network was built for: view by view (ObjectSLAM(max_crops=...) lifts it)
            return False
        return all(0 < len(v[3]) <= 16 and np.asarray(v[1]).shape == shape and np.asarray(v[1]).dtype == np.uint8 for v in views)

    @_on_stream
    def process_views_single(self, views):
        """Single-view evaluation (evaluate.py --nviews 1: reset / process_view / collect_results per reference view, evaluate.py:338-395) of SEVERAL
        independent views in one device call: `views` = [(view_id, img, K, obj_ids, bboxes, model_kps, model_kps_masks, kp_masks), ...].
        Returns [collect_results() of view 0, of view 1, ...] -- what the per-view loop returns: everything downstream of the network is bit for bit
        the per-view loop's on the same network outputs (one geometry launch; the PnP sampler's keys continue from frame to frame as the per-view
        loop advances its seed, csrc/pnp.hip); the shared network call picks its kernels by launch size, so its keypoints agree with the
        per-view calls' to the network's tolerance (1e-5 of the reference either way).  tests/test_gpu_evaluator.py holds both.
        = submit_views_single + collect_views_single; a caller with more batches to come submits the next one BEFORE collecting this one
        (Evaluator.run does), so that the host's share of a batch -- bookkeeping of the results, preparation of the next -- runs under the device's.
        Refuses to run with batches of an earlier submit_views_single still in flight: collect_views_single hands back the OLDEST batch, which would be zipped
        against these views."""
        if self.views_in_flight():
            raise RuntimeError("process_views_single: %d batch(es) submitted earlier are still in flight -- collect_views_single() / drain_views_single() them first"
                               % self.views_in_flight())
        if not self.single_views_take_the_device_chain(views):
            self.drain_views_single()
            out = []
            for v in views:
                self.reset()
                self.process_view(*v[:8])
                out.append(self.collect_results(no_viz=True))
            return out
        self.submit_views_single(views)
        return self.collect_views_single()

    def views_in_flight(self):
        return len(getattr(self, "_tickets", ()))

    @_on_stream
    def drain_views_single(self):
        """Collect (and drop) whatever submit_views_single left in flight."""
        while self.views_in_flight():
            self.collect_views_single()

    def _enqueue_views(self, prep, ff):
        """Network + masks + geometry chain of one prepared batch on the current stream; nothing here waits for the device."""
        import torch
        from .frame_geom import FrameGeometry, kbbox_terms
        from .pkpnet import keypoint_masks
        Ltot, B = ff[-1], len(prep)
        ring = getattr(self, "_fg_ring", None)
        if ring is None:
            ring = self._fg_ring = {"ctx": [None, None], "next": 0}
        k = ring["next"]
        ring["next"] = (k + 1) % 2
        fg = ring["ctx"][k]
        if fg is None or fg.max_crops < Ltot or fg.max_frames < B:
            fg = ring["ctx"][k] = FrameGeometry(max(256, Ltot), max(32, B))
        if getattr(self, "_seed_run", None) is None:
            self._seed_run = torch.zeros(1, dtype=torch.int64, device=self.model.device)      # the sampler's running key, device-resident
            self._seed_base = self._pnp_seed
        K_all = np.concatenate([p[7] for p in prep])
        kinv, camk = kbbox_terms(K_all)
        min_depth = np.array([0.5 * self.mesh_db[o]["diameter"] for p in prep for o in p[3]], dtype=np.float64)
        mm_all = np.concatenate([p[6] for p in prep]).astype(np.uint8)
        kps_all = np.ascontiguousarray(np.concatenate([p[5] for p in prep]), dtype=np.float32)
        pred = self.model.forward_frames([np.ascontiguousarray(p[1]) for p in prep], [np.asarray(p[4], np.float32) for p in prep], check=False,
                                         extra=[mm_all, kps_all])
        mm_dev, kps_dev = pred["extra"]
        vt = 1e30 if self.no_network_cov else self.kp_var_thresh
        masks_dev = keypoint_masks(pred["uv"], pred["cov"], pred["kp_mask"], mm_dev, self.bbox_thresh, vt)
        its = (10, 10, 40, 40) if self.sfm_mode else (10, 10, 10, 10)
        fg.launch(ff, pred["uv"], pred["cov"], masks_dev, kps_dev, kinv, camk, min_depth, seed=self._seed_base, seed_dev=self._seed_run,
                  use_cov=not self.no_network_cov, do_lm=True, its=its)
        return fg, pred

    @_on_stream
    def submit_views_single(self, views):
        """First half of process_views_single: prepare the batch and enqueue its device work.  Up to two batches may be in flight."""
        import torch
        assert self.single_views_take_the_device_chain(views), "submit_views_single: this batch does not take the device chain (process_views_single decides)"
        assert self.views_in_flight() < 2, "two batches are already in flight: collect one first"
        if not self.views_in_flight():
            # nothing in flight: the host's seed is complete -- (re)base the device-resident key on it (another route may have advanced it meanwhile)
            if getattr(self, "_seed_run", None) is not None and getattr(self, "_seed_expect", 0) + self._seed_base != self._pnp_seed:
                self._seed_run.zero_()
                self._seed_base, self._seed_expect = self._pnp_seed, 0
            elif getattr(self, "_seed_run", None) is None:
                self._seed_expect = 0
        prep, ff = [], [0]
        for view_id, img, K, obj_ids, bboxes, model_kps, model_kps_masks, _ in views:
            obj_ids = np.asarray(obj_ids)
            bboxes = np.array(bboxes, dtype=np.float64)
            bboxes[:, [0, 1]] *= 1.0 - self.bbox_inflate                                               # (process_view, :368-369)
            bboxes[:, [2, 3]] *= 1.0 + self.bbox_inflate
            K_bbox = fix_K_for_bbox_ndc_many(K, bboxes).astype(np.float32)
            prep.append((view_id, img, K, obj_ids, bboxes, np.asarray(model_kps), np.asarray(model_kps_masks, dtype=bool), K_bbox))
            ff.append(ff[-1] + len(obj_ids))
        assert ff[-1] <= self.model.max_crops, f"{ff[-1]} crops in one call, the network was built for {self.model.max_crops} (ObjectSLAM(max_crops=...))"
        fg, pred = self._enqueue_views(prep, ff)
        if not hasattr(self, "_tickets"):
            self._tickets = []
        self._tickets.append({"prep": prep, "ff": ff, "fg": fg, "pred": pred, "t0": time()})

    @_on_stream
    def collect_views_single(self):
        """Second half: wait for the OLDEST batch in flight, install its state view by view and return [collect_results() per view]."""
        tk = self._tickets[0]
        r = tk["fg"].fetch(copy=True)
        if self.model.range_exceeded():
            # fp16 form only: an activation left its range -- every batch enqueued and not yet checked is invalid (and so is what they added to the running
            # key).  The network is on bf16x3 now: re-issue all of them in order from the host's seed, which only ever counted valid batches.
            redo = self._tickets
            self._tickets = []
            self.fp16_range_reissues += len(redo)
            import torch
            torch.cuda.synchronize()
            self.model.range_exceeded()
            self._seed_run.zero_()
            self._seed_base, self._seed_expect = self._pnp_seed, 0
            for t in redo:
                fg, pred = self._enqueue_views(t["prep"], t["ff"])
                self._tickets.append({"prep": t["prep"], "ff": t["ff"], "fg": fg, "pred": pred, "t0": t["t0"]})
            tk = self._tickets[0]
            r = tk["fg"].fetch(copy=True)
        self._tickets.pop(0)
        prep, ff = tk["prep"], tk["ff"]
        n_solv = int(np.count_nonzero(r["n_kp"] >= 4))
        self._pnp_seed += n_solv
        self._seed_expect += n_solv
        now = time()
        per_view = (now - max(tk["t0"], getattr(self, "_last_collect", 0.0))) / len(prep)              # batches overlap: the time this batch added to the stream of results
        self._last_collect = now
        out = []
        for f, (view_id, img, K, obj_ids, bboxes, model_kps, model_kps_masks, K_bbox) in enumerate(prep):
            self.reset()
            self.cam_K[view_id] = K
            self.images[view_id] = img
            self.all_time_num_views += 1
            self._ingest_single_view(view_id, obj_ids, bboxes, model_kps, model_kps_masks, K_bbox, r, ff[f], f)
            if self.all_time_num_views > 5:
                self.track_time_meter.update(per_view)
            self.needs_opt = False
            out.append(self.collect_results(no_viz=True))
        return out

    def _cull_after_optimize(self, graph_objs, curr_only, view_curr):
        """object_slam.py:904-930: objects whose centre fell behind 0.5 diameter in the current view, then objects with too few inliers."""
        if not curr_only:
            for o in graph_objs:
                if view_curr in self.cam_poses:
                    p = self.cam_poses[view_curr][:3, :3] @ self.obj_poses[o][:3, 3] + self.cam_poses[view_curr][:3, 3]
                    if p[2] < 0.5 * self.mesh_db[o]["diameter"]:
                        self.remove_obj(o)
        n_inl = defaultdict(int)                                 # obj_num_inliers of every object in one pass over the detections
        for det in self.detections.values():
            for o, d in det.items():
                n_inl[o] += int(np.count_nonzero(d["inliers"]))
        for o in list(self.obj_poses.keys()):
            need = 3 if self.obj_num_dets[o] < 3 else 6
            if n_inl[o] < need:
                self.remove_obj(o)

    # ---------------------------------------------------------------------------------------------
    def _process_objects(self, is_sym, view_id, img, K, obj_ids, bboxes, model_kps, model_kps_masks, kp_masks, uv_gt=None):
        """object_slam.py:464-593."""
        if len(obj_ids) == 0:
            return
        prior_dets = prior_det_uv = None
        if is_sym and (not self.no_prior_det) and (view_id in self.cam_poses):
            prior_dets, prior_det_uv = {}, {}
            T_GtoC = to4x4(self.cam_poses[view_id])
            for k, obj_id in enumerate(obj_ids):
                if obj_id not in self.obj_poses:
                    continue
                m = model_kps_masks[k]
                T_OtoC = T_GtoC @ to4x4(self.obj_poses[obj_id])
                kps_in_C = model_kps[k][m] @ T_OtoC[:3, :3].T + T_OtoC[:3, 3]
                uvd = kps_in_C @ fix_K_for_bbox_ndc(K, bboxes[k]).T
                if np.all(uvd[:, 2] > 0):
                    full = np.zeros((m.shape[0], 2), dtype=np.float32)
                    full[m] = uvd[:, :2] / uvd[:, 2:3]
                    prior_det_uv[obj_id] = full
                    prior_dets[obj_id] = (full, m.astype(np.uint8))        # rendered on the device (pkpnet.PkpNet.forward)
        kp_det = self._run_kp_model(view_id, img, K, obj_ids, bboxes, model_kps, model_kps_masks, kp_masks, uv_gt, prior_dets)
        self._install_kp_detections(view_id, obj_ids, bboxes, model_kps_masks, kp_det, prior_det_uv)

    _VOTE_ON_HOST = object()

    def _install_kp_detections(self, view_id, obj_ids, bboxes, model_kps_masks, kp_det, prior_det_uv, cam_vote=_VOTE_ON_HOST):
        """Second half of __process_objects (object_slam.py:527-593): the detections of a pass into the state, the view's camera pose from the hypothesis vote when it
        has none yet, new objects into the map.  cam_vote: the vote's outcome when it was taken on the device (csrc/slam_vote.hip: the pose or None), else the host votes."""
        if not self.no_network_cov:
            for det in kp_det:
                if det["cov_pred"] is not None and det["cov_pred"].size > 0:
                    std = np.sqrt(det["cov_pred"][..., [0, 1], [0, 1]])
                    self.avg_std_meter.update(std.mean(), std.size)
        detection = {}
        for k, obj_id in enumerate(obj_ids):
            detection[obj_id] = {"bbox": bboxes[k], "model_kp_mask": model_kps_masks[k],
                                 "prior_uv": prior_det_uv.get(obj_id) if prior_det_uv is not None else None}
            detection[obj_id].update(kp_det[k])
            if self.num_views_processed() == 0:
                assert obj_id not in self.obj_poses, f"Object {obj_id} is in detections twice! obj_id must be an instance label."
                if detection[obj_id]["pose"] is not None:
                    T_OtoC = detection[obj_id]["pose"]
                    self.obj_poses[obj_id] = (invert_SE3(to4x4(self.cam_poses[view_id])) @ T_OtoC) if view_id in self.cam_poses else T_OtoC
        if view_id in self.detections:
            for obj_id in obj_ids:
                assert obj_id not in self.detections[view_id], "Object has already been processed for this view"
                self.detections[view_id][obj_id] = detection[obj_id]
        else:
            self.detections[view_id] = detection
        if view_id not in self.cam_poses:
            if self.num_views_processed() == 0:
                self.cam_poses[view_id] = np.eye(4)[:3, :]
            else:
                cam_pose = self._estimate_camera_pose(view_id) if cam_vote is self._VOTE_ON_HOST else cam_vote
                if cam_pose is None:
                    return
                self.cam_poses[view_id] = cam_pose
            self.view_ids.append(view_id)
        for obj_id in obj_ids:
            if obj_id not in self.obj_poses and detection[obj_id]["pose"] is not None:
                self.obj_poses[obj_id] = invert_SE3(to4x4(self.cam_poses[view_id])) @ detection[obj_id]["pose"]

    # ---------------------------------------------------------------------------------------------
    def _frame_on_device(self, img):
        """The frame of the current view on the device: uploaded once (pinned staging + copy kernel, pkpnet.PkpNet._to_device) and handed to
        BOTH network passes of a SLAM view (the reference uploads the full frame per pass, lib/object_slam.py:1092-1098)."""
        import torch
        key = (id(img), getattr(img, "shape", None))
        if getattr(self, "_frame_key", None) != key or self._frame_dev is None:
            host = np.ascontiguousarray(img)
            self._frame_dev = self.model._to_device(torch.from_numpy(host)) if isinstance(host, np.ndarray) and host.dtype == np.uint8 else host
            self._frame_key = key
        return self._frame_dev

    def _run_kp_model(self, view_id, img, K, obj_ids, bboxes, model_kps, model_kps_masks, kp_masks_gt=None, uv_gt=None, prior_dets=None):
        """object_slam.py:1077-1167.  Network + masks on the GPU, then ONE batched PnP launch for all
        objects of the frame (the reference loops lambdatwist.pnp per object)."""
        L = len(obj_ids)
        K_bbox = fix_K_for_bbox_ndc_many(K, bboxes).astype(np.float32)            # float32 container as in the reference (:1082)
        if self.device_chain and self.model is not None and (not self.debug_gt_kp or self.debug_gt_on_device):
            return self._run_kp_model_chain(img, K_bbox, obj_ids, bboxes, model_kps, model_kps_masks, kp_masks_gt, uv_gt, prior_dets)
        cov_uv = None
        if not self.debug_gt_kp or self.run_network_in_debug:
            import torch
            from .pkpnet import keypoint_masks
            prior_uv = prior_mask = None
            if prior_dets:
                # the reference stamps the heat-maps on the host (make_prior_kp_input) and uploads [L,41,256,256];
                # here the projected keypoints go to the device and the stamps are rendered while the crop is staged
                prior_uv = np.zeros((L, NUM_KP, 2), dtype=np.float32)
                prior_mask = np.zeros((L, NUM_KP), dtype=np.uint8)
                for k, obj_id in enumerate(obj_ids):
                    if obj_id in prior_dets:
                        prior_uv[k], prior_mask[k] = prior_dets[obj_id]
            if self.no_network_cov:
                bt, vt = self.bbox_thresh, 1e30
            else:
                bt, vt = self.bbox_thresh, self.kp_var_thresh
            for _attempt in range(2):
                pred = self.model(self._frame_on_device(img), [torch.as_tensor(np.asarray(bboxes, np.float32))], None,
                                  prior_uv=prior_uv, prior_mask=prior_mask, check=False)
                masks_dev = keypoint_masks(pred["uv"], pred["cov"], pred["kp_mask"], model_kps_masks, bt, vt)
                # (three small read-backs, as the reference does, :1100-1111: packing them with torch.cat first costs more host time in
                #  torch's dispatcher -- 4 extra ops, +0.2 ms per call on the GPU boxes -- than the two stream waits it saves)
                exp_uv = pred["uv"].cpu().numpy()
                kp_masks = masks_dev.cpu().numpy().astype(bool)
                if not self.no_network_cov or self.run_network_in_debug:
                    cov_uv = pred["cov"].cpu().numpy()
                if not self.model.range_exceeded():           # (fp16 form only: the read-backs above synchronised; on True the network is on bf16x3 now)
                    break
                self.fp16_range_reissues += 1
        if self.debug_gt_kp:
            assert kp_masks_gt is not None and uv_gt is not None
            kp_masks = np.asarray(kp_masks_gt, dtype=bool)
            cov_uv = None
        per_obj = []
        for k in range(L):
            m = kp_masks[k]
            if not self.debug_gt_kp:
                uv_pred = exp_uv[k][m].astype(np.float64)
            else:
                uv_pred = uv_gt[k][m].astype(np.float64)
                uv_pred = uv_pred + self._rng.normal(scale=0.01, size=uv_pred.shape)       # :1129-1131
            per_obj.append((uv_pred, cov_uv[k][m] if cov_uv is not None else None, model_kps[k][m].astype(np.float64),
                            K_bbox[k].astype(np.float64)))
        # batched PnP: objects with < 4 points are failures by definition (:31)
        idx = [k for k in range(L) if per_obj[k][0].shape[0] >= 4]
        poses = {}
        if idx:
            # normalize_uv per object (:34-36) with the L inverses taken in one stacked call (the same LAPACK routine per matrix)
            KinvT = np.linalg.inv(K_bbox.astype(np.float64)).transpose(0, 2, 1)
            T, status = _lt.pnp_batch([per_obj[k][2] for k in idx], [per_obj[k][0] @ KinvT[k][:2, :2] + KinvT[k][2:3, :2] for k in idx],
                                      0.001, seed=self._pnp_seed)
            self._pnp_seed += len(idx)
            for j, k in enumerate(idx):
                if not np.allclose(T[j], np.eye(4)):
                    poses[k] = T[j]
        ret = []
        for k, obj_id in enumerate(obj_ids):
            uv_pred, cov_pred, kp_model, K_kp = per_obj[k]
            inliers = np.ones(uv_pred.shape[0], dtype=bool)
            pose = None
            if k in poses and poses[k][2, 3] > 0.5 * self.mesh_db[obj_id]["diameter"] and inliers.sum() >= 4:      # :1147-1148
                pose = poses[k]
            self.obj_num_dets[obj_id] += 1
            self.obj_num_det_kps[obj_id] += uv_pred.shape[0]
            ret.append({"pose": pose, "inliers": inliers, "kp_mask": kp_masks[k], "model_kp": kp_model, "uv_gt": uv_gt,
                        "uv_pred": uv_pred, "cov_pred": cov_pred, "K": K_kp,
                        "score": 0.0 if inliers.size == 0 else float(inliers.astype(np.float32).mean())})
        return ret

    def _run_kp_model_chain(self, img, K_bbox, obj_ids, bboxes, model_kps, model_kps_masks, kp_masks_gt, uv_gt, prior_dets):
        """__run_kp_model (object_slam.py:1077-1167) of a SLAM pass with everything between the network and the PnP poses on the device: network -> masks ->
        compaction -> normalisation -> batched PnP -> acceptance as ONE stream-ordered chain (suo_frame_geom_launch with do_lm = 0: the camera hypotheses of
        :975-1072 continue on the host) and ONE read-back, where the host route makes three read-backs, compacts in Python and ships the points back for the PnP
        launch.  Same kernels on the same numbers: PnP poses and statuses are those of the host route bit for bit (tests/test_gpu_frame_geom.py)."""
        import torch
        from .frame_geom import FrameGeometry, kbbox_terms
        from .pkpnet import keypoint_masks
        L = len(obj_ids)
        prior_uv = prior_mask = None
        if prior_dets:
            prior_uv = np.zeros((L, NUM_KP, 2), dtype=np.float32)
            prior_mask = np.zeros((L, NUM_KP), dtype=np.uint8)
            for k, obj_id in enumerate(obj_ids):
                if obj_id in prior_dets:
                    prior_uv[k], prior_mask[k] = prior_dets[obj_id]
        kinv, camk = kbbox_terms(K_bbox)
        min_depth = np.array([0.5 * self.mesh_db[o]["diameter"] for o in obj_ids], dtype=np.float64)
        if self._fg is None or self._fg.max_crops < L:
            self._fg = FrameGeometry(max(16, L), 1)
        vt = 1e30 if self.no_network_cov else self.kp_var_thresh
        gt_uv = gt_mask = None
        if self.debug_gt_kp:                                  # (debug_gt_on_device: the same draws, in the same order, as the host route's :1129-1131)
            gt_mask = np.ascontiguousarray(kp_masks_gt, dtype=np.uint8)
            gt_uv = np.zeros((L, NUM_KP, 2), dtype=np.float32)
            for k in range(L):
                m = gt_mask[k].astype(bool)
                u = uv_gt[k][m].astype(np.float64)
                gt_uv[k][m] = (u + self._rng.normal(scale=0.01, size=u.shape)).astype(np.float32)
        # every small host array of the pass in ONE pinned block and ONE stream-ordered copy kernel, enqueued BEFORE the network: a pageable .to(device) per array
        # blocks the host until it has run -- in front of the network that is tens of microseconds each on the critical path, behind it a wait for the network
        host_arrays = [np.ascontiguousarray(model_kps, dtype=np.float32), np.ascontiguousarray(bboxes, dtype=np.float32),
                       gt_mask if gt_uv is not None else np.ascontiguousarray(model_kps_masks, dtype=np.uint8)]
        if gt_uv is not None:
            host_arrays.append(gt_uv)
        if prior_uv is not None:
            host_arrays += [prior_uv, prior_mask]
        for _attempt in range(2):
            frame = self._frame_on_device(img)
            st = self.model.stage_block(host_arrays)
            kps_dev, bx_dev, mm_dev = st[0], st[1], st[2]
            puv_dev, pmk_dev = (st[-2], st[-1]) if prior_uv is not None else (None, None)
            pred = self.model(frame, [bx_dev], None, prior_uv=puv_dev, prior_mask=pmk_dev, check=False)
            if gt_uv is not None:
                uv_dev, masks_dev = st[3], mm_dev
            else:
                uv_dev, masks_dev = pred["uv"], keypoint_masks(pred["uv"], pred["cov"], pred["kp_mask"], mm_dev, self.bbox_thresh, vt)
            self._fg.launch([0, L], uv_dev, pred["cov"], masks_dev, kps_dev, kinv, camk, min_depth, seed=self._pnp_seed,
                            use_cov=not self.no_network_cov, do_lm=False)
            r = self._fg.fetch(copy=True)
            if not self.model.range_exceeded():               # (fp16 form only: the fetch synchronised; on True the network is on bf16x3 now, once more)
                break
            self.fp16_range_reissues += 1
        self._pnp_seed += int(np.count_nonzero(r["n_kp"] >= 4))
        return self._kp_det_from_chain(r, obj_ids, model_kps, K_bbox, uv_gt)

    def _kp_det_from_chain(self, r, obj_ids, model_kps, K_bbox, uv_gt):
        """What __run_kp_model returns per object (:1150-1165), from a chain read-back."""
        ret = []
        for k, obj_id in enumerate(obj_ids):
            m = r["mask"][k]
            n = int(r["n_kp"][k])
            self.obj_num_dets[obj_id] += 1
            self.obj_num_det_kps[obj_id] += n
            ret.append({"pose": r["T_pnp"][k].copy() if r["accepted"][k] else None, "inliers": np.ones(n, dtype=bool), "kp_mask": m,
                        "model_kp": model_kps[k][m].astype(np.float64), "uv_gt": uv_gt, "uv_pred": r["uv"][k][m].astype(np.float64),
                        "cov_pred": None if self.no_network_cov else r["cov"][k][m], "K": K_bbox[k].astype(np.float64), "score": 0.0 if n == 0 else 1.0})
        return ret

    def _slam_view_takes_the_vote_chain(self, view_id, cam_pose, n_non_sym, n_sym):
        """Both passes of a SLAM view as ONE device chain (pass A -> PnP -> hypothesis vote -> prior projection -> pass B): a tracking view of a running map with
        objects of both kinds, on the routes that keep their keypoints on the device.  SUO_SLAM_VOTE_CHAIN=0: the host votes between the passes (A/B)."""
        return (self.device_chain and self.model is not None and (not self.debug_gt_kp or self.debug_gt_on_device) and not self.single_view_mode and cam_pose is None
                and not self.no_prior_det and 0 < n_non_sym <= 16 and 0 < n_sym <= 16 and self.num_views_processed() > 0 and view_id not in self.cam_poses
                and os.environ.get("SUO_SLAM_VOTE_CHAIN", "1") not in ("", "0"))

    def _process_view_slam_chain(self, view_id, img, K, A, B):
        """A SLAM view's two network passes with NOTHING on the host between them (round 6; lib/object_slam.py:464-593 twice, :975-1072 between): pass A (the
        non-symmetric objects) -> masks -> compaction -> PnP -> acceptance (csrc/frame_geom.hip) -> camera-hypothesis vote + projection of the symmetric objects' prior
        keypoints (csrc/slam_vote.hip) -> pass B with device-rendered priors -> its PnP, enqueued back to back; the host reads pass A's block and the vote while pass B
        runs and does pass A's bookkeeping under it.  A / B = (obj_ids, bboxes, model_kps, model_kps_masks, kp_masks, uv_gt) of the two passes.  Leaves the state the
        two _process_objects calls leave; when no hypothesis reaches four inliers (:1067) it returns False with pass A installed, and the caller continues as the
        reference does (__backup_estimate_camera_pose, then pass B again with that pose)."""
        import ctypes as C
        import torch
        from . import _lib
        from .frame_geom import FrameGeometry, kbbox_terms
        from .pkpnet import keypoint_masks
        lib = _lib.lib()
        ids_a, bb_a, kps_a, mm_a, gtm_a, gtu_a = A
        ids_b, bb_b, kps_b, mm_b, gtm_b, gtu_b = B
        La, Lb = len(ids_a), len(ids_b)
        if self._fg is None or self._fg.max_crops < max(La, Lb):
            self._fg = FrameGeometry(max(16, La, Lb), 1)
        if getattr(self, "_fg2", None) is None or self._fg2.max_crops < max(La, Lb):
            self._fg2 = FrameGeometry(max(16, La, Lb), 1)
        vt = 1e30 if self.no_network_cov else self.kp_var_thresh
        debug = self.debug_gt_kp
        P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

        def gt_arrays(L, gtm, gtu):                           # (debug_gt_on_device: the host route's draws, in its order -- pass A's objects, then pass B's)
            mask = np.ascontiguousarray(gtm, dtype=np.uint8)
            uv = np.zeros((L, NUM_KP, 2), dtype=np.float32)
            for k in range(L):
                m = mask[k].astype(bool)
                u = gtu[k][m].astype(np.float64)
                uv[k][m] = (u + self._rng.normal(scale=0.01, size=u.shape)).astype(np.float32)
            return mask, uv
        # pass A's host arrays now; pass B's and the vote's block are prepared AFTER pass A is enqueued, under its GPU time
        Kb_a = fix_K_for_bbox_ndc_many(K, bb_a).astype(np.float32)            # float32 container (:1082)
        kinv_a, camk_a = kbbox_terms(Kb_a)
        md_a = np.array([0.5 * self.mesh_db[o]["diameter"] for o in ids_a], dtype=np.float64)
        host_a = [np.ascontiguousarray(kps_a, dtype=np.float32), np.ascontiguousarray(bb_a, dtype=np.float32), np.ascontiguousarray(mm_a, dtype=np.uint8)]
        if debug:
            host_a += list(gt_arrays(La, gtm_a, gtu_a))
        rng_after_a = self._rng.bit_generator.state if debug else None      # (a pass B that has to be issued again draws its noise again: from here)
        host_b = None
        for _attempt in range(2):
            frame = self._frame_on_device(img)
            sa = self.model.stage_block(host_a)
            dev = sa[0].device
            seed_run = torch.zeros(1, dtype=torch.int64, device=dev)
            # ---- pass A
            pa = self.model(frame, [sa[1]], None, check=False, out_slot="slam A")
            uv_a, mk_a = (sa[4], sa[3]) if debug else (pa["uv"], keypoint_masks(pa["uv"], pa["cov"], pa["kp_mask"], sa[2], self.bbox_thresh, vt))
            self._fg.launch([0, La], uv_a, pa["cov"], mk_a, sa[0], kinv_a, camk_a, md_a, seed=self._pnp_seed, use_cov=not self.no_network_cov, do_lm=False,
                            seed_dev=seed_run)
            ra_dev = self._fg.device_result()
            if host_b is None:
                Kb_b = fix_K_for_bbox_ndc_many(K, bb_b).astype(np.float32)
                kinv_b, camk_b = kbbox_terms(Kb_b)
                md_b = np.array([0.5 * self.mesh_db[o]["diameter"] for o in ids_b], dtype=np.float64)
                # the vote's host block (include/suo_hip.h: SUO_SLAM_VOTE_BLOCK): map poses and intrinsics of both passes' objects
                blk = np.zeros(704)
                for k, o in enumerate(ids_a):
                    if o in self.obj_poses:
                        blk[k] = 1.0
                        blk[16 + 12 * k:28 + 12 * k] = np.asarray(self.obj_poses[o], dtype=np.float64)[:3, :4].reshape(-1)
                    blk[208 + 9 * k:217 + 9 * k] = Kb_a[k].astype(np.float64).reshape(-1)
                for k, o in enumerate(ids_b):
                    if o in self.obj_poses:
                        blk[352 + k] = 1.0
                        blk[368 + 12 * k:380 + 12 * k] = np.asarray(self.obj_poses[o], dtype=np.float64)[:3, :4].reshape(-1)
                    blk[560 + 9 * k:569 + 9 * k] = fix_K_for_bbox_ndc(K, bb_b[k]).reshape(-1)       # (double, as the reference projects with it: :505)
                host_b = [np.ascontiguousarray(kps_b, dtype=np.float32), np.ascontiguousarray(bb_b, dtype=np.float32), np.ascontiguousarray(mm_b, dtype=np.uint8), blk]
                if debug:
                    host_b += list(gt_arrays(Lb, gtm_b, gtu_b))
            sb = self.model.stage_block(host_b)
            puv = torch.empty((Lb, NUM_KP, 2), dtype=torch.float32, device=dev)
            pmk = torch.empty((Lb, NUM_KP), dtype=torch.uint8, device=dev)
            vout = torch.empty(32, dtype=torch.float64, device=dev)
            # ---- vote + priors, on the stream, behind pass A's PnP
            _lib.check(lib.suo_slam_vote(La, ra_dev.T_pnp, ra_dev.accepted, ra_dev.n_kp, P(uv_a), P(pa["cov"]), P(mk_a), P(sa[0]), P(sb[3]), Lb, P(sb[0]), P(sb[2]),
                                         int(not self.no_network_cov), float(self.manual_kp_std) ** 2, CHI2_2DOF_95, 4, P(puv), P(pmk), P(vout),
                                         C.c_void_p(_lib.current_stream_ptr())), "suo_slam_vote")
            if getattr(self, "_vote_pin", None) is None:
                self._vote_pin = (torch.empty(32, dtype=torch.float64).pin_memory(), torch.empty((16, NUM_KP, 2), dtype=torch.float32).pin_memory(),
                                  torch.empty((16, NUM_KP), dtype=torch.uint8).pin_memory(), torch.cuda.Event())
            v_pin, puv_pin, pmk_pin, v_ev = self._vote_pin
            v_pin.copy_(vout, non_blocking=True)
            puv_pin[:Lb].copy_(puv, non_blocking=True)
            pmk_pin[:Lb].copy_(pmk, non_blocking=True)
            v_ev.record()
            # ---- pass B: priors rendered on the device from what the vote kernel wrote; nothing above has waited
            pb = self.model(frame, [sb[1]], None, prior_uv=puv, prior_mask=pmk, check=False, out_slot="slam B")
            uv_b, mk_b = (sb[5], sb[4]) if debug else (pb["uv"], keypoint_masks(pb["uv"], pb["cov"], pb["kp_mask"], sb[2], self.bbox_thresh, vt))
            self._fg2.launch([0, Lb], uv_b, pb["cov"], mk_b, sb[0], kinv_b, camk_b, md_b, seed=self._pnp_seed, use_cov=not self.no_network_cov, do_lm=False,
                             seed_dev=seed_run)
            # ---- the host, under pass B: pass A's block and the vote
            ra = self._fg.fetch(copy=False)                   # (views into the pinned block: everything the state keeps is copied out per object below)
            v_ev.synchronize()
            vote = v_pin.numpy().copy()
            prior_uv_h, prior_mask_h = puv_pin[:Lb].numpy().copy(), pmk_pin[:Lb].numpy().copy()
            if not self.model.range_exceeded():
                break
            # (fp16 form only) pass A left the range: its results -- and the priors pass B is running on -- are invalid.  Let pass B drain, then both again on bf16x3
            self._fg2.fetch(copy=False)
            self.model.range_exceeded()
            self.fp16_range_reissues += 1
        assert vote[31] == 0.0, "NaN in information matrix"
        n_solv_a = int(np.count_nonzero(ra["n_kp"] >= 4))
        best = int(vote[12])
        hyp_ids = [o for k, o in enumerate(ids_a) if vote[15 + k] >= 0]
        # ---- pass A into the state, with the device's vote -- while pass B runs
        self._pnp_seed += n_solv_a
        det_a = self._kp_det_from_chain(ra, ids_a, kps_a, Kb_a, gtu_a)
        cam = None
        if best >= 0:
            cam = np.eye(4)
            cam[:3, :4] = vote[:12].reshape(3, 4)
        self.last_cam_hypotheses = ({"obj_ids": hyp_ids, "counts": [int(vote[15 + k]) for k in range(La) if vote[15 + k] >= 0], "best_num_inliers": int(vote[14])}
                                    if hyp_ids else None)
        self._install_kp_detections(view_id, ids_a, bb_a, mm_a, det_a, None, cam_vote=cam)
        rb = self._fg2.fetch(copy=False)
        b_invalid = self.model.range_exceeded()               # (fp16 form only: pass B alone left the range)
        if b_invalid:
            self.fp16_range_reissues += 1
        if cam is None or b_invalid:
            # no hypothesis reached four inliers: the reference falls back to the bbox-centroid pose and THEN runs pass B -- with priors this chain did not have.
            # Pass B's speculative results are dropped (its PnP consumed sampler keys past the host's seed, which never counted them) and the caller issues the
            # pass again, one at a time; the noise draws of its ground-truth keypoints are taken back so that it draws them again.
            if rng_after_a is not None:
                self._rng.bit_generator.state = rng_after_a
            return False
        # ---- pass B into the state
        self._pnp_seed += int(np.count_nonzero(rb["n_kp"] >= 4))
        det_b = self._kp_det_from_chain(rb, ids_b, kps_b, Kb_b, gtu_b)
        prior_det_uv = {o: prior_uv_h[k] for k, o in enumerate(ids_b) if prior_mask_h[k].any()}
        self._install_kp_detections(view_id, ids_b, bb_b, mm_b, det_b, prior_det_uv)
        return True

    # ---------------------------------------------------------------------------------------------
    def _estimate_camera_pose(self, view_id, min_num_inliers=4):
        """object_slam.py:975-1072: hypotheses T_GtoC = T_OtoC(pnp) @ T_GtoO per object, scored by chi2 inliers."""
        curr = self.detections[view_id]
        obj_ids = [o for o in curr if curr[o].get("pose") is not None and o in self.obj_poses]
        self.last_cam_hypotheses = None
        if not obj_ids:
            return None
        hyps = [curr[i]["pose"] @ invert_SE3(to4x4(self.obj_poses[i])) for i in obj_ids]
        scored = [j for j in obj_ids if np.count_nonzero(curr[j]["inliers"]) > 0]
        counts = np.zeros(len(hyps), dtype=np.int64)
        if scored:
            T_obj = np.stack([to4x4(self.obj_poses[j]) for j in scored]).astype(np.float32).astype(np.float64)   # float32 container (:1004)
            # all |hypotheses| x |objects| scorings in one launch (csrc/slam_score.hip); the products as one stacked matmul (numpy runs the
            # same 4x4 kernel per pair as the reference's per-pair `@`)
            Ts = (np.stack(hyps)[:, None] @ T_obj[None]).reshape(-1, 4, 4)
            counts = _sc.chi2_counts(self, Ts, [curr[j] for _ in hyps for j in scored], True, self.manual_kp_std,
                                     CHI2_2DOF_95).reshape(len(hyps), len(scored)).sum(axis=1)
        best, best_n = None, -1
        for T_GtoC, n in zip(hyps, counts):
            if n >= min_num_inliers and n > best_n:
                best, best_n = T_GtoC, int(n)
        self.last_cam_hypotheses = {"obj_ids": obj_ids, "counts": [int(n) for n in counts], "best_num_inliers": best_n}
        return best

    def _maybe_reinit_objects(self, view_id, check_n_views=15):
        """object_slam.py:595-697: re-initialise an object from its current PnP pose when that explains
        >= 3 and more than 3x as many keypoints (over the last views) as the map pose."""
        if self.num_views_processed() < 2 or view_id not in self.cam_poses:
            return {}
        check_n_views = min(len(self.view_ids), check_n_views)
        curr = self.detections[view_id]
        obj_ids = [o for o in self.obj_poses if curr.get(o, {}).get("pose") is not None]
        if not obj_ids:
            return {}
        T_CtoG = invert_SE3(to4x4(self.cam_poses[view_id]))
        views = [self.view_ids[-(i + 1)] for i in range(check_n_views)]
        T_cam32 = np.stack([to4x4(self.cam_poses[v]) for v in views]).astype(np.float32)              # float32 containers (:619,:631)
        T_cam = T_cam32.astype(np.float64)
        T_pnp = {o: T_CtoG @ curr[o]["pose"] for o in obj_ids}
        T_pnp_all = np.stack([T_pnp[o] for o in obj_ids])
        T_est32 = np.stack([to4x4(self.obj_poses[o]) for o in obj_ids]).astype(np.float32)
        # every (object, recent view) pair under both poses in one launch (csrc/slam_score.hip); objects are independent of each other.
        # The reference's products keep numpy's promotion: float32 camera @ float64 PnP pose -> float64, but
        # float32 camera @ float32 map pose -> a float32 product (:634-637); as stacked matmuls (the same 4x4 kernel per pair)
        pairs = [(k, i) for k, o in enumerate(obj_ids) for i, v in enumerate(views) if o in self.detections[v]]
        dets = [self.detections[views[i]][obj_ids[k]] for k, i in pairs]
        owner = np.array([k for k, _ in pairs], dtype=np.int64)
        vi = np.array([i for _, i in pairs], dtype=np.int64)
        Ts = np.concatenate([T_cam[vi] @ T_pnp_all[owner], (T_cam32[vi] @ T_est32[owner]).astype(np.float64)])
        counts = _sc.chi2_counts(self, Ts, dets + dets, False, self.manual_kp_std, CHI2_2DOF_95)
        n_pnp = np.bincount(owner, weights=counts[:len(pairs)], minlength=len(obj_ids)).astype(np.int64)
        n_est = np.bincount(owner, weights=counts[len(pairs):], minlength=len(obj_ids)).astype(np.int64)
        report = {}
        for k, o in enumerate(obj_ids):
            reinit = bool(n_pnp[k] >= 3 and n_pnp[k] > 3 * n_est[k])
            report[o] = {"pnp": int(n_pnp[k]), "estim": int(n_est[k]), "reinit": reinit}
            if reinit:
                self.obj_poses[o] = T_pnp[o]
        return report                                  # (the reference returns nothing; the counts are for tests / logging)

    def _backup_estimate_camera_pose(self, view_id, obj_ids_, bboxes):
        """object_slam.py:933-973: bbox-centroid PnP, else constant velocity, else copy the last pose."""
        assert len(self.view_ids) > 0 and view_id not in self.view_ids and view_id not in self.cam_poses
        cents, centers = [], []
        for i, o in enumerate(obj_ids_):
            if o in self.obj_poses:
                cents.append(0.5 * (bboxes[i, :2] + bboxes[i, 2:]))
                centers.append(self.obj_poses[o][:3, 3])
        ret = pnp(np.stack(centers), np.stack(cents), self.cam_K[view_id]) if cents else None
        if ret is not None:
            self.cam_poses[view_id] = ret[0]
        elif len(self.view_ids) > 1:
            T1, T2 = to4x4(self.cam_poses[self.view_ids[-2]]), to4x4(self.cam_poses[self.view_ids[-1]])
            self.cam_poses[view_id] = (T2 @ invert_SE3(T1)) @ T2
        else:
            self.cam_poses[view_id] = self.cam_poses[self.view_ids[-1]]
        self.view_ids.append(view_id)

    # ---------------------------------------------------------------------------------------------
    def build_problem(self, curr_only=False):
        """Graph construction of optimize() (object_slam.py:713-839) as a flat SoA for suo_optimize.
        Returns (problem, bookkeeping) or None when the reference would return early."""
        if len(self.view_ids) == 0:
            return None
        num_cam_edges, num_obj_edges = defaultdict(int), defaultdict(int)
        obj_ids = list(self.obj_poses.keys())
        view_curr = self.view_ids[-1]
        if curr_only:
            if view_curr not in self.cam_poses:
                return None
            dets = {view_curr: self.detections[view_curr]}
        else:
            dets = self.detections
        for v, det in dets.items():
            if v in self.cam_poses:
                for o, d in det.items():
                    if o in obj_ids:
                        n = int(np.count_nonzero(d["inliers"]))
                        num_cam_edges[v] += n
                        num_obj_edges[o] += n
        if curr_only and num_cam_edges[view_curr] < 3:
            return None
        obj_index = {o: j for j, o in enumerate(o for o in obj_ids if num_obj_edges[o] > 0)}
        cam_views = [view_curr] if curr_only else list(self.cam_poses.keys())
        cam_index, cam_fixed = {}, []
        for i, v in enumerate(cam_views):
            if num_cam_edges[v] > 0:
                cam_index[v] = len(cam_index)
                cam_fixed.append(0 if curr_only else int(i == 0))          # gauge: enumeration index 0 only (:774, R11)
        if not cam_index or not obj_index:
            return None
        # one segment of edges per detection, in the reference's enumeration order (views, objects, keypoints); the
        # per-keypoint data comes from the detection's cache instead of a Python loop over keypoints
        segs, seg_cam, seg_obj, seg_n, e_k, e_p, e_uv, e_info, e_inl = [], [], [], [], [], [], [], [], []
        E = 0
        for v, det in dets.items():
            if v not in cam_index:
                continue
            for o, d in det.items():
                if o in obj_index:
                    c = _det_edges(d)
                    n = c["n"]
                    segs.append((v, o, E, n))
                    seg_cam.append(cam_index[v]); seg_obj.append(obj_index[o]); seg_n.append(n)
                    e_k.append(c["camk"]); e_p.append(c["pts"][:n]); e_uv.append(c["uv"][:n]); e_info.append(c["info"])
                    e_inl.append(np.asarray(d["inliers"], dtype=np.uint8))
                    E += n
        seg_n = np.array(seg_n, dtype=np.int64)
        if self.sfm_mode or (self.slam_mode and not curr_only):
            its = (10, 10, 40, 40)
        else:
            its = (10, 10, 10, 10)
        prob = _ba.Problem(np.stack([to4x4(self.cam_poses[v])[:3] for v in cam_index]), np.array(cam_fixed, np.uint8),
                           np.stack([to4x4(self.obj_poses[o])[:3] for o in obj_index]),
                           np.full(len(obj_index), 1 if curr_only else 0, np.uint8),
                           np.repeat(np.array(seg_cam, np.int32), seg_n), np.repeat(np.array(seg_obj, np.int32), seg_n),
                           np.repeat(np.array(e_k, np.float64).reshape(-1, 4), seg_n, axis=0),
                           np.concatenate(e_p).reshape(E, 3), np.concatenate(e_uv).reshape(E, 2),
                           np.concatenate(e_info).reshape(E, 3), np.concatenate(e_inl), its=its,
                           init_with_outliers=bool(self.opt_init_with_outliers and curr_only))
        return prob, (cam_index, obj_index, _EdgeRefs(segs, E), curr_only, view_curr)

    def apply_problem(self, prob, book):
        """Read-back + culling of optimize() (object_slam.py:898-930)."""
        cam_index, obj_index, e_ref, curr_only, view_curr = book
        inl = np.asarray(prob.inlier).astype(bool)
        for v, o, first, n in e_ref.segs:
            self.detections[v][o]["inliers"][:] = inl[first:first + n]
        cam_T = prob.cam_T.reshape(-1, 3, 4)
        obj_T = prob.obj_T.reshape(-1, 3, 4)
        for v, i in cam_index.items():
            self.cam_poses[v] = cam_T[i].copy()
        if not curr_only:
            for o, j in obj_index.items():
                self.obj_poses[o] = obj_T[j].copy()
        self._cull_after_optimize(list(obj_index.keys()), curr_only, view_curr)

    @_on_stream
    def optimize(self, curr_only=False):
        """object_slam.py:703-930 with the g2o graph replaced by one suo_optimize call."""
        built = self.build_problem(curr_only)
        if built is None:
            return
        prob, book = built
        _ba.optimize_batch([prob])
        self.apply_problem(prob, book)

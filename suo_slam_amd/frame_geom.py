"""Device-resident frame geometry (csrc/frame_geom.hip behind suo_frame_geom_*): what the reference does on the host between the
network and a single-view frame's poses -- /root/reference/lib/object_slam.py:1100-1165 (read-back, masks, compaction, pnp() per
object, acceptance) and :703-903 (optimize() with the camera fixed) -- as one stream-ordered chain consuming the network's device
outputs, with ONE device-to-host copy.  There is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .ba import CHI2_THR, HUBER_DELTA
from .weights import NUM_KP


def kbbox_terms(K_bbox):
    """Per-crop host inputs of the chain from the float32-rounded K_bbox [L,3,3] (the reference's container, object_slam.py:1082):
    kinv [L,6] = the entries of inv(K).T that `points_2d @ KinvT[:2,:2] + KinvT[2:3,:2]` uses (:34-36), camk [L,4] = (fx, fy, cx, cy)."""
    Kb = np.asarray(K_bbox, dtype=np.float64).reshape(-1, 3, 3)
    KinvT = np.linalg.inv(Kb).transpose(0, 2, 1)
    kinv = np.stack([KinvT[:, 0, 0], KinvT[:, 1, 0], KinvT[:, 2, 0], KinvT[:, 0, 1], KinvT[:, 1, 1], KinvT[:, 2, 1]], axis=1)
    camk = np.stack([Kb[:, 0, 0], Kb[:, 1, 1], Kb[:, 0, 2], Kb[:, 1, 2]], axis=1)
    return np.ascontiguousarray(kinv), np.ascontiguousarray(camk)


class FrameGeometry:
    """One context = one launch in flight (own device arena + pinned read-back block)."""

    def __init__(self, max_crops, max_frames=1):
        self._lib = _lib.lib()
        _lib.require_gpu()
        self.max_crops, self.max_frames = int(max_crops), int(max_frames)
        self._h = C.c_void_p()
        _lib.check(self._lib.suo_frame_geom_create(self.max_crops, self.max_frames, C.byref(self._h)), "suo_frame_geom_create")

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.suo_frame_geom_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def launch(self, frame_first, uv_dev, cov_dev, mask_dev, model_kps_dev, kinv, camk, min_depth, seed=0, use_cov=True, do_lm=True,
               its=(10, 10, 40, 40), pnp_threshold=1e-3, stream=None, seed_dev=None):
        """Asynchronous.  uv_dev / cov_dev / mask_dev / model_kps_dev: torch CUDA tensors ([L,41,2] f32, [L,41,2,2] f32, [L,41] u8,
        [L,41,3] f32) or raw device pointers; frame_first [F+1]; kinv [L,6], camk [L,4], min_depth [L] float64 (host).
        seed_dev: a device int64 tensor [1] holding a running sampler key -- the launch samples with seed + seed_dev[0] and adds its number of
        solvable problems to it when done (suo_frame_geom_params.seed_dev): the next launch can be enqueued before this one's read-back."""
        ff = np.ascontiguousarray(frame_first, np.int32)
        L = int(ff[-1])
        kinv = np.ascontiguousarray(kinv, np.float64).reshape(L, 6)
        camk = np.ascontiguousarray(camk, np.float64).reshape(L, 4)
        md = np.ascontiguousarray(min_depth, np.float64).reshape(L)
        p = _lib.FrameGeomParams()
        p.pnp_threshold, p.seed, p.use_cov, p.do_lm = float(pnp_threshold), int(seed) % 2 ** 64, int(bool(use_cov)), int(bool(do_lm))
        assert len(its) <= 4
        for i, v in enumerate(its):
            p.its[i] = int(v)
        p.n_rounds, p.chi2_thr, p.huber_delta = len(its), CHI2_THR, HUBER_DELTA
        p.seed_dev = C.c_void_p(seed_dev.data_ptr()) if seed_dev is not None else None

        def ptr(t):
            return C.c_void_p(t.data_ptr()) if hasattr(t, "data_ptr") else C.c_void_p(int(t))
        if stream is None:
            stream = _lib.current_stream_ptr()
        _lib.check(self._lib.suo_frame_geom_launch(self._h, len(ff) - 1, ff.ctypes.data, ptr(uv_dev), ptr(cov_dev), ptr(mask_dev), ptr(model_kps_dev),
                                                   kinv.ctypes.data, camk.ctypes.data, md.ctypes.data, C.byref(p), C.c_void_p(int(stream))),
                   "suo_frame_geom_launch")

    def device_result(self):
        """The last launch's result block where it lies on the DEVICE (raw pointers in a FrameGeomResult; stream-ordered behind the launch, valid until this
        context's next launch): what a kernel continuing the chain reads (suo_slam_vote)."""
        r = _lib.FrameGeomResult()
        _lib.check(self._lib.suo_frame_geom_device_result(self._h, C.byref(r)), "suo_frame_geom_device_result")
        return r

    def ready(self):
        return bool(self._lib.suo_frame_geom_ready(self._h))

    def fetch(self, copy=True):
        """Wait for the launch; dict of numpy arrays (views into the pinned block unless copy): T_pnp [L,4,4], pnp_status [L], accepted
        [L] bool, T_opt [L,3,4], n_kp [L], inlier [L,41] bool / chi2 [L,41] in SLOT order (first n_kp[l] entries of row l = the crop's
        valid keypoints in mask order), uv [L,41,2], cov [L,41,2,2], mask [L,41] bool, lm_stats [F,4], pnp_iterations, pnp_best_inliers."""
        r = _lib.FrameGeomResult()
        _lib.check(self._lib.suo_frame_geom_fetch(self._h, C.byref(r)), "suo_frame_geom_fetch")
        L, F = r.n_crops, r.n_frames

        def arr(p, ctype, shape):
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), shape=shape)
            return a.copy() if copy else a
        return {"T_pnp": arr(r.T_pnp, C.c_double, (L, 4, 4)), "T_opt": arr(r.T_opt, C.c_double, (L, 3, 4)), "chi2": arr(r.chi2, C.c_double, (L, NUM_KP)),
                "pnp_status": arr(r.pnp_status, C.c_int, (L,)), "pnp_best_inliers": arr(r.pnp_best_inliers, C.c_int, (L,)),
                "pnp_iterations": arr(r.pnp_iterations, C.c_int, (L,)), "n_kp": arr(r.n_kp, C.c_int, (L,)),
                "lm_stats": arr(r.lm_stats, C.c_int, (F, 4)), "accepted": arr(r.accepted, C.c_uint8, (L,)).astype(bool),
                "inlier": arr(r.inlier, C.c_uint8, (L, NUM_KP)).astype(bool), "uv": arr(r.uv, C.c_float, (L, NUM_KP, 2)),
                "cov": arr(r.cov, C.c_float, (L, NUM_KP, 2, 2)), "mask": arr(r.mask, C.c_uint8, (L, NUM_KP)).astype(bool)}

"""SLAM-mode hypothesis scoring on the device (csrc/slam_score.hip).

The chi-square inlier counts of ``__estimate_camera_pose`` (/root/reference/lib/object_slam.py:1000-1066) and ``__maybe_reinit_objects``
(:619-690) for all (pose, detection) pairs of a view in one launch.  A detection's immutable arrays (model keypoints, predicted uv,
covariances, K_bbox) are written ONCE into a device-resident store when the detection is first scored; a call ships the poses, the keypoint
selections and the store slots (112 bytes per pair) and reads the counts back.  No CPU fallback: without the HIP library / a GPU this raises."""
import ctypes as C

import numpy as np

from . import _lib
from .weights import NUM_KP

ROW, PAIR = 380, 14                                   # include/suo_hip.h: SUO_SLAM_ROW, SUO_SLAM_PAIR
_PTS, _UV, _COV, _K, _N, _HAS = 0, NUM_KP * 3, NUM_KP * 5, NUM_KP * 9, NUM_KP * 9 + 9, NUM_KP * 9 + 10
assert NUM_KP == 41 and _HAS + 1 == ROW


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class DetectionStore:
    """Device rows of the detections scored so far.  A SLAM sequence re-scores the detections of its last 15 views only, so the store is a soft-bounded ring: when a call
    would take it past ``max_slots`` rows (SUO_SLAM_STORE_SLOTS, default 4096 = 12 MB) every tag is invalidated, the store starts over at row 0 and the call's detections are
    written again -- device memory no longer grows with the length of the sequence.  A single call naming more distinct detections than that (SfM over all
    views) still gets the rows it needs."""

    _uids = 0

    def __init__(self, capacity=1024):
        _lib.require_gpu()
        DetectionStore._uids += 1
        self.uid = DetectionStore._uids                 # (detections remember (uid, slot): a number, so that copies of a detection dict stay plain data)
        self._h = _lib.lib().suo_slam_store_create(int(capacity))
        if not self._h:
            _lib.check(-1, "suo_slam_store_create")
        self.n_slots = 0
        import os
        self.max_slots = int(os.environ.get("SUO_SLAM_STORE_SLOTS", "4096"))
        self.recycled = 0                               # times the ring started over

    def clear(self):
        """Forget every detection (ObjectSLAM.reset: a new scene); the device allocation is kept, stale ``_slot`` tags no longer match."""
        DetectionStore._uids += 1
        self.uid = DetectionStore._uids
        self.n_slots = 0

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().suo_slam_store_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def row_of(d):
        """One detection -> its store row (include/suo_hip.h: suo_slam_store_put)."""
        n = int(d["uv_pred"].shape[0])
        assert n <= NUM_KP, f"a detection carries {n} keypoints, the store holds {NUM_KP}"
        r = np.zeros(ROW)
        r[_PTS:_PTS + 3 * n] = np.asarray(d["model_kp"], dtype=np.float64).reshape(-1)
        r[_UV:_UV + 2 * n] = np.asarray(d["uv_pred"], dtype=np.float64).reshape(-1)
        if d["cov_pred"] is not None:
            r[_COV:_COV + 4 * n] = np.asarray(d["cov_pred"], dtype=np.float64).reshape(-1)        # (float32 from the network -> double, as the host rule reads it)
            r[_HAS] = 1.0
        r[_K:_K + 9] = np.asarray(d["K"], dtype=np.float64).reshape(-1)
        r[_N] = n
        return r

    def slots(self, dets):
        """Store slot and keypoint count of every detection; detections not in the store yet are written in one upload."""
        out = np.empty(len(dets), dtype=np.int64)
        ns = np.empty(len(dets), dtype=np.int64)
        if self.n_slots > 0 and self.max_slots > 0:
            fresh = {id(d) for d in dets if not (d.get("_slot") is not None and d["_slot"][0] == self.uid and d["_slot"][3] is d["uv_pred"])}
            if self.n_slots + len(fresh) > self.max_slots:
                self.clear()                            # every tag is stale now: this call's detections are written again from row 0
                self.recycled += 1
        new = []
        for i, d in enumerate(dets):
            s = d.get("_slot")
            if s is None or s[0] != self.uid or s[3] is not d["uv_pred"]:
                s = (self.uid, self.n_slots + len(new), int(d["uv_pred"].shape[0]), d["uv_pred"])
                d["_slot"] = s
                new.append(self.row_of(d))
            out[i], ns[i] = s[1], s[2]
        if new:
            rows = np.ascontiguousarray(np.stack(new))
            _lib.check(_lib.lib().suo_slam_store_put(self._h, self.n_slots, len(new), _ptr(rows)), "suo_slam_store_put")
            self.n_slots += len(new)
        return out, ns

    def counts(self, T, slots, sel, chi2_max, kp_std2):
        """T [B,4,4] (or [B,3,4]) object -> camera poses, slots [B], sel [B] uint64 keypoint selections -> inlier counts [B]."""
        B = len(slots)
        if B == 0:
            return np.zeros(0, dtype=np.int64)
        buf = np.empty((B, PAIR))
        buf[:, :12] = np.asarray(T, dtype=np.float64)[:, :3, :].reshape(B, 12)
        buf[:, 12] = np.ascontiguousarray(sel, dtype=np.uint64).view(np.float64)                      # (same-width copies: the bits travel)
        buf[:, 13] = np.ascontiguousarray(slots, dtype=np.int64).view(np.float64)
        cnt = np.empty(B + 1, dtype=np.int32)
        _lib.check(_lib.lib().suo_slam_score(self._h, B, _ptr(buf), float(chi2_max), float(kp_std2), _ptr(cnt)), "suo_slam_score")
        assert cnt[B] == 0, "NaN in information matrix"
        return cnt[:B].astype(np.int64)


def chi2_counts(owner, Ts, dets, use_inlier_subset, manual_kp_std, chi2_max):
    """Inlier counts of B (pose, detection) pairs (Ts [B,4,4], dets: B detection dicts; a detection may appear many times).
    ``owner`` keeps the store (``owner._score_store``).  This is the call the CPU tests replace by a host restatement."""
    if len(dets) == 0:
        return np.zeros(0, dtype=np.int64)
    st = getattr(owner, "_score_store", None)
    if st is None:
        st = owner._score_store = DetectionStore()
    slots, ns = st.slots(dets)
    sel = (np.uint64(1) << ns.astype(np.uint64)) - np.uint64(1)
    if use_inlier_subset:
        bits = {}
        for i, d in enumerate(dets):                                  # (few distinct detections: the current view's, once per hypothesis)
            b = bits.get(id(d))
            if b is None:
                b = bits[id(d)] = np.uint64(sum(1 << int(k) for k in np.flatnonzero(d["inliers"])))
            sel[i] &= b
    return st.counts(Ts, slots, sel, chi2_max, manual_kp_std ** 2)

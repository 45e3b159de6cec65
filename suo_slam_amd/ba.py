"""Flat-SoA front end of the HIP pose-refinement / bundle-adjustment kernel (csrc/lm.hip), the
replacement for the g2o graph that ObjectSLAM.optimize builds (/root/reference/lib/object_slam.py:703-903)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

CHI2_THR = 5.991                       # lib/object_slam.py:680,860
HUBER_DELTA = float(np.sqrt(5.991))    # lib/object_slam.py:831


class Problem:
    """One optimize() call.  Arrays are copied; results are read back from .cam_T/.obj_T/.inlier/.chi2/.stats."""

    def __init__(self, cam_T, cam_fixed, obj_T, obj_fixed, edge_cam, edge_obj, edge_camk, edge_p, edge_uv, edge_info,
                 edge_inlier, its=(10, 10, 40, 40), init_with_outliers=False, chi2_thr=CHI2_THR, huber_delta=HUBER_DELTA):
        self.cam_T = np.ascontiguousarray(np.asarray(cam_T, np.float64)[..., :3, :4]).reshape(-1, 12).copy()
        self.obj_T = np.ascontiguousarray(np.asarray(obj_T, np.float64)[..., :3, :4]).reshape(-1, 12).copy()
        self.cam_fixed = np.ascontiguousarray(cam_fixed, np.uint8).copy()
        self.obj_fixed = np.ascontiguousarray(obj_fixed, np.uint8).copy()
        self.edge_cam = np.ascontiguousarray(edge_cam, np.int32).copy()
        self.edge_obj = np.ascontiguousarray(edge_obj, np.int32).copy()
        E = len(self.edge_cam)
        self.edge_camk = np.ascontiguousarray(edge_camk, np.float64).reshape(E, 4).copy()
        self.edge_p = np.ascontiguousarray(edge_p, np.float64).reshape(E, 3).copy()
        self.edge_uv = np.ascontiguousarray(edge_uv, np.float64).reshape(E, 2).copy()
        info = np.asarray(edge_info, np.float64)
        if info.ndim == 3:
            info = np.stack([info[:, 0, 0], info[:, 0, 1], info[:, 1, 1]], -1)
        self.edge_info = np.ascontiguousarray(info).reshape(E, 3).copy()
        self.inlier = np.ascontiguousarray(edge_inlier, np.uint8).copy()
        self.chi2 = np.zeros(max(E, 1))
        self.its = tuple(int(i) for i in its)
        self.init_with_outliers = bool(init_with_outliers)
        self.chi2_thr = float(chi2_thr)
        self.huber_delta = float(huber_delta)
        self.stats = np.zeros(4, np.int32)

    def _fill(self, s):
        s.n_cam, s.n_obj, s.n_edge = len(self.cam_T), len(self.obj_T), len(self.edge_cam)
        for name in ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info"):
            setattr(s, name, getattr(self, name).ctypes.data)
        s.edge_inlier = self.inlier.ctypes.data
        s.edge_chi2 = self.chi2.ctypes.data
        for i, v in enumerate(self.its):
            s.its[i] = v
        s.n_rounds = len(self.its)
        s.init_with_outliers = int(self.init_with_outliers)
        s.chi2_thr = self.chi2_thr
        s.huber_delta = self.huber_delta


def optimize_batch(problems):
    """Run many independent problems (e.g. one per frame) in one launch; results land in each Problem."""
    if not problems:
        return problems
    lib = _lib.lib()
    _lib.require_gpu()
    arr = (_lib.BaProblem * len(problems))()
    for s, p in zip(arr, problems):
        p._fill(s)
    _lib.check(lib.suo_optimize_batch(C.cast(arr, C.c_void_p), len(problems)), "suo_optimize_batch")
    for s, p in zip(arr, problems):
        p.stats[:] = list(s.stats)
        p.chi2 = p.chi2[:len(p.edge_cam)]
    return problems


def optimize(*args, **kw):
    p = Problem(*args, **kw)
    optimize_batch([p])
    return p.cam_T.reshape(-1, 3, 4), p.obj_T.reshape(-1, 3, 4), p.inlier, p.chi2, p.stats

"""ORACLE loader (test infrastructure only): ctypes access to oracle/libsuo_oracle.so (the build's C
restatements) and, when present, oracle/_ref/libp4p_ref.so (the reference's own P3P/P4P)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libsuo_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", HERE, "all"])
        L = C.CDLL(path)
        L.orc_p3p.restype = C.c_int
        L.orc_p3p.argtypes = [_dp] * 8
        L.orc_p4p.restype = None
        L.orc_p4p.argtypes = [_dp, _dp, _ip, _dp, _dp]
        L.orc_get_iterations.restype = C.c_int
        L.orc_get_iterations.argtypes = [C.c_double]
        L.orc_sample4.restype = None
        L.orc_sample4.argtypes = [C.c_uint64, C.c_uint32, C.c_int, _ip]
        L.orc_pnp_ransac.restype = C.c_int
        L.orc_pnp_ransac.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_uint64, C.c_int, _dp, C.POINTER(C.c_int)]
        L.orc_pnp_ransac_draws.restype = C.c_int
        L.orc_pnp_ransac_draws.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_uint64, _ip, C.c_int, C.c_int, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_ref_rng_seed.restype = None
        L.orc_ref_rng_seed.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
        L.orc_ref_randui.restype = C.c_int
        L.orc_ref_randui.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.c_int]
        L.orc_ref_get4.restype = C.c_int
        L.orc_ref_get4.argtypes = [C.POINTER(C.c_uint32), C.c_int, _ip]
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        L.orc_optimize.restype = C.c_int
        L.orc_optimize.argtypes = [C.c_int, C.c_int, C.c_int, _dp, u8p, _dp, u8p, _ip, _ip, _dp, _dp, _dp, _dp, u8p, _dp, _ip,
                                   C.c_int, C.c_int, C.c_double, C.c_double, _ip]
        L.orc_edge_jacobians.restype = None
        L.orc_edge_jacobians.argtypes = [_dp] * 6
        L.orc_pose_oplus.restype = None
        L.orc_pose_oplus.argtypes = [_dp, _dp]
        L.orc_edge_error.restype = None
        L.orc_edge_error.argtypes = [_dp] * 6
        _lib = L
    return _lib


def ref():
    """The reference's compiled P3P/P4P, or None when oracle/_ref was not built (e.g. on the GPU box
    when the prebuilt file did not travel)."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libp4p_ref.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_p4p.restype = None
        L.ref_p4p.argtypes = [_dp, _dp, C.c_int, _ip, _dp]
        L.ref_p3p.restype = C.c_int
        L.ref_p3p.argtypes = [_dp] * 8
        _ref = L
    return _ref


def quat_to_rot(q):
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (a * c + b * d)],
                     [2 * (a * d + b * c), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (a * b + c * d), a * a - b * b - c * c + d * d]])


def p4p(xs, ys, idx):
    xs = np.ascontiguousarray(xs, np.float64)
    ys = np.ascontiguousarray(ys, np.float64)
    q = np.zeros(4)
    t = np.zeros(3)
    lib().orc_p4p(xs, ys, np.ascontiguousarray(idx, np.int32), q, t)
    T = np.eye(4)
    T[:3, :3] = quat_to_rot(q)
    T[:3, 3] = t
    return T


def ref_p4p(xs, ys, idx):
    xs = np.ascontiguousarray(xs, np.float64)
    ys = np.ascontiguousarray(ys, np.float64)
    T = np.zeros(16)
    ref().ref_p4p(xs, ys, len(xs), np.ascontiguousarray(idx, np.int32), T)
    return T.reshape(4, 4)


def pnp(xs, ys, threshold=1e-3, seed=0, refine=True):
    """lambdatwist.pnp(xs, ys, threshold) restated; returns (T[4,4], best_inliers, iterations)."""
    xs = np.ascontiguousarray(xs, np.float64)
    ys = np.ascontiguousarray(ys, np.float64)
    T = np.zeros(16)
    best = C.c_int(0)
    its = lib().orc_pnp_ransac(xs, ys, len(xs), threshold, seed, int(refine), T, C.byref(best))
    return T.reshape(4, 4), best.value, its


def pnp_with_draws(xs, ys, draws, threshold=1e-3, refine=True):
    """The same RANSAC with hypothesis i's 4-point sample read from draws[i] (int32 [n,4]) -- e.g. the reference's own sequence (RefSampler).
    Returns (T[4,4], best_inliers, iterations, winner) with winner = index of the hypothesis that became the result (-1: none)."""
    xs = np.ascontiguousarray(xs, np.float64)
    ys = np.ascontiguousarray(ys, np.float64)
    d = np.ascontiguousarray(draws, np.int32).reshape(-1, 4)
    T = np.zeros(16)
    best, win = C.c_int(0), C.c_int(-1)
    its = lib().orc_pnp_ransac_draws(xs, ys, len(xs), threshold, 0, d, len(d), int(refine), T, C.byref(best), C.byref(win))
    return T.reshape(4, 4), best.value, its, win.value


class RefSampler:
    """The reference's RANSAC sampler restated (pnp_oracle.c: orc_ref_*): std::default_random_engine seeded with 0, consumed by
    get4RandomInRange0 through std::uniform_int_distribution -- one process-global stream that continues from pnp call to pnp call."""

    def __init__(self, seed=0):
        self.state = C.c_uint32(0)
        lib().orc_ref_rng_seed(C.byref(self.state), seed)

    def randui(self, lo, hi):
        return lib().orc_ref_randui(C.byref(self.state), lo, hi)

    def get4(self, n_points, n_samples=1):
        out = np.zeros((n_samples, 4), np.int32)
        for i in range(n_samples):
            lib().orc_ref_get4(C.byref(self.state), n_points, out[i])
        return out

    def fork(self):
        c = RefSampler()
        c.state = C.c_uint32(self.state.value)
        return c


def ref_random():
    """The reference's random.h compiled here (oracle/_ref/librandom_ref.so), or None."""
    path = os.path.join(HERE, "_ref", "librandom_ref.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_rng_reset.restype = None
    L.ref_randui.restype = None
    L.ref_randui.argtypes = [C.c_int, C.c_int, _ip]
    L.ref_get4.restype = None
    L.ref_get4.argtypes = [C.c_uint, C.c_int, _ip]
    return L


CHI2_THR = 5.991          # lib/object_slam.py:680,860
HUBER_DELTA = float(np.sqrt(5.991))


def optimize(cam_T, cam_fixed, obj_T, obj_fixed, edge_cam, edge_obj, edge_camk, edge_p, edge_uv, edge_info, edge_inlier,
             its=(10, 10, 40, 40), init_with_outliers=False, chi2_thr=CHI2_THR, huber_delta=HUBER_DELTA):
    """ObjectSLAM.optimize rounds restated (oracle/lm_oracle.c).  Poses are [n,3,4]; returns
    (cam_T, obj_T, inlier[uint8], chi2, stats) as new arrays."""
    cam_T = np.ascontiguousarray(cam_T, np.float64).reshape(-1, 12).copy()
    obj_T = np.ascontiguousarray(obj_T, np.float64).reshape(-1, 12).copy()
    n_edge = len(edge_cam)
    inl = np.ascontiguousarray(edge_inlier, np.uint8).copy()
    chi2 = np.zeros(max(n_edge, 1))
    stats = np.zeros(4, np.int32)
    info = np.ascontiguousarray(edge_info, np.float64)
    if info.ndim == 3:       # [E,2,2] -> (xx, xy, yy)
        info = np.stack([info[:, 0, 0], info[:, 0, 1], info[:, 1, 1]], -1)
    lib().orc_optimize(len(cam_T), len(obj_T), n_edge, cam_T, np.ascontiguousarray(cam_fixed, np.uint8), obj_T,
                       np.ascontiguousarray(obj_fixed, np.uint8), np.ascontiguousarray(edge_cam, np.int32),
                       np.ascontiguousarray(edge_obj, np.int32), np.ascontiguousarray(edge_camk, np.float64),
                       np.ascontiguousarray(edge_p, np.float64), np.ascontiguousarray(edge_uv, np.float64),
                       np.ascontiguousarray(info), inl, chi2, np.ascontiguousarray(its, np.int32), len(its),
                       int(init_with_outliers), chi2_thr, huber_delta, stats)
    return cam_T.reshape(-1, 3, 4), obj_T.reshape(-1, 3, 4), inl, chi2[:n_edge], stats

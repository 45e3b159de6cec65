"""Drop-in for the reference's ``lambdatwist`` pybind module
(/root/reference/thirdparty/lambdatwist/pnp_python_binding.cpp:57-62), backed by the HIP kernel
csrc/pnp.hip.  ``pnp(xs_in, ys_in, threshold=0.001) -> ndarray[4,4]``; identity on total failure."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

SEED_STRIDE = 0x9E3779B97F4A7C15


def pnp_batch(xs_list, ys_list, threshold=0.001, seed=0, refine=True, return_info=False):
    """All objects of a frame in one launch.  xs_list[o]: [N_o,3] float64, ys_list[o]: [N_o,2] normalised."""
    n_obj = len(xs_list)
    if n_obj == 0:
        return (np.zeros((0, 4, 4)), np.zeros(0, np.int32)) if not return_info else (np.zeros((0, 4, 4)), np.zeros(0, np.int32), {})
    lib = _lib.lib()
    _lib.require_gpu()
    n_pts = np.array([len(x) for x in xs_list], np.int32)
    xs = np.ascontiguousarray(np.concatenate([np.asarray(x, np.float64).reshape(-1, 3) for x in xs_list]))
    ys = np.ascontiguousarray(np.concatenate([np.asarray(y, np.float64).reshape(-1, 2) for y in ys_list]))
    T = np.zeros((n_obj, 4, 4))
    status = np.zeros(n_obj, np.int32)
    best = np.zeros(n_obj, np.int32)
    its = np.zeros(n_obj, np.int32)
    _lib.check(lib.suo_pnp_batch(n_obj, n_pts.ctypes.data, xs.ctypes.data, ys.ctypes.data, float(threshold), C.c_uint64(seed),
                                 int(refine), T.ctypes.data, status.ctypes.data, best.ctypes.data, its.ctypes.data), "suo_pnp_batch")
    if return_info:
        return T, status, {"best_inliers": best, "iterations": its}
    return T, status


def pnp(xs_in, ys_in, threshold=0.001):
    """Legacy signature of lambdatwist.pnp: returns a fresh 4x4 float64 array (identity = failure)."""
    xs = np.ascontiguousarray(xs_in, np.float64)
    ys = np.ascontiguousarray(ys_in, np.float64)
    assert xs.ndim == 2 and xs.shape[1] == 3 and ys.shape == (xs.shape[0], 2)
    lib = _lib.lib()
    _lib.require_gpu()
    T = np.zeros((4, 4))
    _lib.check(lib.suo_pnp(xs.ctypes.data, ys.ctypes.data, xs.shape[0], float(threshold), T.ctypes.data), "suo_pnp")
    return T

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


class ReferenceSampler:
    """The reference's RANSAC sampler, host side: PNP::compute draws its 4-point samples through get4RandomInRange0
    (/root/reference/thirdparty/lambdatwist/pnp_ransac.cpp:161-183) = a std::set<uint> filled with mlib::randui<int>(0, n - 1)
    (utils/random.h:112-116) until it holds four values, read in ascending order; randui is std::uniform_int_distribution<int> over the PROCESS-GLOBAL
    std::default_random_engine seeded once with RANDOM_SEED_VALUE = 0 (random.h:40-42,65-86) -- the stream continues from one pnp call to the next.
    libstdc++: default_random_engine = minstd_rand0 (x <- 16807 x mod 2^31 - 1, seed 0 -> 1); the distribution scales down by rejection
    (scaling = (max - min) / n = (2^31 - 3) / n; draw x - 1 until < n * scaling; divide).  Pinned against the reference's header compiled in the build container
    (tests/test_ref_sampler.py, tests/golden/sampler_golden.npz)."""
    M = 2147483647

    def __init__(self, seed=0):
        x = seed % self.M
        self.state = 1 if x == 0 else x

    def randui(self, lo, hi):
        uerange = hi - lo + 1
        scaling = (self.M - 2) // uerange                                       # urngrange = max() - min() = (M - 1) - 1 = 2^31 - 3 = M - 2
        past = uerange * scaling
        while True:
            self.state = (self.state * 16807) % self.M
            r = self.state - 1
            if r < past:
                return lo + r // scaling

    def get4(self, n_points):
        picked = set()
        while len(picked) < 4:
            picked.add(self.randui(0, n_points - 1))
        return sorted(picked)

    def table(self, n_points, n_samples):
        """The next n_samples samples as int32 [n_samples, 4] WITHOUT consuming the stream (a copy draws them)."""
        c = ReferenceSampler()
        c.state = self.state
        return np.array([c.get4(n_points) for _ in range(n_samples)], np.int32)

    def advance(self, n_points, n_samples):
        for _ in range(n_samples):
            self.get4(n_points)


_reference_sampler = None          # set_reference_sampler(True): pnp() consumes the reference's own draw sequence (index-work parity mode)
MAX_ITERATIONS = 1000              # PnpParams::get_iterations' cap (parameters.h:76-102)


def set_reference_sampler(on=True, seed=0):
    """pnp() -- the legacy one-object entry -- then samples exactly what the reference's process-global generator would hand PNP::compute, call after call
    (the batched fast paths keep their counter-based sampler: hypothesis i of an object is a function of (seed, i) alone, which is what lets the device
    evaluate a thousand of them at once)."""
    global _reference_sampler
    _reference_sampler = ReferenceSampler(seed) if on else None
    return _reference_sampler


def pnp_replay(xs_in, ys_in, draws, threshold=0.001, refine=True):
    """One object with hypothesis i sampling draws[i] (int32 [>= 1000, 4]).  Returns (T[4,4], info) with info = best_inliers, iterations, winner."""
    xs = np.ascontiguousarray(xs_in, np.float64)
    ys = np.ascontiguousarray(ys_in, np.float64)
    d = np.ascontiguousarray(draws, np.int32).reshape(-1, 4)
    assert xs.ndim == 2 and xs.shape[1] == 3 and ys.shape == (xs.shape[0], 2)
    lib = _lib.lib()
    _lib.require_gpu()
    T = np.zeros((4, 4))
    n = np.array([xs.shape[0]], np.int32)
    st, best, its, win = (np.zeros(1, np.int32) for _ in range(4))
    _lib.check(lib.suo_pnp_replay(1, n.ctypes.data, xs.ctypes.data, ys.ctypes.data, float(threshold), d.ctypes.data, int(d.shape[0]), int(refine),
                                  T.ctypes.data, st.ctypes.data, best.ctypes.data, its.ctypes.data, win.ctypes.data), "suo_pnp_replay")
    return T, {"best_inliers": int(best[0]), "iterations": int(its[0]), "winner": int(win[0]), "status": int(st[0])}


def pnp(xs_in, ys_in, threshold=0.001):
    """Legacy signature of lambdatwist.pnp: returns a fresh 4x4 float64 array (identity = failure)."""
    xs = np.ascontiguousarray(xs_in, np.float64)
    ys = np.ascontiguousarray(ys_in, np.float64)
    assert xs.ndim == 2 and xs.shape[1] == 3 and ys.shape == (xs.shape[0], 2)
    if _reference_sampler is not None and xs.shape[0] >= 4:
        T, info = pnp_replay(xs, ys, _reference_sampler.table(xs.shape[0], MAX_ITERATIONS), threshold)
        _reference_sampler.advance(xs.shape[0], info["iterations"])           # PNP::compute drew exactly total_iters samples
        return T
    lib = _lib.lib()
    _lib.require_gpu()
    T = np.zeros((4, 4))
    _lib.check(lib.suo_pnp(xs.ctypes.data, ys.ctypes.data, xs.shape[0], float(threshold), T.ctypes.data), "suo_pnp")
    return T

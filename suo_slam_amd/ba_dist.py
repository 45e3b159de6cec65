"""Multi-GPU global bundle adjustment (BASELINE config 5): cameras (views) are partitioned across
ranks, every rank keeps all object poses, and each Levenberg-Marquardt trial exchanges only the reduced
(Schur-complement) object system -- (6 n_obj)^2 + 6 n_obj doubles -- with one all-reduce, plus one all-reduce of
two scalars for the gain ratio.  With RCCL over xGMI on GPUs (backend "nccl"); "gloo" in the CPU tests.

The per-rank work runs in the phase kernels of csrc/lm_dist.hip (``HipPhases``); this module is the host
schedule: g2o's lambda / nu logic (thirdparty/g2opy/g2o/core/optimization_algorithm_levenberg.cpp:58-150) and
the robust rounds of ObjectSLAM.optimize (lib/object_slam.py:842-896), executed identically on every rank
because every decision is taken on all-reduced quantities.  With one rank it degenerates to the same algorithm
as the single-kernel path (csrc/lm.hip) and is tested against it and against the oracle.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from . import ba as _ba

DBL_MAX = float(np.finfo(np.float64).max)
COLLECTIVES_PER_TRIAL = 2          # [S | r | ok] and [chi2 | scale | ok]; +2 per LM iteration for the linearisation totals
_DIAG21 = (0, 6, 11, 15, 18, 20)


class HipPhases:
    """One rank's share of the graph, resident on its GPU (suo_ba_ctx_* of include/suo_hip.h)."""

    def __init__(self, local: _ba.Problem):
        self.lib = _lib.lib()
        _lib.require_gpu()
        self.local = local
        self._s = _lib.BaProblem()
        local._fill(self._s)
        h = C.c_void_p()
        _lib.check(self.lib.suo_ba_ctx_create(C.byref(self._s), C.byref(h)), "suo_ba_ctx_create")
        self._h = h
        self.n_obj = len(local.obj_T)
        self.ns = int(self.lib.suo_ba_ctx_ns(h))

    def classify(self, keep_all):
        out = np.zeros(1)
        _lib.check(self.lib.suo_ba_classify(self._h, int(keep_all), out.ctypes.data), "suo_ba_classify")
        return float(out[0])

    def linearize(self, robust_on):
        out = np.zeros(2 + 27 * self.n_obj)
        _lib.check(self.lib.suo_ba_linearize(self._h, int(robust_on), out.ctypes.data), "suo_ba_linearize")
        return out

    def schur(self, lam):
        out = np.zeros(self.ns * self.ns + self.ns + 1)
        _lib.check(self.lib.suo_ba_schur(self._h, float(lam), out.ctypes.data), "suo_ba_schur")
        return out

    def solve_update(self, lam, robust_on, totals):
        totals = np.ascontiguousarray(totals, np.float64)
        out = np.zeros(4)
        _lib.check(self.lib.suo_ba_solve_update(self._h, float(lam), int(robust_on), totals.ctypes.data, out.ctypes.data), "suo_ba_solve_update")
        return out

    def restore(self):
        _lib.check(self.lib.suo_ba_restore(self._h), "suo_ba_restore")

    def download(self):
        _lib.check(self.lib.suo_ba_ctx_download(self._h, C.byref(self._s)), "suo_ba_ctx_download")
        return self.local

    def close(self):
        if self._h is not None:
            self.lib.suo_ba_ctx_destroy(self._h)
            self._h = None


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _allreduce(arr, op="sum"):
    """All-reduce a small float64 numpy array (RCCL needs device tensors; gloo works on host tensors)."""
    rank, world = _world()
    a = np.ascontiguousarray(arr, np.float64)
    if world == 1:
        return a.copy()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.from_numpy(a.copy()).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    return t.cpu().numpy()


def split_problem(full: _ba.Problem, rank: int, world: int):
    """Rank `rank`'s share: cameras c with c % world == rank, all objects, the edges of those cameras."""
    cams = [c for c in range(len(full.cam_T)) if c % world == rank]
    remap = {c: i for i, c in enumerate(cams)}
    sel = np.array([e for e in range(len(full.edge_cam)) if int(full.edge_cam[e]) in remap], dtype=np.int64)
    local = _ba.Problem(full.cam_T[cams].reshape(-1, 3, 4) if cams else np.zeros((0, 3, 4)), full.cam_fixed[cams], full.obj_T.reshape(-1, 3, 4),
                        full.obj_fixed, np.array([remap[int(c)] for c in full.edge_cam[sel]], np.int32), full.edge_obj[sel],
                        full.edge_camk[sel], full.edge_p[sel], full.edge_uv[sel], full.edge_info[sel], full.inlier[sel], its=full.its,
                        init_with_outliers=full.init_with_outliers, chi2_thr=full.chi2_thr, huber_delta=full.huber_delta)
    return local, cams, sel


def optimize_distributed(full: _ba.Problem, phases_factory=HipPhases):
    """Run the robust LM rounds of `full` across the ranks of the default process group.  Every rank passes the
    same `full` problem and gets the complete result back (cam_T, obj_T, inlier, chi2, stats) in `full`."""
    rank, world = _world()
    local, cams, sel = split_problem(full, rank, world)
    ph = phases_factory(local)
    ns, O = ph.ns, len(full.obj_T)
    n_edge_total = len(full.edge_cam)
    free_obj = [o for o in range(O) if not full.obj_fixed[o]]
    rounds = lm_its = lm_trials = 0
    if full.init_with_outliers:
        ph.classify(True)
        num_good = n_edge_total
    else:
        num_good = int(round(_allreduce([ph.classify(False)])[0]))
    robust_on = True
    drop = max(1, len(full.its) // 2)
    for rnd, its in enumerate(full.its):
        if n_edge_total < 4 or num_good < 4:
            break
        rounds += 1
        lam, ni = -1.0, 2.0
        for it in range(int(its)):
            lin = ph.linearize(robust_on)
            tot = _allreduce(lin[:1 + 27 * O])
            current_chi, HB = float(tot[0]), tot[1:]
            if it == 0:                                        # computeLambdaInit: tau * max |diag H| over all free vertices
                maxd = float(_allreduce([lin[1 + 27 * O]], "max")[0])
                for o in free_obj:
                    maxd = max(maxd, max(abs(HB[27 * o + d]) for d in _DIAG21))
                lam, ni = 1e-5 * maxd, 2.0
            rho, qmax, lam_finite = 0.0, 0, True
            while True:
                sch = _allreduce(ph.schur(lam))                # the pose-graph reduce: [S | r | ok-count]
                ok2 = int(round(sch[-1])) == world
                temp_chi, scale = DBL_MAX, 0.0
                if ok2:
                    out = ph.solve_update(lam, robust_on, np.concatenate([HB, sch[:ns * ns + ns]]))
                    red = _allreduce([out[0], out[1], out[3]])
                    ok2 = int(round(red[2])) == world
                    if ok2:
                        temp_chi, scale = float(red[0]), float(red[1]) + float(out[2])
                rho = (current_chi - temp_chi) / (scale + 1e-3)
                if rho > 0 and math.isfinite(temp_chi):
                    alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                    lam *= max(1.0 / 3.0, alpha)
                    ni = 2.0
                    current_chi = temp_chi
                else:
                    lam *= ni
                    ni *= 2
                    ph.restore()
                    if not math.isfinite(lam):
                        lam_finite = False
                        break
                qmax += 1
                lm_trials += 1
                if not (rho < 0 and qmax < 10):
                    break
            lm_its += 1
            if qmax == 10 or rho == 0 or not lam_finite:
                break
        num_good = int(round(_allreduce([ph.classify(False)])[0]))
        if rnd == drop:
            robust_on = False
    loc = ph.download()
    # assemble the full result on every rank: each camera / edge is owned by exactly one rank
    cam = np.zeros((len(full.cam_T), 12))
    cam[cams] = loc.cam_T
    inl = np.zeros(n_edge_total)
    chi = np.zeros(n_edge_total)
    inl[sel] = loc.inlier
    chi[sel] = loc.chi2[:len(sel)]
    packed = _allreduce(np.concatenate([cam.ravel(), inl, chi]))
    full.cam_T[:] = packed[:cam.size].reshape(cam.shape)
    full.obj_T[:] = loc.obj_T
    full.inlier[:] = np.round(packed[cam.size:cam.size + n_edge_total]).astype(np.uint8)
    full.chi2 = packed[cam.size + n_edge_total:]
    full.stats[:] = [rounds, lm_its, lm_trials, num_good]
    if hasattr(ph, "close"):
        ph.close()
    return full

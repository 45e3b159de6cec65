"""Multi-GPU global bundle adjustment (BASELINE config 5): cameras (views) are partitioned across
ranks, every rank keeps all object poses, and each Levenberg-Marquardt trial exchanges only the reduced
(Schur-complement) object system -- (6 n_obj)^2 + 6 n_obj doubles -- with one all-reduce, plus one all-reduce of
three scalars for the gain ratio.  With RCCL over xGMI on GPUs (backend "nccl") both reduce device buffers in place --
the phase kernels write them, the collective sums them, the next phase kernel reads them, all ordered on one stream;
"gloo" in the CPU tests.

The per-rank work runs in the phase kernels of csrc/lm_dist.hip (``HipPhases``); this module is the host
schedule: g2o's lambda / nu logic (thirdparty/g2opy/g2o/core/optimization_algorithm_levenberg.cpp:58-150) and
the robust rounds of ObjectSLAM.optimize (lib/object_slam.py:842-896), executed identically on every rank
because every decision is taken on all-reduced quantities.  With one rank it degenerates to the same algorithm
as the single-kernel path (csrc/lm.hip) and is tested against it and against the oracle.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from . import ba as _ba

DBL_MAX = float(np.finfo(np.float64).max)
COLLECTIVES_PER_TRIAL = 2          # [S | r | ok] and [chi2 | scale | ok]; +1 per LM iteration for the linearisation totals
_DIAG21 = (0, 6, 11, 15, 18, 20)


_PINNED = {}


def _pinned(n):
    """A pinned read-back buffer of >= n doubles, kept for the process: pinning one per adjustment is a hipHostMalloc (0.2-0.5 ms) per call."""
    cap = 1 << max(6, (int(n) - 1).bit_length())
    if cap not in _PINNED:
        _PINNED[cap] = torch.empty(cap, dtype=torch.float64).pin_memory()
    return _PINNED[cap]


class HipPhases:
    """One rank's share of the graph, resident on its GPU (suo_ba_ctx_* of include/suo_hip.h), with the exchange buffers
    of the LM schedule as DEVICE tensors that the collectives reduce in place:
        lin  [1 + 27 n_obj + world]  linearisation totals: chi2 | (Hoo 21 + bo 6) per object | one max-|diag Hcc| slot per rank
        sch  [ns^2 + ns + 1]         local Schur complement S_g | r_g | ok
        red  [4]                     after the step: chi2 | scale over own cameras | ok | scale over objects (not summed)
        good [1]                     chi2-inlier count of the own edges
    Every phase is a stream-ordered launch on torch's current stream; nothing is staged through host memory."""

    def __init__(self, local: _ba.Problem, world: int = 1):
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
        dev = torch.device("cuda", torch.cuda.current_device())
        self.lin = torch.zeros(1 + 27 * self.n_obj + world, dtype=torch.float64, device=dev)
        self.sch = torch.zeros(self.ns * self.ns + self.ns + 1, dtype=torch.float64, device=dev)
        self.red = torch.zeros(4, dtype=torch.float64, device=dev)
        self.good = torch.zeros(1, dtype=torch.float64, device=dev)
        self._pin = _pinned(max(self.lin.numel(), 16))
        # device-resident LM schedule (csrc/lm_dist.hip: ctl): this rank's own linearisation totals (the in-place reduce starts from them every
        # unit) and the control block g2o's accept / reject arithmetic lives in
        self.lin_loc = torch.zeros_like(self.lin)
        self.ctl = torch.zeros(16, dtype=torch.float64, device=dev)
        self._graphs, self._graph_failed = {}, False

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def classify(self, keep_all):
        _lib.check(self.lib.suo_ba_classify_dev(self._h, int(keep_all), self._p(self.good), self._stream()), "suo_ba_classify_dev")

    def linearize(self, robust_on, rank, world):
        _lib.check(self.lib.suo_ba_linearize_dev(self._h, int(robust_on), rank, world, self._p(self.lin), self._stream()), "suo_ba_linearize_dev")

    def schur(self, lam):
        _lib.check(self.lib.suo_ba_schur_dev(self._h, float(lam), self._p(self.sch), self._stream()), "suo_ba_schur_dev")

    def solve_update(self, lam, robust_on, world):
        _lib.check(self.lib.suo_ba_solve_update_dev(self._h, float(lam), int(robust_on), world, self._p(self.lin), self._p(self.sch),
                                                    self._p(self.red), self._stream()), "suo_ba_solve_update_dev")

    def restore(self):
        _lib.check(self.lib.suo_ba_restore_dev(self._h, self._stream()), "suo_ba_restore_dev")

    def begin_round(self, its, world):
        _lib.check(self.lib.suo_ba_lm_begin_dev(self._h, self._p(self.ctl), int(its), int(world), self._stream()), "suo_ba_lm_begin_dev")

    def unit(self, robust_on, rank, world, reduce_):
        """One unit of the device-resident schedule: [linearise] -> reduce -> [lambda init] Schur -> reduce -> solve + update + chi2 -> reduce ->
        decide [+ restore]; the phases that the control block does not call for return at once.  No host synchronisation."""
        s = self._stream()
        if world == 1 and not _collectives_forced() and os.environ.get("SUO_BA_FOLD_CTL", "1") not in ("", "0"):
            # nothing is exchanged between the phases: the control steps ride in the tail kernels (csrc/geom_api.hip: suo_ba_lm_unit_one_rank_dev)
            _lib.check(self.lib.suo_ba_lm_unit_one_rank_dev(self._h, int(robust_on), self._p(self.ctl), self._p(self.lin_loc), self._p(self.lin), self._p(self.sch),
                                                            self._p(self.red), s), "suo_ba_lm_unit_one_rank_dev")
            return
        _lib.check(self.lib.suo_ba_lm_linearize_dev(self._h, int(robust_on), rank, world, self._p(self.ctl), self._p(self.lin_loc), self._p(self.lin), s),
                   "suo_ba_lm_linearize_dev")
        reduce_(self.lin)
        _lib.check(self.lib.suo_ba_lm_schur_dev(self._h, self._p(self.ctl), self._p(self.lin), self._p(self.sch), s), "suo_ba_lm_schur_dev")
        reduce_(self.sch)                                  # the pose-graph reduce: [S | r | ok-count], in place
        _lib.check(self.lib.suo_ba_lm_solve_update_dev(self._h, int(robust_on), world, self._p(self.ctl), self._p(self.lin), self._p(self.sch),
                                                       self._p(self.red), s), "suo_ba_lm_solve_update_dev")
        reduce_(self.red[:3])
        _lib.check(self.lib.suo_ba_lm_decide_dev(self._h, self._p(self.ctl), self._p(self.red), s), "suo_ba_lm_decide_dev")

    def unit_replay(self, robust_on, rank, world, reduce_):
        """The same unit as a captured hipGraph (one per robust_on): its launches are identical from trial to trial -- lambda and the
        live / dead decision of every phase come from the control block -- so 17 launches become one graph launch (host 10-16 us instead
        of ~80, kernel boundaries ~1.5 us).  SUO_BA_GRAPH=0: eager; 1 (default): captured when the unit holds no collective (one rank);
        2: captured with its collectives (RCCL inside hipGraph capture).  The graphs live on this object, i.e. for ONE adjustment: capture + instantiation + first replay
        measure 0.30 + 0.25 ms of a 9 ms adjustment (two graphs: robust kernel on / off) -- keeping them across adjustments would need the exchange tensors and the
        context at fixed addresses, for 5 %; not done."""
        mode = int(os.environ.get("SUO_BA_GRAPH", "1"))
        has_collectives = world > 1 or _collectives_forced()
        if mode <= 0 or (has_collectives and mode < 2) or self._graph_failed:
            return self.unit(robust_on, rank, world, reduce_)
        key = (int(robust_on), rank, world, has_collectives)
        g = self._graphs.get(key)
        if g is None:
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):                   # (capture only: nothing of the unit runs here)
                    self.unit(robust_on, rank, world, reduce_)
                self._graphs[key] = g
            except Exception as exc:                        # capture is an optimisation: fall back to eager launches for good -- and say why, once
                import sys
                print(f"suo_slam_amd.ba_dist: hipGraph capture of the LM unit failed ({type(exc).__name__}: {exc}); eager launches from here on", file=sys.stderr)
                self._graph_failed = True
                torch.cuda.synchronize()
                return self.unit(robust_on, rank, world, reduce_)
        g.replay()

    def read(self, t):
        """The host's look at a few reduced scalars (what the LM schedule decides on): one small D2H into pinned memory."""
        n = t.numel()
        self._pin[:n].copy_(t, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self._pin[:n].numpy().copy()

    def download(self):
        torch.cuda.current_stream().synchronize()
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


def _collectives_forced():
    """SUO_FORCE_COLLECTIVES=1: issue every collective even in a process group of ONE rank (where a SUM all-reduce is the identity and
    is otherwise skipped) -- so that a single-GPU box executes the very calls an 8-GPU node will make: the RCCL communicator bring-up,
    the in-place all-reduce of `lin` / `sch` / the `red[:3]` view, and their stream ordering against the phase kernels
    (tests/test_gpu_rccl.py; bench.py's global_ba leg)."""
    return os.environ.get("SUO_FORCE_COLLECTIVES", "0") not in ("", "0") and dist.is_available() and dist.is_initialized()


def _reduce_(t):
    """In-place SUM all-reduce of an exchange buffer.  RCCL ("nccl") reduces the device tensor where it lies; with gloo (CPU
    tests; rehearsals with several ranks on one GPU) a device tensor is staged through the host."""
    rank, world = _world()
    if world == 1 and not _collectives_forced():
        return t
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def _allreduce(arr):
    """SUM all-reduce of a small host array (result assembly after the last round only)."""
    rank, world = _world()
    a = np.ascontiguousarray(arr, np.float64)
    if world == 1 and not _collectives_forced():
        return a.copy()
    t = torch.from_numpy(a.copy())
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
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


def host_schedule_forced():
    """SUO_BA_HOST_SCHEDULE=1: the round-2/3 schedule (the host reads four doubles per trial and decides) also for the HIP phases -- A/B and
    the equivalence test of the device-resident schedule."""
    return os.environ.get("SUO_BA_HOST_SCHEDULE", "0") not in ("", "0")


def optimize_distributed(full: _ba.Problem, phases_factory=HipPhases):
    """Run the robust LM rounds of `full` across the ranks of the default process group.  Every rank passes the
    same `full` problem and gets the complete result back (cam_T, obj_T, inlier, chi2, stats) in `full`.

    Per LM iteration: one collective (linearisation totals; the lambda-init maximum travels in per-rank slots of the same SUM).
    Per LM trial: COLLECTIVES_PER_TRIAL = 2 collectives -- the pose-graph reduce [S | r | ok] and the 3-scalar step result --
    both in place on device buffers.  With the HIP phases g2o's accept / reject decision is taken ON THE DEVICE (control block, csrc/lm_dist.hip)
    and the host reads 16 doubles once per batch of units (a round of n iterations whose trials are all accepted: once); the numpy phases of
    the CPU tests -- and SUO_BA_HOST_SCHEDULE=1 -- keep the schedule on the host, one read of 4 doubles per trial.  A trial on a standing
    linearisation (after a rejected one) repeats the iteration's collective: the units are uniform, a repeat costs 3.5 KB."""
    rank, world = _world()
    local, cams, sel = split_problem(full, rank, world)
    ph = phases_factory(local, world)
    try:
        O = len(full.obj_T)
        n_edge_total = len(full.edge_cam)
        free_obj = [o for o in range(O) if not full.obj_fixed[o]]
        rounds = lm_its = lm_trials = 0

        def classify(keep_all):
            ph.classify(keep_all)
            return int(round(float(ph.read(_reduce_(ph.good))[0])))
        if full.init_with_outliers:
            classify(True)
            num_good = n_edge_total
        else:
            num_good = classify(False)
        robust_on = True
        drop = max(1, len(full.its) // 2)
        device_schedule = hasattr(ph, "unit") and not host_schedule_forced()
        for rnd, its in enumerate(full.its):
            if n_edge_total < 4 or num_good < 4:
                break
            rounds += 1
            if device_schedule:
                # g2o's lambda / nu / gain-ratio logic runs on the device (csrc/lm_dist.hip: ba_ctl_*_kernel); the host enqueues units blindly --
                # `its` of them cover a round whose every trial is accepted, a rejected trial costs one more -- and looks at the control block
                # once per batch.  Every rank enqueues the same units: the decisions are taken on all-reduced quantities.
                ph.begin_round(int(its), world)
                # units between two looks at the control block: a dead unit (enqueued past the end of a round) costs ~50 us of empty launches on one rank, and three
                # all-reduces more on several -- there the batches are half as long
                batch = int(os.environ.get("SUO_BA_UNITS_PER_LOOK", "12" if world == 1 else "6"))
                budget, done = min(int(its), batch), int(its) <= 0
                while not done:
                    for _ in range(budget):
                        (ph.unit_replay if hasattr(ph, "unit_replay") else ph.unit)(robust_on, rank, world, _reduce_)
                    ctl = ph.read(ph.ctl)
                    done = int(ctl[3]) == 2
                    budget = min(max(1, int(its) - int(ctl[4])) + 1, batch)
                lm_its, lm_trials = int(ctl[7]), int(ctl[8])
                num_good = classify(False)
                if rnd == drop:
                    robust_on = False
                continue
            lam, ni = -1.0, 2.0
            for it in range(int(its)):
                ph.linearize(robust_on, rank, world)
                lin = ph.read(_reduce_(ph.lin))
                current_chi, HB = float(lin[0]), lin[1:1 + 27 * O]
                if it == 0:                                        # computeLambdaInit: tau * max |diag H| over all free vertices
                    maxd = float(lin[1 + 27 * O:].max())
                    for o in free_obj:
                        maxd = max(maxd, max(abs(HB[27 * o + d]) for d in _DIAG21))
                    lam, ni = 1e-5 * maxd, 2.0
                rho, qmax, lam_finite = 0.0, 0, True
                while True:
                    ph.schur(lam)
                    _reduce_(ph.sch)                               # the pose-graph reduce: [S | r | ok-count], in place
                    ph.solve_update(lam, robust_on, world)         # refuses the step unless every rank's Schur phase was ok
                    _reduce_(ph.red[:3])
                    red = ph.read(ph.red)                          # [chi2 | scale_cams | ok-count | scale_objs]
                    temp_chi, scale = DBL_MAX, 0.0
                    if int(round(red[2])) == world:
                        temp_chi, scale = float(red[0]), float(red[1]) + float(red[3])
                    rho = (current_chi - temp_chi) / (scale + 1e-3)
                    if rho > 0 and math.isfinite(temp_chi):
                        r21 = 2 * rho - 1
                        alpha = min(1.0 - r21 * r21 * r21, 2.0 / 3.0)        # (the products as the kernels form them: the two schedules agree bit for bit)
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
            num_good = classify(False)
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
        return full
    finally:
        # (also on an exception between two phases: the context's kernels run on the CURRENT stream -- suo_ba_ctx_destroy drains the device before it
        #  parks the buffers, csrc/geom_api.hip)
        if hasattr(ph, "close"):
            ph.close()

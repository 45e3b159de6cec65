#!/usr/bin/env python3
"""Headline benchmark of the suo_slam hot path on MI355X (contract: see the task statement).

One STEP = one network call over `--frames-per-step` (32) consecutive synthetic YCB-V-shaped frames (640x480 uint8,
8 object boxes each = 256 crops) through the whole per-frame path of BASELINE.json configs[1] (single-view eval, no SLAM):
    RoI crop + prior concat -> stacked-hourglass keypoint CNN (fp32 MFMA) -> heat-map decode -> validity masks
    -> D2H of uv / cov / masks (lib/object_slam.py:1100-1109) -> [the step's geometry waits for that read-back]
    -> batched P3P-RANSAC PnP (all objects of the step in one launch) -> uncertainty-weighted LM, rounds [10,10,40,40].
`value` is frames/s = steps * frames_per_step * n_gpus / elapsed; every network call processes exactly the frames that are
counted (no tail call, no partially filled batch).  Frames of the single-view stream are independent (evaluate.py:345-346),
which is what allows batching them; the reference's own call shape (one frame per call) is timed separately after the
timed region and reported as config.latency_mode_fps.
The network has random weights (no checkpoint ships), so -- like the reference's --debug_gt_kp mode
(lib/object_slam.py:1129-1131) -- PnP / LM are driven by projected ground-truth keypoints + N(0, 0.01^2) noise
with random SPD covariances, while the CNN runs on the frame's pixels and its outputs are read back; nothing is skipped or
cached.  Inputs (images, boxes) are resident in HBM before the timed region; PnP / LM take the small host arrays the
reference's FFI hands over (their H2D/D2H is inside the timed region).
Outside the timed region: `roofline` (dominant kernel under HIP events), `global_ba` (BASELINE configs[4]'s exchange step:
one 16-object global pose-graph adjustment with its cameras partitioned over the ranks, reduced system all-reduced over
RCCL per LM trial), `cpu_baseline` (the oracle on the host cores, rank 0 at N=1 only).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_CROP = 31.495          # conv FLOPs, hook-counted on the reference module
# Without priors (this workload: single-view frames, lib/object_slam.py:1094-1097 feeds zeros) 41 of the stem's 44 input
# channels are structural zeros and their MACs are never issued (csrc/net.hip: stem_img_): 2*128*128*64*49*41 per crop.
GFLOP_SKIPPED_PER_CROP = 2 * 128 * 128 * 64 * 49 * 41 / 1e9
FP32_MFMA_PEAK_TF = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md chip table


def winograd_saved_gflop_per_crop(crops_per_call):
    """MACs the Winograd F(2x2,3x3) form does not execute (csrc/conv_wino.hip): the 3x3 convolution of a Residual block (128 -> 128,
    or 64 -> 64 in r1 / r4) runs in that form when its launch has >= 256 tiles of 8 x 16 pixels (conv3x3_wino_pays), at 16 instead
    of 36 products per 2x2 tile.  Such convolutions per crop (hg.py:7-58, 2 stacks): 128 channels -- 9 at 64x64 (r5, up1 and the
    post-hourglass blocks), 12 at 32x32, 12 at 16x16; 64 channels -- r1 at 128x128, r4 at 64x64."""
    saved = 0.0
    for hw, count, ch in ((64, 9, 128), (32, 12, 128), (16, 12, 128), (128, 1, 64), (64, 1, 64)):
        tiles = crops_per_call * (hw // 8) * (hw // 16)
        if tiles >= 256:
            saved += count * 2.0 * hw * hw * ch * ch * 9 * (1 - 1 / 2.25) / 1e9
    return saved


N_OBJ = 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="network calls in the timed region (one step = --frames-per-step frames)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--objects", type=int, default=N_OBJ)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--pool", type=int, default=0, help="number of distinct synthetic frames cycled through (0 = 2 steps' worth)")
    ap.add_argument("--depth", type=int, default=2, help="network calls in flight (independent network instances / streams)")
    ap.add_argument("--frames-per-step", "--frames-per-forward", dest="frames_per_step", type=int, default=32,
                    help="frames of the stream batched into one network call = one step (--objects crops each)")
    ap.add_argument("--only", choices=["all", "cnn", "geometry"], default="all", help="diagnostic: run only one half of the step")
    ap.add_argument("--no-latency-leg", action="store_true", help="skip the one-frame-per-call measurement reported as config.latency_mode_fps")
    ap.add_argument("--no-global-ba-leg", action="store_true", help="skip the (multi-GPU) global pose-graph adjustment reported as global_ba")
    ap.add_argument("--legs-timeout", type=int, default=420, help="seconds the legs after the timed region may take before the line is printed without them")
    return ap.parse_args()


def make_pool(rng, n, L):
    from suo_slam_amd import geometry as geo
    from suo_slam_amd import synthetic as S
    pool = []
    for _ in range(n):
        fr = S.make_frame(rng, L, noise=0.01, outlier_frac=0.05)
        xs, ys = [], []
        for o in range(L):
            m = fr["model_kps_masks"][o]
            xs.append(fr["model_kps"][o][m].astype(np.float64))
            ys.append(geo.normalize_uv(fr["uv"][o][m].astype(np.float64), fr["K_bbox"][o].astype(np.float32).astype(np.float64)))
        fr["pnp_xs"], fr["pnp_ys"] = xs, ys
        fr["ba"] = S.frame_to_ba_problem(fr, np.tile(np.eye(4)[None], (L, 1, 1)))
        pool.append(fr)
    return pool


class FramePipeline:
    """The per-frame product path, called through the C ABI with pre-allocated device buffers.

    One STEP = one network call over F consecutive frames of the stream (L crops each) followed by the geometry of
    exactly those frames:
        forward (RoI crop, CNN, decode) + keypoint masks on the step's stream  ->  D2H of uv / cov / kp_mask / masks into
        pinned host buffers (the three .cpu() of lib/object_slam.py:1100-1109)  ->  [host waits for that copy]  ->
        ONE PnP launch (a wave per object) and ONE LM launch (a workgroup per frame) for the F frames.
    `depth` steps are in flight: step i runs on slot i % depth (own network workspace, hipGraph, stream).  Before a slot is
    reused the host awaits the read-back of the step that used it, hands that step's geometry to the geometry thread (one
    thread, steps in order) and launches the new network call.  Frames of the single-view stream are independent
    (evaluate.py:345-346 resets the SLAM state per frame).  Every network call processes exactly the frames that are
    counted: there is no tail call."""

    def __init__(self, L, pool, F, use_graph=True, only="all", depth=2):
        import torch
        from suo_slam_amd import _lib, ba, lambdatwist, weights
        from suo_slam_amd.pkpnet import PkpNet
        self.only = only
        self.torch, self.lib, self._lib, self.ba, self.lt = torch, _lib.lib(), _lib, ba, lambdatwist
        self.L, self.F, self.depth = L, F, depth
        sd = weights.make_random_state_dict(0, 8.0)
        dev = "cuda"
        self.pool = pool
        assert len(pool) % F == 0, "--pool must be a multiple of --frames-per-step"
        self.n_groups = len(pool) // F
        # the frame stream is resident in HBM as one stack; a network call takes F consecutive frames (L*F crops)
        self.imgs = torch.from_numpy(np.stack([fr["image"] for fr in pool])).to(dev)
        self.g_boxes, self.g_img, self.g_mm = [], [], []
        for g in range(self.n_groups):
            ks = range(g * F, (g + 1) * F)
            self.g_boxes.append(torch.from_numpy(np.concatenate([pool[k]["boxes"] for k in ks])).to(dev))
            self.g_img.append(torch.from_numpy(np.repeat(np.array(list(ks), np.int32), L)).to(dev))
            self.g_mm.append(torch.from_numpy(np.concatenate([pool[k]["model_kps_masks"] for k in ks]).astype(np.uint8)).to(dev))
        LF = L * F
        self.slots = []
        for _ in range(depth):
            net = PkpNet(state_dict=sd, max_crops=LF)
            net.set_graph(use_graph)
            ts = torch.cuda.Stream()      # a real (non-NULL) stream: hipGraph replay is then fully asynchronous
            S = {"net": net, "tstream": ts, "stream": C.c_void_p(ts.cuda_stream), "busy": None, "event": torch.cuda.Event(),
                 "uv": torch.empty((LF, 41, 2), device=dev), "cov": torch.empty((LF, 41, 2, 2), device=dev),
                 "kp": torch.empty((LF, 41), device=dev), "mask": torch.empty((LF, 41), dtype=torch.uint8, device=dev)}
            for k in ("uv", "cov", "kp", "mask"):
                S["h_" + k] = torch.empty(S[k].shape, dtype=S[k].dtype).pin_memory()
            self.slots.append(S)
        # PnP / LM of a step run on ONE worker thread, in step order (the C ABI releases the GIL): the thread that launches network
        # calls never waits for geometry, and a step's geometry still starts only after that step's read-back was awaited
        from concurrent.futures import ThreadPoolExecutor
        self.worker = ThreadPoolExecutor(max_workers=1, initializer=torch.cuda.set_device, initargs=(torch.cuda.current_device(),))   # HIP's current device is per thread
        self.futures = []
        self.reset_metrics()

    def reset_metrics(self):
        self.pose_err, self.n_pose, self.n_inl, self.n_frames, self.n_crops, self.n_net_kp = 0.0, 0, 0, 0, 0, 0

    def step(self, i):
        """Step i: await the outputs of the step that last used slot i % depth, launch this step's network call on the slot,
        THEN run the awaited step's geometry -- so `depth` network calls stay in flight while the host does PnP / LM."""
        S = self.slots[i % self.depth]
        done = self.await_outputs(S)
        g = i % self.n_groups
        if self.only != "geometry":
            P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
            LF = self.L * self.F
            self._lib.check(self.lib.suo_net_forward_frames(S["net"]._h, P(self.imgs), 0, 480, 640, P(self.g_boxes[g]), P(self.g_img[g]), LF, None,
                                                            P(S["uv"]), P(S["cov"]), P(S["kp"]), None, None, S["stream"]), "suo_net_forward_frames")
            self._lib.check(self.lib.suo_keypoint_masks(P(S["uv"]), P(S["cov"]), P(S["kp"]), P(self.g_mm[g]), LF, 0.9, 0.2, P(S["mask"]),
                                                        S["stream"]), "suo_keypoint_masks")
            with self.torch.cuda.stream(S["tstream"]):
                for k in ("uv", "cov", "kp", "mask"):
                    S["h_" + k].copy_(S[k], non_blocking=True)
                S["event"].record(S["tstream"])
            self.n_crops += LF
        S["busy"] = (g, i)
        if done is not None:
            self.futures.append(self.worker.submit(self.geometry, done))

    def await_outputs(self, S):
        """Host waits until a step's uv / cov / kp_mask / masks have arrived in the pinned buffers (the reference's three
        .cpu() calls, lib/object_slam.py:1100-1109) and consumes them; the slot's device buffers are free afterwards."""
        if S["busy"] is None:
            return None
        done, S["busy"] = S["busy"], None
        if self.only != "geometry":
            S["event"].synchronize()
            self.n_net_kp += int(np.count_nonzero(S["h_mask"].numpy()))      # random weights: few keypoints pass the masks
            assert np.isfinite(S["h_uv"].numpy()).all() and np.isfinite(S["h_cov"].numpy()).all()
        return done

    def geometry(self, done):
        """PnP + LM for the F frames of a step whose network outputs have been awaited."""
        if done is None:
            return
        g, i = done
        self.n_frames += self.F
        if self.only == "cnn":
            return
        frames = self.pool[g * self.F:(g + 1) * self.F]
        xs = [x for fr in frames for x in fr["pnp_xs"]]
        ys = [y for fr in frames for y in fr["pnp_ys"]]
        T, status = self.lt.pnp_batch(xs, ys, 1e-3, seed=i)
        L = self.L
        probs = []
        for j, fr in enumerate(frames):
            B = fr["ba"]
            probs.append(self.ba.Problem(B["cam_T"], B["cam_fixed"], T[j * L:(j + 1) * L, :3, :], B["obj_fixed"], B["edge_cam"],
                                         B["edge_obj"], B["edge_camk"], B["edge_p"], B["edge_uv"], B["edge_info"], B["edge_inlier"],
                                         its=(10, 10, 40, 40)))
        self.ba.optimize_batch(probs)
        for j, (fr, prob) in enumerate(zip(frames, probs)):
            obj = prob.obj_T.reshape(-1, 3, 4)
            d = np.linalg.norm(obj[:, :, 3] - fr["T_OtoC"][:, :3, 3], axis=1) / fr["T_OtoC"][:, 2, 3]
            ok = status[j * L:(j + 1) * L] == 0
            self.pose_err += float(d[ok].sum())
            self.n_pose += int(ok.sum())
            self.n_inl += int(prob.inlier.sum())

    def drain(self, next_step):
        """Retire every step still in flight, oldest first."""
        for k in range(self.depth):
            done = self.await_outputs(self.slots[(next_step + k) % self.depth])
            if done is not None:
                self.futures.append(self.worker.submit(self.geometry, done))
        for f in self.futures:
            f.result()                      # (re-raises anything the geometry thread hit)
        self.futures = []


def pack_conv(w, Np, Cp, CK):
    from suo_slam_amd import _lib
    w = np.ascontiguousarray(w, np.float32)
    out = np.empty(2 * Np * ((Cp * w.shape[2] * w.shape[3] + 15) // 16 * 16), np.float32)
    _lib.check(_lib.lib().suo_pack_conv_weight(w.ctypes.data, w.shape[0], w.shape[1], w.shape[2], Np, Cp, CK, out.ctypes.data), "pack_conv")
    return out


def pack_gemm(w, Np, Kp):
    from suo_slam_amd import _lib
    w = np.ascontiguousarray(w, np.float32)
    out = np.empty(2 * Np * Kp, np.float32)
    _lib.check(_lib.lib().suo_pack_gemm_weight(w.ctypes.data, w.shape[0], w.shape[1], Np, Kp, out.ctypes.data), "pack_gemm")
    return out


def dominant_kernel_traffic(L):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (tools/profile_round.sh -> tools/pmc_to_json.py),
    or None when the summary is for another launch shape / kernel."""
    pmc = os.path.join(ROOT, "profiles", "pmc_dominant_conv.json")
    if not os.path.exists(pmc):
        return None
    rec = json.load(open(pmc))
    if rec.get("crops_per_launch") == L and rec.get("kernel", "").replace(" ", "").startswith("wino3x3_kernel<true"):
        return rec.get("hbm_bytes_per_launch")
    return None


def conv_roofline(L, iters=30):
    """Live HIP-event timing of the dominant kernel at the launch shape of the timed region: the tail of a 256 -> 256 Residual block
    at 64x64 in ONE launch -- conv2 (3x3, 128 -> 128, Winograd F(2x2,3x3)) + ReLU, conv3 (1x1, 128 -> 256) + skip:
    wino3x3_kernel<true> (8 launches per network call, ~35 % of its kernel time).  `achieved` counts the ALGORITHMIC FLOPs of the
    two convolutions (2 MACs x 9 taps for the 3x3, as SURVEY.md 8d counts the network's 31.495 GFLOP per crop); the Winograd form
    EXECUTES 2.25x fewer MACs for the 3x3 part, so `frac` can exceed 1 -- the executed MFMA rate is reported beside it, and the
    direct-form kernels (fused and plain 3x3) of the same tile shape are timed in the same process."""
    import torch
    from suo_slam_amd import _lib
    rng = np.random.default_rng(0)
    x = torch.rand((L, 64, 64, 128), device="cuda") - 0.5
    skip = torch.rand((L, 64, 64, 256), device="cuda") - 0.5
    w2 = (rng.standard_normal((128, 128, 3, 3)) / 34.0).astype(np.float32)
    w3 = (rng.standard_normal((256, 128)) / 11.0).astype(np.float32)
    lib = _lib.lib()
    wq = np.empty(16 * 128 * 128, np.float32)
    _lib.check(lib.suo_pack_wino_weight(np.ascontiguousarray(w2).ctypes.data, 128, 128, 128, 128, wq.ctypes.data), "pack_wino")
    wq2 = torch.from_numpy(wq).cuda()
    wp2 = torch.from_numpy(pack_conv(w2, 128, 128, 32)).cuda()
    wp3 = torch.from_numpy(pack_gemm(w3, 256, 128)).cuda()
    b2 = torch.zeros(128, device="cuda")
    b3 = torch.zeros(256, device="cuda")
    mid = torch.empty((L, 64, 64, 128), device="cuda")
    out = torch.empty((L, 64, 64, 256), device="cuda")
    st = torch.cuda.current_stream()
    s = C.c_void_p(st.cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def wino_fused():
        _lib.check(lib.suo_conv3x3_wino_conv1x1_skip(P(x), L, 64, 64, P(wq2), P(b2), P(wp3), P(b3), P(skip), P(out), s), "suo_conv3x3_wino_conv1x1_skip")

    def wino_plain():
        _lib.check(lib.suo_conv3x3_wino(P(x), L, 64, 64, 128, P(wq2), P(b2), P(mid), 128, 1, s), "suo_conv3x3_wino")

    def direct_fused():
        _lib.check(lib.suo_conv3x3_conv1x1_skip(P(x), L, 64, 64, P(wp2), P(b2), P(wp3), P(b3), P(skip), P(out), s), "suo_conv3x3_conv1x1_skip")

    def direct_plain():
        _lib.check(lib.suo_conv_kxk(3, P(x), L, 64, 64, 128, P(wp2), P(b2), P(mid), 128, 1, s), "suo_conv_kxk")

    def timed(f):
        try:
            f()
        except Exception:                                        # (the direct-form fused kernel refuses launches below 1024 tiles)
            return float("nan")
        for _ in range(10):                                      # (the first launches of a kernel in a process run 5-25 % slow)
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            f()
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters
    us, us_wp, us_df, us_dp = timed(wino_fused), timed(wino_plain), timed(direct_fused), timed(direct_plain)
    px = float(L) * 64 * 64
    flop3, flop1 = 2.0 * px * 128 * 128 * 9, 2.0 * px * 128 * 256
    flop = flop3 + flop1
    flop_exec = flop3 / 2.25 + flop1                               # 16 products per 2x2 tile and channel pair instead of 36
    ach = flop / (us * 1e-6) / 1e12
    tf = lambda f, t: round(f / (t * 1e-6) / 1e12, 2) if t == t else None  # noqa: E731
    # HBM traffic per launch: rocprofv3 --pmc passes of this same kernel / launch shape (FETCH_SIZE doubled as the
    # microarch guide prescribes for gfx950, WRITE_SIZE as reported), collected by tools/profile_round.sh, stored under profiles/
    traffic = dominant_kernel_traffic(L)
    return {"bound": "mfma", "kernel": "wino3x3_kernel<true> fused Residual tail: 3x3 128->128 (Winograd F(2x2,3x3)) + ReLU, 1x1 128->256 + skip @64x64, "
                                       "%d crops/launch" % L,
            "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TF, 4),
            "traffic": traffic, "avg_launch_us": round(us, 2), "flop_per_launch": flop,
            "algorithmic_bytes_per_launch": 4.0 * px * (128 + 256 + 256) + 4.0 * (128 * 128 * 16 + 128 * 256),
            "executed_flop_per_launch": flop_exec, "executed_tflops": tf(flop_exec, us), "executed_frac_of_mfma_peak": round(tf(flop_exec, us) / FP32_MFMA_PEAK_TF, 4),
            "same_process": {"wino3x3_kernel<false> (3x3 alone)": {"avg_launch_us": round(us_wp, 2), "algorithmic_tflops": tf(flop3, us_wp),
                                                                     "executed_frac_of_mfma_peak": round(tf(flop3 / 2.25, us_wp) / FP32_MFMA_PEAK_TF, 4)},
                             "convk_kernel<3,1,32,8,16,2,2,2,2,true> (direct, fused tail)": {"avg_launch_us": round(us_df, 2) if us_df == us_df else None, "tflops": tf(flop, us_df),
                                                                                            "frac": round(tf(flop, us_df) / FP32_MFMA_PEAK_TF, 4) if us_df == us_df else None},
                             "convk_kernel<3,1,32,8,16,2,2,2,2,false> (direct 3x3 alone)": {"avg_launch_us": round(us_dp, 2), "tflops": tf(flop3, us_dp),
                                                                                           "frac": round(tf(flop3, us_dp) / FP32_MFMA_PEAK_TF, 4)}}}


def cpu_baseline(pool, L):
    """The oracle (CPU restatement) timed on this box's host cores on a bounded sample of the same workload."""
    import torch
    from oracle import cnn_oracle as O
    from oracle import geometry as G
    from suo_slam_amd import weights
    n = min(len(os.sched_getaffinity(0)), 64)
    torch.set_num_threads(n)
    sd = weights.make_random_state_dict(0, 8.0)
    Pw = O.to_torch(sd)
    t0 = time.perf_counter()
    n_cnn = 0
    while n_cnn < 1 or (time.perf_counter() - t0 < 10.0 and n_cnn < 8):
        fr = pool[n_cnn % len(pool)]
        O.pkpnet_forward(fr["image"], fr["boxes"], None, sd, Pw)
        n_cnn += 1
    t_cnn = (time.perf_counter() - t0) / n_cnn
    t0 = time.perf_counter()
    n_geo = 0
    for rep in range(3):
        for fr in pool:
            init = []
            for o in range(L):
                T, _, _ = G.pnp(fr["pnp_xs"][o], fr["pnp_ys"][o], 1e-3, seed=o)
                init.append(T[:3])
            B = fr["ba"]
            G.optimize(B["cam_T"], B["cam_fixed"], np.array(init), B["obj_fixed"], B["edge_cam"], B["edge_obj"], B["edge_camk"], B["edge_p"],
                       B["edge_uv"], B["edge_info"], B["edge_inlier"])
            n_geo += 1
    t_geo = (time.perf_counter() - t0) / n_geo
    # pose parity of the HIP geometry against this oracle on identical inputs (same sampler seeds): a sample of the pool
    from suo_slam_amd import ba, lambdatwist
    dT = dR = 0.0
    n_cmp = min(8, len(pool))
    for k in range(n_cmp):
        fr = pool[k]
        T, status = lambdatwist.pnp_batch(fr["pnp_xs"], fr["pnp_ys"], 1e-3, seed=k)
        init = []
        for o in range(L):
            To, _, _ = G.pnp(fr["pnp_xs"][o], fr["pnp_ys"][o], 1e-3, seed=(k + o * lambdatwist.SEED_STRIDE) % 2 ** 64)
            dT = max(dT, float(np.abs(T[o] - To).max()))
            init.append(To[:3])
        B = fr["ba"]
        a = (B["cam_T"], B["cam_fixed"], np.array(init), B["obj_fixed"], B["edge_cam"], B["edge_obj"], B["edge_camk"], B["edge_p"], B["edge_uv"],
             B["edge_info"], B["edge_inlier"])
        got, ref = ba.optimize(*a), G.optimize(*a)
        assert np.array_equal(got[2], ref[2]), "HIP and oracle disagree on the inlier flags"
        dR = max(dR, float(np.abs(got[1][:, :, :3] - ref[1][:, :, :3]).max()))
        dT = max(dT, float((np.abs(got[1][:, :, 3] - ref[1][:, :, 3]) / np.abs(ref[1][:, :, 3]).max()).max()))
    parity = {"frames": n_cmp, "objects": n_cmp * L, "max_abs_dR_entry": float(f"{dR:.3e}"), "max_rel_dt": float(f"{dT:.3e}"), "inlier_flags": "identical"}
    return {"pose_parity_vs_oracle": parity, "value": round(1.0 / (t_cnn + t_geo), 4), "unit": "frames/s", "cores": n, "kind": "port",
            "sample": f"{n_cnn} frames x {L} crops through the torch-CPU CNN oracle ({n} threads, {t_cnn * 1e3:.0f} ms/frame) + "
                      f"{n_geo} frames through the C PnP/LM oracle (1 thread, {t_geo * 1e3:.2f} ms/frame)"}


def latency_leg(L, pool, use_graph, seconds=0.6):
    """The reference's call shape (evaluate.py:338-395: one frame per network call), four calls in flight (470 frames/s with two,
    484 with four, six or eight; odd depths lose 8 %): not `value`, reported as config.latency_mode_fps."""
    import torch
    pipe = FramePipeline(L, pool, 1, use_graph=use_graph, depth=4)
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
    return n / (time.perf_counter() - t0)


def global_ba_leg(world, L, n_cam_per_rank=32, reps=3):
    """BASELINE configs[4]'s exchange step: ONE global pose-graph adjustment (first camera fixed, all other cameras and all
    L objects free, lib/object_slam.py:746-778) whose cameras are partitioned over the ranks; each LM trial all-reduces the
    reduced object system over RCCL (suo_slam_amd/ba_dist.py).  Weak scaling: n_cam_per_rank cameras per GPU."""
    from suo_slam_amd import ba, ba_dist
    from suo_slam_amd import synthetic as S
    n_cam = n_cam_per_rank * world
    P = S.make_pose_graph(np.random.default_rng(5), n_cam, L)
    keys = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")
    ts = []
    for _ in range(reps):
        full = ba.Problem(*[P[k].copy() for k in keys])
        t0 = time.perf_counter()
        ba_dist.optimize_distributed(full)
        ts.append(time.perf_counter() - t0)
    err = float(max(np.linalg.norm(full.obj_T.reshape(-1, 3, 4)[o][:, 3] - P["obj_gt"][o][:, 3]) for o in range(L)))
    return {"ranks": world, "cameras": n_cam, "objects": L, "edges": int(len(P["edge_cam"])), "ms": round(1e3 * min(ts), 2),
            "lm_trials": int(full.stats[2]), "collectives_per_trial": ba_dist.COLLECTIVES_PER_TRIAL,
            "reduce_bytes_per_trial": int(8 * ((6 * L) ** 2 + 6 * L + 4)),
            "max_object_translation_err_mm": round(err, 3), "inlier_edges": int(full.inlier.sum())}


def main():
    args = parse()
    import faulthandler
    faulthandler.dump_traceback_later(1500, exit=True)        # a hung run leaves with every thread's stack instead of holding the box
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    # (SUO_LOCAL_DEVICE / SUO_DIST_BACKEND: rehearsal of the multi-process flow on a box with fewer GPUs than ranks --
    #  e.g. two ranks sharing GPU 0 over gloo; RCCL itself refuses duplicate devices.  Not used by the driver.)
    local = int(os.environ.get("SUO_LOCAL_DEVICE", local))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SUO_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    L, F = args.objects, args.frames_per_step
    # frames shard embarrassingly: rank r processes its own stream (weak scaling: K steps = K*F frames per GPU)
    n_pool = args.pool if args.pool > 0 else 2 * F
    pool = make_pool(np.random.default_rng(1000 + rank), n_pool, L)
    pipe = FramePipeline(L, pool, F, use_graph=not args.no_graph, only=args.only, depth=args.depth)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        pipe.step(i)
    pipe.drain(args.warmup)
    pipe.reset_metrics()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        pipe.step(args.warmup + i)
    pipe.drain(args.warmup + args.steps)        # every timed step's read-back and geometry complete inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    assert pipe.n_frames == args.steps * F and (args.only == "geometry" or pipe.n_crops == args.steps * F * L)
    # max-over-ranks time + the only collective of the frame path: metric accumulators (RCCL all-reduce over xGMI)
    from suo_slam_amd import sharding
    dt, (pose_err, n_pose, n_inl, n_net_kp) = sharding.reduce_metrics(dt, [pipe.pose_err, pipe.n_pose, pipe.n_inl, pipe.n_net_kp],
                                                                      device="cuda" if os.environ.get("SUO_DIST_BACKEND", "nccl") == "nccl" else "cpu")
    # ---- the line: built from the timed region first, then extended by the legs that run AFTER it (roofline, global BA,
    # latency mode, cpu baseline).  A watchdog prints what exists and leaves if those legs do not finish in time: they are
    # reported beside `value`, they must never cost it.
    import threading
    line = None
    if rank == 0:
        frames = world * args.steps * F
        fps = frames / dt
        exec_gflop = GFLOP_PER_CROP - GFLOP_SKIPPED_PER_CROP - winograd_saved_gflop_per_crop(L * F)
        line = {
            "metric": "frames/sec (obj-crops/sec) YCB-V 640x480 8-obj; ADD(-S) vs ref",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "YCB-V single-view eval (BASELINE configs[1]): 640x480 frame, %d objects -> RoI crop, hourglass keypoint "
                                   "CNN fp32, decode, masks, D2H of uv/cov/masks, batched PnP, LM rounds [10,10,40,40]" % L,
                       "step": "one network call over frames_per_step consecutive frames + the PnP/LM of those frames (which waits for "
                               "that call's read-back)",
                       "frames_per_step": F, "objects_per_frame": L, "crops_per_step": L * F, "crops_per_s": round(fps * L, 2),
                       "frames_timed": frames, "timed_region_s": round(dt, 4), "steps_in_flight": args.depth,
                       "geometry_inputs": "projected GT keypoints + N(0,0.01^2) NDC noise, 5% outliers (debug_gt_kp mode: random weights "
                                          "give meaningless keypoints); the network's uv/cov/masks are read back and awaited first",
                       "parallelism": f"frame-sharded x{world}, no data-path collective"},
            "cnn_tflops_algorithmic": round(fps * L * GFLOP_PER_CROP / 1e3, 2),           # reference-counted FLOPs per crop x crops/s
            "cnn_tflops_executed": round(fps * L * exec_gflop / 1e3, 2),                  # zero-prior MACs not issued, Winograd 3x3 at 16/36
            "cnn_executed_frac_of_fp32_mfma_peak": round(fps / world * L * exec_gflop / 1e3 / FP32_MFMA_PEAK_TF, 4),
            "cnn_algorithmic_frac_of_fp32_mfma_peak": round(fps / world * L * GFLOP_PER_CROP / 1e3 / FP32_MFMA_PEAK_TF, 4),
            "pose_check": {"mean_rel_translation_err": round(pose_err / max(n_pose, 1), 5), "poses": int(n_pose), "inlier_edges": int(n_inl),
                           "network_keypoints_read_back": int(n_net_kp)},
        }
    printed = threading.Lock()

    def emit(extra=None):
        if not printed.acquire(blocking=False):
            return
        if line is not None:
            if extra:
                line.update(extra)
            print(json.dumps(line), flush=True)

    def give_up():
        emit({"legs_timed_out": "a leg after the timed region (roofline / global_ba / latency / cpu_baseline) did not finish in %d s" % args.legs_timeout})
        os._exit(0 if line is not None else 3)
    dog = threading.Timer(args.legs_timeout, give_up)
    dog.daemon = True
    dog.start()
    if not args.no_global_ba_leg and args.only == "all":
        try:                                    # BASELINE configs[4]'s exchange step, on every rank; never part of `value`
            gba = global_ba_leg(world, 16)
        except Exception as e:                  # reported, not fatal: the frame-path line must survive
            gba = {"error": repr(e)[:300]}
        if line is not None:
            line["global_ba"] = gba
    if rank == 0:
        line["roofline"] = conv_roofline(L * F)      # the launch shape of the timed region
        if world == 1 and args.only == "all":
            if not args.no_latency_leg and F > 1:
                line["config"]["latency_mode_fps"] = round(latency_leg(L, pool, not args.no_graph), 2)
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(pool, L)
    dog.cancel()
    emit()
    faulthandler.cancel_dump_traceback_later()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

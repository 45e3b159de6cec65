"""Host-side legs of bench.py: the CPU baseline (the oracle on this box's cores) and the power / clock sampler of the timed region."""
import os
import time

import numpy as np


def cpu_baseline(pool, L):
    """The oracle (CPU restatement) timed on this box's host cores on a bounded sample of the same workload: the CNN of one 8-crop
    frame per thread count of a sweep (the best is reported), PnP + LM of the pool's frames on one thread (the reference's geometry
    is single-threaded, lib/object_slam.py:440-442)."""
    import torch
    from oracle import cnn_oracle as O
    from oracle import geometry as G
    from suo_slam_amd import geometry as geo
    from suo_slam_amd import synthetic as S
    from suo_slam_amd import weights
    cores = len(os.sched_getaffinity(0))
    quota = cpu_quota()
    sd = weights.make_random_state_dict(0, 8.0)
    Pw = O.to_torch(sd)
    sweep, frames_per_point = {}, {}
    t_start = time.perf_counter()
    # thread counts up to every core the process may run on (8, 16, 32, 64, 128, all); per point: one untimed frame (the intra-op pool's start-up at
    # that size), then up to 3 timed frames, the median reported.  torch's intra-op pool is OpenMP-free (its own work-stealing pool): thread placement
    # is the kernel's, memory is first-touch -- no OMP_PLACES / interleave policy is set, and none would be honoured.  Bounded at ~30 s in all.
    for n in sorted({min(c, cores) for c in (8, 16, 32, 64, 128, cores)}):
        torch.set_num_threads(n)
        fr = pool[0]
        O.pkpnet_forward(fr["image"], fr["boxes"], None, sd, Pw)                      # first call with a thread count: pool start-up
        ts = []
        for k in range(3):
            fk = pool[(1 + k) % len(pool)]
            t0 = time.perf_counter()
            O.pkpnet_forward(fk["image"], fk["boxes"], None, sd, Pw)
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > 30.0:
                break
        sweep[n] = float(np.median(ts))
        frames_per_point[n] = len(ts)
        if time.perf_counter() - t_start > 30.0:
            break
    n_best = min(sweep, key=sweep.get)
    t_cnn = sweep[n_best]
    geo_in = []
    for fr in pool[:16]:
        xs = [fr["model_kps"][o][fr["model_kps_masks"][o]].astype(np.float64) for o in range(L)]
        ys = [geo.normalize_uv(fr["uv"][o][fr["model_kps_masks"][o]].astype(np.float64), fr["K_bbox"][o].astype(np.float32).astype(np.float64)) for o in range(L)]
        geo_in.append((xs, ys, fr))
    t0 = time.perf_counter()
    n_geo = 0
    for rep in range(3):
        for xs, ys, fr in geo_in:
            init = [G.pnp(xs[o], ys[o], 1e-3, seed=o)[0][:3] for o in range(L)]
            B = S.frame_to_ba_problem(fr, np.tile(np.eye(4)[None], (L, 1, 1)))
            G.optimize(B["cam_T"], B["cam_fixed"], np.array(init), B["obj_fixed"], B["edge_cam"], B["edge_obj"], B["edge_camk"], B["edge_p"],
                       B["edge_uv"], B["edge_info"], B["edge_inlier"])
            n_geo += 1
    t_geo = (time.perf_counter() - t0) / n_geo
    # the CPU figures beside the `global_ba` and `slam` legs: the same 32-camera x 16-object pose graph through the dense C oracle (one thread, as the
    # reference's g2o call), and a SLAM view's tracking as the oracle would do it -- two network passes of the frame + the frame's geometry
    from suo_slam_amd import synthetic as S2
    Pg = S2.make_pose_graph(np.random.default_rng(5), 32, 16)
    keys = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")
    t0 = time.perf_counter()
    G.optimize(*[Pg[k].copy() for k in keys])
    t_gba = time.perf_counter() - t0
    return {"value": round(1.0 / (t_cnn + t_geo), 4), "unit": "frames/s", "cores": n_best, "kind": "port",
            "cnn_ms_per_frame_by_threads": {str(k): round(1e3 * v, 1) for k, v in sweep.items()}, "frames_timed_per_thread_count": {str(k): v for k, v in frames_per_point.items()},
            "host_cores_available": cores, "cpu_quota_cores": quota,
            "thread_placement": "torch intra-op pool (not OpenMP): no OMP_PLACES / NUMA interleave policy applies; kernel placement, first-touch memory",
            "global_ba_32x16_ms": round(1e3 * t_gba, 1), "slam_tracking_ms_per_view": round(1e3 * (2 * t_cnn + t_geo), 1),
            "slam_tracking_basis": "2 network passes of an 8-crop frame at the best thread count + PnP/LM of the frame (both measured above); hypotheses / re-initialisation scoring not included",
            "sample": f"up to 3 frames x {L} crops through the torch-CPU CNN oracle per thread count of the sweep (median; best: {n_best} threads, {t_cnn * 1e3:.0f} ms/frame) + "
                      f"{n_geo} frames through the C PnP/LM oracle (1 thread, {t_geo * 1e3:.2f} ms/frame) + one 32 x 16 global adjustment through the C LM oracle (1 thread)"}


def cpu_quota():
    """CPUs the cgroup actually grants (the GPU boxes show 256 and grant 16), or None."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        return None


def sample_power(out, stop=None, bdf=None):
    """Package power / shader clock of GPU 0 while the caller's timed region runs, into `out`: amdgpu's hwmon files every 50 ms until `stop` is set
    (mean / max over the samples), else ONE rocm-smi reading (a child process; it answers ~0.3 s in).  Silent when neither is there."""
    import glob
    import re
    import subprocess
    try:
        hw = [h for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")) if os.path.exists(os.path.join(h, "power1_average")) or os.path.exists(os.path.join(h, "power1_input"))]
        if bdf:                                              # the card of THIS process's GPU (a box shows every card of the node in sysfs)
            hw = [h for h in hw if os.path.basename(os.path.realpath(os.path.join(h, "..", ".."))).lower() == bdf.lower()]
        if hw and stop is not None:
            h = hw[0]
            pf = os.path.join(h, "power1_average") if os.path.exists(os.path.join(h, "power1_average")) else os.path.join(h, "power1_input")
            cap = os.path.join(h, "power1_cap")
            w, f = [], []
            while not stop.is_set():
                w.append(int(open(pf).read()) / 1e6)
                if os.path.exists(os.path.join(h, "freq1_input")):
                    f.append(int(open(os.path.join(h, "freq1_input")).read()) / 1e6)
                stop.wait(0.05)
            if w:
                out.update(package_w_mean=round(sum(w) / len(w), 1), package_w_max=round(max(w), 1), samples=len(w),
                           cap_w=int(open(cap).read()) / 1e6 if os.path.exists(cap) else None, sclk_mhz_mean=round(sum(f) / len(f)) if f else None,
                           source=f"{pf} ({bdf}) every 50 ms while the timed region ran")
            return
        t = time.perf_counter()
        r = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        m = re.search(r"GPU\[0\][^\n]*Current Socket Graphics Package Power \(W\): ([0-9.]+)", r) or re.search(r"GPU\[0\][^\n]*Average Graphics Package Power \(W\): ([0-9.]+)", r)
        c = re.search(r"GPU\[0\][^\n]*Max Graphics Package Power \(W\): ([0-9.]+)", r)
        k = re.search(r"GPU\[0\][^\n]*sclk clock level: \S+ \((\d+)Mhz\)", r)
        if m:
            out.update(package_w=float(m.group(1)), cap_w=float(c.group(1)) if c else None, sclk_mhz=int(k.group(1)) if k else None, t_done=time.perf_counter(),
                       source="rocm-smi, one reading of GPU 0 requested %.2f s into the timed region" % 0.0, t_req=t)
    except Exception:
        pass

#!/usr/bin/env python3
"""Headline benchmark of the suo_slam hot path on MI355X (contract: see the task statement).

One STEP = `--frames-per-step` (32) consecutive synthetic YCB-V-shaped frames (640x480 uint8, 8 object boxes each = 256 crops)
through the whole per-frame path of BASELINE.json configs[1] (single-view eval, no SLAM).  The frames are resident in HBM when the timed region starts (the bench
contract; `--frames-from-host` and the `frames_from_pinned_host` leg time the same region with the frames uploaded inside it -- rounds 1-4's `value`); inside it:
    host: boxes -> K_bbox (fix_K_for_bbox_ndc, float32 container) -> inv(K_bbox) terms; boxes, model keypoints H2D
    -> RoI crop + prior concat -> stacked-hourglass keypoint CNN (fp32 MFMA) -> heat-map decode -> validity masks
    -> [device-resident, csrc/frame_geom.hip] compaction of the valid keypoints -> batched P3P-RANSAC PnP -> acceptance
    -> pose graph -> uncertainty-weighted LM, rounds [10,10,40,40] -> ONE read-back of poses / inlier flags / keypoints.
The geometry consumes exactly what the network emitted (data-dependent keypoint counts and graphs): the weights are random (no
checkpoint ships) with the classifier bias raised and T-LESS-like thresholds so that the masks pass -- the keypoints are then
meaningless as measurements, which makes this the WORST case for the geometry (RANSAC runs to its 1000-iteration cap, LM works
on poorly conditioned graphs); that geometry is right on good measurements is checked after the timed region (`pose_check`:
the same chain on projected ground-truth keypoints + noise, the reference's --debug_gt_kp mode, against the ground truth and
the CPU oracle).  `value` is frames/s = steps * frames_per_step * n_gpus / elapsed; every network call processes exactly the
frames that are counted.  Frames of the single-view stream are independent (evaluate.py:345-346), which is what allows
batching them; the reference's own call shape (one frame per call) is timed separately (`latency`).

Legs after the timed region, never part of `value` and each fenced (an exception or a time-out in one costs only that leg):
`roofline` (dominant kernel + the largest 1x1 GEMM + the latency-mode launch under HIP events), `frames_from_pinned_host`, `pose_check`, `latency`,
`drop_in` (ObjectSLAM.process_view, the thing evaluate.py calls, one frame per call), `slam` (BASELINE configs[2]: a 60-view
sequence through ObjectSLAM.process_view, the reference's two timing meters), `global_ba` (BASELINE configs[4]'s exchange
step over RCCL -- on one GPU over a ONE-rank RCCL group with every collective issued), `fp32_pipe` (the same timed region with every
product on the fp32 matrix pipe, in a child process, + the dominant kernel's error against fp64 for both pipes), `cpu_baseline` (the oracle on the host cores, best of a thread sweep; rank 0 at N=1 only).

    python bench.py --gpus N --steps K --warmup W
N > 1 without a launcher: this process starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
(before anything touches the GPU), relays its output and exit code.  Under a launcher (WORLD_SIZE set) it is one rank.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_legs.common import (BBOX_THRESH, BF16_MFMA_PEAK_TF, DTYPE, DTYPE_NOTE, FP32_MFMA_PEAK_TF, GFLOP_PER_CROP, GFLOP_SKIPPED_PER_CROP, HBM_PEAK_GBPS,  # noqa: E402,F401
                               KP_VAR_THRESH, N_OBJ, _timed, committed_pmc, committed_traffic, confident_state_dict, dominant_kernel_name, dominant_kernel_traffic,
                               make_pool, matrix_pipe, pack_conv, pack_gemm, wino_bf16x3_enabled, winograd_saved_gflop_per_crop)
from bench_legs.host import cpu_baseline, cpu_quota, sample_power  # noqa: E402,F401
from bench_legs.path import drop_in_leg, frames_from_host_leg, global_ba_leg, latency_leg, pose_check_leg, slam_leg, tless_leg  # noqa: E402,F401
from bench_legs.pipeline import FramePipeline  # noqa: E402,F401
from bench_legs.roofline import bf16x3_leg, conv_roofline, fp32_pipe_leg, gemm_roofline, latency_roofline  # noqa: E402,F401


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="steps in the timed region (one step = --frames-per-step frames)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--objects", type=int, default=N_OBJ)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--pool", type=int, default=0, help="number of distinct synthetic frames cycled through (0 = 2 steps' worth)")
    ap.add_argument("--depth", type=int, default=4, help="steps in flight (independent network instances / streams / geometry contexts; "
                                                          "measured 773 / 786 / 795 frames/s at 2 / 3 / 4: 5.4 GB of workspace each)")
    ap.add_argument("--frames-per-step", "--frames-per-forward", dest="frames_per_step", type=int, default=32,
                    help="frames of the stream batched into one network call = one step (--objects crops each)")
    ap.add_argument("--only", choices=["all", "cnn"], default="all", help="diagnostic: network half of the step only")
    ap.add_argument("--frames-from-host", action="store_true", help="frames in pinned host memory, uploaded inside the timed region (rounds 1-4's `value`); default: resident in HBM")
    ap.add_argument("--no-legs", action="store_true", help="timed region only (profiling runs)")
    ap.add_argument("--no-latency-leg", action="store_true")
    ap.add_argument("--no-global-ba-leg", action="store_true")
    ap.add_argument("--no-slam-leg", action="store_true")
    ap.add_argument("--no-tless-leg", action="store_true")
    ap.add_argument("--legs-timeout", type=int, default=600, help="seconds the legs after the timed region may take before the line is printed without the rest")
    ap.add_argument("--no-power-sample", action="store_true", help="do not run rocm-smi beside the timed region")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous only: print the number of ranks seen and leave (no GPU work)")
    return ap.parse_args()


# ---- N > 1 without a launcher: become the launcher's parent (nothing below this line has touched the GPU yet) ------------------


# ---- N > 1 without a launcher: become the launcher's parent (nothing below this line has touched the GPU yet) ------------------
def spawn_ranks(args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // args.gpus)))
    return subprocess.call(cmd, env=env)


_LINE_FD = None


def print_line(text):
    """The contract's one line: to the real stdout (see main)."""
    data = (text + "\n").encode()
    if _LINE_FD is None:
        sys.stdout.write(text + "\n")
        sys.stdout.flush()
    else:
        os.write(_LINE_FD, data)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))                       # (no HIP call has happened in this process)
    # ONE JSON line on stdout, whatever the libraries below print there (RCCL writes its version banner to stdout when a communicator is
    # built): file descriptor 1 is pointed at stderr for the whole run and the line goes to the saved descriptor.
    global _LINE_FD
    sys.stdout.flush()
    _LINE_FD = os.dup(1)
    os.dup2(2, 1)
    import faulthandler
    faulthandler.dump_traceback_later(1700, exit=True)        # a hung run leaves with every thread's stack instead of holding the box
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    # torch's CPU thread pool is not part of the product path; left at its default (one thread per VISIBLE core) it spins through the
    # cgroup's CPU quota on these boxes (256 cores visible, 16 granted) whenever a CPU op wakes it, and host-side waits then stall for
    # tens of milliseconds.  The cpu_baseline leg sets its own thread counts.
    torch.set_num_threads(min(8, len(os.sched_getaffinity(0))))
    # (SUO_LOCAL_DEVICE / SUO_DIST_BACKEND: rehearsal of the multi-process flow on a box with fewer GPUs than ranks --
    #  e.g. two ranks sharing GPU 0 over gloo; RCCL itself refuses duplicate devices.  Not used by the driver.)
    local = int(os.environ.get("SUO_LOCAL_DEVICE", local))
    backend = os.environ.get("SUO_DIST_BACKEND", "nccl")
    on_gpu = not (args.dry_run and backend != "nccl")
    if on_gpu:
        torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    # every rank counts itself: what the line reports as n_ranks_seen is measured, not echoed from the command line
    seen = torch.ones(1, dtype=torch.float64, device="cuda" if (backend == "nccl" and on_gpu) else "cpu")
    if world > 1:
        dist.all_reduce(seen)
    n_ranks_seen = int(seen.item())
    rccl_backend = (dist.get_backend() if world > 1 else None)
    if world == 1 and on_gpu and backend == "nccl" and not args.no_legs and not args.no_global_ba_leg and not args.dry_run:
        # one GPU: a ONE-rank RCCL group, so that the global_ba leg issues the collectives of the multi-GPU schedule for real
        try:
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port1 = sk.getsockname()[1]
            sk.close()
            import datetime
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port1}", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120),
                                    device_id=torch.device("cuda", local))
            one = torch.ones(1, dtype=torch.float64, device="cuda")
            dist.all_reduce(one)
            rccl_backend = dist.get_backend() if int(one.item()) == 1 else None
        except Exception as e:                       # the frame path does not need it: reported, not fatal
            rccl_backend = "unavailable: " + repr(e)[:120]
    if args.dry_run:
        if rank == 0:
            print_line(json.dumps({"dry_run": True, "n_gpus": world, "n_ranks_seen": n_ranks_seen, "rccl_backend": rccl_backend}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    L, F = args.objects, args.frames_per_step
    # frames shard embarrassingly: rank r processes its own stream (weak scaling: K steps = K*F frames per GPU)
    n_pool = args.pool if args.pool > 0 else 2 * F
    pool = make_pool(np.random.default_rng(1000 + rank), n_pool, L)
    pipe = FramePipeline(L, pool, F, use_graph=not args.no_graph, only=args.only, depth=args.depth, resident=not args.frames_from_host)

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
    power = {}
    if rank == 0 and not args.no_power_sample:
        # package power / shader clock WHILE the timed region runs (rocm-smi in a child process started now; it answers ~0.3 s in): the call runs
        # at the package power cap (profiles/r04_power_trace.txt), which is what bounds its matrix-pipe utilisation
        import threading as _th
        power_stop = _th.Event()
        bdf = None
        try:
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        power_thread = _th.Thread(target=sample_power, args=(power, power_stop, bdf), daemon=True)
        power_thread.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        pipe.step(args.warmup + i)
    pipe.drain(args.warmup + args.steps)        # every timed step's read-back is fetched inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    if rank == 0 and not args.no_power_sample:
        power_stop.set()
        power_thread.join(timeout=2.0)
        if "t_done" in power:                                # (the rocm-smi route: keep the reading only if it came back inside the region)
            inside = power.pop("t_done") <= t0 + dt
            power.pop("t_req", None)
            if not inside:
                power.clear()
    assert pipe.n_frames == args.steps * F and pipe.n_crops == args.steps * F * L
    # max-over-ranks time + the only collective of the frame path: metric accumulators (RCCL all-reduce over xGMI)
    from suo_slam_amd import sharding
    dt, (n_kp, n_pose, n_inl, n_trials) = sharding.reduce_metrics(dt, [pipe.n_kp, pipe.n_pose, pipe.n_inl, pipe.n_trials],
                                                                   device="cuda" if backend == "nccl" else "cpu")
    # ---- the line: the timed region first; then the legs, each fenced.  Legs write into `extra` under a lock; the watchdog prints
    # a snapshot of what exists and leaves if they do not finish in time: they are reported beside `value`, they never cost it.
    import threading
    base = None
    if rank == 0:
        frames = world * args.steps * F
        fps = frames / dt
        exec_gflop = GFLOP_PER_CROP - GFLOP_SKIPPED_PER_CROP - winograd_saved_gflop_per_crop(L * F)
        base = {
            "metric": "frames/sec (obj-crops/sec) YCB-V 640x480 8-obj; ADD(-S) vs ref",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[matrix_pipe()], "dtype_note": DTYPE_NOTE[matrix_pipe()], "matrix_pipe": matrix_pipe(),
            # the PCIe-inclusive rate of the same region (frames in pinned host memory, uploaded inside it: the reference's tracking meter includes that copy,
            # lib/object_slam.py:1096-1098): measured by the frames_from_pinned_host leg below, which fills this in; `value` itself follows the bench contract
            "value_pcie_inclusive": round(fps, 3) if args.frames_from_host else None,
            "data": "synthetic", "n_ranks_seen": n_ranks_seen, "rccl_backend": rccl_backend,
            "config": {"workload": "YCB-V single-view eval (BASELINE configs[1]): 640x480 frame, %d objects -> RoI crop, hourglass keypoint "
                                   "CNN (%s), decode, masks, device-resident compaction -> batched PnP -> acceptance -> LM rounds [10,10,40,40], "
                                   "one read-back" % (L, DTYPE[matrix_pipe()]),
                       "step": "frames_per_step consecutive frames (resident in HBM%s): host K_bbox terms + H2D of the boxes / model keypoints + one network "
                               "call + the geometry of those frames on the network's own output" % (" -- NO: --frames-from-host, uploaded inside the step" if args.frames_from_host else ""),
                       "frames_per_step": F, "objects_per_frame": L, "crops_per_step": L * F, "crops_per_s": round(fps * L, 2),
                       "frames_timed": frames, "timed_region_s": round(dt, 4), "steps_in_flight": args.depth,
                       "frames": "pinned host memory, H2D inside the timed region" if args.frames_from_host else "resident in HBM when the timed region starts (bench contract); the PCIe-inclusive rate is the frames_from_pinned_host leg",
                       "inside_timed_region": "fix_K_for_bbox_ndc + inv(K_bbox) per crop (host), boxes / model "
                                              "keypoints H2D, network, masks, compaction, PnP, acceptance, graph build, LM, read-back",
                       "weights": "seeded random, classifier bias + 4, thresholds bbox %.1f / var %.1f so the masks pass: the geometry runs on "
                                  "whatever the network emitted (worst case: RANSAC at its iteration cap)" % (BBOX_THRESH, KP_VAR_THRESH),
                       "parallelism": f"frame-sharded x{world}, no data-path collective"},
            # fp32-equivalent rates of the whole network (zero-prior MACs not issued, Winograd 3x3 at 16/36; with the default bf16x3 form of the
            # Residual 3x3 + tail, part of these products runs on the bf16 pipe: see roofline for the per-kernel accounting)
            "cnn_tflops_executed": round(fps * L * exec_gflop / 1e3, 2),
            "cnn_executed_frac_of_fp32_mfma_peak": round(fps / world * L * exec_gflop / 1e3 / FP32_MFMA_PEAK_TF, 4),
            "cnn_tflops_algorithmic": round(fps * L * GFLOP_PER_CROP / 1e3, 2),           # reference-counted FLOPs per crop x crops/s
            "cnn_algorithmic_over_fp32_mfma_peak": round(fps / world * L * GFLOP_PER_CROP / 1e3 / FP32_MFMA_PEAK_TF, 4),
            # (FramePipeline.check_range raises when a step leaves fp16's range: a line that exists holds no re-issued call)
            "fp16_range_reissues_in_timed_region": 0,
            "power_in_timed_region": dict(power) if power else None,
            "geometry_in_timed_region": {"keypoints_passed_by_the_masks": int(n_kp), "poses_accepted": int(n_pose), "inlier_edges": int(n_inl),
                                         "lm_trials": int(n_trials), "crops": int(world * args.steps * F * L)},
        }
    extra, lock, printed = {}, threading.Lock(), threading.Event()

    def emit(note=None):
        if printed.is_set():
            return
        printed.set()
        if base is not None:
            with lock:
                line = dict(base)
                line.update(json.loads(json.dumps(extra, default=str)))            # a snapshot: the legs may still be writing
            if note:
                line["legs_note"] = note
            print_line(json.dumps(line))

    def give_up():
        try:
            emit("a leg after the timed region did not finish within %d s; the line holds what was measured until then" % args.legs_timeout)
        finally:
            os._exit(0 if base is not None else 3)
    dog = threading.Timer(args.legs_timeout, give_up)
    dog.daemon = True
    dog.start()

    def leg(name, fn, *a, into=None, **kw):
        """Run one leg; whatever it raises becomes its entry."""
        t0 = time.perf_counter()
        try:
            val = fn(*a, **kw)
        except Exception as e:                  # reported, not fatal: the frame-path line must survive
            val = {"error": repr(e)[:300]}
        if isinstance(val, dict):
            val["leg_seconds"] = round(time.perf_counter() - t0, 2)
        with lock:
            (extra if into is None else extra.setdefault(into, {}))[name] = val

    if rank == 0:
        # the whole call against the HBM roof: algorithmic bytes of every launch of one step's network call (a dry run of the schedule that just ran) / step time
        try:
            from bench_legs.roofline import whole_call
            nb = pipe.slots[0]["net"].schedule_bytes(L * F, F, 480, 640)
            with lock:
                extra.setdefault("roofline_all", {})["whole_call"] = whole_call(nb, base["ms_per_step"], args.depth)
        except Exception as e:
            with lock:
                extra.setdefault("roofline_all", {})["whole_call"] = {"error": repr(e)[:300]}
    if not args.no_legs and args.only == "all":
        del pipe                                  # free the timed region's slots before the legs build their own
        if not args.no_global_ba_leg:
            # BASELINE configs[4]'s exchange step, on every rank.  The ranks agree that set-up succeeded before the first
            # collective of the loop (a rank that failed alone would leave the others waiting in all_reduce)
            ok = torch.ones(1, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            try:
                from suo_slam_amd import ba_dist  # noqa: F401
            except Exception:
                ok.zero_()
            if world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() > 0:
                leg("global_ba", global_ba_leg, world, 16)
            else:
                with lock:
                    extra["global_ba"] = {"error": "a rank could not set up the distributed bundle adjustment"}
        if rank == 0:
            leg("dominant_conv", conv_roofline, L * F, into="roofline_all")
            with lock:
                if "error" not in extra["roofline_all"]["dominant_conv"]:
                    extra["roofline"] = dict(extra["roofline_all"]["dominant_conv"])       # the contract's `roofline` = the dominant kernel
                else:
                    extra["roofline"] = extra["roofline_all"]["dominant_conv"]
            leg("largest_gemm", gemm_roofline, L * F, into="roofline_all")
            leg("latency_mode_dominant_conv", latency_roofline, 8, into="roofline_all")
            leg("bf16x3_vs_f32_gemm", bf16x3_leg, L * F)
            if world == 1 and wino_bf16x3_enabled():
                leg("fp32_pipe", fp32_pipe_leg, args, L)
            if world == 1:
                if not args.frames_from_host:
                    leg("frames_from_pinned_host", frames_from_host_leg, L, pool, F, not args.no_graph, args.depth, args.steps, args.warmup, base["value"])
                    with lock:
                        if "frames_per_s" in extra.get("frames_from_pinned_host", {}):
                            base["value_pcie_inclusive"] = extra["frames_from_pinned_host"]["frames_per_s"]
                leg("pose_check", pose_check_leg, L, pool, not args.no_graph)
                if not args.no_latency_leg:
                    leg("latency", latency_leg, L, pool, not args.no_graph)
                    leg("drop_in", drop_in_leg, L, pool)
                if not args.no_slam_leg:
                    leg("slam", slam_leg)
                if not args.no_tless_leg:
                    # (after the one-frame-per-call legs: measured, a SLAM leg that follows this one runs 0.5-1 ms per view slower than one that does not --
                    #  four 256-crop networks' worth of workspace allocated and freed in between; the legs that time host latency go first)
                    leg("tless", tless_leg, not args.no_graph, args.depth)
                if not args.no_cpu_baseline:
                    leg("cpu_baseline", cpu_baseline, pool, L)
                    with lock:                      # the host figures beside the legs they belong to (VERDICT r4 missing #5)
                        cb = extra.get("cpu_baseline", {})
                        if isinstance(extra.get("global_ba"), dict) and "global_ba_32x16_ms" in cb:
                            extra["global_ba"]["cpu_oracle_ms"] = cb["global_ba_32x16_ms"]
                        if isinstance(extra.get("slam"), dict) and "slam_tracking_ms_per_view" in cb:
                            extra["slam"]["cpu_oracle_tracking_ms_per_view"] = cb["slam_tracking_ms_per_view"]
    dog.cancel()
    emit()
    faulthandler.cancel_dump_traceback_later()
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

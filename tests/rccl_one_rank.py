"""Child process of tests/test_gpu_rccl.py: a torch.distributed process group of ONE rank on backend "nccl" (= RCCL on ROCm) on the
box's MI355X, suo_slam_amd.ba_dist.optimize_distributed with SUO_FORCE_COLLECTIVES=1 -- every all-reduce of the multi-GPU schedule
really issued, in place on the device buffers the phase kernels of csrc/lm_dist.hip write -- against the same call with the world-1
short-circuit (no process group, no collective).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suo_slam_amd import ba as BA  # noqa: E402
from suo_slam_amd import ba_dist  # noqa: E402
from tests.test_gpu_geometry import _multi_view_scene  # noqa: E402

KEYS = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")


def main():
    n_cam, n_obj = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 6)
    port = int(sys.argv[3]) if len(sys.argv) > 3 else 29577
    torch.cuda.set_device(0)
    P, _ = _multi_view_scene(np.random.default_rng(7), n_cam, n_obj)
    args = [P[k] for k in KEYS]
    # (1) no process group: world 1, every _reduce_ returns early
    os.environ["SUO_FORCE_COLLECTIVES"] = "0"
    plain = ba_dist.optimize_distributed(BA.Problem(*[x.copy() for x in args]))
    # (2) one-rank RCCL group, collectives forced
    calls = {"n": 0, "cuda": 0, "numel": []}
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls["n"] += 1
        calls["cuda"] += int(t.is_cuda)
        if len(calls["numel"]) < 8:
            calls["numel"].append(int(t.numel()))
        return real(t, *a, **k)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    os.environ["SUO_FORCE_COLLECTIVES"] = "1"
    dist.all_reduce = counting
    t0 = time.perf_counter()
    forced = ba_dist.optimize_distributed(BA.Problem(*[x.copy() for x in args]))
    dt = time.perf_counter() - t0
    dist.all_reduce = real
    backend = dist.get_backend()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    out = {
        "backend": backend, "all_reduce_calls": calls["n"], "on_device": calls["cuda"], "first_numels": calls["numel"],
        "trials": int(forced.stats[2]), "iterations": int(forced.stats[1]), "ms": 1e3 * dt,
        "identical": bool(np.array_equal(plain.cam_T, forced.cam_T) and np.array_equal(plain.obj_T, forced.obj_T)
                          and np.array_equal(plain.inlier, forced.inlier) and np.array_equal(plain.stats, forced.stats)
                          and np.array_equal(plain.chi2, forced.chi2)),
    }
    print("RCCL_ONE_RANK " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

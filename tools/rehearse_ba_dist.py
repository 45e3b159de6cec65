"""Rehearsal of the multi-GPU global bundle adjustment on a box with ONE GPU: every rank of a torch.distributed.run launch
uses GPU 0 and the exchange runs over gloo (RCCL refuses duplicate devices), so the HIP phase kernels of csrc/lm_dist.hip,
the host LM schedule of suo_slam_amd/ba_dist.py and the collectives are exercised together; rank 0 compares with the
single-kernel result.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 \
        tools/rehearse_ba_dist.py [n_cam] [n_obj]
"""
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

n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_obj = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
P, _ = _multi_view_scene(np.random.default_rng(7), n_cam, n_obj)
args = [P[k] for k in ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")]
full = BA.Problem(*[x.copy() for x in args])
ba_dist.optimize_distributed(BA.Problem(*[x.copy() for x in args]))          # warm-up (allocations, first launches)
dt = 1e9
for rep in range(3):
    full = BA.Problem(*[x.copy() for x in args])
    dist.barrier()
    t0 = time.perf_counter()
    ba_dist.optimize_distributed(full)
    dt = min(dt, time.perf_counter() - t0)
if rank == 0:
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        single = BA.optimize(*[x.copy() for x in args])
        ts.append(time.perf_counter() - t0)
    print(f"single cooperative kernel: {1e3 * min(ts):.1f} ms")
    dT = max(np.abs(full.cam_T.reshape(-1, 3, 4) - single[0]).max(), np.abs(full.obj_T.reshape(-1, 3, 4) - single[1]).max())
    same_inl = bool(np.array_equal(full.inlier, single[2]))
    print(f"{world} ranks, {n_cam} cams x {n_obj} objs, {len(P['edge_cam'])} edges: {1e3 * dt:.1f} ms, stats {list(full.stats)}; "
          f"vs single kernel: max |dT| {dT:.2e}, inlier flags equal: {same_inl}, single stats {list(single[4])}")
    assert same_inl and dT < 1e-5
dist.barrier()
dist.destroy_process_group()

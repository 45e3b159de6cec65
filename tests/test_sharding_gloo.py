"""N>1 path on CPU: two gloo ranks shard a frame stream, run the per-frame geometry (through the ORACLE here,
since there is no GPU) and reduce the accumulators exactly like bench.py does with RCCL."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from suo_slam_amd import sharding


def _worker(rank, world, port, n_frames, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import geometry as G
    from suo_slam_amd import geometry as geo
    from suo_slam_amd import synthetic as S
    mine = sharding.shard_frames(n_frames, rank, world)
    n_inl, n_pose = 0, 0
    for f in mine:
        fr = S.make_frame(np.random.default_rng(f), 2, noise=0.002, with_image=False)     # frame content depends on f only
        for o in range(2):
            m = fr["model_kps_masks"][o]
            T, best, _ = G.pnp(fr["model_kps"][o][m].astype(np.float64), geo.normalize_uv(fr["uv"][o][m].astype(np.float64), fr["K_bbox"][o]), seed=f)
            n_inl += best
            n_pose += int(not np.array_equal(T, np.eye(4)))
    tmax, sums = sharding.reduce_metrics(0.1 * (rank + 1), [n_inl, n_pose, len(mine)])
    if rank == 0:
        out.put((tmax, sums))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_frame_sharding_matches_single_process():
    n_frames = 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    tmax, sums = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference
    assert sharding.shard_frames(6, 0, 2) == [0, 2, 4] and sharding.shard_frames(6, 1, 2) == [1, 3, 5]
    from oracle import geometry as G
    from suo_slam_amd import geometry as geo
    from suo_slam_amd import synthetic as S
    n_inl = n_pose = 0
    for f in range(n_frames):
        fr = S.make_frame(np.random.default_rng(f), 2, noise=0.002, with_image=False)
        for o in range(2):
            m = fr["model_kps_masks"][o]
            T, best, _ = G.pnp(fr["model_kps"][o][m].astype(np.float64), geo.normalize_uv(fr["uv"][o][m].astype(np.float64), fr["K_bbox"][o]), seed=f)
            n_inl += best
            n_pose += int(not np.array_equal(T, np.eye(4)))
    assert abs(tmax - 0.2) < 1e-12                      # max over ranks
    assert sums == [float(n_inl), float(n_pose), float(n_frames)]


def test_reduce_metrics_without_group():
    t, s = sharding.reduce_metrics(1.5, [1, 2])
    assert t == 1.5 and s == [1.0, 2.0]

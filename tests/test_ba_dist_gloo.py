"""Multi-rank global bundle adjustment on CPU: two and four gloo ranks partition the cameras, exchange the reduced
object system per LM trial (suo_slam_amd/ba_dist.py) and must reproduce the single-process oracle
(oracle/lm_oracle.c, full dense system).  Phases run in numpy here (tests/ba_numpy_phases.py); on the GPU the
same schedule drives csrc/lm_dist.hip (tests/test_gpu_geometry.py)."""
import os

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import geometry as G
from suo_slam_amd import ba, ba_dist
from suo_slam_amd import synthetic as S


def make_scene(seed, n_cam=7, n_obj=3, noise_px=0.5):
    rng = np.random.default_rng(seed)
    k = np.array([600.0, 600.0, 320.0, 240.0])
    cam_gt = np.zeros((n_cam, 3, 4))
    for c in range(n_cam):
        ang = 0.4 * (c / max(n_cam - 1, 1) - 0.5)
        cam_gt[c, :, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        cam_gt[c, :, 3] = [-250 * (c / max(n_cam - 1, 1) - 0.5), rng.uniform(-20, 20), rng.uniform(-20, 20)]
    obj_gt = np.zeros((n_obj, 3, 4))
    pts = rng.uniform(-60, 60, (n_obj, 9, 3))
    for o in range(n_obj):
        obj_gt[o, :, :3] = S.random_rotation(rng)
        obj_gt[o, :, 3] = [rng.uniform(-200, 200), rng.uniform(-120, 120), rng.uniform(800, 1100)]
    e_cam, e_obj, e_p, e_uv = [], [], [], []
    for c in range(n_cam):
        for o in range(n_obj):
            pw = pts[o] @ obj_gt[o, :, :3].T + obj_gt[o, :, 3]
            pc = pw @ cam_gt[c, :, :3].T + cam_gt[c, :, 3]
            uv = np.c_[k[0] * pc[:, 0] / pc[:, 2] + k[2], k[1] * pc[:, 1] / pc[:, 2] + k[3]] + rng.normal(0, noise_px, (9, 2))
            for j in range(9):
                if rng.random() < 0.06:
                    uv[j] = rng.uniform(0, 480, 2)
                e_cam.append(c); e_obj.append(o); e_p.append(pts[o, j]); e_uv.append(uv[j])
    E = len(e_cam)

    def perturb(T, rot, trans):
        t = np.ascontiguousarray(T.ravel().copy())
        G.lib().orc_pose_oplus(t, np.r_[rng.normal(0, rot, 3), rng.normal(0, trans, 3)])
        return t.reshape(3, 4)
    cam_init = cam_gt.copy()
    for c in range(1, n_cam):
        cam_init[c] = perturb(cam_gt[c], 5e-4, 0.3)
    obj_init = np.stack([perturb(T, 5e-4, 0.3) for T in obj_gt])
    cam_fixed = np.zeros(n_cam, np.uint8)
    cam_fixed[0] = 1
    return dict(cam_T=cam_init, cam_fixed=cam_fixed, obj_T=obj_init, obj_fixed=np.zeros(n_obj, np.uint8),
                edge_cam=np.array(e_cam, np.int32), edge_obj=np.array(e_obj, np.int32), edge_camk=np.tile(k, (E, 1)),
                edge_p=np.array(e_p), edge_uv=np.array(e_uv), edge_info=np.tile([1 / noise_px ** 2, 0, 1 / noise_px ** 2], (E, 1)),
                edge_inlier=np.ones(E, np.uint8))


KEYS = ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.ba_numpy_phases import NumpyPhases
    P = make_scene(3)
    full = ba.Problem(*[P[k] for k in KEYS], its=(10, 10, 20, 20))
    ba_dist.optimize_distributed(full, phases_factory=NumpyPhases)
    q.put((rank, full.cam_T.copy(), full.obj_T.copy(), full.inlier.copy(), full.stats.copy()))
    dist.barrier()
    dist.destroy_process_group()


def _check_against_oracle(cam, obj, inl, stats):
    P = make_scene(3)
    ref = G.optimize(*[P[k] for k in KEYS], its=(10, 10, 20, 20))
    assert np.array_equal(inl, ref[2]) and stats[0] == ref[4][0] and stats[3] == ref[4][3]
    for a, b in zip(cam.reshape(-1, 3, 4), ref[0]):
        assert np.linalg.norm(a[:, :3] - b[:, :3]) < 1e-6 and np.linalg.norm(a[:, 3] - b[:, 3]) < 1e-5 * max(1, np.linalg.norm(b[:, 3]))
    for a, b in zip(obj.reshape(-1, 3, 4), ref[1]):
        assert np.linalg.norm(a[:, :3] - b[:, :3]) < 1e-6 and np.linalg.norm(a[:, 3] - b[:, 3]) < 1e-5 * np.linalg.norm(b[:, 3])


def test_split_problem_partitions_cameras_and_edges():
    P = make_scene(3)
    full = ba.Problem(*[P[k] for k in KEYS])
    seen = []
    for r in range(3):
        loc, cams, sel = ba_dist.split_problem(full, r, 3)
        assert cams == [c for c in range(7) if c % 3 == r] and len(loc.obj_T) == 3
        assert all(int(full.edge_cam[e]) % 3 == r for e in sel) and len(loc.edge_cam) == len(sel)
        seen += sel.tolist()
    assert sorted(seen) == list(range(len(full.edge_cam)))


def test_single_rank_schedule_matches_oracle():
    from tests.ba_numpy_phases import NumpyPhases
    P = make_scene(3)
    full = ba.Problem(*[P[k] for k in KEYS], its=(10, 10, 20, 20))
    ba_dist.optimize_distributed(full, phases_factory=NumpyPhases)
    _check_against_oracle(full.cam_T, full.obj_T, full.inlier, full.stats)


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_gloo_pose_graph_reduce_matches_oracle(world):
    """2 and 4 ranks (7 cameras: shares of 2/2/2/1 at world 4; rank 0 owns the fixed gauge camera)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank holds the identical, complete result
    for other in res[1:]:
        for a, b in zip(res[0][1:], other[1:]):
            assert np.array_equal(a, b)
    _check_against_oracle(*res[0][1:])

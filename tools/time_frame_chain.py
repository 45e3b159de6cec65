"""Where a frame's time goes at one frame per call (8 crops): python tools/time_frame_chain.py [objects]
Phases on one stream, host-timed with a synchronisation after each: H2D + network, + masks, + device geometry chain (on the network's
own output with the bench's confident weights, and on projected ground-truth keypoints)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pool = bench.make_pool(np.random.default_rng(0), 8, L)
for gt in (False, True):
    pipe = bench.FramePipeline(L, pool, 1, depth=1, gt_keypoints=gt)
    for i in range(6):
        pipe.step(i)
    pipe.drain(6)
    torch.cuda.synchronize()
    for only in ("cnn", "all"):
        pipe.only = only
        for i in range(4):
            pipe.step(i)
        pipe.drain(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        for i in range(n):
            pipe.step(i)
        pipe.drain(n)
        torch.cuda.synchronize()
        print(f"gt_keypoints={gt} only={only}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per frame, one in flight")
    S = pipe.slots[0]
    pipe.only = "all"
    pipe.step(0)
    r = pipe.retire(S)
    print("   n_kp", r["n_kp"].tolist(), "pnp iterations", r["pnp_iterations"].tolist(), "lm stats", r["lm_stats"].tolist())
    # the chain alone, re-launched on the buffers the last step left behind
    from suo_slam_amd.frame_geom import kbbox_terms
    kinv, camk = kbbox_terms(np.stack([fr for fr in pool[0]["K_bbox"]]).astype(np.float32))
    md = 0.5 * pool[0]["diameter"]
    for do_lm in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            S["fg"].launch(pipe.first, S["uv"], S["cov"], S["mask"], S["kps"], kinv, camk, md, seed=i, do_lm=do_lm, stream=S["tstream"].cuda_stream)
            S["fg"].fetch(copy=False)
        print(f"   chain alone do_lm={do_lm}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
    del pipe

"""Host time of one FramePipeline step at one frame per call (what bounds the frames in flight): python tools/time_step_host.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
torch.set_num_threads(8)
L = 8
pool = bench.make_pool(np.random.default_rng(0), 8, L)
for depth in (1, 2, 4, 8):
    pipe = bench.FramePipeline(L, pool, 1, depth=depth, only=os.environ.get("ONLY", "all"), gt_keypoints=os.environ.get("GT", "0") == "1")
    for i in range(3 * depth): pipe.step(i)
    pipe.drain(3 * depth); torch.cuda.synchronize()
    t_retire = t_launch = 0.0
    n = 200
    t0 = time.perf_counter()
    orig_retire = pipe.retire
    def retire(S):
        global t_retire
        a = time.perf_counter(); r = orig_retire(S); t_retire += time.perf_counter() - a; return r
    pipe.retire = retire
    for i in range(n): pipe.step(i)
    pipe.drain(n); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"depth {depth}: {n/dt:.1f} frames/s, {dt/n*1e3:.3f} ms per step of which retire (wait + fetch + accounting) {t_retire/n*1e3:.3f} ms, launch side {(dt-t_retire)/n*1e3:.3f} ms")
    del pipe

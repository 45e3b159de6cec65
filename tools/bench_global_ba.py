"""Time one global bundle adjustment (first camera fixed, everything else free) through suo_optimize (one C call: since round 6 the phase kernels under the
device-resident schedule, driven from C) and through suo_slam_amd/ba_dist.py on one rank (the same kernels, Python between the launches):
python tools/bench_global_ba.py [n_cam] [n_obj]   (tuning builds: SUO_LM_PHASES=0 = rounds 4-5's grid-barrier kernel, SUO_LM_GRID_WGS=0/4/8/16/32 its width)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import ba as BA  # noqa: E402
from tests.test_gpu_geometry import _multi_view_scene  # noqa: E402

n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_obj = int(sys.argv[2]) if len(sys.argv) > 2 else 8
P, _ = _multi_view_scene(np.random.default_rng(1), n_cam, n_obj)
args = [P[k] for k in ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p", "edge_uv", "edge_info", "edge_inlier")]
ts = []
for rep in range(4):
    a = [x.copy() for x in args]
    t0 = time.perf_counter()
    out = BA.optimize(*a)
    ts.append(time.perf_counter() - t0)
print(f"suo_optimize (SUO_LM_PHASES={os.environ.get('SUO_LM_PHASES', 'default')}, SUO_LM_GRID_WGS={os.environ.get('SUO_LM_GRID_WGS', 'default')}): {n_cam} cams x {n_obj} objs, "
      f"{len(P['edge_cam'])} edges: {1e3 * min(ts):.2f} ms, {1e6 * min(ts) / max(int(out[4][2]), 1):.1f} us per LM trial (rounds/its/trials/good = {[int(v) for v in out[4]]})")

from suo_slam_amd import _lib  # noqa: E402
if hasattr(_lib.lib(), "suo_debug_lg_prof"):          # -DSUO_LG_PROFILE build (tools/build_variant.sh lgprof -DSUO_LG_PROFILE)
    import ctypes
    buf = (ctypes.c_double * 16)()
    _lib.lib().suo_debug_lg_prof(buf)                 # reset
    BA.optimize(*[x.copy() for x in args])
    _lib.lib().suo_debug_lg_prof(buf)
    names = ["edge pass + chi2 reduce", "pair blocks", "diagonal gather", "push + camera inverses", "Y = Hcc^-1 Hco", "reduced system",
             "x_o write-back", "x_c + update", "edge pass + reduce3", "trial / iteration bookkeeping, reclassification",
             "reduced system -> LDS", "wave Cholesky + substitutions"]
    tot = sum(buf[:12])
    for i, nm in enumerate(names):
        print(f"    {nm:48s} {buf[i] / 1e3:8.2f} ms  {100 * buf[i] / tot:5.1f} %")

# the multi-GPU phase kernels (csrc/lm_dist.hip) under the host schedule, one rank (no exchange)
from suo_slam_amd import ba_dist  # noqa: E402
ts = []
for rep in range(4):
    prob = BA.Problem(*[x.copy() for x in args])
    t0 = time.perf_counter()
    ba_dist.optimize_distributed(prob)
    ts.append(time.perf_counter() - t0)
print(f"ba_dist.py, one rank (Python-driven): {1e3 * min(ts):.2f} ms, {1e6 * min(ts) / max(int(prob.stats[2]), 1):.1f} us per LM trial (rounds/its/trials/good = {[int(v) for v in prob.stats]})")

"""Is the one-launch Residual block slower inside the network (24.2 / 22.1 us at 32x32 / 16x16, 8 crops) than alone (19.9 / 16.8) because its WEIGHTS are cold?  The same
fp16 block kernel looped over ONE weight set (L2-resident after the first launch) and rotated over 24 different sets (0.85 MB each: what a network call streams through).
python tools/bench_res_block_cold.py [crops]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from suo_slam_amd import _lib
from tests import hipops as ops
from tests.test_gpu_res_block import _block_weights
lib = _lib.lib()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NSET = 24
rng = np.random.default_rng(1)
sets = []
for i in range(NSET):
    B = _block_weights(rng)
    h1, h2, h3 = np.empty(2 * 128 * 256, np.uint16), np.empty(2 * 128 * 128 * 9, np.uint16), np.empty(2 * 256 * 128, np.uint16)
    o1, o2, o3 = np.empty(128, np.float32), np.empty(128, np.float32), np.empty(256, np.float32)
    _lib.check(lib.suo_pack_res_block_f16x2(B["w1"].ctypes.data, B["w2"].ctypes.data, None, B["w3"].ctypes.data, h1.ctypes.data, h2.ctypes.data, h3.ctypes.data, o1.ctypes.data, o2.ctypes.data, o3.ctypes.data))
    sets.append(([torch.from_numpy(t.view(np.int16)).cuda() for t in (h1, h2, h3)], [ops.dev(t) for t in (o1, o2, o3)], [ops.dev(t) for t in (B["pro"][0], B["pro"][1], B["b1"], B["b2"], B["b3"])]))
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream(); s = C.c_void_p(st.cuda_stream)
P = ops.P
for H in (32, 16, 8):
    x = torch.rand((L, H, H, 256), device="cuda") - 0.5
    out = torch.empty_like(x)
    k = [0]
    def run(rotate):
        dh, do, d = sets[k[0] % NSET if rotate else 0]
        k[0] += 1
        _lib.check(lib.suo_res_block_f16x2(P(x), L, H, H, 0, P(d[0]), P(d[1]), P(dh[0]), P(do[0]), P(d[2]), P(dh[1]), P(do[1]), P(d[3]), P(dh[2]), P(do[2]), P(d[4]), None, P(out), P(flag), s))
    warm = bench._timed(lambda: run(False), st, 96)
    cold = bench._timed(lambda: run(True), st, 96)
    print(f"{H:3d}x{H:<3d} x {L} crops, fp16 one-launch block: one weight set {warm:6.1f} us   rotating over {NSET} sets {cold:6.1f} us")

"""One Residual block (256 -> 256) at the shapes of a one-frame call: the fused kernel of csrc/res_small.hip against the three per-layer launches
the network used before it (whatever suo_conv1x1 / suo_conv_kxk dispatch to at that size).   python tools/bench_res_block.py [crops]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from suo_slam_amd import _lib
from tests import hipops as ops
from tests.test_gpu_res_block import _block_weights
lib = _lib.lib()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(1)
B = _block_weights(rng)
p1, p2, p3 = np.empty(128 * 256, np.float32), np.empty(128 * 128 * 9, np.float32), np.empty(256 * 128, np.float32)
_lib.check(lib.suo_pack_res_block(B["w1"].ctypes.data, B["w2"].ctypes.data, None, B["w3"].ctypes.data, p1.ctypes.data, p2.ctypes.data, p3.ctypes.data))
d = [ops.dev(t) for t in (B["pro"][0], B["pro"][1], p1, B["b1"], p2, B["b2"], p3, B["b3"])]
q1, q2, q3 = np.empty(3 * 128 * 256, np.uint16), np.empty(3 * 128 * 128 * 9, np.uint16), np.empty(3 * 256 * 128, np.uint16)
_lib.check(lib.suo_pack_res_block_bf16x3(B["w1"].ctypes.data, B["w2"].ctypes.data, None, B["w3"].ctypes.data, q1.ctypes.data, q2.ctypes.data, q3.ctypes.data))
dx = [torch.from_numpy(t.view(np.int16)).cuda() for t in (q1, q2, q3)]
h1, h2, h3 = np.empty(2 * 128 * 256, np.uint16), np.empty(2 * 128 * 128 * 9, np.uint16), np.empty(2 * 256 * 128, np.uint16)
o1, o2_, o3 = np.empty(128, np.float32), np.empty(128, np.float32), np.empty(256, np.float32)
_lib.check(lib.suo_pack_res_block_f16x2(B["w1"].ctypes.data, B["w2"].ctypes.data, None, B["w3"].ctypes.data, h1.ctypes.data, h2.ctypes.data, h3.ctypes.data, o1.ctypes.data, o2_.ctypes.data, o3.ctypes.data))
dh = [torch.from_numpy(t.view(np.int16)).cuda() for t in (h1, h2, h3)]
do = [ops.dev(t) for t in (o1, o2_, o3)]
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
wp1 = ops.dev(ops.pack_gemm(B["w1"], 128, 256)); wp3 = ops.dev(ops.pack_gemm(B["w3"], 256, 128)); wp2 = ops.dev(ops.pack_conv(B["w2"], 128, 128, 32))
b1, b2, b3 = ops.dev(B["b1"]), ops.dev(B["b2"]), ops.dev(B["b3"])
st = torch.cuda.current_stream(); s = C.c_void_p(st.cuda_stream)
P = ops.P
for H in (32, 16, 8, 4):
    x = torch.rand((L, H, H, 256), device="cuda") - 0.5
    xp = torch.rand((L, 2 * H, 2 * H, 256), device="cuda") - 0.5
    up = torch.rand((L, H // 2, H // 2, 256), device="cuda") - 0.5
    out = torch.empty_like(x); m1 = torch.empty((L, H, H, 128), device="cuda"); m2 = torch.empty_like(m1); o2 = torch.empty_like(x)
    M = L * H * H
    fused = lambda: _lib.check(lib.suo_res_block(P(x), L, H, H, 0, P(d[0]), P(d[1]), P(d[2]), P(d[3]), P(d[4]), P(d[5]), P(d[6]), P(d[7]), None, P(out), s))
    fused_pu = lambda: _lib.check(lib.suo_res_block(P(xp), L, H, H, 1, P(d[0]), P(d[1]), P(d[2]), P(d[3]), P(d[4]), P(d[5]), P(d[6]), P(d[7]), P(up), P(out), s))
    ox = torch.empty_like(x)
    x3 = lambda: _lib.check(lib.suo_res_block_bf16x3(P(x), L, H, H, 0, P(d[0]), P(d[1]), P(dx[0]), P(d[3]), P(dx[1]), P(d[5]), P(dx[2]), P(d[7]), None, P(ox), s))
    x3_pu = lambda: _lib.check(lib.suo_res_block_bf16x3(P(xp), L, H, H, 1, P(d[0]), P(d[1]), P(dx[0]), P(d[3]), P(dx[1]), P(d[5]), P(dx[2]), P(d[7]), P(up), P(ox), s))
    oh = torch.empty_like(x)
    f16 = lambda: _lib.check(lib.suo_res_block_f16x2(P(x), L, H, H, 0, P(d[0]), P(d[1]), P(dh[0]), P(do[0]), P(d[3]), P(dh[1]), P(do[1]), P(d[5]), P(dh[2]), P(do[2]), P(d[7]), None, P(oh), P(flag), s))
    f16_pu = lambda: _lib.check(lib.suo_res_block_f16x2(P(xp), L, H, H, 1, P(d[0]), P(d[1]), P(dh[0]), P(do[0]), P(d[3]), P(dh[1]), P(do[1]), P(d[5]), P(dh[2]), P(do[2]), P(d[7]), P(up), P(oh), P(flag), s))
    def sep():
        _lib.check(lib.suo_conv1x1(P(x), 256, 256, P(d[0]), P(d[1]), None, 0, 0, P(wp1), P(b1), None, 0, P(m1), 128, M, 128, 128, 1, 0, s))
        _lib.check(lib.suo_conv_kxk(3, P(m1), L, H, H, 128, P(wp2), P(b2), P(m2), 128, 1, s))
        _lib.check(lib.suo_conv1x1(P(m2), 128, 128, None, None, None, 0, 0, P(wp3), P(b3), P(x), 256, P(o2), 256, M, 256, 256, 0, 0, s))
    fused(); sep(); x3(); torch.cuda.synchronize()
    flop = 2.0 * M * (256 * 128 + 128 * 128 * 9 + 128 * 256)
    dmax = float((out - ox).abs().max())
    tf, tp, ts = bench._timed(fused, st, 50), bench._timed(fused_pu, st, 50), bench._timed(sep, st, 50)
    tx, txp = bench._timed(x3, st, 50), bench._timed(x3_pu, st, 50)
    th, thp = bench._timed(f16, st, 50), bench._timed(f16_pu, st, 50)
    print(f"{H:3d}x{H:<3d} x {L} crops: f16x2 {th:6.1f} us (+ pool-in + up {thp:6.1f})   bf16x3 {tx:6.1f} us ({flop / tx * 1e-6:6.1f} TFLOP/s; + pool-in + up {txp:6.1f})   fp32 fused {tf:6.1f} us (+ pool-in + up {tp:6.1f})   "
          f"three launches {ts:6.1f} us   max |bf16x3 - fp32 fused| {dmax:.2e}")

"""HIP-event time of the fused RoIAlign + stem launch (csrc/stem_x3.hip) against roi_align_concat<*,4> + the fp32-pipe stem, L crops of one frame.
   python tools/bench_stem.py [L]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import _lib  # noqa: E402
from tests import hipops as ops  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.lib()
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.integers(0, 256, (32, 480, 640, 3), dtype=np.uint8)).cuda()
c = rng.uniform([100, 100], [540, 380], (L, 2))
hw = rng.uniform(60, 120, (L, 2))
boxes = ops.dev(np.concatenate([c - hw, c + hw], 1).astype(np.float32))
idx = torch.from_numpy((np.arange(L) // 8 % 32).astype(np.int32)).cuda()
w = (rng.standard_normal((64, 44, 7, 7)) / 12).astype(np.float32)
wx = np.empty(14 * 2 * 3 * 64 * 8, np.uint16)
_lib.check(lib.suo_pack_stem_weight_bf16x3(w.ctypes.data, 44, None, wx.ctypes.data))
wxd, bd = torch.from_numpy(wx.view(np.int16)).cuda(), ops.dev(np.zeros(64, np.float32))
out = torch.empty((L, 128, 128, 64), device="cuda")


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


us = timed(lambda: _lib.check(lib.suo_stem_x3(ops.P(img), 0, 480, 640, ops.P(boxes), ops.P(idx), L, ops.P(wxd), ops.P(bd), ops.P(out), ops.S())))
print(f"stem_x3 (RoIAlign + stem, bf16 pipe) L={L}: {us:.1f} us  ({L * 128 * 128 * 64 * 147 * 2 / us / 1e6:.1f} TFLOP/s of the 147-term products; output {L * 128 * 128 * 64 * 4 / us / 1e3:.0f} GB/s)")
wh, osc = np.empty(14 * 2 * 2 * 64 * 8, np.uint16), np.empty(64, np.float32)
_lib.check(lib.suo_pack_stem_weight_f16x2(w.ctypes.data, 44, None, wh.ctypes.data, osc.ctypes.data))
whd, od, flag = torch.from_numpy(wh.view(np.int16)).cuda(), ops.dev(osc), torch.zeros(1, dtype=torch.int32, device="cuda")
us2 = timed(lambda: _lib.check(lib.suo_stem_f16x2(ops.P(img), 0, 480, 640, ops.P(boxes), ops.P(idx), L, ops.P(whd), ops.P(od), ops.P(bd), ops.P(out), ops.P(flag), ops.S())))
print(f"stem f16x2 (two fp16 terms)          L={L}: {us2:.1f} us  ({L * 128 * 128 * 64 * 147 * 2 / us2 / 1e6:.1f} TFLOP/s; output {L * 128 * 128 * 64 * 4 / us2 / 1e3:.0f} GB/s)  flag {int(flag.item())}")
w1 = (rng.standard_normal((64, 64)) / 8).astype(np.float32)
w1h, o1, _ = ops.pack_gemm_f16x2(w1)
ps, pt, b1 = ops.dev(np.ones(64, np.float32)), ops.dev(np.zeros(64, np.float32)), ops.dev(np.zeros(64, np.float32))
mid = torch.empty((L, 128, 128, 64), device="cuda")
us3 = timed(lambda: _lib.check(lib.suo_stem_f16x2_next(ops.P(img), 0, 480, 640, ops.P(boxes), ops.P(idx), L, ops.P(whd), ops.P(od), ops.P(bd), ops.P(out), ops.P(ps), ops.P(pt),
                                                       ops.P(w1h), ops.P(o1), ops.P(b1), ops.P(mid), ops.P(flag), ops.S())))
a2 = out.reshape(-1, 64)
mid2 = torch.empty((L * 128 * 128, 64), device="cuda")
us4 = timed(lambda: _lib.check(lib.suo_conv1x1_f16x2_ex(ops.P(a2), 64, 64, ops.P(ps), ops.P(pt), None, 0, 0, ops.P(w1h), ops.P(o1), ops.P(b1), None, 0, ops.P(mid2), 64, L * 128 * 128, 64, 1,
                                                        ops.P(flag), ops.S())))
print(f"stem f16x2 + r1's conv1 on the tile  L={L}: {us3:.1f} us  (the stem alone {us2:.1f}; conv1 as its own fp16 GEMM launch {us4:.1f}; the network's former conv1: the fp32-pipe persistent kernel, 440 at 256 crops)")

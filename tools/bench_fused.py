"""Residual tail at the bench launch shape (128 crops, 64x64): conv3x3 alone, conv1x1 + skip alone, the fused launch."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import _lib  # noqa: E402
from tests import hipops as ops  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(0)
lib = _lib.lib()
P = ops.P
x = torch.rand((L, H, W, 128), device="cuda") - 0.5
skip = torch.rand((L, H, W, 256), device="cuda") - 0.5
w2 = (rng.standard_normal((128, 128, 3, 3)) / 34).astype(np.float32)
w3 = (rng.standard_normal((256, 128)) / 11).astype(np.float32)
wp2 = ops.dev(ops.pack_conv(w2, 128, 128, 32))
wp3 = ops.dev(ops.pack_gemm(w3, 256, 128))
b2 = torch.zeros(128, device="cuda")
b3 = torch.zeros(256, device="cuda")
mid = torch.empty((L, H, W, 128), device="cuda")
out = torch.empty((L, H, W, 256), device="cuda")
M = L * H * W
st = torch.cuda.current_stream()
s = C.c_void_p(st.cuda_stream)


def conv():
    lib.suo_conv_kxk(3, P(x), L, H, W, 128, P(wp2), P(b2), P(mid), 128, 1, s)


def gemm():
    lib.suo_conv1x1(P(mid), 128, 128, None, None, None, 0, 0, P(wp3), P(b3), P(skip), 256, P(out), 256, M, 256, 256, 0, 0, s)


def fused():
    _lib.check(lib.suo_conv3x3_conv1x1_skip(P(x), L, H, W, P(wp2), P(b2), P(wp3), P(b3), P(skip), P(out), s), "fused")


def t(f, n=30):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        f()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


packed = np.empty(16 * 128 * 128, np.float32)
lib.suo_pack_wino_weight(np.ascontiguousarray(w2).ctypes.data, 128, 128, 128, 128, packed.ctypes.data)
wq2 = ops.dev(packed)


def wino():
    lib.suo_conv3x3_wino(P(x), L, H, W, 128, P(wq2), P(b2), P(mid), 128, 1, s)


def wino_fused():
    _lib.check(lib.suo_conv3x3_wino_conv1x1_skip(P(x), L, H, W, P(wq2), P(b2), P(wp3), P(b3), P(skip), P(out), s), "wino fused")


tw, twf = t(wino), t(wino_fused)
print(f"L={L} {H}x{W}: winograd 3x3 {tw:.1f} us, + separate 1x1 = {tw:.1f} + gemm;  winograd fused tail {twf:.1f} us ({(2.0 * L * H * W * 128 * (128 * 9 + 256)) / twf / 1e6:.1f} TF algorithmic)")
tc, tg, tf = t(conv), t(gemm), t(fused)
f2, f3 = 2.0 * M * 128 * 128 * 9, 2.0 * M * 128 * 256
print(f"L={L} {H}x{W}: conv3x3 {tc:.1f} us ({f2 / tc / 1e6:.1f} TF)  conv1x1+skip {tg:.1f} us ({f3 / tg / 1e6:.1f} TF)  sum {tc + tg:.1f} us | "
      f"fused {tf:.1f} us ({(f2 + f3) / tf / 1e6:.1f} TF), tail costs {tf - tc:.1f} us = {f3 / (tf - tc) / 1e6:.1f} TF")

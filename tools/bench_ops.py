"""Micro-benchmark single conv launches through the C ABI: python tools/bench_ops.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import _lib  # noqa: E402
from tests import hipops  # noqa: E402

lib = _lib.lib()
P = hipops.P


_WARM = [False]


def timeit(fn, iters=30):
    st = torch.cuda.current_stream()
    # the first measurement of a process reads ~15 % low (clocks still ramping): spend ~50 ms of the same op first
    for _ in range(3 if _WARM[0] else 150):
        fn()
    _WARM[0] = True
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def gemm(M, K1, N, K2=0, res=False, pro=False, relu=False, rotate=8):
    """rotate: cycle through `rotate` distinct buffers so inputs are not L2-resident between launches"""
    rng = np.random.default_rng(0)
    Np = (N + 63) // 64 * 64
    w = np.zeros((Np, K1 + K2), np.float32)
    w[:N] = rng.standard_normal((N, K1 + K2)) / 16
    wp = hipops.dev(hipops.pack_gemm(w, Np, K1 + K2))
    b = torch.zeros(Np, device="cuda")
    a1 = [torch.randn((M, K1), device="cuda") for _ in range(rotate)]
    a2 = [torch.randn((M, K2), device="cuda") for _ in range(rotate)] if K2 else None
    r = [torch.randn((M, N), device="cuda") for _ in range(rotate)] if res else None
    out = [torch.empty((M, N), device="cuda") for _ in range(rotate)]
    sc = torch.ones(K1, device="cuda") if pro else None
    sh = torch.zeros(K1, device="cuda") if pro else None
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    i = [0]

    def fn():
        k = i[0] % rotate
        i[0] += 1
        lib.suo_conv1x1(P(a1[k]), K1, K1, P(sc), P(sh), P(a2[k]) if K2 else None, K2, K2, P(wp), P(b), P(r[k]) if res else None, N,
                        P(out[k]), N, M, Np, N, int(relu), 0, s)
    us = timeit(fn)
    fl = 2.0 * M * N * (K1 + K2)
    by = 4.0 * (M * K1 + M * K2 + M * N * (2 if res else 1))
    print(f"gemm M={M:6d} K={K1}+{K2} N={N:3d} res={int(res)} pro={int(pro)}: {us:8.2f} us  {fl/us/1e6:7.1f} TF  {by/us/1e6:6.2f} TB/s")


def conv3(L, H, C_, N, rotate=8):
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((N, C_, 3, 3)) / 30).astype(np.float32)
    wp = hipops.dev(hipops.pack_conv(w, N, C_, 32))
    b = torch.zeros(N, device="cuda")
    x = [torch.randn((L, H, H, C_), device="cuda") for _ in range(rotate)]
    out = [torch.empty((L, H, H, N), device="cuda") for _ in range(rotate)]
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    i = [0]

    def fn():
        k = i[0] % rotate
        i[0] += 1
        lib.suo_conv_kxk(3, P(x[k]), L, H, H, C_, P(wp), P(b), P(out[k]), N, 1, s)
    us = timeit(fn)
    fl = 2.0 * L * H * H * N * C_ * 9
    print(f"conv3 L={L} {H}x{H} C={C_} N={N}: {us:8.2f} us  {fl/us/1e6:7.1f} TF")


if __name__ == "__main__":
    L = 8
    for H in (64, 32, 16, 8, 4):
        conv3(L, H, 128, 128)
    conv3(L, 128, 64, 64)
    conv3(L, 64, 64, 64)
    for H in (64, 32, 16, 8, 4):
        M = L * H * H
        gemm(M, 256, 128, pro=True, relu=True)          # conv1
        gemm(M, 128, 256, res=True)                     # conv3 + identity skip
    gemm(L * 4096, 128, 256, K2=128)                    # r5: conv3 + conv4
    gemm(L * 4096, 256, 256, relu=True)                 # lin_
    gemm(L * 4096, 256, 256, K2=64, res=True)           # re-injection
    gemm(L * 128 * 128, 64, 64, pro=True, relu=True)    # r1.conv1
    gemm(L * 128 * 128, 64, 128, K2=64)                 # r1.conv3+conv4

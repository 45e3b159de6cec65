"""HIP-event times of the network's large 1x1 launches on the bf16 pipe (csrc/gemm_bf16x3.hip) at L crops -- the shapes of net.hip's batched call:
   python tools/bench_gemm_x3_shapes.py [L]      (SUO_HIP_LIB=<variant> for tuning builds)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suo_slam_amd import _lib  # noqa: E402

lib = _lib.lib()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
st = torch.cuda.current_stream()
s = C.c_void_p(st.cuda_stream)
rng = np.random.default_rng(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def shape(name, H, K1, N, K2=0, res=False, pro=False, relu=False, pool=False, rotate=3):
    M = L * H * H
    w = (rng.standard_normal((N, K1 + K2)) / 16).astype(np.float32)
    w3 = np.empty(3 * N * (K1 + K2), np.uint16)
    _lib.check(lib.suo_pack_gemm_weight_bf16x3(w.ctypes.data, N, K1 + K2, w3.ctypes.data))
    w3d = torch.from_numpy(w3.view(np.int16)).cuda()
    b = torch.zeros(N, device="cuda")
    # SUO_BENCH_DATA=zeros / relu: operand toggling (and with it power, hence the clock) depends on the data
    mk = {"zeros": lambda m, k: torch.zeros((m, k), device="cuda"), "relu": lambda m, k: torch.relu(torch.randn((m, k), device="cuda")),
          "randn": lambda m, k: torch.randn((m, k), device="cuda")}[os.environ.get("SUO_BENCH_DATA", "randn")]
    a1 = [mk(M, K1) for _ in range(rotate)]
    a2 = [mk(M, K2) for _ in range(rotate)] if K2 else None
    r = [torch.randn((M, N), device="cuda") for _ in range(rotate)] if res else None
    out = None if pool else [torch.empty((M, N), device="cuda") for _ in range(rotate)]
    po = [torch.empty((M // 4, N), device="cuda") for _ in range(rotate)] if pool else None
    sc = torch.ones(K1, device="cuda") if pro else None
    sh = torch.zeros(K1, device="cuda") if pro else None
    i = [0]

    def fn():
        k = i[0] % rotate
        i[0] += 1
        if pool:
            _lib.check(lib.suo_conv1x1_bf16x3_pool(P(a1[k]), K1, K1, P(sc), P(sh), P(a2[k]) if K2 else None, K2, K2, P(w3d), P(b), P(r[k]) if res else None, N,
                                                   None, N, M, N, int(relu), H, H, P(po[k]), s), name)
        else:
            _lib.check(lib.suo_conv1x1_bf16x3_ex(P(a1[k]), K1, K1, P(sc), P(sh), P(a2[k]) if K2 else None, K2, K2, P(w3d), P(b), P(r[k]) if res else None, N,
                                                 P(out[k]), N, M, N, int(relu), s), name)
    us = timed(fn)
    by = 4.0 * (M * K1 + M * K2 + M * N * ((0.25 if pool else 1) + (1 if res else 0)))
    print(f"{name:44s} M={M:8d} K={K1}+{K2:<3d} N={N:3d}: {us:8.1f} us  {2.0 * M * N * (K1 + K2) / us / 1e6:6.1f} TFLOP/s  {by / us / 1e6:5.2f} TB/s")
    del a1, a2, r, out, po
    torch.cuda.empty_cache()


print({k: v for k, v in os.environ.items() if k.startswith("SUO_")}, "L =", L)
shape("Residual.conv1 @64 (x9 per call)", 64, 256, 128, pro=True, relu=True)
shape("lin_ @64 (x2)", 64, 256, 256, relu=True)
shape("re-injection @64 + residual + pool (x1)", 64, 256, 256, res=True, pool=True)
shape("r5 conv3 + conv4 @64 + pool (x1)", 64, 128, 256, K2=128, pool=True)
shape("r4 conv3 + residual @64 (x1)", 64, 64, 128, res=True)
shape("r1 conv3 + conv4 @128 + pool (x1)", 128, 64, 128, K2=64, pool=True, rotate=2)
shape("Residual.conv1 @32 (x12)", 32, 256, 128, pro=True, relu=True)

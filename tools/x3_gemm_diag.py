"""Diagnostic: error of the bf16x3 GEMM vs fp64 by shape, in units of 2^-24 sum|x||w| (tools only)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import _lib
from tests import hipops as ops
lib = _lib.lib()
U = 2.0 ** -24
for (M, K, N, relu) in [(4096, 256, 256, 1), (4096, 256, 256, 0), (4096, 256, 128, 0), (8192, 256, 128, 0), (8192, 256, 256, 0)]:
    rng = np.random.default_rng(M + K + N)
    w = (rng.standard_normal((N, K)) / 16.0).astype(np.float32)
    w3 = np.empty(3 * N * K, np.uint16)
    _lib.check(lib.suo_pack_gemm_weight_bf16x3(w.ctypes.data, N, K, w3.ctypes.data))
    w3d = torch.from_numpy(w3.view(np.int16)).cuda()
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    a1 = rng.standard_normal((M, K)).astype(np.float32)
    a1d, bd = ops.dev(a1), ops.dev(b)
    out = torch.full((M, N), -5.0, device="cuda")
    _lib.check(lib.suo_conv1x1_bf16x3_ex(ops.P(a1d), K, K, None, None, None, 0, 0, ops.P(w3d), ops.P(bd), None, N, ops.P(out), N, M, N, relu, ops.S()))
    torch.cuda.synchronize()
    ref = a1.astype(np.float64) @ w.astype(np.float64).T + b
    S = np.abs(a1).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
    if relu:
        ref = np.maximum(ref, 0)
    got = out.cpu().numpy().astype(np.float64)
    e = (got - ref) / (U * S)
    i = np.unravel_index(np.abs(e).argmax(), e.shape)
    print(f"M={M} K={K} N={N} relu={relu}: max {np.abs(e).max():.3f} at {i} (abs {abs(got[i]-ref[i]):.3e}, ref {ref[i]:.4f}, S {S[i]:.3f}); by column half: "
          f"{np.abs(e[:, :128]).max():.3f} / {np.abs(e[:, 128:]).max() if N > 128 else 0:.3f}; mean {e.mean():+.4f} std {e.std():.4f}; max abs err {np.abs(got-ref).max():.3e}")
    f32 = ops.conv1x1(a1d, w, b, relu=bool(relu)).cpu().numpy().astype(np.float64)
    ef = (f32 - ref) / (U * S)
    print(f"      fp32 pipe: max {np.abs(ef).max():.3f} mean {ef.mean():+.4f} std {ef.std():.4f}; max abs err {np.abs(f32-ref).max():.3e}")

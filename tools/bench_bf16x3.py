"""The bf16x3 1x1 convolution (csrc/gemm_bf16x3.hip, what the network launches) against the fp32 MFMA kernel on the network's largest 1x1 shape
(Residual.conv1: K 256 -> N 128, BN + ReLU prologue, + ReLU): error of both against fp64, and time.  python tools/bench_bf16x3.py [crops]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from suo_slam_amd import _lib
lib = _lib.lib()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, K, N = L * 4096, 256, 128
rng = np.random.default_rng(1)
a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).cuda()
w = (rng.standard_normal((N, K)) / 16.0).astype(np.float32)
sc = rng.uniform(0.5, 1.5, K).astype(np.float32); sh = (rng.standard_normal(K) * 0.1).astype(np.float32)
b = (rng.standard_normal(N) * 0.1).astype(np.float32)
wp = torch.from_numpy(bench.pack_gemm(w, N, K)).cuda()
w3 = np.empty(3 * N * K, np.uint16)
_lib.check(lib.suo_pack_gemm_weight_bf16x3(w.ctypes.data, N, K, w3.ctypes.data))
w3d = torch.from_numpy(w3.view(np.int16)).cuda()
scd, shd, bd = torch.from_numpy(sc).cuda(), torch.from_numpy(sh).cuda(), torch.from_numpy(b).cuda()
o32 = torch.empty((M, N), device="cuda"); o3 = torch.empty((M, N), device="cuda")
st = torch.cuda.current_stream(); s = C.c_void_p(st.cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
f32 = lambda: _lib.check(lib.suo_conv1x1(P(a), K, K, P(scd), P(shd), None, 0, 0, P(wp), P(bd), None, 0, P(o32), N, M, N, N, 1, 0, s))
x3 = lambda: _lib.check(lib.suo_conv1x1_bf16x3(P(a), K, K, P(scd), P(shd), P(w3d), P(bd), P(o3), N, M, N, 1, s))
f32(); x3(); torch.cuda.synchronize()
rows = slice(0, 4096)
ref = np.maximum(np.maximum(a[rows].cpu().numpy().astype(np.float64) * sc + sh, 0) @ w.astype(np.float64).T + b, 0)
# the prologue itself is float32 in both kernels: take it as given
pre = np.maximum(a[rows].cpu().numpy() * sc + sh, 0).astype(np.float32).astype(np.float64)
ref2 = np.maximum(pre @ w.astype(np.float64).T + b, 0)
for name, o in (("fp32 MFMA (gemm_persist_kernel)", o32), ("bf16x3 (6 cross terms)", o3)):
    e = np.abs(o[rows].cpu().numpy() - ref2)
    print(f"{name:34s} max abs err {e.max():.3e}  rel to output range {e.max() / np.abs(ref2).max():.3e}  mean abs {e.mean():.3e}")
print("agreement of the two kernels: max abs diff", float((o32 - o3).abs().max()))
for name, f in (("fp32 MFMA", f32), ("bf16x3", x3)):
    us = bench._timed(f, st, 30)
    print(f"{name:10s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s (fp32-equivalent)  = {2.0 * M * N * K / us / 1e6 / 157.3:.3f} of the fp32 MFMA peak")

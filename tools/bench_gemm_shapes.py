"""Time the network's large-map 1x1 GEMM shapes through the C ABI (one process per dispatch setting, e.g.
SUO_GEMM_STAGE=0/1): python tools/bench_gemm_shapes.py [L]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_ops as bo  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
print({k: v for k, v in os.environ.items() if k.startswith("SUO_")}, "L =", L)
for H in (64, 32):
    M = L * H * H
    bo.gemm(M, 256, 128, pro=True, relu=True)      # Residual.conv1
    bo.gemm(M, 128, 256, res=True)                 # Residual.conv3 + identity skip
bo.gemm(L * 4096, 128, 256, K2=128)                # r5: conv3 + conv4
bo.gemm(L * 4096, 256, 256, relu=True)             # lin_
bo.gemm(L * 4096, 256, 256, K2=64, res=True)       # re-injection
bo.gemm(L * 128 * 128, 64, 128, K2=64)             # r1: conv3 + conv4

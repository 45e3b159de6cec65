"""Calibration: the vendor library (rocBLAS / hipBLASLt through torch.mm, true fp32) on this network's GEMM shapes."""
import torch

def t(M, K, N, it=30, rot=8):
    A = [torch.randn(M, K, device="cuda") for _ in range(rot)]
    B = torch.randn(K, N, device="cuda")
    C = [torch.empty(M, N, device="cuda") for _ in range(rot)]
    for i in range(3):
        torch.mm(A[i % rot], B, out=C[i % rot])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(it):
        torch.mm(A[i % rot], B, out=C[i % rot])
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    print(f"torch.mm M={M} K={K} N={N}: {us:8.2f} us {2.0*M*N*K/us/1e6:7.1f} TF  {4.0*(M*K+M*N)/us/1e6:5.2f} TB/s")

torch.backends.cuda.matmul.allow_tf32 = False
for M in (131072, 32768):
    t(M, 256, 128); t(M, 128, 256); t(M, 256, 256)
t(524288, 128, 128)
t(8192, 8192, 8192, it=5, rot=1)

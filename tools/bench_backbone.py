"""Quick timing of the HIP backbone (suo_net_backbone) at L crops: python tools/bench_backbone.py [L] [iters] [graph]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import _lib, weights
from suo_slam_amd.pkpnet import PkpNet

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
graph = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sd = weights.make_random_state_dict(0, 8.0)
net = PkpNet(state_dict=sd, max_crops=L)
net.set_graph(bool(graph))
print("workspace MB", net.workspace_bytes() / 2**20)
x = torch.rand((L, 256, 256, 48), device="cuda")
x[..., 44:] = 0
out = torch.empty((L, 41, 64, 64), device="cuda")
st = torch.cuda.current_stream().cuda_stream
lib = _lib.lib()
for _ in range(3):
    _lib.check(lib.suo_net_backbone(net._h, C.c_void_p(x.data_ptr()), L, C.c_void_p(out.data_ptr()), C.c_void_p(st)))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    _lib.check(lib.suo_net_backbone(net._h, None, L, None, C.c_void_p(st)))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
gf = 31.495e9 * L
print(f"L={L} graph={graph}: {dt*1e3:.3f} ms/frame  {L/dt:.1f} crops/s  {gf/dt/1e12:.1f} TFLOP/s ({gf/dt/157.3e12*100:.1f}% of fp32 MFMA peak)")

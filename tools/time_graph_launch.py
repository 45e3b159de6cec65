"""Host cost of one network call vs its GPU time at L crops: python tools/time_graph_launch.py [L] [nets]
(is one-frame-per-call mode bounded by the host's hipGraphLaunch or by the GPU?)"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import _lib, weights
from suo_slam_amd.pkpnet import PkpNet

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nets = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sd = weights.make_random_state_dict(0, 8.0)
lib = _lib.lib()
N = []
for _ in range(nets):
    net = PkpNet(state_dict=sd, max_crops=L)
    net.set_graph(True)
    ts = torch.cuda.Stream()
    x = torch.rand((L, 256, 256, 48), device="cuda")
    x[..., 44:] = 0
    out = torch.empty((L, 41, 64, 64), device="cuda")
    N.append((net, ts, x, out))
for net, ts, x, out in N:
    for _ in range(3):
        _lib.check(lib.suo_net_backbone(net._h, C.c_void_p(x.data_ptr()), L, C.c_void_p(out.data_ptr()), C.c_void_p(ts.cuda_stream)))
torch.cuda.synchronize()
iters = 300
host = 0.0
t0 = time.perf_counter()
for i in range(iters):
    net, ts, x, out = N[i % nets]
    a = time.perf_counter()
    _lib.check(lib.suo_net_backbone(net._h, None, L, None, C.c_void_p(ts.cuda_stream)))
    host += time.perf_counter() - a
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"L={L} nets={nets}: {dt*1e3:.3f} ms per call end to end, host time inside the call {host/iters*1e3:.3f} ms")
# one call at a time, synchronised: GPU latency of a single graph
t0 = time.perf_counter()
for i in range(100):
    net, ts, x, out = N[0]
    _lib.check(lib.suo_net_backbone(net._h, None, L, None, C.c_void_p(ts.cuda_stream)))
    ts.synchronize()
print(f"  single call, synchronised each time: {(time.perf_counter()-t0)/100*1e3:.3f} ms")
h = 0.0
for i in range(100):
    net, ts, x, out = N[0]
    a = time.perf_counter()
    _lib.check(lib.suo_net_backbone(net._h, None, L, None, C.c_void_p(ts.cuda_stream)))
    h += time.perf_counter() - a
    ts.synchronize()
print(f"  host time of a call into an idle GPU: {h/100*1e3:.3f} ms")

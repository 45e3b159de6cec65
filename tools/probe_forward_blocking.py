import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from suo_slam_amd import _lib
from suo_slam_amd.pkpnet import PkpNet, _ptr, _stream
lib = _lib.lib()
net = PkpNet(state_dict=bench.confident_state_dict(), max_crops=128)
B, L = 16, 128
imgs = torch.randint(0, 255, (B, 480, 640, 3), dtype=torch.uint8, device="cuda")
rng = np.random.default_rng(0)
bx = torch.tensor(np.tile(np.array([[100, 80, 300, 290]], np.float32), (L, 1))).cuda()
idx = torch.arange(L, dtype=torch.int32).cuda() // 8
uv = torch.empty((L, 41, 2), device="cuda"); cov = torch.empty((L, 41, 2, 2), device="cuda"); kpm = torch.empty((L, 41), device="cuda"); kpl = torch.empty((L, 41), device="cuda")
logits = torch.empty((L, 41, 64, 64), device="cuda")
def call(lg):
    t = time.perf_counter()
    _lib.check(lib.suo_net_forward_frames(net._h, _ptr(imgs), 0, 480, 640, _ptr(bx), _ptr(idx), L, None, _ptr(uv), _ptr(cov), _ptr(kpm), _ptr(kpl), _ptr(lg) if lg is not None else None, _stream()))
    return (time.perf_counter() - t) * 1e3
for lg in (logits, None):
    for rep in range(2):
        call(lg); torch.cuda.synchronize()
        ts = [call(lg) for _ in range(4)]
        t = time.perf_counter(); torch.cuda.synchronize(); td = (time.perf_counter() - t) * 1e3
        print("logits" if lg is not None else "no logits", "host ms per back-to-back call:", [round(x, 2) for x in ts], "drain", round(td, 1))
net.set_graph(False)
call(None); torch.cuda.synchronize()
ts = [call(None) for _ in range(4)]
torch.cuda.synchronize()
print("no graph:", [round(x, 2) for x in ts])

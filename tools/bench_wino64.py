"""The 64 -> 64 Winograd 3x3 convolution of the first two Residual blocks (128 x 128 and 64 x 64 maps): bf16x3 form against the fp32-pipe form.
python tools/bench_wino64.py [crops]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from suo_slam_amd import _lib
lib = _lib.lib()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(1)
w = (rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)
b = torch.zeros(64, device="cuda")
pk = np.empty(16 * 64 * 64, np.float32); _lib.check(lib.suo_pack_wino_weight(w.ctypes.data, 64, 64, 64, 64, pk.ctypes.data))
pk3 = np.empty(3 * 16 * 64 * 64, np.uint16); _lib.check(lib.suo_pack_wino_weight_bf16x3(w.ctypes.data, 64, 64, pk3.ctypes.data))
wq, wq3 = torch.from_numpy(pk).cuda(), torch.from_numpy(pk3.view(np.int16)).cuda()
st = torch.cuda.current_stream(); s = C.c_void_p(st.cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
for H in (128, 64):
    x = torch.rand((L, H, H, 64), device="cuda") - 0.5
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    f32 = lambda: _lib.check(lib.suo_conv3x3_wino(P(x), L, H, H, 64, P(wq), P(b), P(o1), 64, 1, s))
    x3 = lambda: _lib.check(lib.suo_conv3x3_wino_x3_n(P(x), L, H, H, 64, P(wq3), P(b), P(o2), 1, s))
    f32(); x3(); torch.cuda.synchronize()
    print(f"{H}x{H} x {L} crops: fp32 pipe {bench._timed(f32, st, 20):8.1f} us   bf16x3 {bench._timed(x3, st, 20):8.1f} us   max abs diff {float((o1 - o2).abs().max()):.2e}")

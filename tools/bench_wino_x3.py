"""The bf16x3 Winograd 3x3 convolution (csrc/conv_wino_x3.hip, the network's default) against the fp32 MFMA Winograd kernel (csrc/conv_wino.hip) on
the Residual block's shape (128 -> 128 channels, 64 x 64 maps): error of both against fp64, and time, plain and with the fused Residual tail.
python tools/bench_wino_x3.py [crops]"""
import ctypes as C, os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from suo_slam_amd import _lib
lib = _lib.lib()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = W = 64
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.standard_normal((L, H, W, 128)).astype(np.float32)).cuda()
skip = torch.from_numpy(rng.standard_normal((L, H, W, 256)).astype(np.float32)).cuda()
low = torch.from_numpy(rng.standard_normal((L, H // 2, W // 2, 256)).astype(np.float32)).cuda()
w2 = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
b2 = (rng.standard_normal(128) * 0.3).astype(np.float32)
w3 = (rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
b3 = rng.standard_normal(256).astype(np.float32)
pk = np.empty(16 * 128 * 128, np.float32); _lib.check(lib.suo_pack_wino_weight(w2.ctypes.data, 128, 128, 128, 128, pk.ctypes.data))
pk3 = np.empty(3 * 16 * 128 * 128, np.uint16); _lib.check(lib.suo_pack_wino_weight_bf16x3(w2.ctypes.data, 128, 128, pk3.ctypes.data))
wq, wq3 = torch.from_numpy(pk).cuda(), torch.from_numpy(pk3.view(np.int16)).cuda()
wp3 = torch.from_numpy(bench.pack_gemm(w3, 256, 128)).cuda()
pk3t = np.empty(3 * 256 * 128, np.uint16); _lib.check(lib.suo_pack_tail_weight_bf16x3(w3.ctypes.data, 256, 128, pk3t.ctypes.data))
wp3x = torch.from_numpy(pk3t.view(np.int16)).cuda()
b2d, b3d = torch.from_numpy(b2).cuda(), torch.from_numpy(b3).cuda()
o_f, o_x = torch.empty((L, H, W, 128), device="cuda"), torch.empty((L, H, W, 128), device="cuda")
t_f, t_x, t_xx = (torch.empty((L, H, W, 256), device="cuda") for _ in range(3))
st = torch.cuda.current_stream(); s = C.c_void_p(st.cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
plain_f = lambda: _lib.check(lib.suo_conv3x3_wino(P(x), L, H, W, 128, P(wq), P(b2d), P(o_f), 128, 1, s))
plain_x = lambda: _lib.check(lib.suo_conv3x3_wino_x3(P(x), L, H, W, P(wq3), P(b2d), P(o_x), 1, s))
tail_f = lambda: _lib.check(lib.suo_conv3x3_wino_conv1x1_skip_up(P(x), L, H, W, P(wq), P(b2d), P(wp3), P(b3d), P(skip), P(low), P(t_f), s))
tail_x = lambda: _lib.check(lib.suo_conv3x3_wino_x3_conv1x1_skip_up(P(x), L, H, W, P(wq3), P(b2d), P(wp3), 0, P(b3d), P(skip), P(low), P(t_x), s))
tail_xx = lambda: _lib.check(lib.suo_conv3x3_wino_x3_conv1x1_skip_up(P(x), L, H, W, P(wq3), P(b2d), P(wp3x), 1, P(b3d), P(skip), P(low), P(t_xx), s))
for f in (plain_f, plain_x, tail_f, tail_x, tail_xx): f()
torch.cuda.synchronize()
for l in (0, L - 1):
    xm = x[l:l + 1].permute(0, 3, 1, 2).double().cpu()
    m = F.relu(F.conv2d(xm, torch.from_numpy(w2).double(), torch.from_numpy(b2).double(), padding=1))
    ref = m.numpy()
    for name, o in (("fp32 Winograd", o_f), ("bf16x3 Winograd", o_x)):
        e = np.abs(o[l:l + 1].permute(0, 3, 1, 2).cpu().numpy() - ref)
        print(f"crop {l:3d} plain {name:16s} max abs err {e.max():.3e}  rel to output range {e.max() / np.abs(ref).max():.3e}  mean abs {e.mean():.3e}")
    ref2 = (F.conv2d(m, torch.from_numpy(w3).double()[:, :, None, None], torch.from_numpy(b3).double()) + skip[l:l + 1].permute(0, 3, 1, 2).double().cpu()
            + low[l:l + 1].permute(0, 3, 1, 2).double().cpu().repeat_interleave(2, 2).repeat_interleave(2, 3)).numpy()
    for name, o in (("fp32 Winograd", t_f), ("bf16x3 Winograd", t_x), ("bf16x3 both", t_xx)):
        e = np.abs(o[l:l + 1].permute(0, 3, 1, 2).cpu().numpy() - ref2)
        print(f"crop {l:3d} tail  {name:16s} max abs err {e.max():.3e}  rel to output range {e.max() / np.abs(ref2).max():.3e}  mean abs {e.mean():.3e}")
print("agreement of the two plain kernels: max abs diff", float((o_f - o_x).abs().max()), " tails:", float((t_f - t_x).abs().max()), float((t_f - t_xx).abs().max()))
flop = 2.0 * L * H * W * 128 * 128 * 9
for name, f, fl in (("fp32 plain", plain_f, flop), ("bf16x3 plain", plain_x, flop), ("fp32 tail+up", tail_f, flop + 2.0 * L * H * W * 128 * 256),
                    ("bf16x3 tail+up", tail_x, flop + 2.0 * L * H * W * 128 * 256), ("bf16x3 both +up", tail_xx, flop + 2.0 * L * H * W * 128 * 256)):
    us = bench._timed(f, st, 30)
    print(f"{name:15s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s (reference-counted)")

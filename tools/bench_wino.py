"""Winograd vs direct 3x3 (128 -> 128) at a launch shape: correctness against fp64 on one crop + timing."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import _lib  # noqa: E402
from tests import hipops as ops  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
CH = int(sys.argv[3]) if len(sys.argv) > 3 else 128          # channels in = out: 128 or 64
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((L, H, W, CH)).astype(np.float32)).cuda()
w = (rng.standard_normal((CH, CH, 3, 3)) / np.sqrt(9 * CH)).astype(np.float32)
b = rng.standard_normal(CH).astype(np.float32)
out = ops.conv3x3_wino(x, w, b, relu=True)
ref_d = ops.conv_kxk(x, w, b, relu=True)
for l in (0, L - 1):
    ref = F.relu(F.conv2d(x[l:l + 1].permute(0, 3, 1, 2).double().cpu(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1))
    got = out[l:l + 1].permute(0, 3, 1, 2).cpu().double()
    dire = ref_d[l:l + 1].permute(0, 3, 1, 2).cpu().double()
    print(f"crop {l}: winograd max err {float((got - ref).abs().max()):.3e}  direct max err {float((dire - ref).abs().max()):.3e}  (range {float(ref.abs().max()):.2f})")
lib = _lib.lib()
P = ops.P
packed = np.empty(16 * CH * CH, np.float32)
lib.suo_pack_wino_weight(w.ctypes.data, CH, CH, CH, CH, packed.ctypes.data)
wpw, wpd, bd = ops.dev(packed), ops.dev(ops.pack_conv(w, CH, CH, 32)), ops.dev(b)
o2 = torch.empty_like(out)
st = torch.cuda.current_stream()
s = C.c_void_p(st.cuda_stream)


def t(f, n=30):
    for _ in range(8):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        f()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tw = t(lambda: lib.suo_conv3x3_wino(P(x), L, H, W, CH, P(wpw), P(bd), P(o2), CH, 1, s))
td = t(lambda: lib.suo_conv_kxk(3, P(x), L, H, W, CH, P(wpd), P(bd), P(o2), CH, 1, s))
fl = 2.0 * L * H * W * CH * CH * 9
print(f"L={L} {H}x{W}: winograd {tw:.1f} us ({fl / tw / 1e6:.1f} TF algorithmic, {fl / 2.25 / tw / 1e6:.1f} TF executed)   direct {td:.1f} us ({fl / td / 1e6:.1f} TF)   speed-up {td / tw:.2f}x")

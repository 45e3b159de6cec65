"""HIP-event time of lin -> head (256 -> 256 -> 41 channels at 64x64) as ONE launch (csrc/gemm_bf16x3.hip: gemm_chain_head_kernel) against the two launches it replaces.
   python tools/bench_chain_head.py [crops]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import _lib  # noqa: E402
from tests import hipops as ops  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, hw = L * 4096, 4096
lib = _lib.lib()
rng = np.random.default_rng(0)
a = torch.randn((M, 256), device="cuda")
w1 = (rng.standard_normal((256, 256)) / 16).astype(np.float32)
w2 = np.zeros((64, 256), np.float32)
w2[:41] = (rng.standard_normal((41, 256)) / 16).astype(np.float32)
w1h, o1, _ = ops.pack_gemm_f16x2(w1)
w2h, o2, _ = ops.pack_gemm_f16x2(w2)
b1, b2 = ops.dev(np.zeros(256, np.float32)), ops.dev(np.zeros(64, np.float32))
ll = torch.empty((M, 256), device="cuda")
out = torch.empty((L, 41, hw), device="cuda")
out64 = torch.empty((M, 64), device="cuda")
flag = ops._flag()
P, S = ops.P, ops.S


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def chain():
    _lib.check(lib.suo_conv1x1_chain_head_f16x2(P(a), 256, M, P(w1h), P(o1), P(b1), P(w2h), P(o2), P(b2), P(out), 41, hw, P(flag), S()))


def lin():
    _lib.check(lib.suo_conv1x1_f16x2_ex(P(a), 256, 256, None, None, None, 0, 0, P(w1h), P(o1), P(b1), None, 0, P(ll), 256, M, 256, 1, P(flag), S()))


def head16():
    _lib.check(lib.suo_conv1x1_f16x2_ex(P(ll), 256, 256, None, None, None, 0, 0, P(w2h), P(o2), P(b2), None, 0, P(out64), 64, M, 64, 0, P(flag), S()))


t_c, t_l, t_h = timed(chain), timed(lin), timed(head16)
gb = (M * 256 * 4 + M * 41 * 4) / 1e9
print(f"{L} crops: lin + head in one launch {t_c:.1f} us ({gb / t_c * 1e6:.0f} GB/s of its {gb:.2f} GB);  lin alone {t_l:.1f} us, head (fp16 GEMM, row-major 64) alone {t_h:.1f} us;  flag {int(flag.item())}")

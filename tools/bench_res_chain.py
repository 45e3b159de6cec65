"""HIP-event time of the twelve Residual blocks of one Hourglass stack's 8x8 / 4x4 levels at L crops: ONE cooperative launch (csrc/res_chain.hip) against the
one-launch block kernel block by block, and the grid barrier's own cost (an empty chain of N barriers).
   python tools/bench_res_chain.py [L]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suo_slam_amd import _lib  # noqa: E402
from tests import hipops as ops  # noqa: E402
from tests.test_gpu_res_chain import _hourglass_tail  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = 8
lib = _lib.lib()
rng = np.random.default_rng(0)
steps, buf = _hourglass_tail(ops, rng, L, H)
flag = ops._flag()
scratch = ops.chain_scratch(L * H * H)
descs = [w.desc(x, out, L, h, h, pool, up) for (w, x, out, h, pool, up) in steps]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def blocks():
    for (w, x, out, h, pool, up) in steps:
        w.launch(x, out, L, h, h, pool, up, flag)


print(f"L={L}: twelve blocks, one-launch block kernel x 12: {timed(blocks):.1f} us")
print(f"L={L}: twelve blocks, ONE cooperative launch:        {timed(lambda: ops.res_chain(descs, scratch, flag)):.1f} us   (flag {int(flag.item())})")
for k in (1, 3, 6):
    print(f"   the first {k} block(s) of the chain: {timed(lambda: ops.res_chain(descs[:k], scratch, flag)):.1f} us")
for nb in (0, 100):
    us = timed(lambda: _lib.check(lib.suo_res_chain_probe(ops.P(scratch), scratch.numel(), ops.P(flag), 0, nb, L * H * H, ops.S())))
    print(f"   empty chain + {nb} grid barriers: {us:.1f} us")
bar = scratch[-16:].view(torch.int32).cpu().numpy()
print("   launches that fell back to the general barrier:", int(bar[3]))

"""csrc/res_chain.hip: the Residual blocks of an Hourglass's 8x8 / 4x4 levels (hg.py:37-58) as ONE cooperative launch on the CUs of one XCD -- against the same blocks
launched one by one (csrc/res_small_x3.hip, NP = 2: the same packed weights and the same two-term fp16 arithmetic, another summation order) and against fp64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from suo_slam_amd import _lib
    _lib.require_gpu()
    from tests import hipops
    return hipops


def _hourglass_tail(ops, rng, L, H):
    """The chain csrc/net.hip hands over for one stack: input x at 2H x 2H; low1 x 2 at H (the first takes the pool), the inner Hourglass (up1[0] at H, low1 x 2 /
    low2 x 2 / low3 x 2 at H / 2, up1[1] at H adding the up-sampled low branch), low3 x 2 at H.  Returns (steps, buffers): steps = (weights, x, out, h, pool, up)."""
    from tests.test_gpu_res_block import _block_weights
    t = lambda h: torch.full((L, h, h, 256), float("nan"), device="cuda")      # noqa: E731
    x = torch.from_numpy(rng.standard_normal((L, 2 * H, 2 * H, 256)).astype(np.float32)).cuda()
    Ws = [ops.ChainWeights(_block_weights(rng)) for _ in range(12)]
    Bs = [w for w in Ws]
    lo_a, lo_b, up_a, inner = t(H), t(H), t(H), t(H)
    q = [t(H // 2) for _ in range(6)]
    l3a, l3b = t(H), t(H)
    steps = [(Bs[0], x, lo_a, H, True, None), (Bs[1], lo_a, lo_b, H, False, None), (Bs[2], lo_b, up_a, H, False, None),
             (Bs[3], lo_b, q[0], H // 2, True, None), (Bs[4], q[0], q[1], H // 2, False, None), (Bs[5], q[1], q[2], H // 2, False, None),
             (Bs[6], q[2], q[3], H // 2, False, None), (Bs[7], q[3], q[4], H // 2, False, None), (Bs[8], q[4], q[5], H // 2, False, None),
             (Bs[9], up_a, inner, H, False, q[5]), (Bs[10], inner, l3a, H, False, None), (Bs[11], l3a, l3b, H, False, None)]
    return steps, dict(x=x, out=l3b, all=[lo_a, lo_b, up_a, inner, l3a, l3b] + q)


@pytest.mark.parametrize("L,H", [(8, 8), (3, 8), (1, 8), (16, 8), (5, 6)])
def test_chain_equals_the_blocks_one_by_one(ops, L, H):
    """Every tensor of the chain (all twelve blocks' outputs) against the one-launch block kernel run block by block on the same inputs: same weights, same fp16 split,
    only the order of the fp32 partial sums differs -- within 2e-6 of each tensor's range per block, and the end of the chain within 1e-5 after twelve."""
    rng = np.random.default_rng(100 * L + H)
    steps, buf = _hourglass_tail(ops, rng, L, H)
    flag = ops._flag()
    scratch = ops.chain_scratch(L * H * H)
    ops.res_chain([w.desc(x, out, L, h, h, pool, up) for (w, x, out, h, pool, up) in steps], scratch, flag, xcd=0)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    got = [t.clone() for t in buf["all"]]
    assert all(torch.isfinite(t).all() for t in got)
    # block by block, each on the CHAIN's own input tensors (so that differences do not accumulate): the chain's tensors are still in place
    for (w, x, out, h, pool, up) in steps:
        ref = torch.empty_like(out)
        w.launch(x, ref, L, h, h, pool, up, flag)
        torch.cuda.synchronize()
        scale = float(ref.abs().max())
        err = float((ref - out).abs().max()) / scale
        assert err < 2e-6, (err, h, pool, up is not None)
    assert int(flag.item()) == 0
    # and the whole chain run block by block from the start
    for (w, x, out, h, pool, up) in steps:
        w.launch(x, out, L, h, h, pool, up, flag)
    torch.cuda.synchronize()
    end = buf["out"]
    assert float((end - got[5]).abs().max()) / float(end.abs().max()) < 1e-5
    bar = scratch[-16:].view(torch.int32).cpu().numpy()
    assert bar[0] == 0 and bar[2] == 0                      # arrivals and the XCC mask are back to zero
    print("\nchain L=%d H=%d: launches that fell back to the general barrier: %d" % (L, H, bar[3]))


@pytest.mark.parametrize("L,H", [(8, 8), (8, 4), (3, 6)])
def test_chain_against_fp64(ops, L, H):
    """One block of the chain (every flavour: plain, pool-in, up-sampled addend) against fp64 on its own input: the bound the one-launch block is held to."""
    from tests.test_gpu_res_block import _fp64, _block_weights
    rng = np.random.default_rng(9 + H)
    flag = ops._flag()
    scratch = ops.chain_scratch(L * H * H)
    for pool, up in ((False, False), (True, False), (False, True), (True, True)):
        B = _block_weights(rng)
        w = ops.ChainWeights(B)
        xin = torch.from_numpy(rng.standard_normal((L, 2 * H, 2 * H, 256) if pool else (L, H, H, 256)).astype(np.float32)).cuda()
        low = torch.from_numpy(rng.standard_normal((L, H // 2, H // 2, 256)).astype(np.float32)).cuda() if up else None
        out = torch.empty((L, H, H, 256), device="cuda")
        ops.res_chain([w.desc(xin, out, L, H, H, pool, low)], scratch, flag)
        torch.cuda.synchronize()
        x = xin.view(L, H, 2, H, 2, 256).amax(dim=(2, 4)) if pool else xin
        ref = _fp64(x, B, low)
        e = np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()
        assert e < 5e-6, (pool, up, e)
    assert int(flag.item()) == 0


def test_chain_range_guard_and_repeat(ops):
    """An inner activation beyond fp16's range raises the flag; the same scratch serves launch after launch (barrier words return to their rest state), and two
    launches of the same chain give the same bits."""
    from tests.test_gpu_res_block import _block_weights
    rng = np.random.default_rng(3)
    L, H = 8, 8
    B = _block_weights(rng)
    x = torch.from_numpy(np.abs(rng.standard_normal((L, H, H, 256))).astype(np.float32)).cuda()
    scratch = ops.chain_scratch(L * H * H)
    w = ops.ChainWeights(B)
    out1, out2 = torch.empty((L, H, H, 256), device="cuda"), torch.empty((L, H, H, 256), device="cuda")
    flag = ops._flag()
    for out in (out1, out2):
        ops.res_chain([w.desc(x, out, L, H, H, False, None)] * 1, scratch, flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0 and torch.equal(out1, out2)
    Bi = dict(B, w1=np.abs(B["w1"]) * np.float32(4000.0), b1=np.abs(B["b1"]), pro=(np.abs(B["pro"][0]), np.abs(B["pro"][1])))
    wi = ops.ChainWeights(Bi)
    ops.res_chain([wi.desc(x, out1, L, H, H, False, None)], scratch, flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1

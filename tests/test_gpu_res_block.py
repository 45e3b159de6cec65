"""csrc/res_small.hip: a whole Residual block (lib/models/layers/Residual.py:20-35, 256 -> 256 channels) in ONE launch on the small
feature maps of a one-frame call -- against the per-layer fp32 kernels launched separately (bit-identical: same summation order) and
against fp64 (the per-kernel bound every convolution is held to)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from suo_slam_amd import _lib
    _lib.require_gpu()
    from tests import hipops
    return hipops


def _block_weights(rng):
    """A Residual block with its BatchNorms (eval mode, eps 1e-5) folded the way csrc/net.hip folds them: bn -> prologue scale / shift,
    bn1 / bn2 into the preceding convolution's rows (float32 products) and bias."""
    def bn(c):
        g, b = rng.uniform(0.5, 1.5, c).astype(np.float32), (rng.standard_normal(c) * 0.1).astype(np.float32)
        m, v = (rng.standard_normal(c) * 0.1).astype(np.float32), rng.uniform(0.5, 1.5, c).astype(np.float32)
        s = (g / np.sqrt(v + np.float32(1e-5))).astype(np.float32)
        return s, (b - m * s).astype(np.float32)
    s0, t0 = bn(256)
    w1 = (rng.standard_normal((128, 256)) / 16).astype(np.float32)
    c1 = (rng.standard_normal(128) * 0.1).astype(np.float32)
    s1, t1 = bn(128)
    w2 = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    c2 = (rng.standard_normal(128) * 0.1).astype(np.float32)
    s2, t2 = bn(128)
    w3 = (rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
    b3 = (rng.standard_normal(256) * 0.1).astype(np.float32)
    return dict(pro=(s0, t0), w1=(w1 * s1[:, None]).astype(np.float32), b1=(c1 * s1 + t1).astype(np.float32),
                w2=(w2 * s2[:, None, None, None]).astype(np.float32), b2=(c2 * s2 + t2).astype(np.float32), w3=w3, b3=b3)


def _separate_launches(ops, x, B, up=None):
    """The same block as three per-layer launches (+ the up-sample add): the one-accumulator fp32 kernels (csrc/conv.hip /
    gemm_persist.hip), reached by repeating the crops until a launch holds > 4096 pixels (below that the entry points dispatch to the
    split-K kernels of csrc/conv_small.hip, which sum in another order); crops are independent."""
    from suo_slam_amd import _lib
    L, H, W, _ = x.shape
    reps = -(-4097 // (L * H * W))
    xr = x.repeat(reps, 1, 1, 1).contiguous()
    Lr = L * reps
    mid1 = ops.conv1x1(xr.reshape(-1, 256), B["w1"], B["b1"], pro=B["pro"], relu=True)
    mid2 = ops.conv_kxk(mid1.reshape(Lr, H, W, 128), B["w2"], B["b2"], relu=True).contiguous()
    out = ops.conv1x1(mid2.reshape(-1, 128), B["w3"], B["b3"], res=xr.reshape(-1, 256)).reshape(Lr, H, W, 256)[:L].contiguous()
    if up is not None:
        want = torch.empty_like(out)
        _lib.check(_lib.lib().suo_upsample2_add(ops.P(out), ops.P(up), ops.P(want), L, H, W, 256, ops.S()), "suo_upsample2_add")
        torch.cuda.synchronize()
        out = want
    return out


def _fp64(x, B, up=None):
    xd = x.permute(0, 3, 1, 2).double().cpu()
    s0, t0 = (torch.from_numpy(t).double() for t in B["pro"])
    a = F.relu(xd * s0[None, :, None, None] + t0[None, :, None, None])
    m1 = F.relu(F.conv2d(a, torch.from_numpy(B["w1"]).double()[:, :, None, None], torch.from_numpy(B["b1"]).double()))
    m2 = F.relu(F.conv2d(m1, torch.from_numpy(B["w2"]).double(), torch.from_numpy(B["b2"]).double(), padding=1))
    o = F.conv2d(m2, torch.from_numpy(B["w3"]).double()[:, :, None, None], torch.from_numpy(B["b3"]).double()) + xd
    if up is not None:
        o = o + up.permute(0, 3, 1, 2).double().cpu().repeat_interleave(2, 2).repeat_interleave(2, 3)
    return o.permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize("L,H,W,up,pool", [
    (8, 32, 32, False, False), (8, 32, 32, True, False),      # one frame at 32x32: 4x8 tiles, 256 workgroups
    (8, 16, 16, False, True), (8, 16, 16, True, False),       # 16x16: 4x4 tiles; the level's first block takes the 2x2 max-pool itself
    (8, 8, 8, True, True), (8, 4, 4, False, True),            # 8x8, 4x4 (one tile per crop, the whole ring outside the map)
    (1, 4, 4, False, False), (3, 12, 20, True, False),        # one crop; ragged: tiles cut by the right / bottom edge
    (16, 32, 32, False, True), (2, 6, 10, False, False),      # two frames; 4x8 tiles forced ragged below
])
def test_res_block_equals_the_separate_launches(ops, L, H, W, up, pool):
    from suo_slam_amd import _lib
    rng = np.random.default_rng(L * 1000 + H * 10 + W + up + 2 * pool)
    B = _block_weights(rng)
    xin = torch.from_numpy(rng.standard_normal((L, 2 * H, 2 * W, 256) if pool else (L, H, W, 256)).astype(np.float32)).cuda()
    low = torch.from_numpy(rng.standard_normal((L, H // 2, W // 2, 256)).astype(np.float32)).cuda() if up else None
    got = ops.res_block(xin, B["pro"], B["w1"], B["b1"], B["w2"], B["b2"], B["w3"], B["b3"], up_nhwc=low, pool_in=pool)
    x = xin
    if pool:
        x = torch.empty((L, H, W, 256), device="cuda")
        _lib.check(_lib.lib().suo_maxpool2(ops.P(xin), ops.P(x), L, 2 * H, 2 * W, 256, ops.S()), "suo_maxpool2")
        torch.cuda.synchronize()
    want = _separate_launches(ops, x, B, low)
    ref = _fp64(x, B, low)
    err = np.abs(got.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 5e-6, err                                       # the bound of every fp32 convolution kernel (observed ~3e-7)
    assert torch.equal(got, want), float((got - want).abs().max())


def test_res_block_tile_shapes_agree(ops, monkeypatch):
    """4x4 and 4x8 pixel tiles of the same block (SUO_RES_TILE is read once per process: both shapes are reached through map sizes
    that select them): a 32x32 map at 8 crops takes 4x8 tiles, at 1 crop 4x4 -- crop 0 must come out the same bits."""
    rng = np.random.default_rng(5)
    B = _block_weights(rng)
    x = torch.from_numpy(rng.standard_normal((8, 32, 32, 256)).astype(np.float32)).cuda()
    a = ops.res_block(x, B["pro"], B["w1"], B["b1"], B["w2"], B["b2"], B["w3"], B["b3"])
    b = ops.res_block(x[:1].contiguous(), B["pro"], B["w1"], B["b1"], B["w2"], B["b2"], B["w3"], B["b3"])
    assert torch.equal(a[:1], b)


@pytest.mark.parametrize("L,H,W,up,pool", [
    (8, 32, 32, False, False), (8, 32, 32, True, True), (8, 16, 16, True, False), (8, 16, 16, False, True),
    (3, 12, 20, True, False), (1, 4, 4, False, False), (2, 6, 10, False, True), (16, 8, 8, False, False),
])
def test_res_block_bf16x3_is_fp32_accurate(ops, L, H, W, up, pool):
    """csrc/res_small_x3.hip against fp64 (5e-6 of the output range, the bound of every fp32 convolution kernel) and against the fp32-pipe
    kernel of csrc/res_small.hip on the same inputs: never worse than 2x its error (+ 1e-7), every crop within 1e-5 of it (stray stores,
    ragged tiles, the pool taken while staging, the up-sampled addend)."""
    rng = np.random.default_rng(L * 1000 + H * 10 + W + up + 2 * pool + 7)
    B = _block_weights(rng)
    xin = torch.from_numpy(rng.standard_normal((L, 2 * H, 2 * W, 256) if pool else (L, H, W, 256)).astype(np.float32)).cuda()
    low = torch.from_numpy(rng.standard_normal((L, H // 2, W // 2, 256)).astype(np.float32)).cuda() if up else None
    args = (xin, B["pro"], B["w1"], B["b1"], B["w2"], B["b2"], B["w3"], B["b3"])
    got = ops.res_block_x3(*args, up_nhwc=low, pool_in=pool)
    f32 = ops.res_block(*args, up_nhwc=low, pool_in=pool)
    x = xin.view(L, H, 2, W, 2, 256).amax(dim=(2, 4)) if pool else xin
    ref = _fp64(x, B, low)
    scale = np.abs(ref).max()
    e_x3 = np.abs(got.cpu().numpy() - ref).max() / scale
    e_f32 = np.abs(f32.cpu().numpy() - ref).max() / scale
    assert e_x3 < 5e-6, e_x3
    assert e_x3 <= 2.0 * e_f32 + 1e-7, (e_x3, e_f32)
    per_crop = (got - f32).abs().reshape(L, -1).amax(1) / float(scale)
    assert float(per_crop.max()) < 1e-5, int(per_crop.argmax())


def test_res_block_bf16x3_forward_error_per_element(ops):
    """The per-element gate of tests/test_gpu_x3_accuracy.py for the whole block on one-signed data (positive x, weights, biases, BN): three
    products deep, every partial result positive, so sum |.||.| IS the result.  |err| <= 2 sqrt(K) 2^-24 of it with K = 256 + 1152 + 128 terms
    per output, never worse than 1.5x the fp32-pipe kernel, signed mean within the matrix pipe's truncation bias."""
    rng = np.random.default_rng(11)
    L, H, W = 4, 32, 32
    B = _block_weights(rng)
    B = dict(B, pro=(np.abs(B["pro"][0]), np.abs(B["pro"][1])), w1=np.abs(B["w1"]), b1=np.abs(B["b1"]), w2=np.abs(B["w2"]), b2=np.abs(B["b2"]),
             w3=np.abs(B["w3"]), b3=np.abs(B["b3"]))
    x = torch.from_numpy(np.abs(rng.standard_normal((L, H, W, 256))).astype(np.float32)).cuda()
    args = (x, B["pro"], B["w1"], B["b1"], B["w2"], B["b2"], B["w3"], B["b3"])
    got = ops.res_block_x3(*args).cpu().numpy().astype(np.float64)
    f32 = ops.res_block(*args).cpu().numpy().astype(np.float64)
    ref = _fp64(x, B)
    U = 2.0 ** -24
    ex, ef = (got - ref) / (U * ref), (f32 - ref) / (U * ref)
    print("\nres_block one-signed: max|err| bf16x3 %.3f fp32 pipe %.3f, mean signed %+.4f %+.4f, std %.3f %.3f   [units of 2^-24 sum|x||w|]"
          % (np.abs(ex).max(), np.abs(ef).max(), ex.mean(), ef.mean(), ex.std(), ef.std()))
    assert np.abs(ex).max() <= 2.0 * np.sqrt(256 + 1152 + 128)
    assert np.abs(ex).max() <= 1.5 * np.abs(ef).max() + 0.5
    assert abs(ex.mean()) <= 1.5                                # (three products deep; 720 truncating MFMA accumulations per wave: ~ -0.9)


# ---- the same block on two fp16 terms per operand (csrc/res_small_x3.hip, NP = 2; csrc/f16x2.h): the network's default since round 5 ----------------
@pytest.mark.parametrize("L,H,W,up,pool", [
    (8, 32, 32, False, False), (8, 32, 32, True, True), (8, 16, 16, True, False), (8, 16, 16, False, True),
    (3, 12, 20, True, False), (1, 4, 4, False, False), (2, 6, 10, False, True), (16, 8, 8, False, False),
])
def test_res_block_f16x2_is_fp32_accurate(ops, L, H, W, up, pool):
    """The bounds the bf16x3 block is held to (5e-6 of the output range against fp64; never worse than 2x the fp32-pipe kernel + 1e-7; every crop within
    1e-5 of it), the range flag down."""
    rng = np.random.default_rng(L * 1000 + H * 10 + W + up + 2 * pool + 7)
    B = _block_weights(rng)
    xin = torch.from_numpy(rng.standard_normal((L, 2 * H, 2 * W, 256) if pool else (L, H, W, 256)).astype(np.float32)).cuda()
    low = torch.from_numpy(rng.standard_normal((L, H // 2, W // 2, 256)).astype(np.float32)).cuda() if up else None
    args = (xin, B["pro"], B["w1"], B["b1"], B["w2"], B["b2"], B["w3"], B["b3"])
    got, flag = ops.res_block_f16x2(*args, up_nhwc=low, pool_in=pool)
    assert flag == 0
    f32 = ops.res_block(*args, up_nhwc=low, pool_in=pool)
    x = xin.view(L, H, 2, W, 2, 256).amax(dim=(2, 4)) if pool else xin
    ref = _fp64(x, B, low)
    scale = np.abs(ref).max()
    e_h = np.abs(got.cpu().numpy() - ref).max() / scale
    e_f32 = np.abs(f32.cpu().numpy() - ref).max() / scale
    assert e_h < 5e-6, e_h
    assert e_h <= 2.0 * e_f32 + 1e-7, (e_h, e_f32)
    per_crop = (got - f32).abs().reshape(L, -1).amax(1) / float(scale)
    assert float(per_crop.max()) < 1e-5, int(per_crop.argmax())


def test_res_block_f16x2_forward_error_per_element_and_range_guard(ops):
    """The per-element gate on one-signed data (as the bf16x3 block's), and the guard: an input, or either INNER activation (relu(conv1), relu(conv2): tensors the
    caller never sees), of 4094 or more raises the flag; just below, the flag stays down and the result is finite."""
    rng = np.random.default_rng(11)
    L, H, W = 4, 32, 32
    B = _block_weights(rng)
    B = dict(B, pro=(np.abs(B["pro"][0]), np.abs(B["pro"][1])), w1=np.abs(B["w1"]), b1=np.abs(B["b1"]), w2=np.abs(B["w2"]), b2=np.abs(B["b2"]),
             w3=np.abs(B["w3"]), b3=np.abs(B["b3"]))
    x = torch.from_numpy(np.abs(rng.standard_normal((L, H, W, 256))).astype(np.float32)).cuda()
    args = (x, B["pro"], B["w1"], B["b1"], B["w2"], B["b2"], B["w3"], B["b3"])
    got, flag = ops.res_block_f16x2(*args)
    assert flag == 0
    got = got.cpu().numpy().astype(np.float64)
    f32 = ops.res_block(*args).cpu().numpy().astype(np.float64)
    ref = _fp64(x, B)
    U = 2.0 ** -24
    ex, ef = (got - ref) / (U * ref), (f32 - ref) / (U * ref)
    print("\nres_block one-signed: max|err| f16x2 %.3f fp32 pipe %.3f, mean signed %+.4f %+.4f, std %.3f %.3f   [units of 2^-24 sum|x||w|]"
          % (np.abs(ex).max(), np.abs(ef).max(), ex.mean(), ef.mean(), ex.std(), ef.std()))
    assert np.abs(ex).max() <= 2.0 * np.sqrt(256 + 1152 + 128)
    assert np.abs(ex).max() <= 1.5 * np.abs(ef).max() + 1.5
    assert abs(ex.mean()) <= 1.5
    # the guard, on the block's own input (after the prologue: scale 1, shift 0 here) ...
    Bg = dict(_block_weights(rng), pro=(np.ones(256, np.float32), np.zeros(256, np.float32)))
    for big, want in ((4093.0, 0), (4095.0, 1), (np.inf, 1)):
        xg = rng.standard_normal((2, 8, 8, 256)).astype(np.float32)
        xg[1, 7, 7, 255] = big
        out, flag = ops.res_block_f16x2(torch.from_numpy(xg).cuda(), Bg["pro"], Bg["w1"], Bg["b1"], Bg["w2"], Bg["b2"], Bg["w3"], Bg["b3"])
        assert flag == want, (big, flag)
        if not want:
            assert torch.isfinite(out).all()
    # ... and on the inner activations: conv1 / conv2 gains that push relu(conv1) / relu(conv2) beyond 4094 while the input stays O(1)
    xg = torch.from_numpy(np.abs(rng.standard_normal((2, 8, 8, 256))).astype(np.float32)).cuda()
    for key, gain in (("w1", 2000.0), ("w2", 3000.0)):
        Bi = dict(B)
        Bi[key] = B[key] * np.float32(gain)
        out, flag = ops.res_block_f16x2(xg, Bi["pro"], Bi["w1"], Bi["b1"], Bi["w2"], Bi["b2"], Bi["w3"], Bi["b3"])
        assert flag == 1, key

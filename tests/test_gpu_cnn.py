"""GPU parity tests of the CNN path: every HIP stage, called through the C ABI, against the oracle
(oracle/cnn_oracle.py) and the committed golden vectors.  Tolerances are stated per test.  The network's
default kernels form their products on the bf16 matrix pipe from operands split into three bf16 terms
(csrc/bf16x3.h: exact split, 6 of 9 cross terms, fp32 accumulation -- fp32-level accuracy by construction,
gated per element in tests/test_gpu_x3_accuracy.py); SUO_WINO_BF16X3=0 builds the fp32-MFMA network (an
exact fp32 FMA chain).  Network-level gates hold BOTH to BASELINE.md section 4.5's 1e-5."""
import ctypes as C
import os

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


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _conv1x1_f32_sequential(ops, a1, w1, bias, pro=None, a2=None, w2=None, res=None, relu=False):
    """The fp32-pipe 1x1 kernel that accumulates like the bf16x3 one does -- ONE accumulator per output, K ascending (csrc/conv.hip,
    csrc/gemm_persist.hip) -- as the yardstick of "no worse than the fp32 pipe".  Below 4097 rows suo_conv1x1 dispatches to the split-K kernel
    of csrc/conv_small.hip, whose four partial sums make it ~2.5x MORE accurate than any one-accumulator kernel (fp32 or bf16x3; the network
    never launches a bf16x3 GEMM below 32768 rows): rows are independent, so the operands are tiled past that size and the first M rows returned."""
    M = a1.shape[0]
    reps = 1 if M > 4096 else -(-4097 // M)
    t = (lambda x: x.repeat(reps, 1).contiguous() if x is not None else None)
    out = ops.conv1x1(t(a1), w1, bias, pro=pro, a2=t(a2), w2=w2, res=t(res), relu=relu)
    return out[:M].cpu().numpy()


@pytest.mark.parametrize("M,K,N", [(128, 64, 64), (4096, 256, 128), (1000, 128, 256), (37, 64, 128), (8192, 128, 64)])
def test_conv1x1_plain(ops, M, K, N):
    rng = np.random.default_rng(M + K + N)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    out = ops.conv1x1(ops.dev(a), w, b).cpu().numpy()
    ref = a.astype(np.float64) @ w.astype(np.float64).T + b
    assert _rel(out, ref) < 5e-6


def test_conv1x1_asymmetric_identity(ops):
    # A = I (first 64 rows) with an asymmetric W catches operand / output transposes
    K = N = 64
    a = np.zeros((64, K), np.float32)
    a[np.arange(64), np.arange(64)] = 1
    w = np.arange(N * K, dtype=np.float32).reshape(N, K)
    out = ops.conv1x1(ops.dev(a), w, np.zeros(N, np.float32)).cpu().numpy()
    assert np.array_equal(out, w.T)


def test_conv1x1_fused_prologue_dual_residual_relu(ops):
    rng = np.random.default_rng(7)
    M, K1, K2, N = 2000, 128, 64, 256
    a1 = rng.standard_normal((M, K1)).astype(np.float32)
    a2 = rng.standard_normal((M, K2)).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    w1 = (rng.standard_normal((N, K1)) / 10).astype(np.float32)
    w2 = (rng.standard_normal((N, K2)) / 10).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, K1).astype(np.float32)
    sh = rng.standard_normal(K1).astype(np.float32)
    out = ops.conv1x1(ops.dev(a1), w1, b, pro=(sc, sh), a2=ops.dev(a2), w2=w2, res=ops.dev(r), relu=True).cpu().numpy()
    act = np.maximum(a1.astype(np.float64) * sc + sh, 0)
    ref = np.maximum(act @ w1.T.astype(np.float64) + a2.astype(np.float64) @ w2.T + b + r, 0)
    assert _rel(out, ref) < 5e-6


@pytest.mark.parametrize("M,K1,K2,N,pro,res,relu", [
    (131072, 128, 0, 256, False, True, False),     # 128x128 tiles, 2048 tiles over 768 persistent workgroups (2-3 each)
    (66048, 256, 0, 128, True, False, True),       # 128x128 tiles, 516 tiles (not a multiple of 8): plain round-robin lists
    (262144, 256, 0, 128, True, False, True),      # 2048 tiles over 768 workgroups, prologue constants from LDS
    (65536, 256, 64, 256, False, True, False),     # dual operand with a 2-chunk tail (re-injection shape)
    (65536, 64, 64, 128, False, False, False),     # 2 + 2 chunks per tile
    (32768, 256, 0, 128, True, False, True),       # 128x64 tiles, one tile per workgroup
    (33152, 128, 0, 128, False, True, False),      # 128x64 tiles, 518 tiles (not a multiple of 8)
    (8192, 256, 0, 128, True, False, True),        # 64x64 tiles
    (8256, 32, 0, 64, False, True, True),          # single-chunk tiles: every step is a tile boundary; 129 tiles
    (4160, 64, 0, 64, False, False, False),        # 65 tiles: odd count, fewer than the resident workgroups
])
def test_conv1x1_persistent_tile_walk(ops, M, K1, K2, N, pro, res, relu):
    """Whole-tile shapes take the persistent kernel (csrc/gemm_persist.hip): tile lists per workgroup, the activation
    prefetch and the weight ring running across tile boundaries, and the parked fetch cursor at the tail must not change
    a single output."""
    rng = np.random.default_rng(M % 1000 + K1 + N)
    a1 = rng.standard_normal((M, K1)).astype(np.float32)
    w1 = (rng.standard_normal((N, K1)) / np.sqrt(K1 + K2)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    kw, ref = {}, None
    act = a1.astype(np.float64)
    if pro:
        sc, sh = rng.uniform(0.5, 1.5, K1).astype(np.float32), rng.standard_normal(K1).astype(np.float32)
        kw["pro"] = (sc, sh)
        act = np.maximum(act * sc + sh, 0)
    ref = act @ w1.T.astype(np.float64) + b
    if K2:
        a2 = rng.standard_normal((M, K2)).astype(np.float32)
        w2 = (rng.standard_normal((N, K2)) / np.sqrt(K1 + K2)).astype(np.float32)
        kw.update(a2=ops.dev(a2), w2=w2)
        ref += a2.astype(np.float64) @ w2.T
    if res:
        r = rng.standard_normal((M, N)).astype(np.float32)
        kw["res"] = ops.dev(r)
        ref += r
    if relu:
        ref = np.maximum(ref, 0)
    out = ops.conv1x1(ops.dev(a1), w1, b, relu=relu, **kw).cpu().numpy()
    assert _rel(out, ref) < 5e-6
    # per-row check as well: a mis-assigned tile would hide in a global norm only if it were tiny, never in a row max
    bad = np.abs(out - ref).max(1) > 1e-3 * (1 + np.abs(ref).max())
    assert not bad.any(), np.flatnonzero(bad)[:8]


def test_conv1x1_nchw_head(ops):
    rng = np.random.default_rng(8)
    L, hw, K, N = 2, 4096, 256, 41
    a = rng.standard_normal((L * hw, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / 16).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    out = ops.conv1x1(ops.dev(a), w, b, nchw_hw=hw).cpu().numpy()          # [L, 41, hw]
    ref = (a.astype(np.float64) @ w.T + b).reshape(L, hw, N).transpose(0, 2, 1)
    assert out.shape == (L, N, hw) and _rel(out, ref) < 5e-6


@pytest.mark.parametrize("L,H,C,N", [(2, 64, 128, 128), (1, 32, 128, 128), (3, 16, 64, 64), (2, 8, 128, 128), (1, 4, 128, 128), (1, 12, 32, 64)])
def test_conv3x3(ops, L, H, C, N):
    rng = np.random.default_rng(L * H + C)
    x = rng.standard_normal((L, C, H, H)).astype(np.float32)
    w = (rng.standard_normal((N, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    out = ops.nchw(ops.conv_kxk(ops.nhwc(x), w, b, relu=True))
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)).numpy()
    assert _rel(out, ref) < 5e-6


@pytest.mark.parametrize("L,H,W,C,N,why", [
    (80, 60, 60, 128, 128, "128x128 tiles (>= 2048 of them), 3 workgroups per CU, ragged bottom / right tiles"),
    (20, 60, 60, 128, 128, "128x64 tiles, 8-wave workgroups, ragged"),
    (20, 120, 104, 64, 64, "N = 64: 128x64 tiles with 4 waves, ragged, non-square"),
    (70, 64, 64, 128, 128, "128x128 tiles, tile count not a multiple of 8 (no XCD remap)"),
])
def test_conv3x3_large_launches(ops, L, H, W, C, N, why):
    """The tile configurations only large launches select (csrc/conv.hip: launch_conv3x3), with the zero padding and the
    out-of-tile pixels left to the buffer descriptors' range check."""
    rng = np.random.default_rng(L + H + C)
    x = rng.standard_normal((L, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((N, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    out = ops.nchw(ops.conv_kxk(ops.nhwc(x), w, b, relu=True))
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)).numpy()
    assert _rel(out, ref) < 5e-6, why
    bad = np.abs(out - ref).reshape(L, -1).max(1) > 1e-3 * (1 + np.abs(ref).max())
    assert not bad.any(), (why, np.flatnonzero(bad)[:8])     # per crop: a stray out-of-tile store lands in a neighbouring row


@pytest.mark.parametrize("L,H,W", [(128, 64, 64), (80, 60, 60), (70, 64, 72), (128, 32, 32)])
def test_fused_residual_tail_equals_the_separate_launches(ops, L, H, W):
    """conv2 (3x3) -> conv3 (1x1) + skip in one launch (csrc/conv.hip: FUSE), the shape class the network uses it for
    (>= 1024 tiles), ragged and non-square maps included: bit-identical to suo_conv_kxk followed by suo_conv1x1 with the
    residual operand (same summation order), and within 5e-6 of the fp64 reference (Residual.py:27-35)."""
    rng = np.random.default_rng(L + H)
    x = torch.from_numpy(rng.standard_normal((L, H, W, 128)).astype(np.float32)).cuda()
    skip = torch.from_numpy(rng.standard_normal((L, H, W, 256)).astype(np.float32)).cuda()
    w2 = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    b2 = rng.standard_normal(128).astype(np.float32) * 0.3
    w3 = (rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
    b3 = rng.standard_normal(256).astype(np.float32)
    out = ops.conv3x3_conv1x1_skip(x, w2, b2, w3, b3, skip)
    mid = ops.conv_kxk(x, w2, b2, relu=True).contiguous()
    sep = ops.conv1x1(mid.reshape(-1, 128), w3, b3, res=skip.reshape(-1, 256)).reshape(L, H, W, 256)
    assert torch.equal(out, sep)
    for l in (0, L // 2, L - 1):                         # fp64 reference on a few crops (the whole launch is covered by `sep`)
        xm = x[l:l + 1].permute(0, 3, 1, 2).double().cpu()
        m = F.relu(F.conv2d(xm, torch.from_numpy(w2).double(), torch.from_numpy(b2).double(), padding=1))
        ref = F.conv2d(m, torch.from_numpy(w3).double()[:, :, None, None], torch.from_numpy(b3).double()) + skip[l:l + 1].permute(0, 3, 1, 2).double().cpu()
        assert _rel(out[l:l + 1].permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 5e-6


@pytest.mark.parametrize("L,H,W,C,N", [(2, 64, 64, 128, 128), (3, 40, 40, 128, 128), (128, 16, 16, 128, 128), (5, 24, 56, 128, 128), (1, 8, 16, 128, 128),
                                        (2, 32, 32, 64, 128), (2, 128, 128, 64, 64), (3, 40, 72, 64, 64), (1, 8, 16, 64, 64), (40, 64, 64, 64, 64)])
def test_conv3x3_winograd(ops, L, H, W, C, N):
    """3x3 convolution in Winograd F(2x2,3x3) form (csrc/conv_wino.hip): within 5e-6 of the fp64 reference (the direct kernel's
    bound; observed ~3x closer than the direct kernel, which sums 9x more terms per output), ragged and non-square maps, maps
    of a single tile, the 64 -> 64 form (components split over wave pairs), per-crop check against stray stores."""
    rng = np.random.default_rng(L * H + W)
    x = rng.standard_normal((L, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((N, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    out = ops.nchw(ops.conv3x3_wino(ops.nhwc(x), w, b, relu=True))
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)).numpy()
    assert _rel(out, ref) < 5e-6
    bad = np.abs(out - ref).reshape(L, -1).max(1) > 1e-3 * (1 + np.abs(ref).max())
    assert not bad.any(), np.flatnonzero(bad)[:8]


@pytest.mark.parametrize("L,H,W", [(128, 64, 64), (20, 60, 60), (7, 64, 72), (128, 16, 16)])
def test_fused_winograd_residual_tail_equals_the_separate_launches(ops, L, H, W):
    """conv2 (3x3, Winograd) -> conv3 (1x1) + skip in one launch -- what the network runs for its 256 -> 256 blocks:
    bit-identical to suo_conv3x3_wino followed by suo_conv1x1 with the residual operand, and within 5e-6 of fp64."""
    rng = np.random.default_rng(L + H)
    x = torch.from_numpy(rng.standard_normal((L, H, W, 128)).astype(np.float32)).cuda()
    skip = torch.from_numpy(rng.standard_normal((L, H, W, 256)).astype(np.float32)).cuda()
    w2 = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    b2 = rng.standard_normal(128).astype(np.float32) * 0.3
    w3 = (rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
    b3 = rng.standard_normal(256).astype(np.float32)
    out = ops.conv3x3_wino_conv1x1_skip(x, w2, b2, w3, b3, skip)
    mid = ops.conv3x3_wino(x, w2, b2, relu=True).contiguous()
    sep = ops.conv1x1(mid.reshape(-1, 128), w3, b3, res=skip.reshape(-1, 256)).reshape(L, H, W, 256)
    assert torch.equal(out, sep)
    for l in (0, L - 1):
        xm = x[l:l + 1].permute(0, 3, 1, 2).double().cpu()
        m = F.relu(F.conv2d(xm, torch.from_numpy(w2).double(), torch.from_numpy(b2).double(), padding=1))
        ref = F.conv2d(m, torch.from_numpy(w3).double()[:, :, None, None], torch.from_numpy(b3).double()) + skip[l:l + 1].permute(0, 3, 1, 2).double().cpu()
        assert _rel(out[l:l + 1].permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) < 5e-6


@pytest.mark.parametrize("L,H,W,scale,C", [(2, 64, 64, 1.0, 128), (3, 40, 40, 1.0, 128), (128, 16, 16, 1.0, 128), (5, 24, 56, 1.0, 128), (1, 8, 16, 1.0, 128),
                                            (2, 64, 64, 1e-12, 128), (2, 32, 32, 1e9, 128), (2, 128, 128, 1.0, 64), (3, 40, 72, 1.0, 64), (1, 8, 16, 1.0, 64),
                                            (40, 64, 64, 1.0, 64)])
def test_conv3x3_winograd_bf16x3_is_fp32_accurate(ops, L, H, W, scale, C):
    """csrc/conv_wino_x3.hip (what the network launches for its 128 -> 128 3x3 convolutions): the Winograd products on the BF16 matrix pipe,
    both operands split into three bf16 terms (round-to-nearest, exact residuals: csrc/bf16x3.h), 6 of the 9 cross terms accumulated in fp32.  Same bound as the
    fp32-pipe kernel (5e-6 of the output range against fp64; observed 4e-7), ragged maps, a single tile, the 64 -> 64 form (components split over wave
    pairs), tiny and huge magnitudes (bf16 has
    fp32's exponent range: the split must not lose the small terms), per-crop check against stray stores."""
    rng = np.random.default_rng(L * H + W)
    x = (rng.standard_normal((L, C, H, W)) * scale).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    b = (rng.standard_normal(C) * scale).astype(np.float32)
    out = ops.nchw(ops.conv3x3_wino_x3(ops.nhwc(x), w, b, relu=True))
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)).numpy()
    assert _rel(out, ref) < 5e-6
    bad = np.abs(out - ref).reshape(L, -1).max(1) > 1e-3 * np.abs(ref).max()
    assert not bad.any(), np.flatnonzero(bad)[:8]
    f32 = ops.nchw(ops.conv3x3_wino(ops.nhwc(x), w, b, relu=True))
    assert _rel(out, f32) < 5e-6                              # and no worse than the fp32-pipe kernel
    assert _rel(out, ref) < 2.0 * _rel(f32, ref) + 1e-7


@pytest.mark.parametrize("L,H,W,up,tail_x3", [(128, 64, 64, False, True), (20, 60, 60, False, True), (7, 64, 72, True, True), (128, 16, 16, True, True),
                                               (9, 40, 56, True, False), (3, 8, 16, False, False)])
def test_fused_winograd_bf16x3_residual_tail(ops, L, H, W, up, tail_x3):
    """conv2 (3x3, Winograd, bf16x3) -> ReLU -> conv3 (1x1, on the bf16 pipe too, or on the fp32 one) + bias + skip [+ up-sampled addend] in one
    launch: within 5e-6 of fp64 on the first and last crop, and within 1e-5 of the fp32-pipe fused kernel on EVERY crop (stray stores, the
    two pixel-row passes of the tail, ragged edges)."""
    rng = np.random.default_rng(L + H)
    x = torch.from_numpy(rng.standard_normal((L, H, W, 128)).astype(np.float32)).cuda()
    skip = torch.from_numpy(rng.standard_normal((L, H, W, 256)).astype(np.float32)).cuda()
    low = torch.from_numpy(rng.standard_normal((L, H // 2, W // 2, 256)).astype(np.float32)).cuda() if up else None
    w2 = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    b2 = rng.standard_normal(128).astype(np.float32) * 0.3
    w3 = (rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
    b3 = rng.standard_normal(256).astype(np.float32)
    out = ops.conv3x3_wino_x3_conv1x1_skip_up(x, w2, b2, w3, b3, skip, low, tail_x3=tail_x3)
    f32 = ops.conv3x3_wino_conv1x1_skip_up(x, w2, b2, w3, b3, skip, low) if up else ops.conv3x3_wino_conv1x1_skip(x, w2, b2, w3, b3, skip)
    scale = float(f32.abs().max())
    per_crop = (out - f32).abs().reshape(L, -1).amax(1) / scale
    assert float(per_crop.max()) < 1e-5, int(per_crop.argmax())
    for l in (0, L - 1):
        xm = x[l:l + 1].permute(0, 3, 1, 2).double().cpu()
        m = F.relu(F.conv2d(xm, torch.from_numpy(w2).double(), torch.from_numpy(b2).double(), padding=1))
        ref = F.conv2d(m, torch.from_numpy(w3).double()[:, :, None, None], torch.from_numpy(b3).double()) + skip[l:l + 1].permute(0, 3, 1, 2).double().cpu()
        if up:
            ref = ref + low[l:l + 1].permute(0, 3, 1, 2).double().cpu().repeat_interleave(2, 2).repeat_interleave(2, 3)
        e_x3 = _rel(out[l:l + 1].permute(0, 3, 1, 2).cpu().numpy(), ref.numpy())
        assert e_x3 < 5e-6
        assert e_x3 <= 2.0 * _rel(f32[l:l + 1].permute(0, 3, 1, 2).cpu().numpy(), ref.numpy()) + 1e-7      # never worse than the fp32-pipe kernel


def test_bf16x3_kernels_at_the_bench_launch_shape_size_independent_properties(ops):
    """The bench's launch shape (256 crops x 64 x 64: 8192 workgroups, 1 M GEMM rows) is too big to check against fp64 on the host, so two
    properties that hold EXACTLY stand in: (1) homogeneity under powers of two -- the round-to-nearest splits commute with a scaling by 2^k, every
    product and sum scales exactly, ReLU commutes with a positive factor: f(4 x; 4 b, 4 skip) == 4 f(x; b, skip) bit for bit; (2) crops are
    independent -- the same crop at every position of the batch gives the same output at every position (stray indexing, tile order)."""
    from suo_slam_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(2026)
    L, H, W = 256, 64, 64
    one = rng.standard_normal((1, H, W, 128)).astype(np.float32)
    skip1 = rng.standard_normal((1, H, W, 256)).astype(np.float32)
    w2 = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    b2 = (rng.standard_normal(128) * 0.25).astype(np.float32)
    w3 = (rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
    b3 = rng.standard_normal(256).astype(np.float32)
    x = torch.from_numpy(one).cuda().expand(L, H, W, 128).contiguous()
    skip = torch.from_numpy(skip1).cuda().expand(L, H, W, 256).contiguous()
    out = ops.conv3x3_wino_x3_conv1x1_skip_up(x, w2, b2, w3, b3, skip, None, tail_x3=True)
    assert bool((out == out[0:1]).all())                                     # (2)
    out4 = ops.conv3x3_wino_x3_conv1x1_skip_up(x * 4, w2, b2 * 4, w3, b3 * 4, skip * 4, None, tail_x3=True)
    assert torch.equal(out4, out * 4)                                        # (1)
    del out4, skip
    # the conv1 GEMM at M = 1 048 576 with its prologue (scale a power of two too: the BN shift scales with the input)
    K, N, M = 256, 128, L * H * W
    w1 = (rng.standard_normal((N, K)) / 16.0).astype(np.float32)
    w1x = np.empty(3 * N * K, np.uint16)
    _lib.check(lib.suo_pack_gemm_weight_bf16x3(w1.ctypes.data, N, K, w1x.ctypes.data))
    w1d = torch.from_numpy(w1x.view(np.int16)).cuda()
    a = out.reshape(M, 256)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, K).astype(np.float32)).cuda()
    sh = torch.from_numpy((rng.standard_normal(K) * 0.1).astype(np.float32)).cuda()
    bias = torch.from_numpy((rng.standard_normal(N) * 0.1).astype(np.float32)).cuda()
    o1, o2 = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
    _lib.check(lib.suo_conv1x1_bf16x3(ops.P(a), K, K, ops.P(sc), ops.P(sh), ops.P(w1d), ops.P(bias), ops.P(o1), N, M, N, 1, ops.S()))
    a8, sh8, bias8 = a * 8, sh * 8, bias * 8
    _lib.check(lib.suo_conv1x1_bf16x3(ops.P(a8), K, K, ops.P(sc), ops.P(sh8), ops.P(w1d), ops.P(bias8), ops.P(o2), N, M, N, 1, ops.S()))
    torch.cuda.synchronize()
    assert torch.equal(o2, o1 * 8)
    assert bool((o1.reshape(L, H * W, N) == o1.reshape(L, H * W, N)[0:1]).all())


@pytest.mark.parametrize("L,H,W", [(64, 64, 64), (9, 40, 56), (128, 16, 16)])
def test_fused_tail_with_upsampled_addend_equals_tail_then_upsample_add(ops, L, H, W):
    """Hourglass "up1 + up2(low3)" (hg.py:56-58) folded into the last up1 block's fused tail: bit-identical to the fused tail
    followed by suo_upsample2_add (the addend goes in last, as that kernel adds it)."""
    rng = np.random.default_rng(L + W)
    x = torch.from_numpy(rng.standard_normal((L, H, W, 128)).astype(np.float32)).cuda()
    skip = torch.from_numpy(rng.standard_normal((L, H, W, 256)).astype(np.float32)).cuda()
    low = torch.from_numpy(rng.standard_normal((L, H // 2, W // 2, 256)).astype(np.float32)).cuda()
    w2 = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    b2 = rng.standard_normal(128).astype(np.float32) * 0.3
    w3 = (rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
    b3 = rng.standard_normal(256).astype(np.float32)
    got = ops.conv3x3_wino_conv1x1_skip_up(x, w2, b2, w3, b3, skip, low)
    tail = ops.conv3x3_wino_conv1x1_skip(x, w2, b2, w3, b3, skip)
    want = torch.empty_like(tail)
    from suo_slam_amd import _lib
    _lib.check(_lib.lib().suo_upsample2_add(ops.P(tail), ops.P(low), ops.P(want), L, H, W, 256, ops.S()), "suo_upsample2_add")
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = tail + low.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    assert torch.equal(want, ref)


@pytest.mark.parametrize("L,H,W,K1,K2,N,res,relu,want_full", [
    (3, 128, 128, 64, 64, 128, False, False, False),     # r1: conv3 + conv4, only the pooled tensor is kept
    (5, 64, 64, 128, 128, 256, False, False, True),      # r5 -> the first Hourglass: x and max_pool(x)
    (2, 64, 64, 256, 64, 256, True, False, True),        # re-injection (x + ll_ + tmpOut_) -> the second Hourglass
    (12, 6, 64, 256, 0, 128, False, True, True),         # ragged height, prologue + ReLU (M > 4096: below that suo_conv1x1 is the split-K kernel)
    (16, 2, 192, 96, 0, 128, True, True, False),         # three column blocks, K not a power of two
])
def test_conv1x1_with_fused_maxpool_equals_conv1x1_then_maxpool(ops, L, H, W, K1, K2, N, res, relu, want_full):
    """nn.MaxPool2d(2, 2) (hg.py:41, the stem's pool) folded into the epilogue of the 1x1 convolution that produces its input:
    both tensors bit-identical to suo_conv1x1 followed by suo_maxpool2."""
    from suo_slam_amd import _lib
    rng = np.random.default_rng(H * W + K1)
    M = L * H * W
    a1 = torch.from_numpy(rng.standard_normal((M, K1)).astype(np.float32)).cuda()
    a2 = torch.from_numpy(rng.standard_normal((M, K2)).astype(np.float32)).cuda() if K2 else None
    r = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda() if res else None
    w1 = (rng.standard_normal((N, K1)) / np.sqrt(K1)).astype(np.float32)
    w2 = (rng.standard_normal((N, K2)) / np.sqrt(K2)).astype(np.float32) if K2 else None
    b = rng.standard_normal(N).astype(np.float32)
    pro = (rng.uniform(0.5, 1.5, K1).astype(np.float32), rng.standard_normal(K1).astype(np.float32) * 0.2) if relu else None
    want = ops.conv1x1(a1, w1, b, pro=pro, a2=a2, w2=w2, res=r, relu=relu)
    want_pool = torch.empty((M // 4, N), device="cuda")
    _lib.check(_lib.lib().suo_maxpool2(ops.P(want), ops.P(want_pool), L, H, W, N, ops.S()), "suo_maxpool2")
    torch.cuda.synchronize()
    ref_pool = want.view(L, H // 2, 2, W // 2, 2, N).amax(dim=(2, 4)).reshape(M // 4, N)
    assert torch.equal(want_pool, ref_pool)
    full, pooled = ops.conv1x1_pool(a1, w1, b, H, W, pro=pro, a2=a2, w2=w2, res=r, relu=relu, want_full=want_full)
    assert torch.equal(pooled, want_pool)
    if want_full:
        assert torch.equal(full, want)


def test_conv1x1_with_fused_maxpool_refuses_other_shapes(ops):
    from suo_slam_amd import _lib
    a1 = torch.zeros((2 * 32 * 32, 64), device="cuda")
    with pytest.raises(RuntimeError, match="fused max-pool"):
        ops.conv1x1_pool(a1, np.zeros((128, 64), np.float32), np.zeros(128, np.float32), 32, 32)


def test_conv7x7_stride2(ops):
    rng = np.random.default_rng(11)
    L, H, C, N = 2, 64, 44, 64
    x = rng.standard_normal((L, C, H, H)).astype(np.float32)
    xp = np.zeros((L, 48, H, H), np.float32)
    xp[:, :C] = x
    w = (rng.standard_normal((N, C, 7, 7)) / np.sqrt(49 * C)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    out = ops.nchw(ops.conv_kxk(ops.nhwc(xp), w, b, relu=False))
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=2, padding=3).numpy()
    assert _rel(out, ref) < 5e-6


@pytest.mark.parametrize("cpad,L,H", [(4, 3, 64), (8, 3, 64), (4, 2, 76)])
def test_conv7x7_image_only_stem(ops, cpad, L, H):
    """The prior-less stem multiplies only the 3 image channels: staged as 4 channels with two taps per MFMA k-group
    (what the network uses) or as one 8-channel chunk; ragged tiles included (76 -> 38 = 4.75 tiles)."""
    rng = np.random.default_rng(21 + cpad)
    x = rng.standard_normal((L, 3, H, H)).astype(np.float32)
    xp = np.zeros((L, cpad, H, H), np.float32)
    xp[:, :3] = x
    xp[:, 3:] = rng.standard_normal((L, cpad - 3, H, H))          # the pad channels carry no weight: must not matter
    w = (rng.standard_normal((64, 3, 7, 7)) / np.sqrt(49 * 3)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    out = ops.nchw(ops.conv_kxk(ops.nhwc(xp), w, b, relu=True))
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=2, padding=3)).numpy()
    assert _rel(out, ref) < 5e-6


@pytest.mark.parametrize("L,C,H", [(2, 64, 16), (3, 48, 12), (5, 256, 64)])       # power-of-two extents (shift path) and not
def test_pool_and_upsample(ops, L, C, H):
    from suo_slam_amd import _lib
    rng = np.random.default_rng(12)
    x = rng.standard_normal((L, C, H, H)).astype(np.float32)
    xd = ops.nhwc(x)
    out = torch.empty((L, H // 2, H // 2, C), device="cuda")
    _lib.check(_lib.lib().suo_maxpool2(ops.P(xd), ops.P(out), L, H, H, C, ops.S()))
    torch.cuda.synchronize()
    assert np.array_equal(ops.nchw(out), F.max_pool2d(torch.from_numpy(x), 2, 2).numpy())
    low = rng.standard_normal((L, C, H // 2, H // 2)).astype(np.float32)
    o2 = torch.empty((L, H, H, C), device="cuda")
    lowd = ops.nhwc(low)
    _lib.check(_lib.lib().suo_upsample2_add(ops.P(xd), ops.P(lowd), ops.P(o2), L, H, H, C, ops.S()))
    torch.cuda.synchronize()
    ref = torch.from_numpy(x) + F.interpolate(torch.from_numpy(low), scale_factor=2)
    assert np.array_equal(ops.nchw(o2), ref.numpy())


def test_roi_align_concat_matches_oracle(ops):
    from oracle import cnn_oracle as O
    from suo_slam_amd import _lib
    rng = np.random.default_rng(13)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    boxes = np.array([[100.3, 50.7, 300.9, 260.2], [0, 0, 640, 480], [-20.5, -10, 90, 120], [600, 400, 700, 520],
                      [320, 240, 320.5, 240.2], [10, 20, 522, 532 - 60]], np.float32)
    pri = rng.uniform(0, 1, (len(boxes), 41, 256, 256)).astype(np.float32)
    ref = O.roi_align(O.image_to_chw(img), boxes)
    for fmt, src in ((0, ops.dev(img, torch.uint8)), (1, ops.dev(O.image_to_chw(img)))):
        for priors in (None, ops.dev(pri)):
            out = torch.full((len(boxes), 256, 256, 48), 7.0, device="cuda")
            bx = ops.dev(boxes)
            _lib.check(_lib.lib().suo_roi_align_concat(ops.P(src), fmt, 480, 640, ops.P(bx), len(boxes), ops.P(priors),
                                                       ops.P(out), ops.S()))
            torch.cuda.synchronize()
            o = out.cpu().numpy()
            np.testing.assert_allclose(o[..., :3].transpose(0, 3, 1, 2), ref, atol=1e-6, rtol=0)
            if priors is None:
                assert np.all(o[..., 3:] == 0)
            else:
                assert np.array_equal(o[..., 3:44].transpose(0, 3, 1, 2), pri) and np.all(o[..., 44:] == 0)


@pytest.mark.parametrize("H,W", [(540, 720), (480, 640)])
def test_roi_align_kernels_match_the_grid_sample_reference_for_large_boxes(ops, H, W):
    """Boxes beyond 256 px a side (2 and 3 samples per bin and axis: T-LESS 720x540 frames reach > 512 px, YCB-V 640x480 the whole frame) and boxes leaving the
    image, against the implementation the builder did not write (tests/roi_ref.py: F.grid_sample at the positions RoIAlign's float32 arithmetic gives, 1e-6) --
    the same reference the oracle is pinned to (tests/test_oracle_cnn.py).  Both device samplers: the staging kernel (roi_align_concat_kernel, uint8 and float
    frames) and, through its output, the fused stem (stem_x3_kernel uses the same device function, csrc/roi_sample.h: tests/test_gpu_stem.py)."""
    from suo_slam_amd import _lib
    from tests import roi_ref as R
    from oracle import cnn_oracle as O
    rng = np.random.default_rng(41 + H)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    boxes = R.large_boxes(rng, 24, H, W)
    boxes[0] = [3.25, 2.5, 3.25 + 700.0 * W / 720, 2.5 + 530.0 * H / 540]      # > 512 px on the T-LESS frame: three samples per bin and axis
    _, _, gw, gh = R.sample_positions(boxes)
    assert max(gw.max(), gh.max()) == 3 and min(gw.min(), gh.min()) == 1
    chw = O.image_to_chw(img)
    ref = R.roi_align_grid_sample(chw, boxes, 256, f32_positions=True)
    for fmt, src in ((0, ops.dev(img, torch.uint8)), (1, ops.dev(chw))):
        out = torch.full((len(boxes), 256, 256, 48), 7.0, device="cuda")
        bx = ops.dev(boxes)
        _lib.check(_lib.lib().suo_roi_align_concat(ops.P(src), fmt, H, W, ops.P(bx), len(boxes), None, ops.P(out), ops.S()))
        torch.cuda.synchronize()
        got = out.cpu().numpy()[..., :3].transpose(0, 3, 1, 2)
        assert np.abs(got - ref).max() < 1e-6, (fmt, np.abs(got - ref).max())


def test_decode_golden_and_masks(ops, cnn_golden, state_dict):
    from oracle import cnn_oracle as O
    from suo_slam_amd import _lib
    lib = _lib.lib()
    for key in ("decode", "backbone"):
        logits = cnn_golden[key + "_in"] if key == "decode" else cnn_golden["backbone_logits"]
        L = logits.shape[0]
        ld = ops.dev(logits)
        uv = torch.empty((L, 41, 2), device="cuda")
        cov = torch.empty((L, 41, 2, 2), device="cuda")
        ml = torch.empty((L, 41), device="cuda")
        _lib.check(lib.suo_decode_heatmaps(ops.P(ld), L, ops.P(uv), ops.P(cov), ops.P(ml), None, None, ops.S()))
        kl = torch.empty((L, 41), device="cuda")
        kp = torch.empty((L, 41), device="cuda")
        wc, bc = ops.dev(state_dict["classifier.2.weight"]), ops.dev(state_dict["classifier.2.bias"])   # keep alive
        _lib.check(lib.suo_classifier(ops.P(ml), ops.P(wc), ops.P(bc), L, ops.P(kl), ops.P(kp), ops.S()))
        torch.cuda.synchronize()
        # tolerance: abs 1e-5 (fp32 reduction-order noise over 4096 terms), SURVEY.md 7.2
        np.testing.assert_allclose(uv.cpu().numpy(), cnn_golden[key + "_uv"], atol=1e-5, rtol=0)
        np.testing.assert_allclose(cov.cpu().numpy(), cnn_golden[key + "_cov"], atol=1e-5, rtol=0)
        np.testing.assert_allclose(kl.cpu().numpy(), cnn_golden[key + "_kp_mask_logits"], atol=2e-5, rtol=0)
        np.testing.assert_allclose(kp.cpu().numpy(), cnn_golden[key + "_kp_mask"], atol=1e-5, rtol=0)
        # boolean masks: bit-exact, away from thresholds by construction of the comparison set
        rng = np.random.default_rng(3)
        mm = rng.random((L, 41)) > 0.2
        for bt, vt in ((0.9, 0.2), (1.0, 0.5)):
            ref = O.keypoint_masks(cnn_golden[key + "_uv"], cnn_golden[key + "_cov"], cnn_golden[key + "_kp_mask"], mm, bt, vt)
            from suo_slam_amd.pkpnet import keypoint_masks
            got = keypoint_masks(uv, cov, kp, mm, bt, vt).cpu().numpy().astype(bool)
            guv, gcov, gkp = cnn_golden[key + "_uv"], cnn_golden[key + "_cov"], cnn_golden[key + "_kp_mask"]
            near = (np.abs(gkp - 0.3) < 1e-4) | (np.abs(np.abs(guv).max(-1) - bt) < 1e-4) \
                | (np.abs(np.sqrt(gcov[..., [0, 1], [0, 1]]) - 2 * vt).min(-1) < 1e-4)
            assert np.array_equal(got[~near], ref[~near])


def test_full_network_golden(ops, cnn_golden, state_dict):
    """One 44x256x256 crop through the whole HIP backbone vs the reference's own output."""
    from suo_slam_amd import _lib
    from suo_slam_amd.pkpnet import PkpNet
    net = PkpNet(state_dict=state_dict, max_crops=2)
    rng = np.random.Generator(np.random.PCG64(int(cnn_golden["backbone_in_seed"])))
    x = rng.uniform(0, 1, (1, 44, 256, 256)).astype(np.float32)
    # feed the pre-cropped tensor: an exact 256x256 box at integer offset makes roi_align an identity
    # only for aligned=True; instead drive the backbone through the staging buffer directly.
    xin = np.zeros((1, 256, 256, 48), np.float32)
    xin[..., :44] = x.transpose(0, 2, 3, 1)
    ref = cnn_golden["backbone_logits"]
    from tests.gpu_backbone import run_backbone_from_staged
    for graph in (False, True):
        net.set_graph(graph)
        logits = run_backbone_from_staged(net, xin)
        rel = np.abs(logits - ref).max() / np.abs(ref).max()
        assert rel < 1e-5, rel   # 186 fp32 conv layers deep against the REFERENCE's own logits; observed ~1e-6


@pytest.mark.parametrize("L", [40, 16])
@pytest.mark.parametrize("pipe", ["f16x2", "bf16x3", "f32"])
def test_full_network_golden_on_the_winograd_path(ops, cnn_golden, state_dict, monkeypatch, pipe, L):
    """All three forms of the Residual blocks' 3x3 + tail and of the large 1x1 convolutions: two fp16 terms per operand (the default, csrc/f16x2.h), three
    bf16 terms (SUO_F16X2=0) and the fp32 pipe (SUO_WINO_BF16X3=0; both read when the network is built).  The same crop repeated 40 times: every 3x3 layer down to the 16x16 maps now has >= 256 tiles, so the Winograd kernels, the
    fused Residual tails, the fused up-sample adds and the pooled GEMMs run (test_full_network_golden at L = 1 takes the direct
    forms only).  Every copy against the REFERENCE's own logits; hard arg-max of the HIP logits = torch.argmax of the reference's
    wherever the runner-up is more than 1e-4 below the maximum."""
    from suo_slam_amd.pkpnet import PkpNet, decode_extras
    monkeypatch.delenv("SUO_WINO_BF16X3", raising=False)
    monkeypatch.delenv("SUO_F16X2", raising=False)
    if pipe == "bf16x3":
        monkeypatch.setenv("SUO_F16X2", "0")
    if pipe == "f32":
        monkeypatch.setenv("SUO_WINO_BF16X3", "0")
    net = PkpNet(state_dict=state_dict, max_crops=L)
    assert net.pipe() == {"f32": 0, "bf16x3": 1, "f16x2": 2}[pipe]
    rng = np.random.Generator(np.random.PCG64(int(cnn_golden["backbone_in_seed"])))
    x = rng.uniform(0, 1, (1, 44, 256, 256)).astype(np.float32)
    xin = np.zeros((L, 256, 256, 48), np.float32)
    xin[..., :44] = x.transpose(0, 2, 3, 1)
    ref = cnn_golden["backbone_logits"]
    from tests.gpu_backbone import run_backbone_from_staged
    logits = run_backbone_from_staged(net, xin)
    assert logits.shape == (L, 41, 64, 64)
    assert not net.range_exceeded() and net.pipe() == {"f32": 0, "bf16x3": 1, "f16x2": 2}[pipe]      # (the fp16 form stayed inside its range: these ARE its results)
    rel = np.abs(logits - ref).max() / np.abs(ref).max()
    assert rel < 1e-5, rel                                    # the reference's own logits, either pipe, every copy
    # ... and what the path hands on: uv / cov / mean logit decoded from the HIP logits vs the reference's decode of ITS logits
    dec = decode_extras(torch.from_numpy(logits[:2]).cuda())
    assert np.abs(dec["uv"].cpu().numpy() - cnn_golden["backbone_uv"]).max() < 1e-5
    assert np.abs(dec["cov"].cpu().numpy() - cnn_golden["backbone_cov"]).max() < 1e-5
    idx = decode_extras(torch.from_numpy(logits).cuda())["argmax"].cpu().numpy()
    sure = cnn_golden["backbone_top2_gap"][0] > 1e-4
    assert sure.sum() >= 40
    for i in (0, L // 2, L - 1):
        np.testing.assert_array_equal(idx[i][sure], cnn_golden["backbone_argmax"][0][sure])


@pytest.mark.parametrize("L", [1, 2, 3, 5, 7])
def test_full_network_golden_at_slam_call_sizes(ops, cnn_golden, state_dict, monkeypatch, L):
    """A SLAM pass is a call of 2-7 crops.  Since round 5 the fp16 pipe takes such calls on the Winograd 3x3 / fused tail / fp16 GEMM kernels as well (csrc/net.hip:
    from 32 tiles / 4096 rows; rounds 1-4 sent them to the direct fp32-pipe kernels below 8 crops), the 32x32 / 16x16 levels on the one-launch blocks, lin -> head as one
    launch: every copy of the crop against the REFERENCE's own logits at 1e-5, with and without the captured graph, the range flag down."""
    from suo_slam_amd.pkpnet import PkpNet
    from tests.gpu_backbone import run_backbone_from_staged
    monkeypatch.delenv("SUO_WINO_BF16X3", raising=False)
    monkeypatch.delenv("SUO_F16X2", raising=False)
    net = PkpNet(state_dict=state_dict, max_crops=8)
    assert net.pipe() == 2
    rng = np.random.Generator(np.random.PCG64(int(cnn_golden["backbone_in_seed"])))
    x = rng.uniform(0, 1, (1, 44, 256, 256)).astype(np.float32)
    xin = np.zeros((L, 256, 256, 48), np.float32)
    xin[..., :44] = x.transpose(0, 2, 3, 1)
    ref = cnn_golden["backbone_logits"]
    for graph in (False, True):
        net.set_graph(graph)
        logits = run_backbone_from_staged(net, xin)
        assert logits.shape == (L, 41, 64, 64) and not net.range_exceeded() and net.pipe() == 2
        rel = np.abs(logits - ref).max() / np.abs(ref).max()
        assert rel < 1e-5, (L, graph, rel)


def test_the_two_matrix_pipe_forms_are_both_reachable_and_agree(ops, state_dict, monkeypatch):
    """SUO_WINO_BF16X3 / SUO_F16X2 select the form when a network is built; "1" below = the split forms with the default SUO_F16X2, i.e. the fp16 one.
    The networks' logits on the same input agree to fp32-rounding level but are not bit-identical (= different kernels really ran)."""
    from suo_slam_amd.pkpnet import PkpNet
    from tests.gpu_backbone import run_backbone_from_staged
    L = 40
    xin = np.zeros((L, 256, 256, 48), np.float32)
    xin[..., :44] = np.random.default_rng(5).uniform(0, 1, (L, 256, 256, 44)).astype(np.float32)
    outs = {}
    for mode in (None, "0", "1"):
        if mode is None:
            monkeypatch.delenv("SUO_WINO_BF16X3", raising=False)
        else:
            monkeypatch.setenv("SUO_WINO_BF16X3", mode)
        outs[mode] = run_backbone_from_staged(PkpNet(state_dict=state_dict, max_crops=L), xin)
    assert np.array_equal(outs[None], outs["1"])
    assert not np.array_equal(outs["0"], outs["1"])
    assert np.abs(outs["0"] - outs["1"]).max() / np.abs(outs["0"]).max() < 1e-5
    # what the path hands to the solvers is the same on both pipes: uv / cov to 1e-5, the boolean keypoint masks (object_slam.py:1100-1115)
    # bit-identical wherever no input of the comparison sits within 1e-4 of its threshold, both threshold sets (evaluate.py:58-76)
    from suo_slam_amd import _lib
    from suo_slam_amd.pkpnet import decode_extras, keypoint_masks
    wc, bc = ops.dev(state_dict["classifier.2.weight"]), ops.dev(state_dict["classifier.2.bias"])
    dec = {}
    for mode in ("0", "1"):
        d = decode_extras(torch.from_numpy(outs[mode]).cuda())
        kl, kp = torch.empty((L, 41), device="cuda"), torch.empty((L, 41), device="cuda")
        _lib.check(_lib.lib().suo_classifier(ops.P(d["mean_logit"]), ops.P(wc), ops.P(bc), L, ops.P(kl), ops.P(kp), ops.S()))
        torch.cuda.synchronize()
        dec[mode] = (d["uv"], d["cov"], kp)
    for a, b in zip(dec["0"], dec["1"]):
        assert float((a - b).abs().max()) < 1e-5
    mm = np.random.default_rng(4).random((L, 41)) > 0.2
    guv, gcov, gkp = (t.cpu().numpy() for t in dec["0"])
    n_checked = 0
    for bt, vt in ((0.9, 0.2), (1.0, 0.5)):
        m0 = keypoint_masks(*dec["0"], mm, bt, vt).cpu().numpy().astype(bool)
        m1 = keypoint_masks(*dec["1"], mm, bt, vt).cpu().numpy().astype(bool)
        near = (np.abs(gkp - 0.3) < 1e-4) | (np.abs(np.abs(guv).max(-1) - bt) < 1e-4) | (np.abs(np.sqrt(gcov[..., [0, 1], [0, 1]]) - 2 * vt).min(-1) < 1e-4)
        assert np.array_equal(m0[~near], m1[~near])
        n_checked += int((~near).sum())
    assert n_checked > 0.95 * 2 * L * 41


def test_decode_hard_argmax_is_bit_exact_and_prob_is_the_softmax(ops, cnn_golden):
    """Optional outputs of suo_decode_heatmaps.  argmax_idx: int32 flat index h*64+w of the FIRST maximum -- bit-exact against
    torch.argmax of the same logits as the reference computed it (synthetic peaked maps, the reference backbone's own logits, and
    maps quantised to 8 levels = full of ties).  prob: the reference's spatial_softmax (pkpnet.py:13-17), rel 2e-6."""
    from suo_slam_amd import _lib
    lib = _lib.lib()
    tie = np.floor(np.random.Generator(np.random.PCG64(int(cnn_golden["tie_seed"]))).uniform(0, 8, (2, 41, 64, 64))).astype(np.float32)
    for logits, want, prob_want in ((cnn_golden["decode_in"], cnn_golden["decode_argmax"], cnn_golden["decode_prob_sample"]),
                                    (cnn_golden["backbone_logits"], cnn_golden["backbone_argmax"], cnn_golden["backbone_prob_sample"]),
                                    (tie, cnn_golden["tie_argmax"], None)):
        L = logits.shape[0]
        ld = ops.dev(logits)
        uv = torch.empty((L, 41, 2), device="cuda")
        cov = torch.empty((L, 41, 2, 2), device="cuda")
        ml = torch.empty((L, 41), device="cuda")
        idx = torch.full((L, 41), -7, dtype=torch.int32, device="cuda")
        prob = torch.empty((L, 41, 64, 64), device="cuda")
        _lib.check(lib.suo_decode_heatmaps(ops.P(ld), L, ops.P(uv), ops.P(cov), ops.P(ml), ops.P(idx), ops.P(prob), ops.S()))
        uv2, cov2 = torch.empty_like(uv), torch.empty_like(cov)
        _lib.check(lib.suo_decode_heatmaps(ops.P(ld), L, ops.P(uv2), ops.P(cov2), ops.P(ml), None, None, ops.S()))
        torch.cuda.synchronize()
        got = idx.cpu().numpy()
        assert got.dtype == np.int32
        np.testing.assert_array_equal(got, want)                                           # bit-exact integer keypoint indices
        np.testing.assert_array_equal(got, logits.reshape(L, 41, -1).argmax(-1))           # (numpy agrees on the convention)
        assert torch.equal(uv, uv2) and torch.equal(cov, cov2)                              # the extras do not perturb the decode
        p = prob.cpu().numpy()
        np.testing.assert_allclose(p.reshape(L, 41, -1).sum(-1), 1.0, rtol=0, atol=2e-6)
        if prob_want is not None:
            np.testing.assert_allclose(p[:, ::5, ::4, ::4], prob_want, rtol=2e-6, atol=1e-12)


def test_schedule_options_do_not_change_the_result():
    """The schedule knobs of csrc/net.hip (side streams for the up1 branch, hipGraph replay, the fused up-sample add, the max-pool
    fused into the producing GEMM) only move work between launches and streams: the logits are bit-identical under every setting.
    They are read once per process, hence one child process per setting.  34 crops: the 16x16 maps reach the 256-tile threshold of
    the fused Winograd tail, so the up-sample fusion is exercised on three levels."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    settings = [({}, 0), ({}, 1), ({"SUO_NET_SIDE_STREAMS": "2"}, 1), ({"SUO_NET_SIDE_STREAMS": "1"}, 0),
                ({"SUO_FUSE_UPSAMPLE": "0"}, 0), ({"SUO_FUSE_POOL": "0"}, 0)]
    digests = []
    for env_extra, graph in settings:
        env = dict(os.environ)
        env.update(env_extra)
        out = subprocess.run([sys.executable, os.path.join(here, "gpu_backbone_digest.py"), "34", str(graph)], env=env, capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [x for x in out.stdout.splitlines() if x.startswith("DIGEST")][-1]
        digests.append(line.split()[1])
    assert len(set(digests)) == 1, list(zip(settings, digests))


def test_forward_frames_equals_per_frame_calls(ops, state_dict):
    """Two frames in one call (crops of both frames batched) == two single-frame calls.  Not bit for bit: the
    tile / split-K configuration is chosen from the number of pixels in the launch, so accumulation order differs."""
    from suo_slam_amd.pkpnet import PkpNet
    rng = np.random.default_rng(33)
    imgs = (rng.uniform(0, 1, (2, 480, 640, 3)) * 255).astype(np.uint8)
    boxes = [np.array([[100, 80, 300, 290], [350.5, 100.25, 600, 400]], np.float32), np.array([[10, 200, 130, 330]], np.float32)]
    net = PkpNet(state_dict=state_dict, max_crops=4)
    both = net.forward_frames(imgs, boxes)
    off = 0
    for f in range(2):
        one = net(imgs[f], [torch.from_numpy(boxes[f])], None)
        n = len(boxes[f])
        lg = one["prob_logits"]
        assert float((both["prob_logits"][off:off + n] - lg).abs().max() / lg.abs().max()) < 1e-4
        for key in ("uv", "cov", "kp_mask"):
            assert float((both[key][off:off + n] - one[key]).abs().max()) < 1e-4, (f, key)
        off += n


def test_forward_matches_oracle_end_to_end(ops, state_dict):
    """uint8 frame + boxes -> uv/cov/kp_mask: HIP vs the CPU oracle on the same seeded inputs."""
    from oracle import cnn_oracle as O
    from suo_slam_amd.pkpnet import PkpNet
    rng = np.random.default_rng(21)
    img = (rng.uniform(0, 1, (480, 640, 3)) * 255).astype(np.uint8)
    boxes = np.array([[100, 80, 300, 290], [350.5, 100.25, 600, 400], [10, 200, 130, 330]], np.float32)
    net = PkpNet(state_dict=state_dict, max_crops=4)
    out = net(img, [torch.from_numpy(boxes)], None)
    ref = O.pkpnet_forward(img, boxes, None, state_dict)
    lg, lr = out["prob_logits"].cpu().numpy(), ref["prob_logits"].numpy()
    # uv / cov abs <= 1e-5 on the fp32 path (lib/models/hg.py:95-119 + pkpnet.py:106-118); logits 1e-5 of their range
    assert np.abs(lg - lr).max() / np.abs(lr).max() < 1e-5
    np.testing.assert_allclose(out["uv"].cpu().numpy(), ref["uv"].numpy(), atol=1e-5, rtol=0)
    np.testing.assert_allclose(out["cov"].cpu().numpy(), ref["cov"].numpy(), atol=1e-5, rtol=0)
    np.testing.assert_allclose(out["kp_mask"].cpu().numpy(), ref["kp_mask"].numpy(), atol=1e-5, rtol=0)
    # float CHW entry (the tensor PkpNet.forward receives in the reference) gives the same result
    out2 = net(torch.from_numpy(O.image_to_chw(img))[None], [torch.from_numpy(boxes)], None)
    assert torch.equal(out2["prob_logits"], out["prob_logits"])
    # priors given as zeros == priors omitted (pkpnet.py:95-97).  Two different first launches here -- the 44-channel stem on the fp32 pipe
    # with priors, the fused RoIAlign + 3-channel stem on the bf16 pipe without (csrc/stem_x3.hip) -- so: held to the oracle like `out`, and to
    # each other far inside that tolerance (bit-identical with SUO_STEM_X3=0: tests/test_gpu_stem.py runs that mode)
    out3 = net(img, [torch.from_numpy(boxes)], [torch.zeros(3, 41, 256, 256)])
    lg3 = out3["prob_logits"].cpu().numpy()
    assert np.abs(lg3 - lr).max() / np.abs(lr).max() < 1e-5
    assert np.abs(lg3 - lg).max() / np.abs(lr).max() < 3e-6


def test_render_priors_matches_host_restatement_and_reference_windows(ops):
    """make_prior_kp_input on the device (csrc/misc.hip) vs the host restatement (object_slam.make_prior_kp_input) and
    vs the paste windows recorded from the REFERENCE's utils.make_prior_kp_input (tests/golden/host_golden.npz):
    NDC->pixel, round-half-even, clipping, non-finite and masked keypoints."""
    import os
    from suo_slam_amd import object_slam as OS
    from suo_slam_amd.pkpnet import render_priors
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "host_golden.npz"))
    kp16, m16 = gold["prior_kp"], gold["prior_mask"]
    rng = np.random.default_rng(4)
    uv = np.zeros((3, 41, 2), np.float32)
    mask = np.zeros((3, 41), bool)
    uv[0, :16], mask[0, :16] = kp16, m16
    uv[1] = rng.uniform(-1.2, 1.2, (41, 2))
    mask[1] = rng.random(41) < 0.7
    uv[2] = rng.uniform(-1, 1, (41, 2))
    mask[2, ::3] = True
    uv[2, 5] = [np.inf, 0.0]
    out = render_priors(uv, mask).cpu().numpy()
    assert out.shape == (3, 41, 256, 256)
    for l in range(3):
        host = OS.make_prior_kp_input(uv[l], mask[l], (256, 256), ndc=True)
        assert np.abs(out[l] - host).max() <= 2.4e-7, l                      # <= 2 float32 ulp at 1.0 (exp in libm vs numpy)
        assert np.array_equal(out[l] != 0, host != 0)
    # the reference's windows: bounding rectangle and peak position of every stamped channel
    def rects(x):
        r = np.full((x.shape[0], 6), -1, np.int32)
        for c in range(x.shape[0]):
            ys, xs = np.nonzero(x[c])
            if len(ys):
                my, mx = np.unravel_index(np.argmax(x[c]), x[c].shape)
                r[c] = [ys.min(), ys.max() + 1, xs.min(), xs.max() + 1, my, mx]
        return r
    assert np.array_equal(rects(out[0, :16]), gold["prior_ndc_rect"])
    # values against the reference's make_prior_kp_input run over a cv2 stand-in WITHOUT OpenCV's border reflection (float16 fixture): equal to
    # float16 resolution everywhere inside the patch; on its outermost ring (values <= 1.2e-2) the reflection the product models -- and
    # tests/test_oracle_cnn.py::test_prior_stamp_equals_conv2d_of_an_impulse_with_opencvs_kernel pins to 1e-6 -- is the whole difference
    g = gold["prior_ndc"].astype(np.float32)
    d = np.abs(out[0, :16] - g)
    assert d[g > 1.2e-2].max() < 5e-4 and d.max() < 1.3e-2


def test_forward_with_prior_keypoints_equals_dense_priors(ops, state_dict):
    """The SLAM pass: handing the projected keypoints to suo_net_forward_prior_kp gives bit-identical outputs to
    handing the dense heat-maps (rendered by the same device code) to suo_net_forward, and matches host-rendered
    priors to rounding."""
    from suo_slam_amd import object_slam as OS
    from suo_slam_amd.pkpnet import PkpNet, render_priors
    rng = np.random.default_rng(9)
    img = (rng.uniform(0, 1, (480, 640, 3)) * 255).astype(np.uint8)
    boxes = np.array([[100, 80, 300, 290], [350.5, 100.25, 600, 400]], np.float32)
    uv = rng.uniform(-0.9, 0.9, (2, 41, 2)).astype(np.float32)
    mask = rng.random((2, 41)) < 0.4
    net = PkpNet(state_dict=state_dict, max_crops=2)
    a = net(img, [torch.from_numpy(boxes)], None, prior_uv=uv, prior_mask=mask)
    dense = render_priors(uv, mask)
    b = net(img, [torch.from_numpy(boxes)], [dense])
    for k in ("prob_logits", "uv", "cov", "kp_mask"):
        assert torch.equal(a[k], b[k]), k
    host = np.stack([OS.make_prior_kp_input(uv[l], mask[l], (256, 256), ndc=True) for l in range(2)])
    c = net(img, [torch.from_numpy(boxes)], [torch.from_numpy(host)])
    lg = a["prob_logits"].cpu().numpy()
    assert np.abs(lg - c["prob_logits"].cpu().numpy()).max() / np.abs(lg).max() < 1e-5
    # priors do change the answer (the path is not silently ignoring them)
    z = net(img, [torch.from_numpy(boxes)], None)
    assert not torch.equal(z["prob_logits"], a["prob_logits"])
    net.close()


@pytest.mark.parametrize("reps", [4, 8])
def test_bench_launch_shape_frames_batched_per_call(ops, state_dict, reps):
    """The launch shapes of bench.py (BASELINE configs[1] batched 32 frames per call = 256 crops: 3.2 GB staged input, 43 GB
    workspace; 16 frames = 128 crops for the shorter runs) through size-independent properties: the frames repeat frames 0..3,
    so their crops must come out BIT-identical to their originals (every crop is computed by the same code path wherever it sits
    in the launch -- this catches index overflow and tile-walk errors at full size), and the first frame agrees with a call of
    its own."""
    from suo_slam_amd.pkpnet import PkpNet
    rng = np.random.default_rng(77)
    base = (rng.uniform(0, 1, (4, 480, 640, 3)) * 255).astype(np.uint8)
    imgs = np.concatenate([base] * reps)
    bx4 = [np.column_stack([x1 := rng.uniform(0, 400, 8), y1 := rng.uniform(0, 240, 8), x1 + rng.uniform(60, 240, 8), y1 + rng.uniform(60, 240, 8)]).astype(np.float32)
           for _ in range(4)]
    boxes = bx4 * reps
    net = PkpNet(state_dict=state_dict, max_crops=32 * reps)
    out = net.forward_frames(imgs, boxes)
    assert out["prob_logits"].shape == (32 * reps, 41, 64, 64)
    assert torch.isfinite(out["prob_logits"]).all()
    for key in ("prob_logits", "uv", "cov", "kp_mask"):
        v = out[key].reshape(reps, 32, *out[key].shape[1:])
        for rep in range(1, reps):
            assert torch.equal(v[rep], v[0]), (key, rep)
    one = net(base[0], [torch.from_numpy(bx4[0])], None)
    lg = one["prob_logits"]
    # (both calls are held to the reference at 1e-5 elsewhere -- at this launch shape directly in tests/test_gpu_golden_wide.py; observed here ~1e-6)
    assert float((out["prob_logits"][:8] - lg).abs().max() / lg.abs().max()) < 1e-5
    net.close()


def test_fused_kernels_on_random_shapes(ops):
    """39 random shapes (the sweep that used to live in tools/): the max-pool GEMM against GEMM + pool, bit for bit, over random
    K1 / K2 / N / residual / prologue / map sizes; the Winograd 3x3 against fp64 (5e-6) and the fused Winograd tail (+ the
    up-sample variant) against its separate launches over ragged maps (H, W not multiples of the 8 x 16 tile)."""
    rng = np.random.default_rng(123)
    for it in range(25):
        W = int(rng.choice([64, 128, 192]))
        H = int(2 * rng.integers(1, 9))
        L = int(rng.integers(1, 6))
        while L * H * W <= 4096:
            L += 1
        K1 = int(32 * rng.integers(1, 9))
        K2 = int(rng.choice([0, 32, 64, 128]))
        N = int(rng.choice([128, 256]))
        res, relu, full = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        M = L * H * W
        a1 = torch.from_numpy(rng.standard_normal((M, K1)).astype(np.float32)).cuda()
        a2 = torch.from_numpy(rng.standard_normal((M, K2)).astype(np.float32)).cuda() if K2 else None
        r = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda() if res else None
        w1 = (rng.standard_normal((N, K1)) / np.sqrt(K1)).astype(np.float32)
        w2 = (rng.standard_normal((N, K2)) / np.sqrt(K2)).astype(np.float32) if K2 else None
        b = rng.standard_normal(N).astype(np.float32)
        pro = (rng.uniform(0.5, 1.5, K1).astype(np.float32), rng.standard_normal(K1).astype(np.float32) * 0.2) if relu else None
        want = ops.conv1x1(a1, w1, b, pro=pro, a2=a2, w2=w2, res=r, relu=relu)
        wp = want.view(L, H // 2, 2, W // 2, 2, N).amax(dim=(2, 4)).reshape(M // 4, N)
        got_full, got_pool = ops.conv1x1_pool(a1, w1, b, H, W, pro=pro, a2=a2, w2=w2, res=r, relu=relu, want_full=full)
        assert torch.equal(got_pool, wp) and (not full or torch.equal(got_full, want)), ("pool", L, H, W, K1, K2, N, res, relu, full)
    for it in range(14):
        H, W, L = int(rng.integers(8, 70)), int(rng.integers(16, 70)), int(rng.integers(1, 12))
        x = torch.from_numpy(rng.standard_normal((L, H, W, 128)).astype(np.float32)).cuda()
        skip = torch.from_numpy(rng.standard_normal((L, H, W, 256)).astype(np.float32)).cuda()
        w2 = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
        b2 = rng.standard_normal(128).astype(np.float32) * 0.3
        w3 = (rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
        b3 = rng.standard_normal(256).astype(np.float32)
        mid = ops.conv3x3_wino(x, w2, b2, relu=True)
        ref = torch.nn.functional.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), torch.from_numpy(w2).double(),
                                                                  torch.from_numpy(b2).double(), padding=1)).permute(0, 2, 3, 1)
        e = float((mid.cpu().double() - ref).abs().max() / ref.abs().max())
        assert e < 5e-6, ("wino", L, H, W, e)
        want = ops.conv1x1(mid.reshape(-1, 128), w3, b3, res=skip.reshape(-1, 256)).reshape(L, H, W, 256)
        got = ops.conv3x3_wino_conv1x1_skip(x, w2, b2, w3, b3, skip)
        # (<= 4096 pixels: suo_conv1x1 is the split-K kernel, another summation order -- compare by value there)
        same = torch.equal(got, want) if L * H * W > 4096 else bool(((got - want).abs().max() < 1e-5 * want.abs().max()).item())
        assert same, ("wino tail", L, H, W)
        if H % 2 == 0 and W % 2 == 0:
            low = torch.from_numpy(rng.standard_normal((L, H // 2, W // 2, 256)).astype(np.float32)).cuda()
            gu = ops.conv3x3_wino_conv1x1_skip_up(x, w2, b2, w3, b3, skip, low)
            assert torch.equal(gu, got + low.repeat_interleave(2, 1).repeat_interleave(2, 2)), ("wino tail up", L, H, W)


@pytest.mark.parametrize("M,K1,K2,N,res,relu", [(4096, 256, 0, 256, False, 1), (128 * 9 + 77, 128, 128, 256, False, 0), (5000, 256, 0, 256, True, 0),
                                                 (1000, 64, 64, 128, False, 0), (4096, 128, 128, 256, True, 1), (3001, 64, 64, 64, True, 1), (700, 128, 0, 64, False, 0)])
def test_bf16x3_gemm_general_form(ops, M, K1, K2, N, res, relu):
    """suo_conv1x1_bf16x3_ex (csrc/gemm_bf16x3.hip; what the network launches for lin, the re-injection and conv3 + conv4 at 64x64): N = 256 as two
    column tiles, a second K segment on a second operand, the residual operand; within 5e-6 of fp64 and of the fp32-pipe kernel on the same inputs."""
    from suo_slam_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(M + K1 + N)
    K = K1 + K2
    w = (rng.standard_normal((N, K)) / 16.0).astype(np.float32)
    w3 = np.empty(3 * N * K, np.uint16)
    _lib.check(lib.suo_pack_gemm_weight_bf16x3(w.ctypes.data, N, K, w3.ctypes.data))
    w3d = torch.from_numpy(w3.view(np.int16)).cuda()
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    a1 = rng.standard_normal((M, K1)).astype(np.float32) * 2
    a2 = rng.standard_normal((M, K2)).astype(np.float32) if K2 else None
    r = rng.standard_normal((M, N)).astype(np.float32) if res else None
    a1d, bd = ops.dev(a1), ops.dev(b)
    a2d = ops.dev(a2) if K2 else None
    rd = ops.dev(r) if res else None
    out = torch.full((M + 3, N), -5.0, device="cuda")
    _lib.check(lib.suo_conv1x1_bf16x3_ex(ops.P(a1d), K1, K1, None, None, ops.P(a2d), K2, K2, ops.P(w3d), ops.P(bd), ops.P(rd), N, ops.P(out), N, M, N, relu, ops.S()))
    torch.cuda.synchronize()
    ref = a1.astype(np.float64) @ w[:, :K1].astype(np.float64).T + b
    if K2:
        ref = ref + a2.astype(np.float64) @ w[:, K1:].astype(np.float64).T
    if res:
        ref = ref + r
    if relu:
        ref = np.maximum(ref, 0)
    got = out.cpu().numpy()
    assert np.abs(got[:M] - ref).max() < 5e-6 * np.abs(ref).max()
    assert (got[M:] == -5.0).all()
    f32 = _conv1x1_f32_sequential(ops, a1d, w[:, :K1], b, a2=a2d, w2=w[:, K1:] if K2 else None, res=rd, relu=bool(relu))
    assert np.abs(got[:M] - f32).max() < 5e-6 * np.abs(ref).max()
    assert np.abs(got[:M] - ref).max() <= 2.0 * np.abs(f32 - ref).max() + 1e-7 * np.abs(ref).max()      # never worse than the fp32-pipe kernel


@pytest.mark.parametrize("L,H,W,K1,K2,N,res,want_full", [(3, 64, 64, 256, 0, 256, True, True), (2, 64, 64, 128, 128, 256, False, True), (1, 128, 128, 64, 64, 128, False, False),
                                                          (2, 16, 64, 256, 0, 128, False, True)])
def test_bf16x3_gemm_with_the_pool_in_its_epilogue(ops, L, H, W, K1, K2, N, res, want_full):
    """suo_conv1x1_bf16x3_pool: tiles of two image rows x 64 columns, the 2x2 maximum taken in the epilogue (lane exchange + LDS) -- the full result (when asked
    for) and the pooled one against fp64, the pooled one exactly the 2x2 maximum of the full one, nothing written elsewhere."""
    from suo_slam_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(H + K1 + N)
    M, K = L * H * W, K1 + K2
    w = (rng.standard_normal((N, K)) / 16.0).astype(np.float32)
    w3 = np.empty(3 * N * K, np.uint16)
    _lib.check(lib.suo_pack_gemm_weight_bf16x3(w.ctypes.data, N, K, w3.ctypes.data))
    w3d = torch.from_numpy(w3.view(np.int16)).cuda()
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    a1 = rng.standard_normal((M, K1)).astype(np.float32)
    a2 = rng.standard_normal((M, K2)).astype(np.float32) if K2 else None
    r = rng.standard_normal((M, N)).astype(np.float32) if res else None
    a1d, bd = ops.dev(a1), ops.dev(b)
    a2d = ops.dev(a2) if K2 else None
    rd = ops.dev(r) if res else None
    out = torch.full((M + 2, N), -5.0, device="cuda") if want_full else None
    pooled = torch.full((M // 4 + 2, N), -7.0, device="cuda")
    _lib.check(lib.suo_conv1x1_bf16x3_pool(ops.P(a1d), K1, K1, None, None, ops.P(a2d), K2, K2, ops.P(w3d), ops.P(bd), ops.P(rd), N, ops.P(out), N, M, N, 0,
                                           H, W, ops.P(pooled), ops.S()))
    torch.cuda.synchronize()
    ref = a1.astype(np.float64) @ w[:, :K1].astype(np.float64).T + b
    if K2:
        ref = ref + a2.astype(np.float64) @ w[:, K1:].astype(np.float64).T
    if res:
        ref = ref + r
    refp = ref.reshape(L, H // 2, 2, W // 2, 2, N).max(axis=(2, 4)).reshape(M // 4, N)
    gp = pooled.cpu().numpy()
    assert np.abs(gp[:M // 4] - refp).max() < 5e-6 * np.abs(ref).max()
    assert (gp[M // 4:] == -7.0).all()
    f32 = _conv1x1_f32_sequential(ops, a1d, w[:, :K1], b, a2=a2d, w2=w[:, K1:] if K2 else None, res=rd)      # the fp32-pipe kernel on the same inputs
    f32p = f32.reshape(L, H // 2, 2, W // 2, 2, N).max(axis=(2, 4)).reshape(M // 4, N)
    assert np.abs(gp[:M // 4] - refp).max() <= 2.0 * np.abs(f32p - refp).max() + 1e-7 * np.abs(ref).max()
    if want_full:
        go = out.cpu().numpy()
        assert np.abs(go[:M] - ref).max() < 5e-6 * np.abs(ref).max()
        assert np.abs(go[:M] - ref).max() <= 2.0 * np.abs(f32 - ref).max() + 1e-7 * np.abs(ref).max()
        assert (go[M:] == -5.0).all()
        np.testing.assert_array_equal(gp[:M // 4], go[:M].reshape(L, H // 2, 2, W // 2, 2, N).max(axis=(2, 4)).reshape(M // 4, N))


@pytest.mark.parametrize("K,N", [(256, 128), (128, 128), (64, 128), (64, 64), (128, 64)])
def test_bf16x3_gemm_is_fp32_accurate(ops, K, N):
    """csrc/gemm_bf16x3.hip (what the network launches for conv1 of its Residual blocks at >= 32768 pixels): the 1x1 convolution on the bf16 matrix pipe with both operands
    split into three bf16 terms and 6 of 9 cross products accumulated in fp32 must be as accurate as the fp32 MFMA kernel: same
    bound against fp64 (rel 5e-6 of the output range, the bound every fp32 conv kernel is held to), with and without the BN + ReLU
    prologue, ragged M (rows beyond the last full 128-row tile), never worse than 2x the fp32-pipe kernel's own error, and the split of the weights exact (w0 + w1 + w2 == w).
    N = 64: the 64-column tiles of the first Residual blocks' 64-channel 1x1 convolutions (r1 / r4 conv1)."""
    import ctypes as C
    from suo_slam_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(77)
    w = (rng.standard_normal((N, K)) / 16.0).astype(np.float32)
    w3 = np.empty(3 * N * K, np.uint16)
    _lib.check(lib.suo_pack_gemm_weight_bf16x3(w.ctypes.data, N, K, w3.ctypes.data))
    planes = (w3.reshape(K // 16, N // 32, 3, 2, 32, 8).astype(np.uint32) << 16).view(np.float32)      # [ks][nb][plane][k half][n % 32][8]: B-operand order
    back = planes.astype(np.float64).sum(2).transpose(1, 3, 0, 2, 4).reshape(N, K)
    assert np.array_equal(back.astype(np.float32), w) and np.abs(back - w).max() == 0.0                   # three bf16 terms carry the 24 bits: exact (csrc/bf16x3.h)
    w3d = torch.from_numpy(w3.view(np.int16)).cuda()
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    bd = ops.dev(b)
    for M, pro, relu in ((4096, True, 1), (1000, False, 0), (128 * 513 + 5, True, 1)):
        a = rng.standard_normal((M, K)).astype(np.float32) * 3
        sc, sh = rng.uniform(0.5, 1.5, K).astype(np.float32), (rng.standard_normal(K) * 0.1).astype(np.float32)
        ad, scd, shd = ops.dev(a), ops.dev(sc), ops.dev(sh)
        out = torch.full((M + 7, N), -5.0, device="cuda")
        _lib.check(lib.suo_conv1x1_bf16x3(ops.P(ad), K, K, ops.P(scd) if pro else None, ops.P(shd) if pro else None, ops.P(w3d), ops.P(bd),
                                          ops.P(out), N, M, N, relu, ops.S()))
        torch.cuda.synchronize()
        x = np.maximum(a * sc + sh, 0).astype(np.float32).astype(np.float64) if pro else a.astype(np.float64)
        ref = x @ w.astype(np.float64).T + b
        if relu:
            ref = np.maximum(ref, 0)
        got = out.cpu().numpy()
        assert np.abs(got[:M] - ref).max() < 5e-6 * np.abs(ref).max(), (M, np.abs(got[:M] - ref).max() / np.abs(ref).max())
        assert (got[M:] == -5.0).all()                                                                  # nothing written past M
        f32 = _conv1x1_f32_sequential(ops, ad, w, b, pro=(sc, sh) if pro else None, relu=bool(relu))      # the fp32-pipe kernel, same inputs
        assert np.abs(got[:M] - ref).max() <= 2.0 * np.abs(f32 - ref).max() + 1e-7 * np.abs(ref).max()
    from suo_slam_amd._lib import SuoError
    with pytest.raises(SuoError):
        _lib.check(lib.suo_conv1x1_bf16x3(ops.P(ad), K, 48, None, None, ops.P(w3d), ops.P(bd), ops.P(out), N, 128, N, 0, ops.S()))

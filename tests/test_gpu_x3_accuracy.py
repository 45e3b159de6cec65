"""Per-element forward-error gates of the bf16x3 kernels (the network's default matrix-pipe form, csrc/bf16x3.h).

The other kernel tests bound the LARGEST error by the output RANGE; a split that lost its small terms only where they sit next to large
ones, or that pulled every product toward zero, would pass those.  Here every output element i is held to a bound that scales with ITS
OWN dot product,
        |out_i - fp64_i|  <=  c(K) * 2^-24 * sum_k |x_ik| |w_k| ,
on inputs built to expose a bad split: (i) rows that cancel almost completely, (ii) rows whose terms span 2^-20 .. 2^20 inside ONE dot
product, (iii) all-positive (post-ReLU) activations against one-signed weights, where a truncating split shows up as a signed MEAN
error.  The fp32-pipe kernel of the same operator runs on the same inputs: the bf16x3 kernel may not be worse than it.

c(K): both pipes accumulate K terms one after the other in ONE fp32 accumulator (v_mfma_f32_32x32x2_f32 is an fmaf chain; the bf16
form rounds once per MFMA, 6 K / 16 times), so on one-signed data the rounding errors random-walk: measured standard deviation 2.3-4.1
units of 2^-24 * sum|x||w| at K = 128 ... 1152, worst of ~10^6 outputs 12-19 (4.6 sigma), for EITHER pipe.  c(K) = 2 sqrt(K) covers
that with < 2x to spare at K = 128; VERDICT r3 proposed c = 8, which the fp32-MFMA kernels themselves exceed (19.0 at K = 256); the
deterministic bound (c = K) would gate nothing.  On sign-mixed data the observed error is 3-5x below the one-signed one.

Signed mean on one-signed data (BIAS_MAX): the fp32 pipe has none (|mean| < 0.01).  The bf16 matrix pipe truncates when it aligns its
16 products with the accumulator, which shows as -0.11 ... -0.18 (fused tail, two products deep: -0.30) with the round-to-nearest
operand split the product uses; the truncating split of rounds 1-3 sat at -0.78 ... -0.81 (-1.50) -- measured side by side on one
box, profiles/r04_bias_ab_split.txt (tools/bias_ab.sh).  The gate (0.45) passes the first and fails the second."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

U = 2.0 ** -24
MAX_VS_F32 = 1.5          # worst element of the bf16x3 kernel vs the worst element of the fp32-pipe kernel, same inputs
BIAS_MAX = 0.45           # |mean signed error| on one-signed data, units of 2^-24 sum|x||w| (see the module docstring: measured 0.11-0.30)


@pytest.fixture(scope="module")
def ops():
    from suo_slam_amd import _lib
    _lib.require_gpu()
    from tests import hipops
    return hipops


def _gemm_x3(ops, a, w, b):
    from suo_slam_amd import _lib
    lib = _lib.lib()
    N, K = w.shape
    M = a.shape[0]
    w3 = np.empty(3 * N * K, np.uint16)
    _lib.check(lib.suo_pack_gemm_weight_bf16x3(np.ascontiguousarray(w).ctypes.data, N, K, w3.ctypes.data))
    w3d = torch.from_numpy(w3.view(np.int16)).cuda()
    ad, bd = ops.dev(a), ops.dev(b)
    out = torch.empty((M, N), device="cuda")
    _lib.check(lib.suo_conv1x1_bf16x3(ops.P(ad), K, K, None, None, ops.P(w3d), ops.P(bd), ops.P(out), N, M, N, 0, ops.S()))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _gemm_cases(rng, M, K, N):
    """(name, a [M,K], w [N,K]) -- see the module docstring"""
    cases = []
    # (i) heavy cancellation: consecutive k pairs carry +v, -v (1 + d) with equal weights -- the row sums to ~d of its terms
    v = rng.standard_normal((M, K // 2)).astype(np.float32)
    d = (rng.standard_normal((M, K // 2)) * 1e-4).astype(np.float32)
    a = np.empty((M, K), np.float32)
    a[:, 0::2] = v
    a[:, 1::2] = -v * (1 + d)
    w = np.repeat((rng.standard_normal((N, K // 2)) / 16).astype(np.float32), 2, axis=1)
    cases.append(("cancellation", a, w))
    # (ii) 2^-20 .. 2^20 inside one dot product (activations; the weights keep their usual scale)
    e = rng.integers(-20, 21, (M, K))
    a = (rng.standard_normal((M, K)) * np.exp2(e)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / 16).astype(np.float32)
    cases.append(("mixed_magnitudes", a, w))
    # (ii') the same in the weights
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / 16 * np.exp2(rng.integers(-20, 21, (N, K)))).astype(np.float32)
    cases.append(("mixed_weights", a, w))
    # (iii) post-ReLU activations x one-signed weights: every product positive
    a = np.abs(rng.standard_normal((M, K))).astype(np.float32)
    w = np.abs(rng.standard_normal((N, K)) / 16).astype(np.float32)
    cases.append(("one_signed", a, w))
    # and the ordinary case
    cases.append(("random", rng.standard_normal((M, K)).astype(np.float32), (rng.standard_normal((N, K)) / 16).astype(np.float32)))
    return cases


@pytest.mark.parametrize("K", [256, 128])
def test_gemm_bf16x3_forward_error_per_element(ops, K):
    rng = np.random.default_rng(100 + K)
    M, N = 8192, 128
    c = 2.0 * np.sqrt(K)
    b = np.zeros(N, np.float32)
    report = []
    for name, a, w in _gemm_cases(rng, M, K, N):
        ref = a.astype(np.float64) @ w.astype(np.float64).T
        S = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
        x3 = _gemm_x3(ops, a, w, b).astype(np.float64)
        f32 = ops.conv1x1(ops.dev(a), w, b).cpu().numpy().astype(np.float64)
        ex, ef = (x3 - ref) / (U * S), (f32 - ref) / (U * S)
        report.append((name, np.abs(ex).max(), np.abs(ef).max(), ex.mean(), ef.mean(), ex.std(), ef.std()))
    print(f"\nGEMM K={K}: case, max|err| bf16x3 / fp32 pipe, mean signed err bf16x3 / fp32 pipe, std bf16x3 / fp32 pipe   [units of 2^-24 sum|x||w|]")
    for r in report:
        print("   %-18s %7.3f %7.3f   %+8.4f %+8.4f   %7.4f %7.4f" % r)
    for name, mx, mf, bx, bf, sx, sf in report:
        assert mx <= c, (name, mx)
        assert mx <= MAX_VS_F32 * mf + 0.5, (name, mx, mf)                  # against the fp32 pipe on the same inputs
        if name == "one_signed":                                            # signed mean over 10^6 outputs of one-signed products
            assert abs(bx) <= BIAS_MAX, (name, bx)


def _wino_cases(rng, L, H, W, C):
    cases = []
    # (i) cancellation across channel pairs
    v = rng.standard_normal((L, C // 2, H, W)).astype(np.float32)
    d = (rng.standard_normal((L, C // 2, H, W)) * 1e-4).astype(np.float32)
    x = np.empty((L, C, H, W), np.float32)
    x[:, 0::2] = v
    x[:, 1::2] = -v * (1 + d)
    w = np.repeat((rng.standard_normal((C, C // 2, 3, 3)) / np.sqrt(9 * C)).astype(np.float32), 2, axis=1)
    cases.append(("cancellation", x, w))
    # (ii) channels of one pixel neighbourhood span 2^-20 .. 2^20 (per channel, so that the Winograd input transform -- sums of four
    # neighbours of ONE channel -- stays well conditioned: the operator's own error, not the transform's, is what is compared)
    x = (rng.standard_normal((L, C, H, W)) * np.exp2(rng.integers(-20, 21, (1, C, 1, 1)))).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    cases.append(("mixed_magnitudes", x, w))
    # (iii) one-signed
    x = np.abs(rng.standard_normal((L, C, H, W))).astype(np.float32)
    w = np.abs(rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    cases.append(("one_signed", x, w))
    cases.append(("random", rng.standard_normal((L, C, H, W)).astype(np.float32), (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)))
    return cases


@pytest.mark.parametrize("C", [128, 64])
def test_winograd_bf16x3_forward_error_per_element(ops, C):
    """The same gate for the Winograd 3x3 convolution.  K = 9 C terms per output; the Winograd form itself (either pipe) carries the
    input / output transforms' roundings on top of the accumulation, so its constant is larger than a direct convolution's: the bound is
    stated against the fp32-pipe Winograd kernel (never worse than 1.5x of it) and absolutely as c = 2 sqrt(9 C)."""
    rng = np.random.default_rng(300 + C)
    L, H, W = 4, 32, 32
    c = 2.0 * np.sqrt(9 * C)
    b = np.zeros(C, np.float32)
    report = []
    for name, x, w in _wino_cases(rng, L, H, W, C):
        xt, wt = torch.from_numpy(x).double(), torch.from_numpy(w).double()
        ref = F.conv2d(xt, wt, padding=1).numpy()
        S = F.conv2d(xt.abs(), wt.abs(), padding=1).numpy()
        x3 = ops.nchw(ops.conv3x3_wino_x3(ops.nhwc(x), w, b)).astype(np.float64)
        f32 = ops.nchw(ops.conv3x3_wino(ops.nhwc(x), w, b)).astype(np.float64)
        ex, ef = (x3 - ref) / (U * S), (f32 - ref) / (U * S)
        report.append((name, np.abs(ex).max(), np.abs(ef).max(), ex.mean(), ef.mean(), ex.std(), ef.std()))
    print(f"\nWinograd 3x3 C={C}: case, max|err| bf16x3 / fp32 pipe, mean signed err bf16x3 / fp32 pipe, std bf16x3 / fp32 pipe   [units of 2^-24 sum|x||w|]")
    for r in report:
        print("   %-18s %7.3f %7.3f   %+8.4f %+8.4f   %7.4f %7.4f" % r)
    for name, mx, mf, bx, bf, sx, sf in report:
        assert mx <= c, (name, mx)
        assert mx <= MAX_VS_F32 * mf + 0.5, (name, mx, mf)
        if name == "one_signed":
            assert abs(bx) <= BIAS_MAX, (name, bx)


def test_fused_tail_bf16x3_forward_error_per_element(ops):
    """conv2 (3x3 Winograd) -> ReLU -> conv3 (1x1) + skip in one launch, both products on the bf16 pipe: per element against fp64 with the
    bound of the LAST product (K = 128 over the conv2 tile the kernel itself produced is not observable from outside, so the reference
    is the fp64 chain and the scale sum |relu(conv2)| |w3| + |skip|), on one-signed data, against the fp32-pipe fused kernel."""
    rng = np.random.default_rng(9)
    L, H, W = 6, 32, 32
    x = np.abs(rng.standard_normal((L, H, W, 128))).astype(np.float32)
    skip = np.abs(rng.standard_normal((L, H, W, 256))).astype(np.float32)
    w2 = np.abs(rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    b2 = np.abs(rng.standard_normal(128) * 0.3).astype(np.float32)
    w3 = np.abs(rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
    b3 = np.abs(rng.standard_normal(256)).astype(np.float32)
    xd, sd = torch.from_numpy(x).cuda(), torch.from_numpy(skip).cuda()
    x3 = ops.conv3x3_wino_x3_conv1x1_skip_up(xd, w2, b2, w3, b3, sd, None, tail_x3=True).cpu().numpy().astype(np.float64)
    f32 = ops.conv3x3_wino_conv1x1_skip(xd, w2, b2, w3, b3, sd).cpu().numpy().astype(np.float64)
    xm = torch.from_numpy(x).permute(0, 3, 1, 2).double()
    m = F.relu(F.conv2d(xm, torch.from_numpy(w2).double(), torch.from_numpy(b2).double(), padding=1))
    ref = (F.conv2d(m, torch.from_numpy(w3).double()[:, :, None, None], torch.from_numpy(b3).double())).permute(0, 2, 3, 1).numpy() + skip
    S = ref                                                                 # every term is positive: the sum of magnitudes IS the result
    ex, ef = (x3 - ref) / (U * S), (f32 - ref) / (U * S)
    print("\nfused tail, one-signed: max|err| bf16x3 %.3f fp32 pipe %.3f, mean signed %+.4f %+.4f   [units of 2^-24 sum|x||w|]"
          % (np.abs(ex).max(), np.abs(ef).max(), ex.mean(), ef.mean()))
    assert np.abs(ex).max() <= 2.0 * np.sqrt(9 * 128 + 128)
    assert np.abs(ex).max() <= MAX_VS_F32 * np.abs(ef).max() + 0.5
    assert abs(ex.mean()) <= BIAS_MAX

"""The two-term fp16 form of the matrix-pipe kernels (suo_slam_amd/csrc/f16x2.h; the network's default since round 5): three fp16 MFMAs per product block
where the bf16x3 form issues six.

(1) Accuracy: the SAME per-element gates tests/test_gpu_x3_accuracy.py holds the bf16x3 kernels to, on the same adversarial inputs (imported from there):
    |out_i - fp64_i| <= 2 sqrt(K) * 2^-24 * sum_k |x_ik||w_k|, never worse than 1.5x the fp32-pipe kernel (+0.5), |signed mean| <= 0.45 on one-signed data.
(2) Range guard: fp16 ends at 65504 and activations enter times 16, so an input of magnitude >= 4094 (3x3 Winograd: >= 1023.5, its guard bounds
    |B^T d B| by 4 max |d|) cannot be computed in this form.  The kernels must raise the caller's flag -- and must NOT raise it just below the limit; cases
    of the x3 suite that span 2^-20 .. 2^20 are expected to raise it (their results are then not compared: the network re-issues such a call on bf16x3).
(3) Network: the default network runs this form (suo_net_get_pipe == 2) and agrees with the bf16x3 network to the network's tolerance; a network whose
    activations leave the range falls back by itself and then returns bit for bit what a bf16x3 network returns."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_x3_accuracy import BIAS_MAX, MAX_VS_F32, U, _gemm_cases, _wino_cases

pytestmark = pytest.mark.gpu

# Against the fp32 pipe on the same inputs: never worse than 1.5x its worst element, plus REPR.  The additive term is the one thing this form has that the
# fp32 pipe has not: an operand is held to 2^-22 relative in the worst case (|x - hi - lo| <= 2^-11 of a residual of <= 2^-11 |x|), i.e. up to 4 units of
# 2^-24 |x w| when ONE product dominates a dot product.  It only shows where the accumulation noise is absent -- the "cancellation" case, fp32 pipe 0.2-0.3
# units: measured 0.6 (K = 256) and 1.3 (K = 128), standard deviation 0.09-0.13.  Everywhere else this form is below the fp32 pipe outright.
REPR = 1.5
LIMIT_1X1 = 65504.0 / 16.0
LIMIT_3X3 = 65504.0 / 64.0


@pytest.fixture(scope="module")
def ops():
    from suo_slam_amd import _lib
    _lib.require_gpu()
    from tests import hipops
    return hipops


@pytest.mark.parametrize("K", [256, 128])
def test_gemm_f16x2_forward_error_per_element(ops, K):
    rng = np.random.default_rng(100 + K)
    M, N = 8192, 128
    c = 2.0 * np.sqrt(K)
    b = np.zeros(N, np.float32)
    cases = _gemm_cases(rng, M, K, N)
    # the in-range relative of "mixed_magnitudes": 2^-20 .. 2^6 inside one dot product
    a = (rng.standard_normal((M, K)) * np.exp2(rng.integers(-20, 7, (M, K)))).astype(np.float32)
    cases.append(("mixed_in_range", a, (rng.standard_normal((N, K)) / 16).astype(np.float32)))
    report = []
    for name, a, w in cases:
        out, flag = ops.conv1x1_f16x2(ops.dev(a), w, b)
        over = bool(np.abs(a).max() >= LIMIT_1X1)
        assert flag == int(over), (name, flag, np.abs(a).max())
        if over:
            assert name == "mixed_magnitudes"
            continue
        ref = a.astype(np.float64) @ w.astype(np.float64).T
        S = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
        h = out.cpu().numpy().astype(np.float64)
        f32 = ops.conv1x1(ops.dev(a), w, b).cpu().numpy().astype(np.float64)
        assert np.isfinite(h).all()
        ex, ef = (h - ref) / (U * S), (f32 - ref) / (U * S)
        report.append((name, np.abs(ex).max(), np.abs(ef).max(), ex.mean(), ef.mean(), ex.std(), ef.std()))
    print(f"\nGEMM K={K}: case, max|err| f16x2 / fp32 pipe, mean signed err f16x2 / fp32 pipe, std f16x2 / fp32 pipe   [units of 2^-24 sum|x||w|]")
    for r in report:
        print("   %-18s %7.3f %7.3f   %+8.4f %+8.4f   %7.4f %7.4f" % r)
    assert len(report) == len(cases) - 1
    for name, mx, mf, bx, bf, sx, sf in report:
        assert mx <= c, (name, mx)
        assert mx <= MAX_VS_F32 * mf + REPR, (name, mx, mf)
        if name == "one_signed":
            assert abs(bx) <= BIAS_MAX, (name, bx)


def test_gemm_f16x2_variants_match_fp64(ops):
    """Every operand path of the kernel (BN + ReLU prologue, second K segment, residual operand, ReLU, fused 2x2 max-pool, 64-row tiles, 64 / 256 columns)
    against fp64 with the per-element gate; the pooled output is exactly the max-pool of the un-pooled one."""
    rng = np.random.default_rng(7)
    K = 256

    def gate(out, ref, S, Kt, what):
        e = np.abs(out.astype(np.float64) - ref) / (U * S)
        assert e.max() <= 2.0 * np.sqrt(Kt), (what, e.max())

    # prologue + relu, N = 128, M not a multiple of the tile
    M = 128 * 70 + 37
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((128, K)) / 16).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, K).astype(np.float32), (rng.standard_normal(K) * 0.3).astype(np.float32)
    out, flag = ops.conv1x1_f16x2(ops.dev(a), w, b, pro=(sc, sh), relu=True)
    x = np.maximum(a.astype(np.float64) * sc + sh, 0)
    pre = x @ w.astype(np.float64).T + b
    assert flag == 0
    gate(out.cpu().numpy(), np.maximum(pre, 0), np.abs(x) @ np.abs(w).astype(np.float64).T + np.abs(b), K, "prologue")
    # 64-row tiles (few tiles), N = 256, residual
    M = 64 * 9
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((256, K)) / 16).astype(np.float32)
    b = rng.standard_normal(256).astype(np.float32)
    r = rng.standard_normal((M, 256)).astype(np.float32)
    out, flag = ops.conv1x1_f16x2(ops.dev(a), w, b, res=ops.dev(r))
    assert flag == 0
    gate(out.cpu().numpy(), a.astype(np.float64) @ w.astype(np.float64).T + b + r, np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b) + np.abs(r), K, "residual")
    # second K segment (conv3 + conv4) with the fused pool: 2 crops of 16 x 64 pixels
    Hh, Ww, L = 16, 64, 2
    M = L * Hh * Ww
    a1 = rng.standard_normal((M, 128)).astype(np.float32)
    a2 = rng.standard_normal((M, 64)).astype(np.float32)
    w1 = (rng.standard_normal((128, 128)) / 12).astype(np.float32)
    w2 = (rng.standard_normal((128, 64)) / 8).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    (out, pooled), flag = ops.conv1x1_f16x2(ops.dev(a1), w1, b, a2=ops.dev(a2), w2=w2, pool_hw=(Hh, Ww))
    assert flag == 0
    ref = a1.astype(np.float64) @ w1.astype(np.float64).T + a2.astype(np.float64) @ w2.astype(np.float64).T + b
    S = np.abs(a1).astype(np.float64) @ np.abs(w1).astype(np.float64).T + np.abs(a2).astype(np.float64) @ np.abs(w2).astype(np.float64).T + np.abs(b)
    gate(out.cpu().numpy(), ref, S, 192, "dual")
    o = out.reshape(L, Hh, Ww, 128).permute(0, 3, 1, 2)
    assert torch.equal(pooled.reshape(L, Hh // 2, Ww // 2, 128), F.max_pool2d(o, 2, 2).permute(0, 2, 3, 1))
    # 64 output columns
    w = (rng.standard_normal((64, K)) / 16).astype(np.float32)
    a = rng.standard_normal((1000, K)).astype(np.float32)
    out, flag = ops.conv1x1_f16x2(ops.dev(a), w, np.zeros(64, np.float32))
    assert flag == 0
    gate(out.cpu().numpy(), a.astype(np.float64) @ w.astype(np.float64).T, np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T, K, "n64")


def test_gemm_f16x2_weight_rows_of_any_magnitude(ops):
    """Rows of the weight matrix 2^-30 ... 2^30 apart (BatchNorm folded into a row can do that): the per-row power-of-two scale keeps every row in
    fp16's range and the epilogue's factor brings it back -- same relative accuracy for every row; an all-zero row stays zero."""
    rng = np.random.default_rng(17)
    M, K, N = 2048, 128, 128
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / 12).astype(np.float32) * np.exp2(rng.integers(-30, 31, (N, 1))).astype(np.float32)
    w[5] = 0
    out, flag = ops.conv1x1_f16x2(ops.dev(a), w, np.zeros(N, np.float32))
    assert flag == 0
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    S = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
    o = out.cpu().numpy().astype(np.float64)
    assert np.all(o[:, 5] == 0)
    S[:, 5] = 1
    assert (np.abs(o - ref) / (U * S)).max() <= 2 * np.sqrt(K)


@pytest.mark.parametrize("pro", [False, True])
def test_gemm_f16x2_range_guard(ops, pro):
    """Exactly at the limit: the largest activation the form can hold is 65504 / 16 = 4094; one element just below leaves the flag down and the result
    finite and accurate, one element at / above raises it (with a prologue: the value AFTER relu(x * scale + shift) counts).  inf raises it too."""
    rng = np.random.default_rng(3)
    M, K, N = 4096, 128, 128
    w = (rng.standard_normal((N, K)) / 12).astype(np.float32)
    b = np.zeros(N, np.float32)
    sc, sh = np.full(K, 2.0, np.float32), np.full(K, 1.0, np.float32)
    for big, want in ((4093.0, 0), (4094.0, 1), (5000.0, 1), (1e30, 1), (np.inf, 1)):
        a = np.abs(rng.standard_normal((M, K))).astype(np.float32)
        a[M - 7, K - 3] = (big - 1.0) / 2.0 if pro else big              # the last k-step of a late tile: every step is tracked
        out, flag = ops.conv1x1_f16x2(ops.dev(a), w, b, pro=(sc, sh) if pro else None)
        assert flag == want, (big, flag)
        if not want:
            x = a.astype(np.float64) * 2 + 1 if pro else a.astype(np.float64)
            ref = x @ w.astype(np.float64).T
            S = np.abs(x) @ np.abs(w).astype(np.float64).T
            o = out.cpu().numpy().astype(np.float64)
            assert np.isfinite(o).all() and (np.abs(o - ref) / (U * S)).max() <= 2 * np.sqrt(K)


@pytest.mark.parametrize("C", [128, 64])
def test_winograd_f16x2_forward_error_per_element(ops, C):
    rng = np.random.default_rng(300 + C)
    L, H, W = 4, 32, 32
    c = 2.0 * np.sqrt(9 * C)
    b = np.zeros(C, np.float32)
    cases = _wino_cases(rng, L, H, W, C)
    x = (rng.standard_normal((L, C, H, W)) * np.exp2(rng.integers(-20, 5, (1, C, 1, 1)))).astype(np.float32)
    cases.append(("mixed_in_range", x, (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)))
    report = []
    for name, x, w in cases:
        out, flag = ops.conv3x3_wino_f16x2(ops.nhwc(x), w, b)
        over = bool(np.abs(x).max() >= LIMIT_3X3)
        assert flag == int(over), (name, flag, np.abs(x).max())
        if over:
            assert name == "mixed_magnitudes"
            continue
        xt, wt = torch.from_numpy(x).double(), torch.from_numpy(w).double()
        ref = F.conv2d(xt, wt, padding=1).numpy()
        S = F.conv2d(xt.abs(), wt.abs(), padding=1).numpy()
        h = ops.nchw(out).astype(np.float64)
        f32 = ops.nchw(ops.conv3x3_wino(ops.nhwc(x), w, b)).astype(np.float64)
        assert np.isfinite(h).all()
        ex, ef = (h - ref) / (U * S), (f32 - ref) / (U * S)
        report.append((name, np.abs(ex).max(), np.abs(ef).max(), ex.mean(), ef.mean(), ex.std(), ef.std()))
    print(f"\nWinograd 3x3 C={C}: case, max|err| f16x2 / fp32 pipe, mean signed err f16x2 / fp32 pipe, std f16x2 / fp32 pipe   [units of 2^-24 sum|x||w|]")
    for r in report:
        print("   %-18s %7.3f %7.3f   %+8.4f %+8.4f   %7.4f %7.4f" % r)
    assert len(report) == len(cases) - 1
    for name, mx, mf, bx, bf, sx, sf in report:
        assert mx <= c, (name, mx)
        assert mx <= MAX_VS_F32 * mf + REPR, (name, mx, mf)
        if name == "one_signed":
            assert abs(bx) <= BIAS_MAX, (name, bx)


@pytest.mark.parametrize("up", [False, True])
def test_fused_tail_f16x2_forward_error_per_element(ops, up):
    """conv2 (3x3 Winograd) -> ReLU -> conv3 (1x1) + skip [+ up-sampled addend] in one launch, both products on two fp16 terms: the gate of the bf16x3 tail
    (tests/test_gpu_x3_accuracy.py) on the same one-signed data, and ragged map sizes (tiles that leave the map) against the bf16x3 tail."""
    rng = np.random.default_rng(9)
    L, H, W = 6, 32, 32
    x = np.abs(rng.standard_normal((L, H, W, 128))).astype(np.float32)
    skip = np.abs(rng.standard_normal((L, H, W, 256))).astype(np.float32)
    upv = np.abs(rng.standard_normal((L, H // 2, W // 2, 256))).astype(np.float32) if up else None
    w2 = np.abs(rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    b2 = np.abs(rng.standard_normal(128) * 0.3).astype(np.float32)
    w3 = np.abs(rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
    b3 = np.abs(rng.standard_normal(256)).astype(np.float32)
    xd, sd = torch.from_numpy(x).cuda(), torch.from_numpy(skip).cuda()
    ud = torch.from_numpy(upv).cuda() if up else None
    out, flag = ops.conv3x3_wino_f16x2_conv1x1_skip_up(xd, w2, b2, w3, b3, sd, ud)
    assert flag == 0
    h = out.cpu().numpy().astype(np.float64)
    f32 = (ops.conv3x3_wino_conv1x1_skip_up(xd, w2, b2, w3, b3, sd, ud) if up else ops.conv3x3_wino_conv1x1_skip(xd, w2, b2, w3, b3, sd)).cpu().numpy().astype(np.float64)
    xm = torch.from_numpy(x).permute(0, 3, 1, 2).double()
    m = F.relu(F.conv2d(xm, torch.from_numpy(w2).double(), torch.from_numpy(b2).double(), padding=1))
    ref = (F.conv2d(m, torch.from_numpy(w3).double()[:, :, None, None], torch.from_numpy(b3).double())).permute(0, 2, 3, 1).numpy() + skip
    if up:
        ref = ref + np.repeat(np.repeat(upv.astype(np.float64), 2, axis=1), 2, axis=2)
    ex, ef = (h - ref) / (U * ref), (f32 - ref) / (U * ref)                 # every term positive: the sum of magnitudes IS the result
    print("\nfused tail (up=%s), one-signed: max|err| f16x2 %.3f fp32 pipe %.3f, mean signed %+.4f %+.4f   [units of 2^-24 sum|x||w|]"
          % (up, np.abs(ex).max(), np.abs(ef).max(), ex.mean(), ef.mean()))
    assert np.abs(ex).max() <= 2.0 * np.sqrt(9 * 128 + 128)
    assert np.abs(ex).max() <= MAX_VS_F32 * np.abs(ef).max() + REPR
    assert abs(ex.mean()) <= BIAS_MAX
    # ragged maps: 20 x 24 (tiles of 8 x 16 leave the map on both axes), sign-mixed data, against the bf16x3 tail
    Hr, Wr = 20, 24
    x = rng.standard_normal((3, Hr, Wr, 128)).astype(np.float32)
    skip = rng.standard_normal((3, Hr, Wr, 256)).astype(np.float32)
    w2 = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    w3 = (rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
    xd, sd = torch.from_numpy(x).cuda(), torch.from_numpy(skip).cuda()
    ud = torch.from_numpy(rng.standard_normal((3, Hr // 2, Wr // 2, 256)).astype(np.float32)).cuda() if up else None
    out, flag = ops.conv3x3_wino_f16x2_conv1x1_skip_up(xd, w2, b2, w3, b3, sd, ud)
    x3 = ops.conv3x3_wino_x3_conv1x1_skip_up(xd, w2, b2, w3, b3, sd, ud, tail_x3=True)
    assert flag == 0
    assert (out - x3).abs().max().item() <= 2e-6 * x3.abs().max().item()


def test_winograd_and_tail_range_guard(ops):
    """The 3x3 form bounds |B^T d B| <= 4 max |d| and raises the flag from 1023.5 on (never late, at most 4x early); the tail raises it when
    relu(conv2 + b2) itself -- an activation the caller never sees -- reaches 4094."""
    rng = np.random.default_rng(21)
    w = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    b = np.zeros(128, np.float32)
    for big, want in ((1000.0, 0), (1023.0, 0), (1024.0, 1), (3000.0, 1), (np.inf, 1)):
        x = rng.standard_normal((2, 16, 32, 128)).astype(np.float32)
        x[1, 15, 31, 127] = big                                            # the last pixel, the last channel chunk
        out, flag = ops.conv3x3_wino_f16x2(torch.from_numpy(x).cuda(), w, b)
        assert flag == want, (big, flag)
        if not want:
            assert torch.isfinite(out).all()
    # the tail: inputs of O(1), conv2 weights large enough that relu(conv2 + b2) exceeds 4094 while conv2's own input stays far inside the range
    x = np.abs(rng.standard_normal((2, 16, 32, 128))).astype(np.float32)
    skip = rng.standard_normal((2, 16, 32, 256)).astype(np.float32)
    w3 = (rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
    b3 = np.zeros(256, np.float32)
    for gain, want in ((1.0, 0), (100.0, 0), (250.0, 1)):
        w2 = np.abs(rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32) * np.float32(gain)
        xd = torch.from_numpy(x).cuda()
        m = F.relu(F.conv2d(xd.permute(0, 3, 1, 2), torch.from_numpy(w2).cuda(), padding=1))
        assert (m.max().item() >= LIMIT_1X1) == bool(want), m.max().item()
        out, flag = ops.conv3x3_wino_f16x2_conv1x1_skip_up(xd, w2, b, w3, b3, torch.from_numpy(skip).cuda(), None)
        assert flag == want, (gain, flag)


def _net_outputs(sd, img, boxes, env, monkeypatch, max_crops=None):
    from suo_slam_amd.pkpnet import PkpNet
    for k in ("SUO_F16X2", "SUO_WINO_BF16X3"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    net = PkpNet(state_dict=sd, max_crops=max_crops or len(boxes))
    out = net(img, [torch.from_numpy(boxes)], None)
    torch.cuda.synchronize()
    return net, {k: v.cpu().numpy() for k, v in out.items()}


def test_network_default_pipe_is_f16x2_and_agrees_with_bf16x3(monkeypatch):
    from suo_slam_amd import weights
    sd = weights.make_random_state_dict(seed=0, logit_gain=8.0)
    rng = np.random.default_rng(4)
    img = (rng.uniform(0, 1, (480, 640, 3)) * 255).astype(np.uint8)
    boxes = np.tile(np.array([[100, 80, 300, 290], [350.5, 100.25, 600, 400], [10, 200, 130, 330], [200, 150, 420, 330]], np.float32), (10, 1))      # 40 crops: the large-launch kernels
    n16, o16 = _net_outputs(sd, img, boxes, {}, monkeypatch)
    assert n16.pipe() == 2 and not n16.range_exceeded()
    n3, o3 = _net_outputs(sd, img, boxes, {"SUO_F16X2": "0"}, monkeypatch)
    assert n3.pipe() == 1
    scale = np.abs(o3["prob_logits"]).max()
    assert np.abs(o16["prob_logits"] - o3["prob_logits"]).max() <= 1e-5 * scale
    assert np.abs(o16["uv"] - o3["uv"]).max() <= 1e-5 and np.abs(o16["cov"] - o3["cov"]).max() <= 1e-5
    assert not np.array_equal(o16["prob_logits"], o3["prob_logits"])       # (they ARE different kernels)


def test_network_falls_back_when_activations_leave_the_fp16_range(monkeypatch, capfd):
    """conv1_ scaled by 3e4 makes the stem's output -- the first operand of a 1x1 convolution on the matrix pipe -- ~1e4 > 4094: the fp16 form cannot hold it.
    PkpNet notices at its first synchronisation point (suo_net_range_exceeded), the network moves to bf16x3 and the call is re-issued: the caller gets, bit
    for bit, what a network created on bf16x3 returns, and says so on stderr."""
    from suo_slam_amd import weights
    sd = weights.make_random_state_dict(seed=0, logit_gain=8.0)
    sd = dict(sd)
    sd["backbone.conv1_.weight"] = sd["backbone.conv1_.weight"] * np.float32(3e4)
    rng = np.random.default_rng(4)
    img = (rng.uniform(0, 1, (480, 640, 3)) * 255).astype(np.uint8)
    boxes = np.tile(np.array([[100, 80, 300, 290], [350.5, 100.25, 600, 400]], np.float32), (20, 1))
    n16, o16 = _net_outputs(sd, img, boxes, {}, monkeypatch)
    assert n16.pipe() == 1                                                  # fell back
    assert "left the fp16 range" in capfd.readouterr().err
    n3, o3 = _net_outputs(sd, img, boxes, {"SUO_F16X2": "0"}, monkeypatch)
    for k in o3:
        assert np.isfinite(o3[k]).all() and np.array_equal(o16[k], o3[k]), k
    # and the next call of the same object stays on bf16x3 without another round trip
    out = n16(img, [torch.from_numpy(boxes)], None)
    torch.cuda.synchronize()
    assert np.array_equal(out["prob_logits"].cpu().numpy(), o3["prob_logits"]) and not n16.range_exceeded()


@pytest.mark.parametrize("up", [False, True])
@pytest.mark.parametrize("L,H,W", [(6, 32, 32), (3, 20, 24), (2, 64, 64)])
def test_fused_tail_with_the_next_blocks_conv1_equals_the_separate_launches(ops, monkeypatch, up, L, H, W):
    """csrc/conv_wino_x3.hip, NEXT: the block's output is split for the next block's conv1 where it is stored and multiplied on the spot.  Bit for bit: `out` = the
    fused tail without NEXT; `next` = suo_conv1x1_f16x2_ex (BN + ReLU prologue, ReLU) on that `out` -- the same products in the same order.  Ragged maps (tiles that
    leave the map write nothing outside it), the up-sampled addend, and the range guard on the inner operand relu(bn_next(out)).
    (Bit-identity is between the FOUR-wave forms: the eight-wave form small launches take pairs its last additions differently, test_eight_wave_form_of_small_launches.)"""
    monkeypatch.setenv("SUO_WINO_W8", "0")
    rng = np.random.default_rng(31 + H + up)
    x = torch.from_numpy(rng.standard_normal((L, H, W, 128)).astype(np.float32)).cuda()
    skip = torch.from_numpy(rng.standard_normal((L, H, W, 256)).astype(np.float32)).cuda()
    upd = torch.from_numpy(rng.standard_normal((L, H // 2, W // 2, 256)).astype(np.float32)).cuda() if up else None
    w2 = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    b2 = (rng.standard_normal(128) * 0.3).astype(np.float32)
    w3 = (rng.standard_normal((256, 128)) / np.sqrt(128)).astype(np.float32)
    b3 = rng.standard_normal(256).astype(np.float32)
    ns, nt = rng.uniform(0.5, 1.5, 256).astype(np.float32), (rng.standard_normal(256) * 0.3).astype(np.float32)
    w1 = (rng.standard_normal((128, 256)) / 16).astype(np.float32)
    b1 = (rng.standard_normal(128) * 0.2).astype(np.float32)
    if up:                                                                  # not built: that variant has no registers left (profiles/REJECTED.md)
        from suo_slam_amd import _lib
        with pytest.raises(_lib.SuoError, match="up-sampled addend"):
            ops.conv3x3_wino_f16x2_tail_next(x, w2, b2, w3, b3, skip, upd, (ns, nt), w1, b1)
        return
    out, nxt, flag = ops.conv3x3_wino_f16x2_tail_next(x, w2, b2, w3, b3, skip, upd, (ns, nt), w1, b1)
    ref_out, f0 = ops.conv3x3_wino_f16x2_conv1x1_skip_up(x, w2, b2, w3, b3, skip, upd)
    assert flag == 0 and f0 == 0
    assert torch.equal(out, ref_out)
    ref_nxt, f1 = ops.conv1x1_f16x2(ref_out.reshape(-1, 256), w1, b1, pro=(ns, nt), relu=True)
    assert f1 == 0
    assert torch.equal(nxt.reshape(-1, 128), ref_nxt)
    # the guard: a next-block BatchNorm scale that pushes relu(bn_next(out)) beyond 4094 raises the flag (out itself stays valid and in range)
    ns_big = ns * np.float32(3000.0)
    _, _, flag = ops.conv3x3_wino_f16x2_tail_next(x, w2, b2, w3, b3, skip, upd, (ns_big, nt), w1, b1)
    assert flag == 1


def test_device_split_equals_numpy_float16_rounding(ops):
    """The device's operand split (v_cvt_pk_f16_f32 for hi, v_fma_mixlo / mixhi_f16 for the residual: csrc/f16x2.h s2_lo_pack) against numpy's float16 rounding,
    bit for bit: with W = identity the kernel returns (hi + lo) / 16 of every input exactly (one product per output, powers of two everywhere), so the output
    IS the split.  Inputs cover ties, the normal / subnormal boundary of both terms, and values whose residual is below fp16's smallest subnormal."""
    rng = np.random.default_rng(5)
    K = N = 128
    M = 4096
    a = rng.standard_normal((M, K)).astype(np.float32)
    a[:64] *= np.exp2(rng.integers(-24, 8, (64, K))).astype(np.float32)
    a[64, :16] = np.array([1 + 2.0 ** -11, 1 + 3 * 2.0 ** -11, 2.0 ** -18, 2.0 ** -18 * (1 + 2.0 ** -10), 2.0 ** -28, 2.0 ** -29, 3 * 2.0 ** -29, 255.9375,
                           -(1 + 2.0 ** -11), 4093.999, 2047.5 / 16, 65503.0 / 16, 1e-12, -1e-7, 0.0, 1023.5 * 2.0 ** -28], np.float32)
    w = np.eye(N, K, dtype=np.float32)
    out, flag = ops.conv1x1_f16x2(ops.dev(a), w, np.zeros(N, np.float32))
    assert flag == 0
    x = (a * np.float32(16.0)).astype(np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    want = ((hi.astype(np.float32) + lo.astype(np.float32)) / np.float32(16.0)).astype(np.float32) + np.float32(0.0)      # (+ 0: an input that vanishes in BOTH terms gives the accumulator's +0, not -0)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))



# ---- lin -> head in one launch (csrc/gemm_bf16x3.hip: gemm_chain_head_kernel) -------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,hw,lda", [(8192, 4096, 256), (64, 64, 256), (4096 * 5, 4096, 256), (1024, 256, 320)])
def test_chain_head_is_bit_identical_to_the_two_launches(ops, M, hw, lda):
    """logits = W2 relu(W1 a + b1) + b2 with the 256-channel tensor kept in LDS: the same products in the same order as suo_conv1x1_f16x2_ex launched twice (the second
    on the stored tensor) -- every bit equal; NCHW, only the first 41 of the 64 padded channels written; the flag down."""
    rng = np.random.default_rng(M + hw)
    a_full = torch.from_numpy(rng.standard_normal((M, lda)).astype(np.float32)).cuda()
    a = a_full[:, :256]
    w1 = (rng.standard_normal((256, 256)) / 16).astype(np.float32)
    b1 = (rng.standard_normal(256) * 0.2).astype(np.float32)
    w2 = np.zeros((64, 256), np.float32)
    w2[:41] = (rng.standard_normal((41, 256)) / 16).astype(np.float32)
    w2[7] *= 300.0                                                 # rows of very different size: the per-row shift
    w2[9] *= 1e-3
    b2 = np.zeros(64, np.float32)
    b2[:41] = rng.standard_normal(41).astype(np.float32)
    got, flag = ops.conv1x1_chain_head_f16x2(a, w1, b1, w2, b2, 41, hw)
    assert flag == 0 and torch.isfinite(got).all()
    ll, f1 = ops.conv1x1_f16x2(a, w1, b1, relu=True)
    two, f2 = ops.conv1x1_f16x2(ll, w2, b2)
    assert f1 == 0 and f2 == 0
    want = two.reshape(M // hw, hw, 64)[:, :, :41].permute(0, 2, 1).contiguous()
    assert torch.equal(got, want)
    # and against fp64
    ref = (np.maximum(a.double().cpu().numpy() @ w1.astype(np.float64).T + b1, 0) @ w2.astype(np.float64).T + b2)[:, :41]
    ref = ref.reshape(M // hw, hw, 41).transpose(0, 2, 1)
    scale = np.abs(ref).max(axis=(0, 2), keepdims=True)
    assert (np.abs(got.cpu().numpy() - ref) / scale).max() < 5e-6


def test_chain_head_range_guard(ops):
    """The tensor that never leaves the CU is covered by the guard: relu(lin) of 4094 or more raises the flag, as does the input."""
    rng = np.random.default_rng(2)
    M, hw = 4096, 4096
    a = torch.from_numpy(np.abs(rng.standard_normal((M, 256))).astype(np.float32)).cuda()
    w1 = np.abs(rng.standard_normal((256, 256)) / 16).astype(np.float32)
    w2 = np.zeros((64, 256), np.float32)
    w2[:41] = (rng.standard_normal((41, 256)) / 16).astype(np.float32)
    b = np.zeros(256, np.float32)
    assert ops.conv1x1_chain_head_f16x2(a, w1, b, w2, np.zeros(64, np.float32), 41, hw)[1] == 0
    assert ops.conv1x1_chain_head_f16x2(a, w1 * np.float32(400.0), b, w2, np.zeros(64, np.float32), 41, hw)[1] == 1       # the inner tensor
    a2 = a.clone()
    a2[100, 3] = 5000.0
    assert ops.conv1x1_chain_head_f16x2(a2, w1, b, w2, np.zeros(64, np.float32), 41, hw)[1] == 1                          # the input


@pytest.mark.parametrize("L,H,W", [(1, 64, 64), (5, 64, 64), (8, 64, 64), (3, 32, 32), (2, 40, 24)])
def test_eight_wave_form_of_small_launches(ops, monkeypatch, L, H, W):
    """Round 6: launches of <= 256 tiles (a one-frame call, the passes of a SLAM view) run the fp16 Winograd kernels with EIGHT waves per tile
    (csrc/conv_wino_x3.hip: W8 -- two waves per 32-channel slice, half the components each, no fold; conv3 one n-tile per wave).  Against fp64 at the split forms'
    tolerance, against the four-wave form (SUO_WINO_W8=0) to fp32 rounding of the last additions, and -- ragged maps -- nothing written outside the map."""
    import torch.nn.functional as Fn
    rng = np.random.default_rng(100 * L + H)
    x = torch.from_numpy(rng.standard_normal((L, H, W, 128)).astype(np.float32)).cuda()
    skip = torch.from_numpy(rng.standard_normal((L, H, W, 256)).astype(np.float32)).cuda()
    w2 = (rng.standard_normal((128, 128, 3, 3)) / 34).astype(np.float32)
    b2 = (rng.standard_normal(128) * 0.2).astype(np.float32)
    w3 = (rng.standard_normal((256, 128)) / 11).astype(np.float32)
    b3 = rng.standard_normal(256).astype(np.float32)
    even = H % 2 == 0 and W % 2 == 0
    low = torch.from_numpy(rng.standard_normal((L, H // 2, W // 2, 256)).astype(np.float32)).cuda() if even else None

    def run():
        r = [ops.conv3x3_wino_f16x2_conv1x1_skip_up(x, w2, b2, w3, b3, skip), ops.conv3x3_wino_f16x2(x, w2, b2, relu=True)]
        if even:
            r.append(ops.conv3x3_wino_f16x2_conv1x1_skip_up(x, w2, b2, w3, b3, skip, low))
        assert all(f == 0 for _, f in r)
        return [t for t, _ in r]
    monkeypatch.setenv("SUO_WINO_W8", "0")
    four = run()
    monkeypatch.setenv("SUO_WINO_W8", "1")
    eight = run()
    xm = x.cpu().permute(0, 3, 1, 2).double()
    mid = Fn.relu(Fn.conv2d(xm, torch.from_numpy(w2).double(), torch.from_numpy(b2).double(), padding=1))
    ref_tail = Fn.conv2d(mid, torch.from_numpy(w3).double()[:, :, None, None], torch.from_numpy(b3).double()).permute(0, 2, 3, 1) + skip.cpu().double()
    refs = [ref_tail, mid.permute(0, 2, 3, 1)]
    if even:
        refs.append(ref_tail + low.cpu().double().repeat_interleave(2, 1).repeat_interleave(2, 2))
    for k, (a, b, r) in enumerate(zip(four, eight, refs)):
        scale = float(r.abs().max())
        assert torch.isfinite(b).all()
        assert float((b.cpu().double() - r).abs().max()) < 5e-6 * scale, (k, "vs fp64")
        assert float((b - a).abs().max()) < 2e-6 * scale, (k, "vs the four-wave form")
        assert not torch.equal(a, b) or L * H * W < 64                      # (the eight-wave form really ran: its last additions pair differently)

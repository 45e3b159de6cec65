"""csrc/stem_x3.hip: RoIAlign of the frame + stem conv 7x7 / stride 2 (3 image channels) + BN + ReLU in one launch on the bf16 matrix pipe --
against the two launches it replaces (roi_align_concat -> convk<7,2,4> on the fp32 pipe) and against fp64 on the crop kernel's own samples."""
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


def _stem_inputs(rng, fmt):
    img = rng.integers(0, 256, (2, 480, 640, 3), dtype=np.uint8)
    boxes = np.array([[100.3, 50.7, 300.9, 260.2], [0, 0, 640, 480], [-20.5, -10, 90, 120], [600, 400, 700, 520], [320, 240, 320.5, 240.2],
                      [10, 20, 522, 472], [200, 100, 330, 420]], np.float32)          # incl. > 256 px (4 samples per bin), out of image, degenerate
    idx = np.array([0, 1, 0, 1, 1, 0, 1], np.int32)
    if fmt == 1:
        imgd = torch.from_numpy((img.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2).copy()).cuda()
    else:
        imgd = torch.from_numpy(img).cuda()
    return imgd, boxes, idx


@pytest.mark.parametrize("fmt", [0, 1])
def test_fused_stem_equals_roi_align_then_stem(ops, fmt):
    """The staged values are the crop kernel's (same device function, csrc/roi_sample.h); the convolution on the bf16 pipe is within 5e-6 of
    fp64 on those values and never worse than 2x the fp32-pipe stem (+1e-7); every crop is checked (frame index per crop, boxes larger than
    256 px, boxes leaving the image)."""
    from suo_slam_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(31 + fmt)
    imgd, boxes, idx = _stem_inputs(rng, fmt)
    L = len(boxes)
    w = (rng.standard_normal((64, 44, 7, 7)) / np.sqrt(49 * 3)).astype(np.float32)       # the checkpoint's conv1_ (44 input channels): only 3 are used
    scale = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    bias = (rng.standard_normal(64) * 0.3).astype(np.float32)
    wx = np.empty(14 * 2 * 3 * 64 * 8, np.uint16)
    _lib.check(lib.suo_pack_stem_weight_bf16x3(w.ctypes.data, 44, scale.ctypes.data, wx.ctypes.data), "suo_pack_stem_weight_bf16x3")
    wxd, bd, boxd, idxd = torch.from_numpy(wx.view(np.int16)).cuda(), ops.dev(bias), ops.dev(boxes), torch.from_numpy(idx).cuda()
    out = torch.full((L, 128, 128, 64), -7.0, device="cuda")
    _lib.check(lib.suo_stem_x3(ops.P(imgd), fmt, 480, 640, ops.P(boxd), ops.P(idxd), L, ops.P(wxd), ops.P(bd), ops.P(out), ops.S()), "suo_stem_x3")
    torch.cuda.synchronize()
    # the two launches it replaces: staged crop (4 channels) -> fp32-pipe stem with the same folded weights
    crops = []
    for i in range(L):                                           # (suo_roi_align_concat takes one frame: crop by crop)
        st = torch.zeros((1, 256, 256, 48), device="cuda")
        one = imgd[idx[i]].contiguous()
        _lib.check(lib.suo_roi_align_concat(ops.P(one), fmt, 480, 640, ops.P(boxd[i:i + 1].contiguous()), 1, None, ops.P(st), ops.S()), "suo_roi_align_concat")
        crops.append(st[..., :4])
    torch.cuda.synchronize()
    staged = torch.cat(crops).contiguous()
    wf = (w[:, :3] * scale[:, None, None, None]).astype(np.float32)
    f32 = ops.conv_kxk(staged, wf, bias, relu=True)                         # convk_kernel<7,2,4,...>
    x = staged[..., :3].permute(0, 3, 1, 2).double().cpu()
    ref = F.relu(F.conv2d(x, torch.from_numpy(wf).double(), torch.from_numpy(bias).double(), stride=2, padding=3)).permute(0, 2, 3, 1).numpy()
    got, f = out.cpu().numpy(), f32.cpu().numpy()
    rng_ = np.abs(ref).max()
    for i in range(L):
        e3, e32 = np.abs(got[i] - ref[i]).max() / rng_, np.abs(f[i] - ref[i]).max() / rng_
        assert e3 < 5e-6, (i, e3)
        assert e3 <= 2.0 * e32 + 1e-7, (i, e3, e32)


@pytest.mark.parametrize("fmt", [0, 1])
def test_fused_stem_on_two_fp16_terms(ops, fmt):
    """The same launch with NP = 2 (csrc/f16x2.h; the network's default pipe since round 5): samples times 16 as hi + lo fp16, weight rows scaled into
    [2^12, 2^13), three MFMAs per product block.  Same gate as the bf16 form against fp64 on the crop kernel's samples (5e-6 of each channel's range); the worst channel within 1.5x the bf16 form's
    worst + 1e-7 and every channel within 2.5x its own bf16 error + 1e-7 (measured: 1.37e-6 against 1.36e-6 overall, worst single channel 1.95x); the guard
    stays down for a frame in [0, 1]."""
    from suo_slam_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(41 + fmt)
    imgd, boxes, idx = _stem_inputs(rng, fmt)
    L = len(boxes)
    w = (rng.standard_normal((64, 44, 7, 7)) / np.sqrt(49 * 3)).astype(np.float32)
    w[5] *= 1e-3                                                 # channels of very different size: the per-channel shift
    w[9] *= 40.0
    w[11] = 0.0
    scale = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    bias = (rng.standard_normal(64) * 0.3).astype(np.float32)
    wx = np.empty(14 * 2 * 3 * 64 * 8, np.uint16)
    wh = np.empty(14 * 2 * 2 * 64 * 8, np.uint16)
    osc = np.empty(64, np.float32)
    _lib.check(lib.suo_pack_stem_weight_bf16x3(w.ctypes.data, 44, scale.ctypes.data, wx.ctypes.data), "suo_pack_stem_weight_bf16x3")
    _lib.check(lib.suo_pack_stem_weight_f16x2(w.ctypes.data, 44, scale.ctypes.data, wh.ctypes.data, osc.ctypes.data), "suo_pack_stem_weight_f16x2")
    assert np.all(np.log2(osc) == np.round(np.log2(osc)))        # exact powers of two
    wxd, whd = torch.from_numpy(wx.view(np.int16)).cuda(), torch.from_numpy(wh.view(np.int16)).cuda()
    bd, od, boxd, idxd = ops.dev(bias), ops.dev(osc), ops.dev(boxes), torch.from_numpy(idx).cuda()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    o3 = torch.full((L, 128, 128, 64), -7.0, device="cuda")
    o2 = torch.full((L, 128, 128, 64), -7.0, device="cuda")
    _lib.check(lib.suo_stem_x3(ops.P(imgd), fmt, 480, 640, ops.P(boxd), ops.P(idxd), L, ops.P(wxd), ops.P(bd), ops.P(o3), ops.S()), "suo_stem_x3")
    _lib.check(lib.suo_stem_f16x2(ops.P(imgd), fmt, 480, 640, ops.P(boxd), ops.P(idxd), L, ops.P(whd), ops.P(od), ops.P(bd), ops.P(o2), ops.P(flag), ops.S()),
               "suo_stem_f16x2")
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    crops = []
    for i in range(L):
        st = torch.zeros((1, 256, 256, 48), device="cuda")
        one = imgd[idx[i]].contiguous()
        _lib.check(lib.suo_roi_align_concat(ops.P(one), fmt, 480, 640, ops.P(boxd[i:i + 1].contiguous()), 1, None, ops.P(st), ops.S()), "suo_roi_align_concat")
        crops.append(st[..., :3])
    torch.cuda.synchronize()
    x = torch.cat(crops).permute(0, 3, 1, 2).double().cpu()
    wf = (w[:, :3] * scale[:, None, None, None]).astype(np.float32)
    ref = F.relu(F.conv2d(x, torch.from_numpy(wf).double(), torch.from_numpy(bias).double(), stride=2, padding=3)).permute(0, 2, 3, 1).numpy()
    g2, g3 = o2.cpu().numpy(), o3.cpu().numpy()
    # per output channel (their magnitudes differ by 4e4): error against that channel's own range
    rng_c = np.maximum(np.abs(ref).max(axis=(0, 1, 2)), 1e-30)
    e2 = np.abs(g2 - ref).max(axis=(0, 1, 2)) / rng_c
    e3 = np.abs(g3 - ref).max(axis=(0, 1, 2)) / rng_c
    assert e2.max() < 5e-6, e2.max()
    print("stem f16x2 / bf16x3 per-channel max error / range:", e2.max(), e3.max(), "worst ratio", (e2 / np.maximum(e3, 1e-12)).max(), "at", int(np.argmax(e2 / np.maximum(e3, 1e-12))))
    assert e2.max() <= 1.5 * e3.max() + 1e-7, (e2.max(), e3.max())
    assert np.all(e2 <= 2.5 * e3 + 1e-7), (e2 / np.maximum(e3, 1e-12)).max()     # (a max over 1e5 outputs per channel: the ratio of two such maxima scatters)
    assert np.array_equal(g2[..., 11], np.maximum(np.broadcast_to(bias[11], g2[..., 11].shape), 0))     # an all-zero channel: relu(bias) exactly


@pytest.mark.parametrize("fmt", [0, 1])
def test_fused_stem_with_the_first_blocks_conv1_is_bit_identical_to_the_two_launches(ops, fmt):
    """csrc/stem_x3.hip, NEXT: r1's conv1 (BatchNorm + ReLU prologue, 1x1 64 -> 64, folded BatchNorm, ReLU) computed on the tile in the stem's epilogue patch.  The stem's
    own output is unchanged bit for bit; the second tensor equals suo_conv1x1_f16x2_ex launched on the stored stem output (same products, same order)."""
    from suo_slam_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(61 + fmt)
    imgd, boxes, idx = _stem_inputs(rng, fmt)
    L = len(boxes)
    w = (rng.standard_normal((64, 44, 7, 7)) / np.sqrt(49 * 3)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    bias = (rng.standard_normal(64) * 0.3).astype(np.float32)
    wh, osc = np.empty(14 * 2 * 2 * 64 * 8, np.uint16), np.empty(64, np.float32)
    _lib.check(lib.suo_pack_stem_weight_f16x2(w.ctypes.data, 44, scale.ctypes.data, wh.ctypes.data, osc.ctypes.data), "suo_pack_stem_weight_f16x2")
    w1 = (rng.standard_normal((64, 64)) / 8).astype(np.float32)
    w1[3] *= 50.0
    b1 = (rng.standard_normal(64) * 0.2).astype(np.float32)
    ps, pt = rng.uniform(0.5, 1.5, 64).astype(np.float32), (rng.standard_normal(64) * 0.2).astype(np.float32)
    w1h, o1, _ = ops.pack_gemm_f16x2(w1)
    whd, od, bd = torch.from_numpy(wh.view(np.int16)).cuda(), ops.dev(osc), ops.dev(bias)
    boxd, idxd = ops.dev(boxes), torch.from_numpy(idx).cuda()
    psd, ptd, b1d = ops.dev(ps), ops.dev(pt), ops.dev(b1)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    plain = torch.empty((L, 128, 128, 64), device="cuda")
    _lib.check(lib.suo_stem_f16x2(ops.P(imgd), fmt, 480, 640, ops.P(boxd), ops.P(idxd), L, ops.P(whd), ops.P(od), ops.P(bd), ops.P(plain), ops.P(flag), ops.S()), "suo_stem_f16x2")
    out = torch.full((L, 128, 128, 64), -7.0, device="cuda")
    mid = torch.full((L, 128, 128, 64), -7.0, device="cuda")
    _lib.check(lib.suo_stem_f16x2_next(ops.P(imgd), fmt, 480, 640, ops.P(boxd), ops.P(idxd), L, ops.P(whd), ops.P(od), ops.P(bd), ops.P(out), ops.P(psd), ops.P(ptd),
                                       ops.P(w1h), ops.P(o1), ops.P(b1d), ops.P(mid), ops.P(flag), ops.S()), "suo_stem_f16x2_next")
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    assert torch.equal(out, plain)
    want, f2 = ops.conv1x1_f16x2(plain.reshape(-1, 64), w1, b1, pro=(ps, pt), relu=True)
    assert f2 == 0
    assert torch.equal(mid.reshape(-1, 64), want)


def test_fused_stem_fp16_guard_rises_for_a_float_frame_beyond_range(ops):
    """A float frame is whatever the caller hands over: one sample of 40000 (a bilinear weight of 1/4 or more of it, times 16, is beyond 65504) inside a box raises the flag; the same frame with the
    sample outside every box does not."""
    from suo_slam_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 1, (1, 3, 480, 640)).astype(np.float32)
    img[0, 1, 100, 200] = 40000.0
    w = (rng.standard_normal((64, 3, 7, 7)) / 12).astype(np.float32)
    wh, osc = np.empty(14 * 2 * 2 * 64 * 8, np.uint16), np.empty(64, np.float32)
    _lib.check(lib.suo_pack_stem_weight_f16x2(w.ctypes.data, 3, None, wh.ctypes.data, osc.ctypes.data), "suo_pack_stem_weight_f16x2")
    whd, od, bd = torch.from_numpy(wh.view(np.int16)).cuda(), ops.dev(osc), ops.dev(np.zeros(64, np.float32))
    imgd = torch.from_numpy(img).cuda()
    idxd = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = torch.empty((1, 128, 128, 64), device="cuda")
    for box, want in (([150, 50, 300, 200], 1), ([300, 250, 500, 400], 0)):
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        boxd = ops.dev(np.array([box], np.float32))
        _lib.check(lib.suo_stem_f16x2(ops.P(imgd), 1, 480, 640, ops.P(boxd), ops.P(idxd), 1, ops.P(whd), ops.P(od), ops.P(bd), ops.P(out), ops.P(flag), ops.S()),
                   "suo_stem_f16x2")
        torch.cuda.synchronize()
        assert int(flag.item()) == want, box


def test_network_with_and_without_the_fused_stem(monkeypatch):
    """SUO_STEM_X3 (read once per process: child processes) switches the prior-less pass between the fused stem and roi_align + fp32 stem; the
    network's outputs agree to the 1e-5 the path is held to (tests/test_gpu_cnn.py holds either against the reference's golden logits)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from suo_slam_amd import weights; from suo_slam_amd.pkpnet import PkpNet\n"
            "sd = weights.make_random_state_dict(seed=0, logit_gain=8.0); rng = np.random.default_rng(3)\n"
            "img = (rng.uniform(0, 1, (480, 640, 3)) * 255).astype(np.uint8)\n"
            "boxes = np.array([[100, 80, 300, 290], [350.5, 100.25, 600, 400], [10, 200, 130, 330]], np.float32)\n"
            "out = PkpNet(state_dict=sd, max_crops=4)(img, [torch.from_numpy(boxes)], None); torch.cuda.synchronize()\n"
            "z = PkpNet(state_dict=sd, max_crops=4)(img, [torch.from_numpy(boxes)], [torch.zeros(3, 41, 256, 256)])['prob_logits'].cpu().numpy()\n"
            "np.savez(sys.argv[1], zeros_prior=z, **{k: out[k].cpu().numpy() for k in ('prob_logits', 'uv', 'cov', 'kp_mask')})\n") % root
    outs = {}
    for mode in ("1", "0"):
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"suo_stem_{os.getpid()}_{mode}.npz")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, SUO_STEM_X3=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = dict(np.load(path))
        os.remove(path)
    lg = outs["0"]["prob_logits"]
    assert np.array_equal(outs["0"]["zeros_prior"], lg)                           # same stem kernel either way: priors of zeros == no priors, bit for bit
    assert not np.array_equal(outs["1"]["prob_logits"], lg)                        # (different kernels really ran)
    assert np.abs(outs["1"]["prob_logits"] - lg).max() < 1e-5 * np.abs(lg).max()
    for k in ("uv", "cov", "kp_mask"):
        assert np.abs(outs["1"][k] - outs["0"][k]).max() < 1e-5, k

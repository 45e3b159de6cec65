"""Pin the CNN oracle (oracle/cnn_oracle.py) against vectors produced by the reference's own
modules (tests/golden/make_golden.py -> cnn_golden.npz).  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cnn_oracle as O
from suo_slam_amd import weights as W


def test_state_dict_matches_reference_layout(cnn_golden, state_dict):
    # 1274 entries in the reference state_dict = float tensors + one num_batches_tracked per BN
    n_bn = sum(1 for k in state_dict if k.endswith("running_mean"))
    assert len(state_dict) + n_bn == int(cnn_golden["n_state_entries"]) == 1274
    assert W.num_params(state_dict) == int(cnn_golden["n_params"]) == 12726732


def test_residual_blocks(cnn_golden, state_dict):
    P = O.to_torch(state_dict)
    for name in ("backbone.r1", "backbone.r4", "backbone.hourglass.0.up1_.0"):
        x = torch.from_numpy(cnn_golden[f"res_in:{name}"])
        y = O.residual(x, P, name).numpy()
        np.testing.assert_allclose(y, cnn_golden[f"res_out:{name}"], rtol=1e-5, atol=1e-5)


def test_hourglass(cnn_golden, state_dict):
    P = O.to_torch(state_dict)
    y = O.hourglass(torch.from_numpy(cnn_golden["hg_in"]), P, "backbone.hourglass.1", 4).numpy()
    np.testing.assert_allclose(y, cnn_golden["hg_out"], rtol=1e-5, atol=1e-4)


def test_full_backbone_and_decode(cnn_golden, state_dict):
    P = O.to_torch(state_dict)
    rng = np.random.Generator(np.random.PCG64(int(cnn_golden["backbone_in_seed"])))
    x = rng.uniform(0, 1, (1, 44, 256, 256)).astype(np.float32)
    with torch.no_grad():
        raw = O.hourglass_net(torch.from_numpy(x), P)
    ref = cnn_golden["backbone_logits"]
    scale = np.abs(ref).max()
    assert np.abs(raw.numpy() - ref).max() <= 2e-5 * scale
    d = O.decode(torch.from_numpy(ref), P)
    np.testing.assert_allclose(d["uv"].numpy(), cnn_golden["backbone_uv"], atol=2e-6)
    np.testing.assert_allclose(d["cov"].numpy(), cnn_golden["backbone_cov"], atol=2e-6)
    np.testing.assert_allclose(d["kp_mask_logits"].numpy(), cnn_golden["backbone_kp_mask_logits"], atol=1e-5)
    np.testing.assert_allclose(d["kp_mask"].numpy(), cnn_golden["backbone_kp_mask"], atol=1e-6)


def test_decode_peaked(cnn_golden, state_dict):
    P = O.to_torch(state_dict)
    d = O.decode(torch.from_numpy(cnn_golden["decode_in"]), P)
    np.testing.assert_allclose(d["uv"].numpy(), cnn_golden["decode_uv"], atol=2e-6)
    np.testing.assert_allclose(d["cov"].numpy(), cnn_golden["decode_cov"], atol=2e-6)
    np.testing.assert_allclose(d["kp_mask"].numpy(), cnn_golden["decode_kp_mask"], atol=1e-6)
    xx, yy = O.mesh_grid(64, 64)
    assert np.array_equal(xx.numpy(), cnn_golden["mesh_xx"])
    assert np.array_equal(yy.numpy(), cnn_golden["mesh_yy"])


def test_roi_align_identity_box():
    # integer-aligned box of exactly the output size: bin centres fall at pixel+0.5 => average of
    # the 2x2 neighbourhood (aligned=False convention: pixel centres at integer coordinates)
    rng = np.random.Generator(np.random.PCG64(5))
    img = rng.uniform(0, 1, (3, 40, 50)).astype(np.float32)
    out = O.roi_align(img, np.array([[4, 6, 20, 22]], np.float32), (16, 16))
    exp = 0.25 * (img[:, 6:22, 4:20] + img[:, 7:23, 4:20] + img[:, 6:22, 5:21] + img[:, 7:23, 5:21])
    np.testing.assert_allclose(out[0], exp, atol=1e-6)
    # samples beyond [-1, H] contribute zero; the box far outside gives zeros
    out = O.roi_align(img, np.array([[-100, -100, -50, -50]], np.float32), (4, 4))
    assert np.all(out == 0)
    # 2x downsample: grid 2x2 per bin -- bin (ph, pw) covers pixels [2 ph, 2 ph + 2) and its four samples sit at 2 ph + {.5, 1.5}:
    # each the mean of a 2x2 pixel neighbourhood, by hand
    out = O.roi_align(img, np.array([[0, 0, 32, 32]], np.float32), (16, 16))
    m = 0.25 * (img[:, :-1, :-1] + img[:, 1:, :-1] + img[:, :-1, 1:] + img[:, 1:, 1:])      # m[y, x] = the sample at (y + .5, x + .5)
    exp = 0.25 * (m[:, 0:32:2, 0:32:2] + m[:, 1:32:2, 0:32:2] + m[:, 0:32:2, 1:32:2] + m[:, 1:32:2, 1:32:2])
    np.testing.assert_allclose(out[0], exp, atol=1e-6)


# ---- independent cross-checks of the two "builder's reading only" pieces -----------------------------------------
def test_roi_align_equals_grid_sample_for_boxes_up_to_256_px():
    """For boxes of at most 256 px a side torchvision's RoIAlign (aligned=False, sampling_ratio=-1 -> one sample per bin,
    lib/models/pkpnet.py:93) is a plain bilinear resample at the bin centres x1 + (pw + .5) w / 256 in the pixel-centres-at-integers
    convention -- exactly F.grid_sample(align_corners=True) at g = 2 x / (W - 1) - 1.  torch's grid_sample is an implementation the
    builder did not write: 200 random in-image boxes of the synthetic stream's range (w, h ~ U[60, 240]), coordinates
    formed independently in float64."""
    from suo_slam_amd import synthetic as S
    rng = np.random.default_rng(2024)
    img = S.make_frame(rng, 1, noise=0.0)["image"]                       # uint8 [480,640,3], low-pass texture
    chw = O.image_to_chw(img)                                            # float32 [3,480,640] / 255
    H, W = chw.shape[1:]
    n = 200
    w = rng.uniform(60, 240, n)
    h = rng.uniform(60, 240, n)
    h[:20] = rng.uniform(1.5, 60, 20)                                    # a few small boxes too (> 1 px: the min-size clamp is not grid_sample's business)
    w[10:30] = rng.uniform(1.5, 60, 20)
    w[30], h[30] = 256.0, 256.0                                          # the largest one-sample-per-bin box
    x1 = rng.uniform(0, W - 1 - w)
    y1 = rng.uniform(0, H - 1 - h)
    boxes = np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    got = O.roi_align(chw, boxes, (256, 256))
    src = torch.from_numpy(chw).double()[None]

    def resample(xs, ys):                                                # xs, ys: float64 [n,256] sample coordinates in pixels
        ref = np.empty_like(got)
        grid = np.stack(np.broadcast_arrays((2 * xs / (W - 1) - 1)[:, None, :], (2 * ys / (H - 1) - 1)[:, :, None]), -1)      # [n,256,256,(x,y)]
        for i in range(0, n, 25):
            g = torch.from_numpy(grid[i:i + 25])
            ref[i:i + 25] = F.grid_sample(src.expand(g.shape[0], -1, -1, -1), g, mode="bilinear", padding_mode="zeros", align_corners=True).float().numpy()
        return ref

    # (A) the interpolation: sample coordinates rounded as RoIAlign's float32 arithmetic rounds them (start + ph * bin +
    #     .5 * bin), the interpolation itself by grid_sample in float64 -- atol 1e-6, the bound the HIP kernel is held to against this oracle
    f32 = np.float32
    bw = ((boxes[:, 2] - boxes[:, 0]) / f32(256)).astype(f32)
    bh = ((boxes[:, 3] - boxes[:, 1]) / f32(256)).astype(f32)
    p = np.arange(256, dtype=f32)
    xs32 = (boxes[:, 0:1] + p[None] * bw[:, None] + (f32(0.5) * bw)[:, None]).astype(f32)
    ys32 = (boxes[:, 1:2] + p[None] * bh[:, None] + (f32(0.5) * bh)[:, None]).astype(f32)
    ref = resample(xs32.astype(np.float64), ys32.astype(np.float64))
    assert np.abs(got - ref).max() < 1e-6, np.abs(got - ref).max()
    # (B) the sampling positions: bin centres x1 + (pw + .5) w / 256 formed in float64 from the definition.  float32 places a sample
    #     within 640 * 2^-23 = 8e-5 px of that; times the steepest gradient of the texture
    b = boxes.astype(np.float64)
    centres = (np.arange(256) + 0.5) / 256.0
    ref = resample(b[:, 0:1] + centres[None] * (b[:, 2:3] - b[:, 0:1]), b[:, 1:2] + centres[None] * (b[:, 3:4] - b[:, 1:2]))
    grad = max(np.abs(np.diff(chw, axis=1)).max(), np.abs(np.diff(chw, axis=2)).max())
    assert np.abs(got - ref).max() < 1e-6 + 2 * 8e-5 * grad, (np.abs(got - ref).max(), grad)


@pytest.mark.parametrize("H,W", [(480, 640), (540, 720)])
def test_roi_align_equals_grid_sample_for_boxes_beyond_256_px(H, W):
    """Boxes of more than 256 px a side -- YCB-V 640x480 and T-LESS 720x540 detections reach the whole frame -- take ceil(roi / 256) = 2 or
    3 samples per bin and axis (/root/reference/lib/models/pkpnet.py:93), and boxes leaving the image meet the `y < -1 or
    y > H -> 0` and clamp-to-the-border rules.  200 random boxes, a third of them leaving the image; every sample interpolated by
    F.grid_sample(align_corners=True) in float64 and averaged (tests/roi_ref.py), (A) at the positions RoIAlign's float32 arithmetic
    gives, (B) at positions formed in float64 from the definition."""
    from tests import roi_ref as R
    from suo_slam_amd import synthetic as S
    rng = np.random.default_rng(77 + H)
    img = S.make_frame(rng, 1, noise=0.0)["image"]                       # uint8 [480,640,3], low-pass texture
    if (H, W) != img.shape[:2]:                                          # T-LESS frame size: the same texture, tiled and cut
        img = np.tile(img, (2, 2, 1))[:H, :W]
    chw = O.image_to_chw(img)
    boxes = R.large_boxes(rng, 200, H, W)
    _, _, gw, gh = R.sample_positions(boxes)
    assert set(gw) | set(gh) == {1, 2, 3}                                # every grid count the frame sizes can produce
    got = O.roi_align(chw, boxes, (256, 256))
    ref = R.roi_align_grid_sample(chw, boxes, 256, f32_positions=True)
    assert np.abs(got - ref).max() < 1e-6, np.abs(got - ref).max()
    ref64 = R.roi_align_grid_sample(chw, boxes, 256, f32_positions=False)
    grad = max(np.abs(np.diff(chw, axis=1)).max(), np.abs(np.diff(chw, axis=2)).max())
    # a float32 position is within ~1e-4 px of the float64 one (720 * 2^-23 per operation); a sample within that of the y = -1 / y = H
    # rule may count on one side only: compare where the two agree on validity, and require that to be all but a handful of bins
    d = np.abs(got - ref64)
    close = d < 1e-6 + 4 * 1e-4 * grad
    assert close.mean() > 1 - 1e-5 and d[close].max() < 1e-6 + 4 * 1e-4 * grad, (close.mean(), d.max())


def test_prior_stamp_equals_conv2d_of_an_impulse_with_opencvs_kernel():
    """gaussian_2d(91) (lib/utils/utils.py:356-361) = cv2.GaussianBlur(impulse, (91, 91), 0) / max.  OpenCV's documented pieces --
    getGaussianKernel(91, sigma <= 0 -> 0.3 ((91 - 1) / 2 - 1) + 0.8 = 14, float32 coefficients normalised to sum 1), a separable
    filter, BORDER_REFLECT_101 -- assembled from torch's own reflect padding and F.conv2d (machinery the builder did not write)
    against the closed form the product and the device kernel evaluate (object_slam._gaussian_patch; csrc/misc.hip prior_value)."""
    from suo_slam_amd import object_slam as OS
    n = 91
    sigma = 0.3 * ((n - 1) * 0.5 - 1) + 0.8
    assert sigma == 14.0
    x = np.arange(n, dtype=np.float64) - (n - 1) * 0.5
    cf = np.exp(-0.5 / (sigma * sigma) * x * x).astype(np.float32)       # getGaussianKernel, CV_32F
    k = (cf.astype(np.float64) * (1.0 / cf.astype(np.float64).sum())).astype(np.float32)
    imp = torch.zeros(1, 1, n, n)
    imp[0, 0, n // 2, n // 2] = 1
    padded = F.pad(imp, (n // 2,) * 4, mode="reflect")                   # torch's "reflect" = OpenCV's BORDER_REFLECT_101 (edge sample not repeated)
    kt = torch.from_numpy(k)
    rows = F.conv2d(padded, kt.reshape(1, 1, 1, n))
    blur = F.conv2d(rows, kt.reshape(1, 1, n, 1))[0, 0].numpy()
    assert blur.shape == (n, n)
    ref = blur / blur.max()
    got = OS._gaussian_patch(n)
    assert np.abs(got - ref).max() < 1e-6, np.abs(got - ref).max()
    # the reflection really is what doubles the outer ring (without it the corner would be a quarter of this)
    assert abs(ref[0, 0] / (4 * (k[0] / k[n // 2]) ** 2) - 1) < 1e-5


# ---- round 6: the wide reference-generated goldens (tests/golden/make_golden_wide.py) ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def wide():
    import os
    from tests.conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "cnn_golden_wide.npz"))


def test_oracle_backbone_on_five_crops_of_different_statistics(wide, state_dict):
    """The oracle against the REFERENCE's logits / decode on five crops (texture with zero priors, with stamped priors, heavy-tailed, dark under dense priors,
    saturated blocks): 2e-5 of each crop's own logit range, the decode on the reference's logits to float32 rounding."""
    from tests.golden import cnn_inputs as I
    P = O.to_torch(state_dict)
    assert list(wide["kinds"]) == list(I.CROP_KINDS)
    for i, kind in enumerate(I.CROP_KINDS):
        with torch.no_grad():
            raw = O.hourglass_net(torch.from_numpy(I.crop(kind)[None]), P).numpy()
        ref = wide["logits"][i:i + 1]
        assert np.abs(raw - ref).max() <= 2e-5 * np.abs(ref).max(), kind
    d = O.decode(torch.from_numpy(wide["logits"]), P)
    np.testing.assert_allclose(d["uv"].numpy(), wide["uv"], atol=2e-6)
    np.testing.assert_allclose(d["cov"].numpy(), wide["cov"], atol=2e-6)
    np.testing.assert_allclose(d["kp_mask"].numpy(), wide["kp_mask"], atol=1e-6)


def test_oracle_residual_blocks_at_network_map_sizes(wide, state_dict):
    """256 -> 256 at 64x64 / 32x32, r4 (128 -> 128) at 64x64 / 32x32, r5 (128 -> 256, conv4 skip) at 64x64 against the reference's Residual module."""
    from tests.golden import cnn_inputs as I
    P = O.to_torch(state_dict)
    for i, (name, cin, cout, hw) in enumerate(I.BLOCKS):
        with torch.no_grad():
            y = O.residual(torch.from_numpy(I.block_input(i)), P, name).numpy()
        ref = wide["block%d_rows" % i]
        assert y.shape == (1, cout, hw, hw)
        assert np.abs(y[:, :, I.block_rows(hw), :] - ref).max() <= 1e-5 * float(wide["block%d_absmax" % i]), (name, hw)

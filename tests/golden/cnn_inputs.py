"""Seeded network inputs shared by tests/golden/make_golden_wide.py (which runs the REFERENCE's modules on them, build container only) and by the tests (which
regenerate them instead of storing 11.5 MB per crop).  Five [44,256,256] crops with different statistics, and the inputs of the per-block goldens."""
import numpy as np

CROP_KINDS = ("texture_zero_priors", "texture_stamped_priors", "heavy_tailed", "dark_dense_priors", "saturated_blocks")


def _texture(rng):
    """Smooth colour texture in [0,1]: low-pass filtered noise (what a RoIAlign-ed crop of a real frame looks like to the stem)."""
    t = rng.uniform(0, 1, (3, 34, 34))
    t = np.kron(t, np.ones((1, 8, 8)))[:, :256 + 8, :256 + 8]
    k = np.ones(9) / 9
    for ax in (1, 2):
        t = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, t)
    return t[:, 4:260, 4:260]


def _stamps(rng, frac=0.6):
    """41 prior channels as lib/utils/utils.py:356-411 renders them: a Gaussian bump (sigma 14, peak 1) at a random pixel for `frac` of the keypoints, else zeros."""
    p = np.zeros((41, 256, 256))
    yy, xx = np.mgrid[0:256, 0:256]
    for c in range(41):
        if rng.random() < frac:
            cy, cx = rng.uniform(0, 256, 2)
            g = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 14.0 ** 2))
            g[(np.abs(yy - cy) > 45) | (np.abs(xx - cx) > 45)] = 0.0
            p[c] = g
    return p


def crop(kind, seed=20260):
    """One [44,256,256] float32 network input of the named kind."""
    rng = np.random.Generator(np.random.PCG64(seed + CROP_KINDS.index(kind)))
    x = np.zeros((44, 256, 256))
    if kind == "texture_zero_priors":            # the single-view pass (lib/object_slam.py:1094-1097 feeds zero priors)
        x[:3] = _texture(rng)
    elif kind == "texture_stamped_priors":       # the SLAM prior pass
        x[:3] = _texture(rng)
        x[3:] = _stamps(rng)
    elif kind == "heavy_tailed":                 # magnitudes over three decades: what the split-operand forms' range handling has to survive
        x[:] = np.abs(rng.standard_t(2.5, x.shape)) * 0.25
    elif kind == "dark_dense_priors":            # a nearly black, low-contrast crop under dense prior channels
        x[:3] = 0.02 + 0.01 * rng.standard_normal((3, 256, 256))
        x[3:] = rng.uniform(0, 1, (41, 256, 256))
    elif kind == "saturated_blocks":             # 0 / 1 blocks of 16 x 16 pixels: every edge is full contrast
        x[:3] = np.kron(rng.integers(0, 2, (3, 16, 16)), np.ones((16, 16)))
    else:
        raise KeyError(kind)
    return x.astype(np.float32)


def staged(kinds):
    """[L,256,256,48] NHWC staging tensor (suo_net_backbone's input) of the named crops."""
    xin = np.zeros((len(kinds), 256, 256, 48), np.float32)
    for i, k in enumerate(kinds):
        xin[i, ..., :44] = crop(k).transpose(1, 2, 0)
    return xin


# per-block goldens: (state_dict prefix, cin, cout, H = W); inputs ~ what the block sees in the network (post-ReLU-like, positive-skewed) from a seed
BLOCKS = (("backbone.hourglass.0.up1_.0", 256, 256, 64), ("backbone.hourglass.0.low1_.0", 256, 256, 32), ("backbone.r4", 128, 128, 64), ("backbone.r4", 128, 128, 32),
          ("backbone.r5", 128, 256, 64), ("backbone.Residual.1", 256, 256, 64))
BLOCK_ROWS = (0, 1, 7, 8, 16, 31, 32, 47, 63)      # output rows stored (borders, tile seams of 8 / 16 / 32, interior); rows >= H dropped


def block_input(i):
    name, cin, cout, hw = BLOCKS[i]
    rng = np.random.Generator(np.random.PCG64(7700 + i))
    x = rng.standard_normal((1, cin, hw, hw)) * rng.uniform(0.3, 2.0, (1, cin, 1, 1)) + rng.uniform(-0.5, 1.0, (1, cin, 1, 1))
    return x.astype(np.float32)


def block_rows(hw):
    return [r for r in BLOCK_ROWS if r < hw]

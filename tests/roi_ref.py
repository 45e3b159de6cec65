"""An independent RoIAlign for the tests: torchvision's definition (the call at
/root/reference/lib/models/pkpnet.py:93: aligned=False, sampling_ratio=-1 -> ceil(roi / 256) samples per bin and axis) with the bilinear
interpolation done by torch's F.grid_sample(align_corners=True) in float64 -- machinery the builder did not write.  Two ways to form the
sample positions: as RoIAlign's float32 arithmetic rounds them (tight comparison) or in float64 from the definition."""
import numpy as np
import torch
import torch.nn.functional as F


def sample_positions(boxes, out=256, f32=True):
    """[n, grid_max, out] positions per axis + grid counts.  Returns (xs, ys, gw, gh): xs[i][:gw[i]] are the ix-th sample columns."""
    b = np.asarray(boxes, np.float32)
    n = len(b)
    T = np.float32 if f32 else np.float64
    bt = b.astype(T)
    roi_w = np.maximum(bt[:, 2] - bt[:, 0], T(1.0))
    roi_h = np.maximum(bt[:, 3] - bt[:, 1], T(1.0))
    bin_w = (roi_w / T(out)).astype(T)
    bin_h = (roi_h / T(out)).astype(T)
    gw = np.ceil(roi_w / T(out)).astype(int)
    gh = np.ceil(roi_h / T(out)).astype(int)
    p = np.arange(out, dtype=T)
    gm = int(max(gw.max(), gh.max()))
    xs = np.zeros((n, gm, out), T)
    ys = np.zeros((n, gm, out), T)
    for i in range(n):
        for ix in range(gw[i]):     # start + p * bin + (ix + .5) * bin / grid, evaluated left to right in T
            xs[i, ix] = ((bt[i, 0] + p * bin_w[i]).astype(T) + (T(ix + 0.5) * bin_w[i]).astype(T) / T(gw[i])).astype(T)
        for iy in range(gh[i]):
            ys[i, iy] = ((bt[i, 1] + p * bin_h[i]).astype(T) + (T(iy + 0.5) * bin_h[i]).astype(T) / T(gh[i])).astype(T)
    return xs, ys, gw, gh


def roi_align_grid_sample(chw, boxes, out=256, f32_positions=True):
    """float32 [n, C, out, out].  Per sample: 0 when y < -1 or y > H (x likewise), else the coordinate clamped into [0, H - 1] and
    interpolated bilinearly; the bin's value is the mean of its grid_h * grid_w samples."""
    chw = np.asarray(chw, np.float32)
    C, H, W = chw.shape
    src = torch.from_numpy(chw).double()[None]
    xs, ys, gw, gh = sample_positions(boxes, out, f32_positions)
    res = np.zeros((len(boxes), C, out, out), np.float32)
    for i in range(len(boxes)):
        acc = torch.zeros((C, out, out), dtype=torch.float64)
        for iy in range(gh[i]):
            y = ys[i, iy].astype(np.float64)
            vy = ~((y < -1.0) | (y > H))
            yc = np.clip(y, 0.0, H - 1.0)
            for ix in range(gw[i]):
                x = xs[i, ix].astype(np.float64)
                vx = ~((x < -1.0) | (x > W))
                xc = np.clip(x, 0.0, W - 1.0)
                gx, gy = np.meshgrid(2 * xc / (W - 1) - 1, 2 * yc / (H - 1) - 1)
                g = torch.from_numpy(np.stack([gx, gy], -1))[None]
                v = F.grid_sample(src, g, mode="bilinear", padding_mode="border", align_corners=True)[0]
                acc += v * torch.from_numpy((vy[:, None] & vx[None, :]).astype(np.float64))
        res[i] = (acc / float(gh[i] * gw[i])).float().numpy()
    return res


def large_boxes(rng, n, H, W):
    """n boxes for an H x W frame: sides up to the whole frame (1-3 samples per bin and axis), a third of them leaving the image."""
    w = rng.uniform(200, W, n)
    h = rng.uniform(200, H, n)
    x1 = rng.uniform(0, np.maximum(W - 1 - w, 1e-3))
    y1 = rng.uniform(0, np.maximum(H - 1 - h, 1e-3))
    k = n // 3
    x1[:k] += rng.uniform(-0.4, 0.4, k) * W                               # leaving the image on any side
    y1[:k] += rng.uniform(-0.4, 0.4, k) * H
    boxes = np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    boxes[k] = [0, 0, W, H]                                               # the whole frame
    boxes[k + 1] = [-30.5, -20.25, W + 12.75, H + 40.5]                   # beyond it on every side
    return boxes

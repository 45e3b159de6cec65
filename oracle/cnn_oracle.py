"""ORACLE (test infrastructure only) -- CPU restatement of the keypoint-CNN path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product path (``suo_slam_amd``) never does.

Restates, in plain PyTorch-CPU fp32 functional ops, the reference path
    PkpNet.forward           /root/reference/lib/models/pkpnet.py:80-119
    HourglassNet.forward     /root/reference/lib/models/hg.py:95-119
    Hourglass.forward        /root/reference/lib/models/hg.py:37-58
    Residual.forward         /root/reference/lib/models/layers/Residual.py:20-35
    spatial_softmax          /root/reference/lib/models/pkpnet.py:13-17
    mesh_grid                /root/reference/lib/models/pkpnet.py:19-26
    post_process_kp          /root/reference/lib/models/pkpnet.py:28-63
    keypoint mask logic      /root/reference/lib/object_slam.py:1100-1115
over a *state_dict* with the reference's keys (suo_slam_amd/weights.py).

Pinning: tests/golden/make_golden.py imports the reference's own
lib/models/{hg,pkpnet}.py in the build container, loads the same state_dict and
stores its outputs; tests/test_oracle_cnn.py checks this restatement against
those fixtures.  ``roi_align`` is the exception: torchvision 0.9.1 is a
third-party dependency that is absent from /root/reference and from this image
(requirements.txt:4) => roi_align is **parity unpinned**; it restates the
published RoIAlign algorithm (aligned=False, sampling_ratio=-1), SURVEY.md B1.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
NUM_KP = 41


def to_torch(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def _bn(x, P, p):
    return F.batch_norm(x, P[p + ".running_mean"], P[p + ".running_var"],
                        P[p + ".weight"], P[p + ".bias"], False, 0.0, BN_EPS)


def _conv(x, P, p, stride=1, padding=0):
    return F.conv2d(x, P[p + ".weight"], P[p + ".bias"], stride=stride, padding=padding)


def residual(x, P, p):
    """Residual.forward (layers/Residual.py:20-35): pre-activation bottleneck."""
    out = F.relu(_bn(x, P, p + ".bn"))
    out = _conv(out, P, p + ".conv1")
    out = F.relu(_bn(out, P, p + ".bn1"))
    out = _conv(out, P, p + ".conv2", padding=1)
    out = F.relu(_bn(out, P, p + ".bn2"))
    out = _conv(out, P, p + ".conv3")
    res = _conv(x, P, p + ".conv4") if (p + ".conv4.weight") in P else x
    return out + res


def hourglass(x, P, p, n, n_modules=2):
    """Hourglass.forward (hg.py:37-58)."""
    up1 = x
    for j in range(n_modules):
        up1 = residual(up1, P, f"{p}.up1_.{j}")
    low1 = F.max_pool2d(x, 2, 2)
    for j in range(n_modules):
        low1 = residual(low1, P, f"{p}.low1_.{j}")
    if n > 1:
        low2 = hourglass(low1, P, p + ".low2", n - 1, n_modules)
    else:
        low2 = low1
        for j in range(n_modules):
            low2 = residual(low2, P, f"{p}.low2_.{j}")
    low3 = low2
    for j in range(n_modules):
        low3 = residual(low3, P, f"{p}.low3_.{j}")
    up2 = F.interpolate(low3, scale_factor=2)          # default mode: nearest (hg.py:56)
    return up1 + up2


def hourglass_net(x, P, p="backbone", n_stack=2, n_modules=2, depth=4, all_stacks=False):
    """HourglassNet.forward (hg.py:95-119); returns the LAST stack's head output."""
    x = _conv(x, P, p + ".conv1_", stride=2, padding=3)
    x = F.relu(_bn(x, P, p + ".bn1"))
    x = residual(x, P, p + ".r1")
    x = F.max_pool2d(x, 2, 2)
    x = residual(x, P, p + ".r4")
    x = residual(x, P, p + ".r5")
    out = []
    for i in range(n_stack):
        ll = hourglass(x, P, f"{p}.hourglass.{i}", depth, n_modules)
        for j in range(n_modules):
            ll = residual(ll, P, f"{p}.Residual.{i * n_modules + j}")
        ll = F.relu(_bn(_conv(ll, P, f"{p}.lin_.{i}.0"), P, f"{p}.lin_.{i}.1"))
        tmp_out = _conv(ll, P, f"{p}.tmpOut.{i}")
        out.append(tmp_out)
        if i < n_stack - 1:
            x = x + _conv(ll, P, f"{p}.ll_.{i}") + _conv(tmp_out, P, f"{p}.tmpOut_.{i}")
    return out if all_stacks else out[-1]


def spatial_softmax(raw):
    """pkpnet.py:13-17."""
    b, k, h, w = raw.shape
    return F.softmax(raw.reshape(b, k, h * w), dim=-1).reshape(raw.shape)


def mesh_grid(h, w):
    """pkpnet.py:19-26.  NOTE the 'ij' indexing quirk:
    xx[i, j] = r[i] (row axis), yy[i, j] = -r[j] (column axis, negated)."""
    assert h == w
    r = torch.arange(0.5, h, 1) / (h / 2) - 1
    xx = r[:, None].expand(h, w)
    yy = (-r)[None, :].expand(h, w)
    return xx.to(torch.float32), yy.to(torch.float32)


def post_process_kp(prob):
    """pkpnet.py:28-63 with calc_sigma=True, z=None."""
    k, vh, vw = prob.shape[1:]
    xx, yy = mesh_grid(vh, vw)
    sx = torch.sum(prob * xx, [2, 3])
    sy = torch.sum(prob * yy, [2, 3])
    uv = torch.stack([sx, sy], -1)
    dx = xx[None, None] - sx[..., None, None]
    dy = yy[None, None] - sy[..., None, None]
    cxx = torch.sum(prob * dx * dx, [2, 3])
    cxy = torch.sum(prob * dx * dy, [2, 3])
    cyy = torch.sum(prob * dy * dy, [2, 3])
    cov = torch.stack([torch.stack([cxx, cxy], -1), torch.stack([cxy, cyy], -1)], -2)
    return uv, cov


def classifier(raw, P):
    """pkpnet.py:74-78,116-118 in eval mode (Dropout = identity)."""
    m = raw.mean(3).mean(2)
    logits = F.linear(F.relu(m), P["classifier.2.weight"], P["classifier.2.bias"])
    return logits, torch.sigmoid(logits)


def decode(raw, P):
    """Everything after the backbone: softmax -> uv/cov -> validity probability."""
    prob = spatial_softmax(raw)
    uv, cov = post_process_kp(prob)
    logits, kp_mask = classifier(raw, P)
    return {"uv": uv, "cov": cov, "prob": prob, "kp_mask_logits": logits, "kp_mask": kp_mask}


def roi_align(image, boxes, out_hw=(256, 256)):
    """torchvision.ops.roi_align(image[1,C,H,W], [boxes[L,4]], output_size, spatial_scale=1,
    sampling_ratio=-1, aligned=False) -- PARITY UNPINNED restatement.

    Vectorised numpy; image float32 [C,H,W]; boxes float32 [L,4] xyxy pixels.
    Returns float32 [L,C,oh,ow].
    """
    img = np.asarray(image, dtype=np.float32)
    C, H, W = img.shape
    boxes = np.asarray(boxes, dtype=np.float32)
    oh, ow = out_hw
    out = np.zeros((boxes.shape[0], C, oh, ow), dtype=np.float32)
    f32 = np.float32
    for l, (x1, y1, x2, y2) in enumerate(boxes):
        roi_w = max(f32(x2) - f32(x1), f32(1.0))
        roi_h = max(f32(y2) - f32(y1), f32(1.0))
        bin_h = f32(roi_h / f32(oh))
        bin_w = f32(roi_w / f32(ow))
        gh = int(np.ceil(roi_h / f32(oh)))
        gw = int(np.ceil(roi_w / f32(ow)))
        acc = np.zeros((C, oh, ow), dtype=np.float32)
        ph = np.arange(oh, dtype=np.float32)
        pw = np.arange(ow, dtype=np.float32)
        for iy in range(gh):
            y = (f32(y1) + ph * bin_h + f32(iy + 0.5) * bin_h / f32(gh)).astype(np.float32)
            for ix in range(gw):
                x = (f32(x1) + pw * bin_w + f32(ix + 0.5) * bin_w / f32(gw)).astype(np.float32)
                acc += _bilinear(img, y, x)
        out[l] = acc / f32(gh * gw)
    return out


def _bilinear(img, y, x):
    C, H, W = img.shape
    vy = ~((y < -1.0) | (y > H))
    vx = ~((x < -1.0) | (x > W))
    y = np.maximum(y, 0).astype(np.float32)
    x = np.maximum(x, 0).astype(np.float32)
    y0 = np.floor(y).astype(np.int64)
    x0 = np.floor(x).astype(np.int64)
    cy = y0 >= H - 1
    cx = x0 >= W - 1
    y0 = np.where(cy, H - 1, y0)
    x0 = np.where(cx, W - 1, x0)
    y1 = np.where(cy, H - 1, y0 + 1)
    x1 = np.where(cx, W - 1, x0 + 1)
    y = np.where(cy, y0.astype(np.float32), y)
    x = np.where(cx, x0.astype(np.float32), x)
    ly = (y - y0).astype(np.float32)[:, None]
    lx = (x - x0).astype(np.float32)[None, :]
    hy = np.float32(1.0) - ly
    hx = np.float32(1.0) - lx
    v00 = img[:, y0][:, :, x0]
    v01 = img[:, y0][:, :, x1]
    v10 = img[:, y1][:, :, x0]
    v11 = img[:, y1][:, :, x1]
    val = (hy * hx) * v00 + (hy * lx) * v01 + (ly * hx) * v10 + (ly * lx) * v11
    val = val * (vy[:, None] & vx[None, :])
    return val.astype(np.float32)


def image_to_chw(img_u8):
    """object_slam.py:1092 -- HWC uint8 -> CHW float32 / 255 (no channel swap)."""
    return (np.asarray(img_u8).transpose(2, 0, 1).astype(np.float32) / np.float32(255))


def pkpnet_forward(img_u8, boxes, priors, sd, P=None):
    """PkpNet.forward on one image (pkpnet.py:80-119) as called from
    object_slam.py:1092-1099.  ``priors`` None => zeros (pkpnet.py:95-97)."""
    P = P if P is not None else to_torch(sd)
    crops = roi_align(image_to_chw(img_u8), boxes)
    L = crops.shape[0]
    if priors is None:
        priors = np.zeros((L, NUM_KP, crops.shape[2], crops.shape[3]), np.float32)
    x = torch.from_numpy(np.concatenate([crops, np.asarray(priors, np.float32)], axis=1))
    with torch.no_grad():
        raw = hourglass_net(x, P)
        ret = decode(raw, P)
    ret["prob_logits"] = raw
    return ret


def keypoint_masks(uv, cov, kp_mask_prob, model_kps_masks, bbox_thresh=0.9, kp_var_thresh=0.2):
    """object_slam.py:1100-1115 (numpy, float32 inputs)."""
    uv = np.asarray(uv, np.float32)
    cov = np.asarray(cov, np.float32)
    m = (np.asarray(kp_mask_prob, np.float32) > 0.3) & np.asarray(model_kps_masks, bool)
    m = m & (np.min(uv, -1) > -bbox_thresh) & (np.max(uv, -1) < bbox_thresh)
    std = np.sqrt(cov[..., [0, 1], [0, 1]])
    m = m & np.all(std < 2 * kp_var_thresh, axis=-1)
    return m

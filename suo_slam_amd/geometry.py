"""Host geometry helpers of the hot path (float64 numpy), mirroring the reference's utilities.

  fix_K_for_bbox_ndc   /root/reference/lib/utils/utils.py:416-429
  invert_SE3           /root/reference/lib/utils/utils.py:431-435
  transform_pts        /root/reference/lib/utils/utils.py:455-460
  normalize_uv         the K^-T normalisation inside pnp(), /root/reference/lib/object_slam.py:34-36
"""
from __future__ import annotations

import numpy as np


def fix_K_for_bbox_ndc(K_, bbox):
    """K_bbox = S @ T @ K: camera matrix projecting a camera-frame point to the bbox's NDC in [-1,1]
    (x right, y up): T shifts by (-x1,-y1); S = diag(2/w, -2/h, 1) with offsets (-1, +1)."""
    x1, y1, x2, y2 = bbox
    w, h = x2 - x1, y2 - y1
    K = np.array(K_, dtype=np.float64)
    T = np.eye(3)
    T[:2, 2] = -np.array([x1, y1], dtype=np.float64)
    S = np.eye(3)
    S[0, :] *= 2.0 / w
    S[1, :] *= -2.0 / h
    S[0, 2] -= 1
    S[1, 2] += 1
    return S @ T @ K


def fix_K_for_bbox_ndc_many(K_, bboxes):
    """``fix_K_for_bbox_ndc`` for all boxes of a frame at once: the same products as stacked matmuls (numpy runs the same 3x3 kernel per box:
    bit-identical, tests/test_host_logic.py)."""
    b = np.asarray(bboxes, dtype=np.float64).reshape(-1, 4)
    n = b.shape[0]
    T = np.tile(np.eye(3), (n, 1, 1))
    T[:, 0, 2] = -b[:, 0]
    T[:, 1, 2] = -b[:, 1]
    S = np.tile(np.eye(3), (n, 1, 1))
    S[:, 0, :] *= (2.0 / (b[:, 2] - b[:, 0]))[:, None]
    S[:, 1, :] *= (-2.0 / (b[:, 3] - b[:, 1]))[:, None]
    S[:, 0, 2] -= 1
    S[:, 1, 2] += 1
    return S @ T @ np.array(K_, dtype=np.float64)


def invert_SE3(T):
    Tinv = np.eye(4)
    Tinv[:3, :3] = T[:3, :3].T
    Tinv[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return Tinv


def to4x4(T):
    T = np.asarray(T, dtype=np.float64)
    if T.shape == (4, 4):
        return T
    out = np.empty((4, 4))
    out[:3] = T[:3, :4]
    out[3] = (0.0, 0.0, 0.0, 1.0)
    return out


def transform_pts(T, pts):
    T = np.asarray(T)
    return pts @ T[..., :3, :3].swapaxes(-1, -2) + T[..., :3, 3][..., None, :] if T.ndim > 2 else pts @ T[:3, :3].T + T[:3, 3]


def normalize_uv(points_2d, camera_matrix):
    """points_2d @ KinvT[:2,:2] + KinvT[2:3,:2]  (object_slam.py:34-36)."""
    KinvT = np.linalg.inv(camera_matrix).T
    return points_2d @ KinvT[:2, :2] + KinvT[2:3, :2]


def project_ndc(K_bbox, T_OtoC, pts):
    """uv (NDC) of object-frame points under pose T_OtoC and the bbox camera matrix."""
    pc = pts @ T_OtoC[:3, :3].T + T_OtoC[:3, 3]
    uvw = pc @ np.asarray(K_bbox).T
    return uvw[:, :2] / uvw[:, 2:3], pc[:, 2]

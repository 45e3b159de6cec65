"""Host restatement of the chi-square inlier counts (the per-pair numpy of /root/reference/lib/object_slam.py:1032-1066 and :648-681) -- TEST
infrastructure: the product scores on the device (suo_slam_amd/slam_score.py -> csrc/slam_score.hip).  tests/conftest.py routes
``suo_slam_amd.slam_score.chi2_counts`` here when a test runs without a GPU (the host rules around the scoring -- hypotheses, float32 containers,
the 3x rule -- are what those tests check); tests/test_gpu_slam_score.py holds the kernel against these functions and against oracle/slam_rules.py."""
import numpy as np

from suo_slam_amd.object_slam import CHI2_2DOF_95, _det_cache
from suo_slam_amd.weights import NUM_KP


def _chi2_inliers(T_OtoC, det, use_inlier_subset, manual_kp_std):
    """Shared scoring of __estimate_camera_pose (:1032-1066) and __maybe_reinit_objects (:648-681):
    number of keypoints whose re-projection has chi2 <= 5.991 under the detection's covariance."""
    sel = det["inliers"] if use_inlier_subset else np.ones(len(det["model_kp"]), bool)
    pts = det["model_kp"][sel]
    if pts.shape[0] == 0:
        return 0
    p = pts @ T_OtoC[:3, :3].T + T_OtoC[:3, 3]
    uvw = p @ det["K"].T
    pos = uvw[:, 2] > 0
    if not np.any(pos):
        return 0
    uv_proj = (uvw[:, :2] / uvw[:, 2:3])[pos]
    res = det["uv_pred"][sel][pos] - uv_proj
    rx, ry = res[:, 0], res[:, 1]
    cov = det["cov_pred"]
    if cov is not None:
        cov = np.asarray(cov[sel][pos], dtype=np.float64)
        a = np.maximum(cov[:, 0, 0], 1e-4)                                     # ensure invertible (:669,:1054)
        d = np.maximum(cov[:, 1, 1], 1e-4)
        b, c = cov[:, 0, 1], cov[:, 1, 0]
        det2 = a * d - b * c
        # r^T inv([[a,b],[c,d]]) r in closed form (this runs O(objects^2 + 15 objects) times per SLAM view)
        chi2 = (d * rx * rx - (b + c) * rx * ry + a * ry * ry) / det2
        assert not np.any(np.isnan(chi2)), "NaN in information matrix"
    else:
        chi2 = (rx * rx + ry * ry) / manual_kp_std ** 2
    return int(np.count_nonzero(chi2 <= CHI2_2DOF_95))


def _chi2_inliers_many(Ts, dets, use_inlier_subset, manual_kp_std):
    """``_chi2_inliers`` for B (pose, detection) pairs at once: the same arithmetic on arrays padded to NUM_KP
    keypoints.  Returns B counts."""
    B = len(dets)
    if B == 0:
        return np.zeros(0, dtype=np.int64)
    cs = [_det_cache(d) for d in dets]
    has_cov = cs[0]["cov"] is not None
    if any(((c["cov"] is not None) != has_cov) or c["n"] > NUM_KP for c in cs):
        return np.array([_chi2_inliers(T, d, use_inlier_subset, manual_kp_std) for T, d in zip(Ts, dets)], dtype=np.int64)
    n = np.array([c["n"] for c in cs])
    sel = np.arange(NUM_KP)[None, :] < n[:, None]
    if use_inlier_subset:
        inl = np.zeros((B, NUM_KP), dtype=bool)
        for i, d in enumerate(dets):
            inl[i, :n[i]] = d["inliers"]
        sel &= inl
    pts = np.stack([c["pts"] for c in cs])
    uv = np.stack([c["uv"] for c in cs])
    Ks = np.stack([c["K"] for c in cs])
    Ts = np.asarray(Ts, dtype=np.float64)
    p = pts @ Ts[:, :3, :3].transpose(0, 2, 1) + Ts[:, None, :3, 3]
    uvw = p @ Ks.transpose(0, 2, 1)
    pos = uvw[..., 2] > 0
    z = np.where(pos, uvw[..., 2], 1.0)
    rx = uv[..., 0] - uvw[..., 0] / z
    ry = uv[..., 1] - uvw[..., 1] / z
    if has_cov:
        cov = np.stack([c["cov"] for c in cs])
        a = np.maximum(cov[..., 0, 0], 1e-4)                                   # ensure invertible (:669,:1054)
        dd = np.maximum(cov[..., 1, 1], 1e-4)
        b, cc = cov[..., 0, 1], cov[..., 1, 0]
        chi2 = (dd * rx * rx - (b + cc) * rx * ry + a * ry * ry) / (a * dd - b * cc)
    else:
        chi2 = (rx * rx + ry * ry) / manual_kp_std ** 2
    ok = sel & pos
    assert not np.any(np.isnan(chi2[ok])), "NaN in information matrix"
    return np.count_nonzero(ok & (chi2 <= CHI2_2DOF_95), axis=1)


def chi2_counts(owner, Ts, dets, use_inlier_subset, manual_kp_std, chi2_max=CHI2_2DOF_95):
    """Signature of suo_slam_amd.slam_score.chi2_counts."""
    assert chi2_max == CHI2_2DOF_95
    return _chi2_inliers_many(Ts, dets, use_inlier_subset, manual_kp_std)

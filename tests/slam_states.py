"""Seeded SLAM states and sequences shared by tests/golden/make_slam_golden.py (which feeds them to the REFERENCE's own
ObjectSLAM, imported from /root/reference in the build container) and by the tests that replay them through
suo_slam_amd/object_slam.py and oracle/slam_rules.py.  Plain numpy + dicts: nothing here imports the product, the oracle or
the reference, so that neither side can leak into the inputs.  The fixture stores a digest of every generated state: if a
numpy release ever changed a generator's stream the tests would say so instead of comparing against stale outputs.

A *state* is the reference's own bookkeeping (lib/object_slam.py:125-153): ``detections[view][obj]`` dicts with the keys
``__run_kp_model`` writes (:1150-1165) plus ``bbox`` / ``model_kp_mask`` / ``prior_uv`` (:538-547), ``cam_poses[view]``
([3,4] and [4,4] both occur, as in the reference), ``obj_poses[obj]``, ``view_ids``, ``cam_K[view]``, ``obj_num_dets``.
"""
import copy
import hashlib
from collections import defaultdict

import numpy as np

K_YCBV = np.array([[1066.778, 0.0, 312.9869], [0.0, 1067.487, 241.3109], [0.0, 0.0, 1.0]])
NUM_KP = 41


def fix_K_for_bbox_ndc(K_, bbox):
    """Inputs only (the function under test is the product's / the reference's own): bbox NDC intrinsics."""
    x1, y1, x2, y2 = bbox
    w, h = x2 - x1, y2 - y1
    T = np.eye(3)
    T[:2, 2] = [-x1, -y1]
    S = np.eye(3)
    S[0, :] *= 2.0 / w
    S[1, :] *= -2.0 / h
    S[0, 2] -= 1
    S[1, 2] += 1
    return S @ T @ np.array(K_, dtype=np.float64)


def random_rotation(rng):
    A = rng.standard_normal((3, 3))
    Q, R = np.linalg.qr(A)
    Q = Q @ np.diag(np.sign(np.diag(R)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] *= -1
    return Q


def small_motion(rng, rot, trans):
    w = rng.normal(0, rot, 3)
    th = np.linalg.norm(w)
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + (np.sin(th) / th) * Wx + ((1 - np.cos(th)) / (th * th)) * (Wx @ Wx) if th > 1e-12 else np.eye(3)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.normal(0, trans, 3)
    return T


def _pose(rng, z=(700, 1100)):
    T = np.eye(4)
    T[:3, :3] = random_rotation(rng)
    T[:3, 3] = [rng.uniform(-150, 150), rng.uniform(-100, 100), rng.uniform(*z)]
    return T


def make_state(seed, n_obj=5, n_views=4, use_cov=True, noise=0.01, pnp_rot=5e-4, pnp_trans=0.5, map_rot=5e-4, map_trans=0.5,
               miss=0.2, drop_pose=0.15, kp_range=(6, 14), tiny_cov=False, inlier_rate=0.85, drop_map=0.1, outlier_rate=0.15,
               mixed_shapes=True):
    """n_views views with camera poses (the last one is the current view), detections with float32 covariances, a PnP pose
    per detection (ground truth perturbed; None with probability drop_pose), inlier flags, map poses for most objects.
    tiny_cov: covariances far below the reference's 1e-4 clamp (:669, :1054)."""
    rng = np.random.default_rng(seed)
    K = K_YCBV
    T_OtoG = {o: _pose(rng) for o in range(1, n_obj + 1)}
    n_kp = {o: int(rng.integers(*kp_range)) for o in T_OtoG}
    kps = {o: rng.uniform(-60, 60, (n_kp[o], 3)) for o in T_OtoG}
    st = {"detections": {}, "cam_poses": {}, "obj_poses": {}, "view_ids": [], "cam_K": {}, "manual_kp_std": 0.01,
          "mesh_db": {o: {"diameter": 120.0, "is_symmetric": bool(o % 3 == 0)} for o in T_OtoG},
          "obj_num_dets": {}, "use_cov": bool(use_cov)}
    for v in range(n_views):
        T_GtoC = small_motion(rng, 0.05, 30.0)
        T_est = T_GtoC @ small_motion(rng, 1e-3, 1.0)
        st["cam_poses"][v] = T_est[:3].copy() if (v % 2 and mixed_shapes) else T_est
        st["view_ids"].append(v)
        st["cam_K"][v] = K.copy()
        st["detections"][v] = {}
        for o in T_OtoG:
            if rng.random() < miss and v != n_views - 1:
                continue
            T_OtoC = T_GtoC @ T_OtoG[o]
            pc = kps[o] @ T_OtoC[:3, :3].T + T_OtoC[:3, 3]
            px = pc @ K.T
            px = px[:, :2] / px[:, 2:3]
            bbox = np.array([px[:, 0].min() - 8, px[:, 1].min() - 8, px[:, 0].max() + 8, px[:, 1].max() + 8])
            Kb = fix_K_for_bbox_ndc(K, bbox)
            uvw = pc @ Kb.T
            uv = uvw[:, :2] / uvw[:, 2:3] + rng.normal(0, noise, (n_kp[o], 2))
            out = rng.random(n_kp[o]) < outlier_rate
            uv[out] += rng.uniform(-0.5, 0.5, (int(out.sum()), 2))
            cov = None
            if use_cov:
                A = rng.normal(0, 0.3, (n_kp[o], 2, 2)) + np.eye(2)
                s = 1e-3 if tiny_cov else noise
                cov = ((A @ A.transpose(0, 2, 1)) * s * s).astype(np.float32)
            pose = None if rng.random() < drop_pose else small_motion(rng, pnp_rot, pnp_trans) @ T_OtoC
            kp_mask = np.zeros(NUM_KP, bool)
            kp_mask[:n_kp[o]] = True
            st["detections"][v][o] = {"pose": pose, "inliers": rng.random(n_kp[o]) < inlier_rate, "model_kp": kps[o].copy(), "uv_pred": uv,
                                      "cov_pred": cov, "K": Kb.astype(np.float32).astype(np.float64), "bbox": bbox, "kp_mask": kp_mask,
                                      "model_kp_mask": kp_mask.copy(), "prior_uv": None, "uv_gt": None,
                                      "score": 1.0}
    for o in T_OtoG:
        st["obj_num_dets"][o] = int(rng.integers(1, 6))
        if rng.random() >= drop_map:
            T = small_motion(rng, map_rot, map_trans) @ T_OtoG[o]
            st["obj_poses"][o] = T[:3].copy() if (o % 2 and mixed_shapes) else T
    return st


def install(slam, st):
    """Put a (deep) copy of the state into an ObjectSLAM-shaped object (the reference's class or the product's)."""
    st = copy.deepcopy(st)
    slam.detections = st["detections"]
    slam.cam_poses = st["cam_poses"]
    slam.obj_poses = st["obj_poses"]
    slam.view_ids = st["view_ids"]
    slam.cam_K = st["cam_K"]
    slam.images = {v: None for v in st["view_ids"]}
    slam.mesh_db = st["mesh_db"]
    slam.manual_kp_std = st["manual_kp_std"]
    slam.obj_num_dets = defaultdict(int, st["obj_num_dets"])
    slam.obj_num_det_kps = defaultdict(int)
    slam.remove_penalty = defaultdict(int)
    slam.needs_opt = True
    slam.no_network_cov = not st["use_cov"]
    return slam


def _feed(h, x):
    if x is None:
        h.update(b"N")
    elif isinstance(x, dict):
        for k in sorted(x, key=str):
            h.update(str(k).encode())
            _feed(h, x[k])
    elif isinstance(x, (list, tuple)):
        for y in x:
            _feed(h, y)
    elif isinstance(x, np.ndarray):
        h.update(str(x.dtype).encode() + str(x.shape).encode())
        h.update(np.ascontiguousarray(x).tobytes())
    else:
        h.update(repr(x).encode())


def digest(x):
    h = hashlib.sha256()
    _feed(h, x)
    return h.hexdigest()[:16]


# ---- whole sequences for process_view in the reference's --debug_gt_kp mode (lib/object_slam.py:1117-1131) ---------------
def make_sequence(seed, n_views=12, n_obj=5, sym_every=3, kp_range=(8, 16), miss=0.1, vis_drop=0.1, first_view_all=True):
    """A synthetic scene: n_obj objects in the world frame (= first camera), a smooth camera path, per view the inputs of
    ObjectSLAM.process_view (:327-328): obj_ids, bboxes (xyxy, float64), model_kps [L,41,3] float32, model_kps_masks, kp_masks
    (ground-truth visibility), uv_gt [L,41,2] (NDC of the bbox).  Every sym_every-th object is symmetric (gets the prior pass)."""
    rng = np.random.default_rng(seed)
    K = K_YCBV
    objs = list(range(1, n_obj + 1))
    T_OtoG = {}
    for o in objs:
        T = np.eye(4)
        T[:3, :3] = random_rotation(rng)
        T[:3, 3] = [rng.uniform(-220, 220), rng.uniform(-140, 140), rng.uniform(800, 1100)]
        T_OtoG[o] = T
    kps, mmask = {}, {}
    for o in objs:
        n = int(rng.integers(*kp_range))
        m = np.zeros(NUM_KP, bool)
        m[rng.choice(NUM_KP, n, replace=False)] = True
        mmask[o] = m
        kps[o] = (rng.uniform(-1, 1, (NUM_KP, 3)) * rng.uniform(40, 80, 3)).astype(np.float32)
    mesh_db = {o: {"diameter": float(2 * np.abs(kps[o]).max()), "is_symmetric": bool(sym_every and o % sym_every == 0)} for o in objs}
    views = []
    for v in range(n_views):
        s = v / max(n_views - 1, 1)
        ang = 0.35 * s
        T_GtoC = np.eye(4)
        if v > 0:
            T_GtoC[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
            T_GtoC[:3, 3] = [-260 * s, rng.uniform(-10, 10), rng.uniform(-10, 10) + 60 * s]
        ids, boxes, mk, mm, vis, uvg = [], [], [], [], [], []
        for o in objs:
            if rng.random() < miss and not (v == 0 and first_view_all):
                continue
            T_OtoC = T_GtoC @ T_OtoG[o]
            pts = kps[o].astype(np.float64)
            pc = pts @ T_OtoC[:3, :3].T + T_OtoC[:3, 3]
            px = pc @ K.T
            px = px[:, :2] / px[:, 2:3]
            sel = px[mmask[o]]
            bbox = np.array([sel[:, 0].min() - 10, sel[:, 1].min() - 10, sel[:, 0].max() + 10, sel[:, 1].max() + 10])
            Kb = fix_K_for_bbox_ndc(K, bbox)
            uvw = pc @ Kb.T
            ids.append(o)
            boxes.append(bbox)
            mk.append(kps[o])
            mm.append(mmask[o])
            vis.append(mmask[o] & (rng.random(NUM_KP) >= vis_drop))
            uvg.append((uvw[:, :2] / uvw[:, 2:3]).astype(np.float32))
        views.append({"view_id": 10 * v, "K": K.copy(), "obj_ids": np.array(ids), "bboxes": np.array(boxes, np.float64),
                      "model_kps": np.array(mk, np.float32), "model_kps_masks": np.array(mm, bool), "kp_masks": np.array(vis, bool),
                      "uv_gt": np.array(uvg, np.float32), "T_GtoC_gt": T_GtoC})
    return {"mesh_db": mesh_db, "views": views, "T_OtoG_gt": T_OtoG}

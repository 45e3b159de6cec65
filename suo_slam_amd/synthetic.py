"""Synthetic YCB-V-shaped inputs: 640x480 frames, L objects per frame with known poses,
per-class keypoint subsets of the 41-channel vocabulary, bounding boxes, and keypoint measurements
generated like the reference's --debug_gt_kp mode (projected GT keypoints + N(0, 0.01^2) NDC noise,
/root/reference/lib/object_slam.py:1129-1131).  No dataset or network is involved."""
from __future__ import annotations

import numpy as np

from .geometry import fix_K_for_bbox_ndc, project_ndc

NUM_KP = 41
# channel groups of the keypoint vocabulary (lib/labeling/kp_config.py:7-72): box 8, cylinder 10, tool 6,
# grip 4, spout 1, brand 4, nutrition 4, barcode 4
GROUPS = {"box_like": (0, 8), "cylinder_like": (8, 18), "hand_tool": (18, 24), "grip": (24, 28), "spout": (28, 29),
          "brand_name": (29, 33), "nutrition_facts": (33, 37), "bar_code": (37, 41)}
# (class, has_grip, has_spout, has_brand, has_nutrition, has_barcode) rows shaped like kp_configs/ycbv_kp_config.csv:2-22
YCBV_LIKE = [("cylinder_like", 0, 0, 1, 0, 1), ("box_like", 0, 0, 1, 1, 1), ("box_like", 0, 0, 1, 1, 1),
             ("cylinder_like", 0, 0, 1, 1, 1), ("cylinder_like", 0, 0, 1, 1, 1), ("cylinder_like", 0, 0, 1, 1, 1),
             ("box_like", 0, 0, 1, 1, 1), ("box_like", 0, 0, 1, 1, 1), ("box_like", 0, 0, 1, 0, 1), ("hand_tool", 0, 0, 0, 0, 0),
             ("cylinder_like", 1, 1, 1, 0, 0), ("cylinder_like", 0, 0, 1, 0, 0), ("cylinder_like", 0, 0, 0, 0, 0),
             ("cylinder_like", 1, 0, 0, 0, 0), ("hand_tool", 1, 0, 0, 0, 0), ("box_like", 0, 0, 0, 0, 0),
             ("hand_tool", 1, 0, 0, 0, 0), ("hand_tool", 0, 0, 0, 0, 0), ("hand_tool", 1, 0, 0, 0, 0),
             ("hand_tool", 1, 0, 0, 0, 0), ("box_like", 0, 0, 0, 0, 0)]

K_YCBV = np.array([[1066.778, 0.0, 312.9869], [0.0, 1067.487, 241.3109], [0.0, 0.0, 1.0]])


def class_mask(obj_class_idx):
    cls, grip, spout, brand, nutr, bar = YCBV_LIKE[obj_class_idx % len(YCBV_LIKE)]
    m = np.zeros(NUM_KP, bool)
    m[GROUPS[cls][0]:GROUPS[cls][1]] = True
    for flag, name in ((grip, "grip"), (spout, "spout"), (brand, "brand_name"), (nutr, "nutrition_facts"), (bar, "bar_code")):
        if flag:
            m[GROUPS[name][0]:GROUPS[name][1]] = True
    return m


def random_rotation(rng):
    A = rng.standard_normal((3, 3))
    Q, R = np.linalg.qr(A)
    Q = Q @ np.diag(np.sign(np.diag(R)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] *= -1
    return Q


def make_texture(rng, H=480, W=640):
    """Smooth random uint8 image (low-pass filtered noise) so bilinear sampling is non-trivial."""
    img = rng.uniform(0, 1, (H // 8 + 2, W // 8 + 2, 3))
    img = np.kron(img, np.ones((8, 8, 1)))[:H, :W]
    k = np.ones(5) / 5
    for ax in (0, 1):
        img = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, img)
    img += rng.normal(0, 0.02, img.shape)
    # C-contiguous like a frame from cv2.imread (bop.py:438): apply_along_axis leaves a permuted layout behind, and a strided 0.9 MB copy
    # per network call (0.7-2.5 ms on the host) is not something the reference's frames ever cost
    return np.ascontiguousarray((np.clip(img, 0, 1) * 255).astype(np.uint8))


def make_frame(rng, n_obj=8, K=K_YCBV, H=480, W=640, noise=0.01, outlier_frac=0.0, with_image=True, z_range=(600.0, 1200.0), class_rows=None, off_centre=(0.25, 0.18)):
    """One synthetic single-view frame.  Returns a dict with image, K, per-object class masks, model
    keypoints [L,41,3] (mm), GT poses T_OtoC [L,4,4], boxes [L,4] xyxy, K_bbox [L,3,3], measured uv
    [L,41,2] (NDC, float32), cov [L,41,2,2] (float32), visibility/validity masks [L,41] and diameters.
    z_range: object depth in mm (closer objects = larger boxes); class_rows: indices into YCBV_LIKE to draw the keypoint classes from."""
    model_kps = np.zeros((n_obj, NUM_KP, 3), np.float32)
    masks = np.zeros((n_obj, NUM_KP), bool)
    poses = np.zeros((n_obj, 4, 4))
    boxes = np.zeros((n_obj, 4), np.float32)
    K_bbox = np.zeros((n_obj, 3, 3))
    uv = np.zeros((n_obj, NUM_KP, 2), np.float32)
    cov = np.zeros((n_obj, NUM_KP, 2, 2), np.float32)
    diam = np.zeros(n_obj)
    for o in range(n_obj):
        masks[o] = class_mask(int(rng.integers(0, len(YCBV_LIKE))) if class_rows is None else int(class_rows[int(rng.integers(0, len(class_rows)))]))
        ext = rng.uniform(40, 100, 3)
        model_kps[o] = (rng.uniform(-1, 1, (NUM_KP, 3)) * ext).astype(np.float32)
        diam[o] = 2 * np.linalg.norm(ext)
        for _ in range(100):
            T = np.eye(4)
            T[:3, :3] = random_rotation(rng)
            z = rng.uniform(z_range[0], z_range[1])
            T[:3, 3] = [rng.uniform(-off_centre[0], off_centre[0]) * z, rng.uniform(-off_centre[1], off_centre[1]) * z, z]
            pts = model_kps[o].astype(np.float64)
            pc = pts @ T[:3, :3].T + T[:3, 3]
            px = pc @ K.T
            px = px[:, :2] / px[:, 2:3]
            x1, y1 = px.min(0) - rng.uniform(5, 15, 2)
            x2, y2 = px.max(0) + rng.uniform(5, 15, 2)
            if x1 >= 0 and y1 >= 0 and x2 <= W - 1 and y2 <= H - 1 and x2 - x1 >= 40 and y2 - y1 >= 40:
                break
        poses[o] = T
        boxes[o] = [x1, y1, x2, y2]
        K_bbox[o] = fix_K_for_bbox_ndc(K, boxes[o].astype(np.float64))
        uv_gt, _ = project_ndc(K_bbox[o], T, model_kps[o].astype(np.float64))
        uv_o = uv_gt + rng.normal(0, noise, uv_gt.shape)
        if outlier_frac > 0:
            out = rng.random(NUM_KP) < outlier_frac
            uv_o[out] = rng.uniform(-0.8, 0.8, (int(out.sum()), 2))
        uv[o] = uv_o.astype(np.float32)
        s = max(noise, 1e-3)
        A = rng.normal(0, 0.3, (NUM_KP, 2, 2)) + np.eye(2)
        cov[o] = ((A @ A.transpose(0, 2, 1)) * s * s).astype(np.float32)
    frame = {"K": K.copy(), "model_kps": model_kps, "model_kps_masks": masks, "T_OtoC": poses, "boxes": boxes,
             "K_bbox": K_bbox, "uv": uv, "cov": cov, "diameter": diam, "obj_ids": list(range(1, n_obj + 1))}
    if with_image:
        frame["image"] = make_texture(rng, H, W)
    return frame


# T-LESS (BASELINE configs[3]): 720 x 540 Primesense frames (bop.py reads K per image from scene_camera.json; this is the dataset's nominal test camera), objects
# with 8 (box-like) or 10 (cylinder-like) keypoints and no texture groups (kp_configs/tless_kp_config.csv:2-31)
K_TLESS = np.array([[1075.65091572, 0.0, 360.0], [0.0, 1073.90347929, 270.0], [0.0, 0.0, 1.0]])
TLESS_ROWS = (12, 15)              # YCBV_LIKE rows with the bare class groups: cylinder_like (10 keypoints), box_like (8)


def make_frame_tless(rng, n_small=5, n_mid=2, n_big=1, noise=0.01, outlier_frac=0.0, with_image=True):
    """One T-LESS-shaped frame: 720 x 540, n_small objects at 600-1100 mm (boxes <= 256 px: one RoIAlign sample per bin), n_mid at 300-480 mm (256-512 px: two
    samples per bin and axis, torchvision's adaptive ceil(roi / 256)) and n_big at 170-240 mm (> 512 px where the draw fits the frame: three) -- the box sizes
    saved T-LESS detections produce (evaluate.py:104-125)."""
    parts = []
    for n, zr in ((n_small, (600.0, 1100.0)), (n_mid, (300.0, 480.0))):
        if n > 0:
            parts.append(make_frame(rng, n, K_TLESS, 540, 720, noise, outlier_frac, with_image and not parts, z_range=zr, class_rows=TLESS_ROWS))
    for _ in range(n_big):                                   # one at a time: the depth is searched so that the box's longer side lands in (512, 538] inside the frame
        z = 220.0
        for _try in range(60):
            p = make_frame(rng, 1, K_TLESS, 540, 720, noise, outlier_frac, with_image and not parts, z_range=(z, z), class_rows=TLESS_ROWS, off_centre=(0.01, 0.01))
            b = p["boxes"][0]
            side = max(b[2] - b[0], b[3] - b[1])
            if 512 < side <= 538 and b[0] >= 0 and b[1] >= 0 and b[2] <= 719 and b[3] <= 539:
                break
            z *= float(side) / 526.0
        parts.append(p)
    out = dict(parts[0])
    for k in ("model_kps", "model_kps_masks", "T_OtoC", "boxes", "K_bbox", "uv", "cov", "diameter"):
        out[k] = np.concatenate([p[k] for p in parts])
    out["obj_ids"] = list(range(1, len(out["boxes"]) + 1))
    return out


def frame_to_ba_problem(frame, init_poses, use_cov=True):
    """Flat SoA of ObjectSLAM.optimize's single-view graph (object_slam.py:746-839): one fixed camera at
    identity, one free vertex per object, one edge per valid keypoint."""
    L = len(frame["boxes"])
    e_cam, e_obj, e_k, e_p, e_uv, e_info = [], [], [], [], [], []
    for o in range(L):
        Kb = frame["K_bbox"][o]
        for k in np.nonzero(frame["model_kps_masks"][o])[0]:
            e_cam.append(0)
            e_obj.append(o)
            e_k.append([Kb[0, 0], Kb[1, 1], Kb[0, 2], Kb[1, 2]])
            e_p.append(frame["model_kps"][o, k].astype(np.float64))
            e_uv.append(frame["uv"][o, k].astype(np.float64))
            Om = np.linalg.inv(frame["cov"][o, k].astype(np.float64)) if use_cov else np.eye(2)
            e_info.append([Om[0, 0], Om[0, 1], Om[1, 1]])
    return {"cam_T": np.eye(4)[None, :3, :], "cam_fixed": np.array([1], np.uint8),
            "obj_T": np.asarray(init_poses, np.float64)[:, :3, :], "obj_fixed": np.zeros(L, np.uint8),
            "edge_cam": np.array(e_cam, np.int32), "edge_obj": np.array(e_obj, np.int32), "edge_camk": np.array(e_k),
            "edge_p": np.array(e_p), "edge_uv": np.array(e_uv), "edge_info": np.array(e_info),
            "edge_inlier": np.ones(len(e_cam), np.uint8)}


def _perturb_pose(T34, rng, rot, trans):
    """Left-multiply a 3x4 pose by a small random rotation (axis-angle ~ N(0, rot)) and add N(0, trans) mm."""
    w = rng.normal(0, rot, 3)
    th = np.linalg.norm(w)
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + (np.sin(th) / th) * Wx + ((1 - np.cos(th)) / (th * th)) * (Wx @ Wx) if th > 1e-12 else np.eye(3)
    out = np.array(T34, dtype=np.float64)
    out[:, :3] = R @ out[:, :3]
    out[:, 3] = R @ out[:, 3] + rng.normal(0, trans, 3)
    return out


def make_pose_graph(rng, n_cam, n_obj, kp_per_obj=10, noise_px=0.5, miss=0.15, outlier_frac=0.05, rot=5e-4, trans=0.3):
    """Flat SoA of ObjectSLAM.optimize's GLOBAL graph (object_slam.py:746-839) for a synthetic sequence: n_cam views on a
    smooth arc looking at n_obj objects with kp_per_obj keypoints each; camera 0 fixed (the gauge, :774), every other
    camera and every object free and perturbed from the ground truth; each object is missed in `miss` of the views and
    `outlier_frac` of the measurements are gross outliers.  Also returns the ground truth as "cam_gt" / "obj_gt"."""
    k = np.array([600.0, 600.0, 320.0, 240.0])
    cam_gt = np.zeros((n_cam, 3, 4))
    for c in range(n_cam):
        s = c / max(n_cam - 1, 1) - 0.5
        ang = 0.5 * s
        cam_gt[c, :, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        cam_gt[c, :, 3] = [-300 * s, rng.uniform(-20, 20), rng.uniform(-20, 20)]
    obj_gt = np.zeros((n_obj, 3, 4))
    pts = rng.uniform(-60, 60, (n_obj, kp_per_obj, 3))
    for o in range(n_obj):
        obj_gt[o, :, :3] = random_rotation(rng)
        obj_gt[o, :, 3] = [rng.uniform(-250, 250), rng.uniform(-150, 150), rng.uniform(800, 1100)]
    e_cam, e_obj, e_p, e_uv = [], [], [], []
    for c in range(n_cam):
        for o in range(n_obj):
            if rng.random() < miss:
                continue
            pw = pts[o] @ obj_gt[o, :, :3].T + obj_gt[o, :, 3]
            pc = pw @ cam_gt[c, :, :3].T + cam_gt[c, :, 3]
            uv = np.c_[k[0] * pc[:, 0] / pc[:, 2] + k[2], k[1] * pc[:, 1] / pc[:, 2] + k[3]] + rng.normal(0, noise_px, (kp_per_obj, 2))
            out = rng.random(kp_per_obj) < outlier_frac
            uv[out] = rng.uniform(0, 480, (int(out.sum()), 2))
            e_cam += [c] * kp_per_obj
            e_obj += [o] * kp_per_obj
            e_p.append(pts[o])
            e_uv.append(uv)
    E = len(e_cam)
    cam_fixed = np.zeros(n_cam, np.uint8)
    cam_fixed[0] = 1
    cam_init = cam_gt.copy()
    for c in range(1, n_cam):
        cam_init[c] = _perturb_pose(cam_gt[c], rng, rot, trans)
    obj_init = np.stack([_perturb_pose(T, rng, rot, trans) for T in obj_gt])
    return {"cam_T": cam_init, "cam_fixed": cam_fixed, "obj_T": obj_init, "obj_fixed": np.zeros(n_obj, np.uint8),
            "edge_cam": np.array(e_cam, np.int32), "edge_obj": np.array(e_obj, np.int32), "edge_camk": np.tile(k, (E, 1)),
            "edge_p": np.concatenate(e_p), "edge_uv": np.concatenate(e_uv),
            "edge_info": np.tile([1.0 / noise_px ** 2, 0, 1.0 / noise_px ** 2], (E, 1)), "edge_inlier": np.ones(E, np.uint8),
            "cam_gt": cam_gt, "obj_gt": obj_gt}


def make_slam_sequence(rng, n_views=60, n_obj=8, sym_every=3, miss=0.05, vis_drop=0.05, with_image=True):
    """A synthetic SLAM sequence (BASELINE configs[2]): n_obj objects in the world frame (= first camera), a smooth camera arc,
    per view the arguments of ObjectSLAM.process_view (object_slam.py:327-328) -- obj_ids, xyxy boxes, model keypoints [L,41,3],
    class masks, ground-truth visibility and the ground-truth NDC keypoints the reference's --debug_gt_kp mode consumes (:1129-1131).
    Every sym_every-th object is symmetric, i.e. goes through the second network pass with rendered priors (:486-519)."""
    K = K_YCBV
    objs = list(range(1, n_obj + 1))
    T_OtoG, kps, mmask = {}, {}, {}
    for o in objs:
        T = np.eye(4)
        T[:3, :3] = random_rotation(rng)
        T[:3, 3] = [rng.uniform(-260, 260), rng.uniform(-160, 160), rng.uniform(850, 1150)]
        T_OtoG[o] = T
        mmask[o] = class_mask(int(rng.integers(0, len(YCBV_LIKE))))
        kps[o] = (rng.uniform(-1, 1, (NUM_KP, 3)) * rng.uniform(35, 70, 3)).astype(np.float32)
    mesh_db = {o: {"diameter": float(2 * np.abs(kps[o]).max()), "is_symmetric": bool(sym_every and o % sym_every == 0)} for o in objs}
    image = make_texture(rng) if with_image else None
    views = []
    for v in range(n_views):
        s = v / max(n_views - 1, 1)
        ang = 0.3 * s
        T_GtoC = np.eye(4)
        if v > 0:
            T_GtoC[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
            T_GtoC[:3, 3] = [-240 * s, rng.uniform(-8, 8), rng.uniform(-8, 8) + 50 * s]
        ids, boxes, mk, mm, vis, uvg = [], [], [], [], [], []
        for o in objs:
            if v > 0 and rng.random() < miss:
                continue
            T_OtoC = T_GtoC @ T_OtoG[o]
            pc = kps[o].astype(np.float64) @ T_OtoC[:3, :3].T + T_OtoC[:3, 3]
            px = pc @ K.T
            px = px[:, :2] / px[:, 2:3]
            sel = px[mmask[o]]
            bbox = np.array([sel[:, 0].min() - 10, sel[:, 1].min() - 10, sel[:, 0].max() + 10, sel[:, 1].max() + 10])
            uvw = pc @ fix_K_for_bbox_ndc(K, bbox).T
            ids.append(o)
            boxes.append(bbox)
            mk.append(kps[o])
            mm.append(mmask[o])
            vis.append(mmask[o] & (rng.random(NUM_KP) >= vis_drop))
            uvg.append((uvw[:, :2] / uvw[:, 2:3]).astype(np.float32))
        views.append({"view_id": v, "K": K.copy(), "image": image, "obj_ids": np.array(ids), "bboxes": np.array(boxes, np.float64),
                      "model_kps": np.array(mk, np.float32), "model_kps_masks": np.array(mm, bool), "kp_masks": np.array(vis, bool),
                      "uv_gt": np.array(uvg, np.float32), "T_GtoC_gt": T_GtoC})
    return {"mesh_db": mesh_db, "views": views, "T_OtoG_gt": T_OtoG}

"""Builds a tiny BOP-format dataset tree (YCB-V or T-LESS shaped) from a seed -- test data for the N3 rows.

The layout is the one the reference reads (lib/datasets/bop.py:55-248, lib/utils/mesh_database.py:7-45,
lib/utils/utils.py:481-569):

    <bop_root>/<dset>/                    data_root
        keyframe.txt | all_target_tless.json
        kp_info/obj_%06d_kp_info.json     {"keypoints": {name: {"pos_mean": [3], "pos_cov": [9]}}, "view_pose": [16]}
        models_bop-compat/models_info.json | models_cad/models_info.json          (symmetries)
        models_bop-compat_eval/ | models_eval/   models_info.json + obj_%06d.ply  (evaluation meshes)
        <split>/%06d/scene_camera.json scene_gt.json scene_gt_info.json rgb/%06d.png
    <bop_root>/<dset>/offsets.txt, <bop_root>/saved_detections/*.pkl             (only when asked for)

Used by tests/golden/make_bop_golden.py (which runs the REFERENCE's reader on the tree) and by the tests (which run
this repository's reader on the identical tree).  Nothing here comes from the reference: the scene is synthetic.
"""
import json
import os
import struct

import numpy as np
from PIL import Image

from suo_slam_amd import kp_config

K_YCBV = [1066.778, 0.0, 312.9869, 0.0, 1067.487, 241.3109, 0.0, 0.0, 1.0]


def _rot(rng):
    A = rng.standard_normal((3, 3))
    Q, R = np.linalg.qr(A)
    Q = Q @ np.diag(np.sign(np.diag(R)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] *= -1
    return Q


def _write_ply(path, pts, faces, binary):
    with open(path, "wb") as f:
        fmt = "binary_little_endian" if binary else "ascii"
        hdr = f"ply\nformat {fmt} 1.0\ncomment synthetic\nelement vertex {len(pts)}\nproperty float x\nproperty float y\nproperty float z\n"
        hdr += f"element face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n"
        f.write(hdr.encode())
        if binary:
            for p in pts:
                f.write(struct.pack("<fff", *p))
            for t in faces:
                f.write(struct.pack("<Biii", 3, *t))
        else:
            for p in pts:
                f.write((" ".join(repr(float(np.float32(v))) for v in p) + "\n").encode())
            for t in faces:
                f.write(("3 " + " ".join(str(int(v)) for v in t) + "\n").encode())


def build(bop_root, dset="ycbv", seed=0, n_scenes=2, n_views=3, objs_per_view=3, img_hw=(480, 640)):
    """Returns a dict describing what was written (scene ids, view ids, object ids per view, model extents)."""
    rng = np.random.default_rng(seed)
    table = kp_config.TABLES[dset]
    n_obj_total = len(table)
    split = "test" if dset == "ycbv" else "test_primesense"
    data_root = os.path.join(bop_root, dset)
    os.makedirs(os.path.join(data_root, "kp_info"), exist_ok=True)
    sym_models = "models_bop-compat" if dset == "ycbv" else "models_cad"
    eval_models = "models_bop-compat_eval" if dset == "ycbv" else "models_eval"
    os.makedirs(os.path.join(data_root, sym_models), exist_ok=True)
    os.makedirs(os.path.join(data_root, eval_models), exist_ok=True)

    # ---- models: keypoints, symmetries, evaluation meshes -------------------------------------------
    models_info, kp3d = {}, {}
    for oid in range(1, n_obj_total + 1):
        names = kp_config.kp_list_of(dset, oid)
        ext = rng.uniform(30, 90, 3)
        pts = rng.uniform(-1, 1, (len(names), 3)) * ext
        kp3d[oid] = pts
        view_pose = np.eye(4)
        view_pose[:3, :3] = _rot(rng)
        view_pose[:3, 3] = [0, 0, 700]
        info = {"keypoints": {n: {"pos_mean": pts[i].tolist(), "pos_cov": (np.eye(3) * 0.5).ravel().tolist()} for i, n in enumerate(names)},
                "view_pose": view_pose.ravel().tolist()}
        with open(os.path.join(data_root, "kp_info", f"obj_{oid:06d}_kp_info.json"), "w") as f:
            json.dump(info, f)
        mi = {"diameter": float(2 * np.linalg.norm(ext)), "min_x": -ext[0], "min_y": -ext[1], "min_z": -ext[2],
              "size_x": 2 * ext[0], "size_y": 2 * ext[1], "size_z": 2 * ext[2]}
        if oid % 5 == 0:
            mi["symmetries_discrete"] = [np.diag([-1.0, -1.0, 1.0, 1.0]).ravel().tolist()]
        if oid % 7 == 0:
            mi["symmetries_continuous"] = [{"axis": [0, 0, 1], "offset": [0, 0, 0]}]
        models_info[str(oid)] = mi
        nv = int(rng.integers(60, 200))
        verts = (rng.uniform(-1, 1, (nv, 3)) * ext).astype(np.float32)
        faces = rng.integers(0, nv, (nv // 2, 3))
        _write_ply(os.path.join(data_root, eval_models, f"obj_{oid:06d}.ply"), verts, faces, binary=bool(oid % 2))
    for d in (sym_models, eval_models):
        with open(os.path.join(data_root, d, "models_info.json"), "w") as f:
            json.dump(models_info, f)

    # ---- scenes ------------------------------------------------------------------------------------
    H, W = img_hw
    desc = {"dset": dset, "split": split, "data_root": data_root, "scenes": {}, "kp3d": kp3d, "models_info": models_info}
    keyframes, targets = [], []
    scene_ids = [48 + 2 * s for s in range(n_scenes)] if dset == "ycbv" else [1 + 3 * s for s in range(n_scenes)]
    for scene_id in scene_ids:
        sdir = os.path.join(data_root, split, f"{scene_id:06d}")
        os.makedirs(os.path.join(sdir, "rgb"), exist_ok=True)
        cam, gt, gt_info = {}, {}, {}
        scene_objs = rng.choice(np.arange(1, n_obj_total + 1), objs_per_view + 1, replace=False).tolist()
        view_ids = sorted(rng.choice(np.arange(1, 400), n_views + 1, replace=False).tolist())
        desc["scenes"][scene_id] = {}
        for vi, view_id in enumerate(view_ids):
            Rw, tw = _rot(rng), rng.standard_normal(3) * 100
            cam[str(view_id)] = {"cam_K": K_YCBV, "depth_scale": 0.1, "cam_R_w2c": Rw.ravel().tolist(), "cam_t_w2c": tw.tolist()}
            objs = scene_objs[:objs_per_view] if vi % 2 == 0 else scene_objs[1:]
            gts, infos = [], []
            for k, oid in enumerate(objs):
                R = _rot(rng)
                t = np.array([rng.uniform(-150, 150), rng.uniform(-100, 100), rng.uniform(650, 1100)])
                Kmat = np.array(K_YCBV).reshape(3, 3)
                uv = (Kmat @ (kp3d[oid] @ R.T + t).T).T
                uv = uv[:, :2] / uv[:, 2:3]
                x0, y0 = np.floor(uv.min(0) - rng.uniform(2, 12, 2))
                x1, y1 = np.ceil(uv.max(0) + rng.uniform(2, 12, 2))
                if k == 1 and vi == 0:
                    x0, y0, x1, y1 = x0 + 30, y0 + 25, x0 + 36, y0 + 33          # a tiny box: the 10 px minimum applies, kps fall outside
                gts.append({"cam_R_m2c": R.ravel().tolist(), "cam_t_m2c": t.tolist(), "obj_id": int(oid)})
                vis = 0.05 if (k == 2 and vi == 1) else float(rng.uniform(0.3, 1.0))
                infos.append({"bbox_obj": [int(x0), int(y0), int(x1 - x0), int(y1 - y0)], "bbox_visib": [int(x0), int(y0), int(x1 - x0), int(y1 - y0)],
                              "px_count_all": 1000, "px_count_valid": 1000, "px_count_visib": int(1000 * vis), "visib_fract": vis})
                if not (vi == 1 and k == 0):
                    targets.append({"im_id": int(view_id), "inst_count": 1, "obj_id": int(oid), "scene_id": int(scene_id)})
            gt[str(view_id)], gt_info[str(view_id)] = gts, infos
            img = (rng.uniform(0, 1, (H // 16 + 1, W // 16 + 1, 3)) * 255).astype(np.uint8)
            img = np.kron(img, np.ones((16, 16, 1), np.uint8))[:H, :W]
            Image.fromarray(img).save(os.path.join(sdir, "rgb", f"{view_id:06d}.png"))
            if vi < n_views:                                                   # the last view is not a keyframe
                keyframes.append(f"{scene_id:04d}/{view_id:06d}")
            desc["scenes"][scene_id][view_id] = objs
        for name, obj in (("scene_camera.json", cam), ("scene_gt.json", gt), ("scene_gt_info.json", gt_info)):
            with open(os.path.join(sdir, name), "w") as f:
                json.dump(obj, f)
    if dset == "ycbv":
        with open(os.path.join(data_root, "keyframe.txt"), "w") as f:
            f.write("\n".join(keyframes) + "\n")
    else:
        with open(os.path.join(data_root, "all_target_tless.json"), "w") as f:
            json.dump(targets, f)
    return desc


def write_saved_detections(bop_root, desc, dataset, seed=0, trans_noise_mm=3.0, box_jitter_px=2.0, drop_every=5):
    """PoseCNN-format (YCB-V) saved detections for the tree: ground-truth poses perturbed by ``trans_noise_mm``,
    ground-truth boxes jittered, every ``drop_every``-th detection missing.  ``dataset`` is any reader exposing
    scene_ids / view_ids / obj_ids / get_obj_pose / data[...]["objects"][...]["bbox"].  Returns the offsets used."""
    import pickle
    assert desc["dset"] == "ycbv"
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(bop_root, "saved_detections"), exist_ok=True)
    offsets = {i: (rng.standard_normal(3) * 10).round(3) for i in range(1, 22)}
    with open(os.path.join(desc["data_root"], "offsets.txt"), "w") as f:
        f.write("\n".join(f"{i:02d} {json.dumps(o.tolist())}" for i, o in offsets.items()) + "\n")
    results, k = {}, 0
    for s in dataset.scene_ids():
        for v in dataset.view_ids(s):
            rois, poses = [], []
            for o in dataset.obj_ids(s, v):
                k += 1
                if k % drop_every == 0:
                    continue
                T = dataset.get_obj_pose(s, v, o)
                x, y, w, h = dataset.data[s][v]["objects"][o]["bbox"]
                box = np.array([x, y, x + w, y + h], np.float64) + rng.uniform(-box_jitter_px, box_jitter_px, 4)
                rois.append([0, o, *box, 1.0])
                R = T[:3, :3]
                qw = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
                if qw > 1e-3:
                    q = np.array([qw, (R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw)])
                else:                                                  # 180-degree rotations: take the axis from R + I
                    A = (R + np.eye(3)) / 2
                    i = int(np.argmax(np.diag(A)))
                    ax = A[:, i] / np.sqrt(A[i, i])
                    q = np.array([0.0, *ax])
                t_bop = T[:3, 3] + rng.standard_normal(3) * trans_noise_mm
                t_orig = t_bop + R @ offsets[o]                        # the loader applies Trans(-offset)
                poses.append([*q, *(t_orig / 1000.0)])
            if rois:
                results[f"{s}/{v}"] = {"rois": np.array(rois, np.float32), "poses": np.array(poses, np.float32)}
    with open(os.path.join(bop_root, "saved_detections", "ycbv_posecnn.pkl"), "wb") as f:
        pickle.dump(results, f)
    return offsets


def write_saved_detections_pix2pose(bop_root, desc, dataset, seed=0, trans_noise_mm=3.0, box_jitter_px=2.0, drop_every=5):
    """Pix2Pose / RetinaNet-format (T-LESS) saved detections (lib/utils/utils.py:538-569): per "scene/view" ``rois[n,4]`` stored as
    (y1, x1, y2, x2) -- the loader swaps them to xyxy (:557-561) --, ``labels_txt[n]`` ending in "_<obj id>", ``poses[n]`` = 4x4
    [R|t] with t in metres.  Ground-truth poses perturbed by ``trans_noise_mm``, boxes jittered, every ``drop_every``-th missing."""
    import pickle
    assert desc["dset"] == "tless"
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(bop_root, "saved_detections"), exist_ok=True)
    results, k = {}, 0
    for s in dataset.scene_ids():
        for v in dataset.view_ids(s):
            rois, poses, labels = [], [], []
            for o in dataset.obj_ids(s, v):
                k += 1
                if k % drop_every == 0:
                    continue
                T = np.array(dataset.get_obj_pose(s, v, o), np.float64)
                x, y, w, h = dataset.data[s][v]["objects"][o]["bbox"]
                x1, y1, x2, y2 = np.array([x, y, x + w, y + h], np.float64) + rng.uniform(-box_jitter_px, box_jitter_px, 4)
                rois.append([y1, x1, y2, x2])
                P = np.eye(4)
                P[:3, :3] = T[:3, :3]
                P[:3, 3] = (T[:3, 3] + rng.standard_normal(3) * trans_noise_mm) / 1000.0
                poses.append(P)
                labels.append(f"obj_{o:02d}")
            if rois:
                results[f"{s}/{v}"] = {"rois": np.array(rois, np.float64), "poses": poses, "labels_txt": labels}
    with open(os.path.join(bop_root, "saved_detections", "tless_pix2pose_retinanet_siso_top1.pkl"), "wb") as f:
        pickle.dump(results, f)


def build_sequence(bop_root, seed=0, n_views=12, n_objs=5, img_hw=(480, 640), dset="ycbv"):
    """A YCB-V-shaped tree with ONE geometrically consistent scene for the SLAM mode: fixed object poses in the world,
    a camera moving on a smooth arc, every frame a keyframe.  Reuses the model / kp_info files of ``build``."""
    desc = build(bop_root, dset=dset, seed=seed, n_scenes=1, n_views=1)
    rng = np.random.default_rng(seed + 1000)
    data_root, split = desc["data_root"], desc["split"]
    import shutil
    shutil.rmtree(os.path.join(data_root, split))
    scene_id = 59
    sdir = os.path.join(data_root, split, f"{scene_id:06d}")
    os.makedirs(os.path.join(sdir, "rgb"))
    obj_ids = rng.choice(np.arange(1, len(kp_config.TABLES[dset]) + 1), n_objs, replace=False).tolist()
    T_OtoW = {}
    for k, o in enumerate(obj_ids):
        T = np.eye(4)
        T[:3, :3] = _rot(rng)
        T[:3, 3] = [(k - (n_objs - 1) / 2) * 170.0, rng.uniform(-60, 60), rng.uniform(-60, 60)]
        T_OtoW[o] = T
    Kmat = np.array(K_YCBV).reshape(3, 3)
    H, W = img_hw
    cam, gt, gt_info, keyframes, targets = {}, {}, {}, [], []
    for vi in range(n_views):
        view_id = 1 + 7 * vi
        ang = np.deg2rad(-20 + 40 * vi / max(1, n_views - 1))
        Rc = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        T_WtoC = np.eye(4)
        T_WtoC[:3, :3] = Rc
        T_WtoC[:3, 3] = [0, 0, 1000.0 + 10 * vi]
        cam[str(view_id)] = {"cam_K": K_YCBV, "depth_scale": 0.1, "cam_R_w2c": Rc.ravel().tolist(), "cam_t_w2c": T_WtoC[:3, 3].tolist()}
        gts, infos = [], []
        for o in obj_ids:
            T = T_WtoC @ T_OtoW[o]
            uv = (Kmat @ (desc["kp3d"][o] @ T[:3, :3].T + T[:3, 3]).T).T
            uv = uv[:, :2] / uv[:, 2:3]
            x0, y0 = np.floor(uv.min(0) - 6)
            x1, y1 = np.ceil(uv.max(0) + 6)
            gts.append({"cam_R_m2c": T[:3, :3].ravel().tolist(), "cam_t_m2c": T[:3, 3].tolist(), "obj_id": int(o)})
            infos.append({"bbox_obj": [int(x0), int(y0), int(x1 - x0), int(y1 - y0)], "bbox_visib": [int(x0), int(y0), int(x1 - x0), int(y1 - y0)],
                          "px_count_all": 1000, "px_count_valid": 1000, "px_count_visib": 900, "visib_fract": 0.9})
            targets.append({"im_id": int(view_id), "inst_count": 1, "obj_id": int(o), "scene_id": int(scene_id)})
        gt[str(view_id)], gt_info[str(view_id)] = gts, infos
        img = (rng.uniform(0, 1, (H // 16 + 1, W // 16 + 1, 3)) * 255).astype(np.uint8)
        Image.fromarray(np.kron(img, np.ones((16, 16, 1), np.uint8))[:H, :W]).save(os.path.join(sdir, "rgb", f"{view_id:06d}.png"))
        keyframes.append(f"{scene_id:04d}/{view_id:06d}")
    for name, obj in (("scene_camera.json", cam), ("scene_gt.json", gt), ("scene_gt_info.json", gt_info)):
        with open(os.path.join(sdir, name), "w") as f:
            json.dump(obj, f)
    if dset == "ycbv":
        with open(os.path.join(data_root, "keyframe.txt"), "w") as f:
            f.write("\n".join(keyframes) + "\n")
    else:
        with open(os.path.join(data_root, "all_target_tless.json"), "w") as f:
            json.dump(targets, f)
    desc["scenes"] = {scene_id: {1 + 7 * vi: obj_ids for vi in range(n_views)}}
    desc["T_OtoW"] = T_OtoW
    return desc

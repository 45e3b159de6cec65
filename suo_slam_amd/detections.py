"""Saved-detection box / pose formats on the input side of the hot path.

Host mirrors of the reference's loaders (same names, same returned dict of parallel lists):

    load_posecnn_results(bop_root)     lib/utils/utils.py:481-536   YCB-V PoseCNN detections
    load_pix2pose_results(bop_root)    lib/utils/utils.py:538-569   T-LESS Pix2Pose / RetinaNet detections
    build_detection_map(...)           evaluate.py:106-124          scene -> view -> obj_id -> detection index

Returned keys: ``scene_ids, view_ids, scores, obj_ids, poses, bboxes``; boxes are xyxy pixels in the 640x480
image (what ObjectSLAM.process_view / suo_net_forward take), poses are object-to-camera with mm translations.
Pure host data plumbing -- nothing here touches the GPU.
"""
from __future__ import annotations

import json
import os
import pickle

import numpy as np

_KEYS = ("scene_ids", "view_ids", "scores", "obj_ids", "poses", "bboxes")


def _quat_wxyz_to_R(q):
    q = np.asarray(q, np.float64)
    w, x, y, z = (q / np.linalg.norm(q)).tolist()
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _split_key(k):
    scene, view = k.split("/")
    return int(scene), int(view)


def load_posecnn_results(bop_root):
    """``saved_detections/ycbv_posecnn.pkl``: per "scene/view" a dict with ``rois[n,>=6]`` (col 1 = class id, cols
    2..5 = xyxy box) and ``poses[n,7]`` = quaternion wxyz + translation in metres, expressed for the ORIGINAL YCB
    model frames.  ``ycbv/offsets.txt`` ("NN [x, y, z]" per line, mm) moves them to the BOP model frames:
    T_bop = T_orig * Trans(-offset).  The "score" is the class id, as in the reference (utils.py:519)."""
    with open(os.path.join(bop_root, "saved_detections/ycbv_posecnn.pkl"), "rb") as f:
        results = pickle.load(f)
    offsets = {}
    with open(os.path.join(bop_root, "ycbv/offsets.txt"), "r") as f:
        for line in f.read().strip().split("\n"):
            offsets[int(line[:2])] = np.array(json.loads(line[3:]))
    data = {k: [] for k in _KEYS}
    for key, res in results.items():
        scene_id, view_id = _split_key(key)
        rois, poses = res["rois"], res["poses"]
        for n in range(rois.shape[0]):
            obj_id = int(rois[n, 1])
            T = np.zeros((3, 4))
            T[:, :3] = _quat_wxyz_to_R(poses[n][:4])
            T[:, 3] = np.asarray(poses[n][4:7], np.float64) * 1000.0          # m -> mm
            T[:, 3] -= T[:, :3] @ offsets[obj_id]                              # compose with Trans(-offset)
            data["scene_ids"].append(scene_id)
            data["view_ids"].append(view_id)
            data["scores"].append(rois[n, 1])
            data["obj_ids"].append(obj_id)
            data["bboxes"].append(rois[n, 2:6])
            data["poses"].append(T)
    return data


def load_pix2pose_results(bop_root):
    """``saved_detections/tless_pix2pose_retinanet_siso_top1.pkl``: ``rois[n,4]`` stored as (y1, x1, y2, x2) and
    swapped to xyxy here (utils.py:557-561), ``labels_txt[n]`` = "..._<obj id>", ``poses[n]`` = [R|t] with t in m."""
    with open(os.path.join(bop_root, "saved_detections/tless_pix2pose_retinanet_siso_top1.pkl"), "rb") as f:
        results = pickle.load(f)
    data = {k: [] for k in _KEYS}
    for key, res in results.items():
        scene_id, view_id = _split_key(key)
        rois = np.asarray(res["rois"])
        xyxy = rois[:, [1, 0, 3, 2]].astype(np.float32)
        for n in range(rois.shape[0]):
            T = np.array(res["poses"][n], np.float64)
            T[:3, 3] *= 1000.0
            data["scene_ids"].append(scene_id)
            data["view_ids"].append(view_id)
            data["scores"].append(rois[n, 1])
            data["obj_ids"].append(int(res["labels_txt"][n].split("_")[-1]))
            data["bboxes"].append(xyxy[n])
            data["poses"].append(T)
    return data


def build_detection_map(detections, targets=None):
    """scene_id -> view_id -> obj_id -> index into the detection lists (evaluate.py:106-124).  ``targets``:
    optional scene -> view -> [obj ids] filter (BOP target list); a duplicate (scene, view, object) is an error."""
    out = {}
    for i, (scene_id, view_id, obj_id) in enumerate(zip(detections["scene_ids"], detections["view_ids"], detections["obj_ids"])):
        per_view = out.setdefault(scene_id, {}).setdefault(view_id, {})
        assert obj_id not in per_view, "Found duplicate object in saved detections"
        if targets is None or obj_id in targets.get(scene_id, {}).get(view_id, []):
            per_view[obj_id] = i
    return out

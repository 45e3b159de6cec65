"""BOP-format input side of the hot path: what turns a YCB-V / T-LESS test tree into the
arguments of ``ObjectSLAM.process_view``.

Test-time mirror of the reference's ``BopDataset`` (lib/datasets/bop.py:28-723) -- same constructor keywords that
matter at evaluation, same accessors (``scene_ids / view_ids / obj_ids / get_obj_pose / get_cam_pose / is_target /
read_img / get_raw``), same ``get_raw`` sample dictionary -- and of ``load_mesh_db`` (lib/utils/mesh_database.py:17-45).
Training-only behaviour (background pasting, augmentation, occlusion masks from depth, random priors) is out of scope
and raises instead of silently differing.  Pure host code: file parsing and a handful of 3x3 products per object.
"""
from __future__ import annotations

import json
import os
import struct
from collections import defaultdict

import numpy as np

from . import kp_config
from .geometry import fix_K_for_bbox_ndc

IMAGE_SIZE = (256, 256)


# ---- meshes ------------------------------------------------------------------------------------------
_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h", "ushort": "H", "uint16": "H",
              "int": "i", "int32": "i", "uint": "I", "uint32": "I", "float": "f", "float32": "f", "double": "d", "float64": "d"}


def load_ply_points(path):
    """Vertex positions [n,3] float64 of a PLY file (ascii or binary little/big endian); other vertex properties are
    skipped, faces are not read.  Stands in for bop_toolkit's ``inout.load_ply(path)["pts"]``."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply", f"{path}: not a PLY file"
        fmt, n_vert, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            assert line, f"{path}: unterminated PLY header"
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vert = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                assert tok[1] != "list", f"{path}: list property on vertices"
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        ix = [names.index(a) for a in ("x", "y", "z")]
        if fmt == "ascii":
            rows = np.array([f.readline().split() for _ in range(n_vert)], dtype=np.float64).reshape(n_vert, len(props))
            return rows[:, ix]
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, end + c) for n, c in props])
        rec = np.frombuffer(f.read(dt.itemsize * n_vert), dtype=dt, count=n_vert)
        return np.stack([rec[a].astype(np.float64) for a in ("x", "y", "z")], axis=1)


def load_mesh_db(model_dir):
    """``{obj_id: {"is_symmetric", "continuous_sym", "diameter", "points"}}`` from ``models_info.json`` +
    ``obj_%06d.ply`` (mesh_database.py:17-45).  ``points`` is a float32 numpy array [P,3] in mm; ``EvalMeter`` uploads
    it to the GPU once (the reference keeps a CUDA tensor here)."""
    with open(os.path.join(model_dir, "models_info.json"), "r") as f:
        model_info = json.load(f)
    mesh_db = {}
    for key, info in model_info.items():
        obj_id = int(key)
        pts = load_ply_points(os.path.join(model_dir, f"obj_{obj_id:06d}.ply")).astype(np.float32)
        cont = info.get("symmetries_continuous", [])
        mesh_db[obj_id] = {
            "is_symmetric": len(info.get("symmetries_discrete", [])) > 0 or len(cont) > 0,
            "continuous_sym": cont if len(cont) > 0 else [],
            "diameter": info["diameter"],
            "points": pts,
        }
    return mesh_db


# ---- dataset -----------------------------------------------------------------------------------------
class BopDataset:
    def __init__(self, data_root, split, bop_dset="ycbv", map_by="view", mask_occluded=False, ignore_symmetry=False, no_aug=False,
                 det_type="gt", keep_all=False, kp_config_file=None, rng=None):
        """``data_root``: the dataset directory (contains ``<split>/``, ``kp_info/``, ``keyframe.txt`` ...).
        ``kp_config_file``: optional path of a reference-format CSV; default is the built-in table of ``bop_dset``.
        ``rng``: numpy Generator for ``det_type="gt+noise"`` (the reference draws from the global numpy state)."""
        assert bop_dset in ("ycbv", "tless")
        assert "train" not in split, "training splits (augmentation, background pasting) are out of scope of this reader"
        assert not mask_occluded, "occlusion masks from depth are a training-time option of the reference; not provided"
        assert ignore_symmetry, "evaluation uses ignore_symmetry=True (evaluate.py:77); symmetry picking is training-only"
        assert det_type in ("gt", "gt+noise")
        assert map_by == "view" or "obj" in map_by
        self.data_root, self.split, self.bop_dset, self.det_type = data_root, split, bop_dset, det_type
        self.map_by = map_by
        self.single_obj = int(map_by.split("_")[1]) if "obj_" in map_by else None
        self.keep_all = keep_all
        self.mask_occluded, self.ignore_symmetry, self.no_aug = False, True, True
        self.kp_path = os.path.join(data_root, "kp_info")
        self.bop_root = os.path.realpath(os.path.join(data_root, ".."))
        self.curr_root = os.path.join(data_root, split)
        self._rng = rng if rng is not None else np.random.default_rng(0)
        self._table = kp_config.load_kp_config_csv(kp_config_file) if kp_config_file else kp_config.TABLES[bop_dset]
        self.kp_map_per_object = [kp_config.load_kp_config(self._table, i + 1) for i in range(len(self._table))]
        self.kp_list_per_object = [kp_config.kp_list_of(self._table, i + 1) for i in range(len(self._table))]
        self._load_kp()
        self._index(min_visib_fract=0.1 if bop_dset == "tless" else -1)

    def num_obj(self):
        return len(self.kp_map_per_object)

    def _load_kp(self):
        """kp_info JSON -> kp_avg [n,3] in the object's channel order, kp_cov [n,3,3], view_pose [4,4] (bop.py:287-308)."""
        self.gt_kp = []
        for idx in range(self.num_obj()):
            path = os.path.join(self.kp_path, f"obj_{idx + 1:06d}_kp_info.json")
            assert os.path.exists(path), f"No keypoint file {path} found."
            with open(path, "r") as f:
                d = json.load(f)
            names = self.kp_list_per_object[idx]
            self.gt_kp.append({
                "kp_avg": np.array([d["keypoints"][n]["pos_mean"] for n in names], np.float64).reshape(len(names), 3),
                "kp_cov": np.array([d["keypoints"][n]["pos_cov"] for n in names], np.float64).reshape(len(names), 3, 3),
                "view_pose": np.array(d["view_pose"], np.float64).reshape(4, 4),
            })

    def _index(self, min_visib_fract):
        """Walk ``<split>/<scene>/scene_{camera,gt_info,gt}.json`` (bop.py:111-248): YCB-V test keeps the frames of
        ``keyframe.txt``; T-LESS test keeps the (scene, image, object) triples of ``all_target_tless.json`` and drops
        objects below 10 % visibility."""
        keyframes, self.targets, self.targets_filename = None, None, None
        if "test" in self.split:
            if self.bop_dset == "ycbv":
                with open(os.path.join(self.data_root, "keyframe.txt"), "r") as f:
                    keyframes = {tuple(int(v) for v in ln.split("/")) for ln in f.read().split("\n")[:-1]}
            else:
                self.targets_filename = os.path.join(self.data_root, "all_target_tless.json")
                with open(self.targets_filename, "r") as f:
                    self.targets = defaultdict(dict)
                    for t in json.load(f):
                        assert t["inst_count"] == 1
                        self.targets[t["scene_id"]].setdefault(t["im_id"], []).append(t["obj_id"])
        self.data = {}
        self.object_index_map = {"scene_ids": [], "view_ids": [], "obj_ids": []}
        self.view_index_map = {"scene_ids": [], "view_ids": []}
        for scene_str in sorted(os.listdir(self.curr_root)):
            scene_dir = os.path.join(self.curr_root, scene_str)
            if not os.path.isdir(scene_dir):
                continue
            scene_id, scene = int(scene_str), {}
            cam_infos, gt_infos, gt_poses = (json.load(open(os.path.join(scene_dir, n), "r")) for n in ("scene_camera.json", "scene_gt_info.json", "scene_gt.json"))
            for view_str, cam in cam_infos.items():
                view_id, keep, only = int(view_str), True, None
                if keyframes is not None:
                    keep = (scene_id, view_id) in keyframes
                elif self.targets is not None:
                    keep = scene_id in self.targets and view_id in self.targets[scene_id]
                    only = self.targets[scene_id][view_id] if keep else None
                if self.single_obj is not None:
                    only = [self.single_obj]
                if not keep:
                    continue
                frame = {"objects": {}, "K": np.array(cam["cam_K"], np.float64).reshape(3, 3), "depth_scale": cam["depth_scale"]}
                if "cam_R_w2c" in cam:
                    frame["cam_pose"] = np.concatenate((np.array(cam["cam_R_w2c"], np.float64).reshape(3, 3), np.array(cam["cam_t_w2c"], np.float64).reshape(3, 1)), axis=-1)
                for obj_idx, (g, info) in enumerate(zip(gt_poses[view_str], gt_infos[view_str])):
                    obj_id = g["obj_id"]
                    if info["visib_fract"] < min_visib_fract or (only is not None and obj_id not in only):
                        continue
                    self.object_index_map["scene_ids"].append(scene_id)
                    self.object_index_map["view_ids"].append(view_id)
                    self.object_index_map["obj_ids"].append(obj_id)
                    frame["objects"][obj_id] = {
                        "mask_path": os.path.join(self.curr_root, f"{scene_id:06d}", "mask_visib", view_str.zfill(6) + f"_{obj_idx:06d}.png"),
                        "bbox": info["bbox_visib"],
                        "pose": np.concatenate((np.array(g["cam_R_m2c"], np.float64).reshape(3, 3), np.array(g["cam_t_m2c"], np.float64).reshape(3, 1)), axis=-1),
                    }
                if frame["objects"]:
                    scene[view_id] = frame
                    self.view_index_map["scene_ids"].append(scene_id)
                    self.view_index_map["view_ids"].append(view_id)
            if scene:
                self.data[scene_id] = scene

    # ---- accessors (bop.py:257-259,387-414) ----
    def is_target(self, scene_id, view_id, obj_id):
        return self.targets is None or obj_id in self.targets.get(scene_id, {}).get(view_id, [])

    def __len__(self):
        return len(self.view_index_map["scene_ids"]) if self.map_by == "view" else len(self.object_index_map["scene_ids"])

    def __getitem__(self, index):
        if self.map_by == "view":
            return self.get_all_obj(self.view_index_map["scene_ids"][index], self.view_index_map["view_ids"][index])
        m = self.object_index_map
        return self.get_raw(m["scene_ids"][index], m["view_ids"][index], [m["obj_ids"][index]])

    def get_cam_pose(self, scene_id, view_id=-1):
        if view_id < 0:
            view_id = min(self.data[scene_id].keys())
        return self.data[scene_id][view_id].get("cam_pose")

    def get_obj_pose(self, scene_id, view_id, obj_id):
        return self.data[scene_id][view_id]["objects"][obj_id]["pose"]

    def scene_ids(self):
        return list(self.data.keys())

    def view_ids(self, scene_id):
        return list(self.data[scene_id].keys())

    def obj_ids(self, scene_id, view_id):
        return list(self.data[scene_id][view_id]["objects"].keys())

    def get_all_obj(self, scene_id, view_id):
        return self.get_raw(scene_id, view_id, self.obj_ids(scene_id, view_id))

    def read_img(self, scene_id, view_id):
        """uint8 [H,W,3] in BGR channel order, as ``cv2.imread`` delivers it (bop.py:430-441)."""
        from PIL import Image
        ext = ".jpg" if "pbr" in self.split else ".png"
        path = os.path.join(self.curr_root, f"{scene_id:06d}", "rgb", f"{view_id:06d}{ext}")
        img = np.asarray(Image.open(path).convert("RGB"), np.uint8)
        assert img.size > 0, f"Empty image {path}"
        return np.ascontiguousarray(img[:, :, ::-1])

    def get_raw(self, scene_id, view_id, obj_ids):
        """The sample dictionary of bop.py:469-723 at test time (torch tensors, same keys, dtypes and shapes), minus
        the random idealised priors (``priors / prior_uvs``; ``has_prior`` is all False): evaluation renders its
        priors from the estimated poses (object_slam.py:486-519)."""
        import torch
        frame = self.data[scene_id][view_id]
        img0, K, n, NK = self.read_img(scene_id, view_id), frame["K"], len(obj_ids), kp_config.num_kp()
        bboxes = np.zeros((n, 4), np.float32)
        for i, obj_id in enumerate(obj_ids):
            xywh = np.array(frame["objects"][obj_id]["bbox"], np.float32)
            if "+noise" in self.det_type:
                xywh += self._rng.normal(scale=20, size=(4,)).astype(np.float32)
            x, y, w, h = xywh
            w, h = max(10, w), max(10, h)                                  # bop.py:551
            bboxes[i] = np.array([x, y, x + w, y + h], np.float32)
        poses = np.zeros((n, 3, 4), np.float32)
        K_kps = np.zeros((n, 3, 3), np.float32)
        kp_uvs = np.zeros((n, NK, 2), np.float32)
        kp_masks = np.zeros((n, NK), bool)
        model_kps = np.zeros((n, NK, 3), np.float32)
        kp_model_masks = np.zeros((n, NK), bool)
        for i, obj_id in enumerate(obj_ids):
            T = frame["objects"][obj_id]["pose"]
            poses[i] = T.astype(np.float32)
            kp3d = self.gt_kp[obj_id - 1]["kp_avg"]
            ch = np.array([self.kp_map_per_object[obj_id - 1][nm] for nm in self.kp_list_per_object[obj_id - 1]], np.int64)
            uvz = (kp3d @ T[:3, :3].T + T[:3, 3]) @ K.T                   # camera frame -> image plane (bop.py:626-631)
            uv = uvz[:, :2] / uvz[:, 2:3]
            x, y, x2, y2 = bboxes[i]
            w, h = x2 - x, y2 - y                                          # float32, like the reference
            ndc = uv - np.array([x, y], np.float64)[None, :]
            ndc[:, 0] = 2 * ndc[:, 0] / w - 1
            ndc[:, 1] = 1 - 2 * ndc[:, 1] / h
            K_kps[i] = fix_K_for_bbox_ndc(K, bboxes[i]).astype(np.float32)
            inside = np.all((ndc >= -1) & (ndc <= 1), axis=1)
            kp_uvs[i, ch] = ndc.astype(np.float32)
            model_kps[i, ch] = kp3d.astype(np.float32)
            kp_model_masks[i, ch] = True
            kp_masks[i, ch] = inside
        return {
            "img": torch.from_numpy(img0).permute(2, 0, 1).to(torch.float32) / 255,
            "K": torch.tensor(K.astype(np.float32)),
            "obj_ids": torch.tensor(list(obj_ids), dtype=torch.long),
            "bboxes": torch.tensor(bboxes),
            "poses": torch.tensor(poses),
            "has_prior": torch.zeros(n, dtype=torch.bool),
            "K_kps": torch.tensor(K_kps),
            "kp_uvs": torch.tensor(kp_uvs),
            "kp_masks": torch.tensor(kp_masks),
            "model_kps": torch.tensor(model_kps),
            "kp_model_masks": torch.tensor(kp_model_masks),
        }

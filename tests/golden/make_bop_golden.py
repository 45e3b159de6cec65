#!/usr/bin/env python3
"""Golden vectors for the BOP input contract, produced by running the REFERENCE's own reader.

Run in the build container only (needs /root/reference):
    python tests/golden/make_bop_golden.py

tests/bop_tree.py writes a small synthetic BOP tree (YCB-V-shaped and T-LESS-shaped, seeded); this script imports,
unmodified, /root/reference/lib/datasets/bop.py (BopDataset) and lib/utils/mesh_database.py (load_mesh_db, which uses
the vendored thirdparty/bop_toolkit load_ply), runs them on that tree and stores what they return:

    index      scene ids / view ids / object ids per view, targets, poses, camera poses
    get_raw    K, bboxes, poses, K_kps, kp_uvs, kp_masks, model_kps, kp_model_masks, img checksum
    mesh_db    points, is_symmetric, diameter

Import-time shims for what this image lacks: ``cv2`` (imread via PIL, BGR order), ``torchvision``, ``imageio``,
``png``; numpy-2 aliases.  The reference resolves ``./kp_configs/*.csv`` relative to the working directory, so the
script chdirs to /root/reference (read-only use).  The fixture is data only.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

for name, val in (("int", int), ("bool", bool), ("float", float)):
    if not hasattr(np, name):
        setattr(np, name, val)
if not hasattr(np, "math"):
    import math
    np.math = math


def _imread(path, flags=None):
    from PIL import Image
    if not os.path.exists(path):
        return None
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), np.uint8)[:, :, ::-1])


cv2 = types.ModuleType("cv2")
cv2.setNumThreads = lambda n: None
cv2.imread = _imread
cv2.IMREAD_ANYDEPTH, cv2.IMREAD_GRAYSCALE = 2, 0
cv2.GaussianBlur = lambda img, ksize, sigma: img      # only feeds the random training priors, which are not recorded
sys.modules["cv2"] = cv2
tv = types.ModuleType("torchvision")
tv.ops = types.ModuleType("torchvision.ops")
tv.datasets = types.ModuleType("torchvision.datasets")
tv.datasets.ImageFolder = object
for k, m in (("torchvision", tv), ("torchvision.ops", tv.ops), ("torchvision.datasets", tv.datasets), ("imageio", types.ModuleType("imageio")),
             ("png", types.ModuleType("png"))):
    sys.modules[k] = m
import matplotlib  # noqa: E402
matplotlib.use("Agg")
sys.path.insert(0, REF)
os.chdir(REF)
from lib.datasets import bop as ref_bop  # noqa: E402
from lib.utils import mesh_database as ref_mesh  # noqa: E402

from tests import bop_tree  # noqa: E402

SEEDS = {"ycbv": 11, "tless": 12}


def main():
    out = {}
    for dset, seed in SEEDS.items():
        with tempfile.TemporaryDirectory() as root:
            desc = bop_tree.build(root, dset=dset, seed=seed)
            ds = ref_bop.BopDataset(desc["data_root"], desc["split"], bop_dset=dset, ignore_symmetry=True)
            rows = []
            for s in ds.scene_ids():
                for v in ds.view_ids(s):
                    for o in ds.obj_ids(s, v):
                        rows.append([s, v, o, int(ds.is_target(s, v, o))])
            out[f"{dset}_index"] = np.array(rows, np.int64)
            out[f"{dset}_len"] = np.array(len(ds))
            out[f"{dset}_obj_index"] = np.array([ds.object_index_map[k] for k in ("scene_ids", "view_ids", "obj_ids")], np.int64)
            out[f"{dset}_bop_root_is_parent"] = np.array(os.path.realpath(root) == ds.bop_root)
            k = 0
            for s in ds.scene_ids():
                out[f"{dset}_campose_first_{s}"] = ds.get_cam_pose(s)
                for v in ds.view_ids(s):
                    ids = ds.obj_ids(s, v)
                    sample = ds.get_raw(s, v, ids)
                    out[f"{dset}_raw_{k}_key"] = np.array([s, v] + ids, np.int64)
                    for name in ("K", "obj_ids", "bboxes", "poses", "K_kps", "kp_uvs", "kp_masks", "model_kps", "kp_model_masks"):
                        out[f"{dset}_raw_{k}_{name}"] = sample[name].numpy()
                    img = sample["img"].numpy()
                    out[f"{dset}_raw_{k}_img_shape"] = np.array(img.shape)
                    out[f"{dset}_raw_{k}_img_sum"] = np.array([float(img[c].astype(np.float64).sum()) for c in range(3)])
                    out[f"{dset}_raw_{k}_img_probe"] = img[:, ::97, ::101].copy()
                    out[f"{dset}_raw_{k}_campose"] = ds.get_cam_pose(s, v)
                    out[f"{dset}_raw_{k}_objpose0"] = ds.get_obj_pose(s, v, ids[0])
                    k += 1
            out[f"{dset}_n_raw"] = np.array(k)
            # a subset request in a different order (evaluate.py passes the detected subset)
            s = ds.scene_ids()[0]
            v = ds.view_ids(s)[0]
            ids = ds.obj_ids(s, v)[::-1][:2]
            sample = ds.get_raw(s, v, ids)
            out[f"{dset}_subset_key"] = np.array([s, v] + ids, np.int64)
            for name in ("bboxes", "kp_uvs", "kp_masks", "model_kps"):
                out[f"{dset}_subset_{name}"] = sample[name].numpy()
            models = "models_bop-compat_eval" if dset == "ycbv" else "models_eval"
            db = ref_mesh.load_mesh_db(os.path.join(desc["data_root"], models))
            out[f"{dset}_mesh_ids"] = np.array(sorted(db.keys()))
            out[f"{dset}_mesh_sym"] = np.array([int(db[o]["is_symmetric"]) for o in sorted(db.keys())])
            out[f"{dset}_mesh_diam"] = np.array([db[o]["diameter"] for o in sorted(db.keys())], np.float64)
            for o in sorted(db.keys())[:6]:
                out[f"{dset}_mesh_pts_{o}"] = db[o]["points"].cpu().numpy()
    path = os.path.join(HERE, "bop_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()

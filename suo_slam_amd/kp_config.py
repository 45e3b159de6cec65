"""The 41-channel keypoint vocabulary and the per-object keypoint subsets.

Mirror of the reference's ``lib/labeling/kp_config.py:7-147`` (vocabulary, ``num_kp``, ``get_kps``,
``load_kp_config``) plus the two class tables it reads from ``kp_configs/{ycbv,tless}_kp_config.csv``.  The channel
ORDER is an interface constant: channel c of the network output / of ``model_kps[L,41,3]`` is ``KP_LIST[c]``, and the
names are the keys of the ``kp_info`` JSON files (lib/datasets/bop.py:297-303).

Channel layout: 8 box corners | 10 cylinder points | 6 hand-tool points | 4 grip | 1 spout | 4 brand name |
4 nutrition facts | 4 bar code.
"""
from __future__ import annotations

import csv

_CORNERS = ("tl", "tr", "br", "bl")
_SHAPE = {
    "box_like": [f"box_corner_{face}_{c}" for face in ("front", "back") for c in _CORNERS],
    "cylinder_like": ["cyl_top_center", "cyl_bottom_center"] + [f"cyl_rim_{end}_{side}" for end in ("top", "bottom") for side in ("front", "back", "right", "left")],
    "hand_tool": ["tactile_point", "rotation_axis"] + [f"tool_base_{a}_{b}" for a in ("front", "back") for b in ("left", "right")],
}
_INSTANCE = {
    "grip": [f"grip_{f}" for f in ("thumb", "palm", "index", "pinky")],
    "spout": ["spout"],
    "brand_name": [f"brand_name_{c}" for c in _CORNERS],
    "nutrition_facts": [f"nutrition_facts_{c}" for c in _CORNERS],
    "bar_code": [f"bar_code_{c}" for c in _CORNERS],
}
SHAPE_CLASSES = ("box_like", "cylinder_like", "hand_tool")
INSTANCE_GROUPS = ("grip", "spout", "brand_name", "nutrition_facts", "bar_code")

KP_LIST = [n for k in SHAPE_CLASSES for n in _SHAPE[k]] + [n for k in INSTANCE_GROUPS for n in _INSTANCE[k]]
kp_list = KP_LIST                                         # the reference's name
assert len(KP_LIST) == 41 and len(set(KP_LIST)) == 41
_INDEX = {n: i for i, n in enumerate(KP_LIST)}


def num_kp():
    return len(KP_LIST)


def get_kps(class_str, has_grip, has_spout, has_brand_name, has_nutrition_facts, has_bar_code):
    """name -> channel for one object (kp_config.py:104-133); insertion order = channel order within each group."""
    assert class_str in _SHAPE, f"Shape class {class_str} is invalid! Options are {list(_SHAPE)}"
    names = list(_SHAPE[class_str])
    for flag, group in zip((has_grip, has_spout, has_brand_name, has_nutrition_facts, has_bar_code), INSTANCE_GROUPS):
        if int(flag):
            names += _INSTANCE[group]
    return {n: _INDEX[n] for n in names}


# (class, has_grip, has_spout, has_brand_name, has_nutrition_facts, has_bar_code) per object id 1..N -- the content of
# kp_configs/ycbv_kp_config.csv:2-22 and kp_configs/tless_kp_config.csv:2-31
_C, _B, _T = "cylinder_like", "box_like", "hand_tool"
YCBV_TABLE = [
    (_C, 0, 0, 1, 0, 1), (_B, 0, 0, 1, 1, 1), (_B, 0, 0, 1, 1, 1), (_C, 0, 0, 1, 1, 1), (_B, 0, 1, 1, 1, 1), (_C, 0, 0, 1, 1, 1), (_B, 0, 0, 1, 1, 1),
    (_B, 0, 0, 1, 1, 1), (_B, 0, 0, 1, 1, 1), (_C, 1, 0, 0, 0, 0), (_C, 1, 1, 0, 0, 0), (_B, 0, 1, 1, 0, 0), (_C, 0, 0, 0, 0, 0), (_C, 1, 0, 0, 0, 0),
    (_T, 1, 0, 1, 0, 0), (_B, 0, 0, 0, 0, 0), (_T, 1, 0, 0, 0, 0), (_C, 0, 0, 1, 0, 0), (_T, 1, 0, 0, 0, 0), (_T, 1, 0, 0, 0, 0), (_B, 0, 0, 0, 0, 0),
]
_TLESS_CYL = {1, 2, 3, 4, 13, 14, 15, 16, 17, 18, 24}
TLESS_TABLE = [((_C if i in _TLESS_CYL else _B), 0, 0, 0, 0, 0) for i in range(1, 31)]
TABLES = {"ycbv": YCBV_TABLE, "tless": TLESS_TABLE}


def load_kp_config_csv(path):
    """Read a ``kp_configs/*.csv`` of the reference (header line, then instance,class,5 flags per object)."""
    rows = []
    with open(path, newline="") as f:
        for i, r in enumerate(csv.reader(f)):
            if i == 0 or not r:
                continue
            rows.append((r[1].strip(),) + tuple(int(v) for v in r[2:7]))
    return rows


def load_kp_config(table, object_id):
    """name -> channel of object ``object_id`` (1-based, as in BOP) -- kp_config.py:138-146 with the table in place of
    the pandas frame.  ``table``: "ycbv" / "tless" or rows from load_kp_config_csv."""
    rows = TABLES[table] if isinstance(table, str) else table
    return get_kps(*rows[object_id - 1])


def kp_list_of(table, object_id):
    """The object's keypoint names in CHANNEL order (lib/datasets/bop.py:270-279) -- the row order of its
    ``kp_avg`` array."""
    m = load_kp_config(table, object_id)
    return [n for n in KP_LIST if n in m]


def model_mask(table, object_id):
    import numpy as np
    mask = np.zeros(num_kp(), bool)
    mask[list(load_kp_config(table, object_id).values())] = True
    return mask

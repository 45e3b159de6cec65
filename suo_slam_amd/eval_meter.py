"""ADD / ADD-S / ADD(-S) AUC evaluation meter -- host mirror of the reference's ``lib/utils/eval_meter.py``
 over the HIP distance kernels of ``csrc/eval.hip``.

Same names and call shapes as the reference so ``evaluate.py``-style harnesses read the same:

    compute_auc_posecnn(errors)                          eval_meter.py:22-45
    AverageMeter                                         eval_meter.py:47-64
    AddAucMeter(obj_avg).update / average                eval_meter.py:66-95
    EvalMeter(mesh_db, sample_n_points=None, d=0.1)      eval_meter.py:97-231
        .update(obj_ids, poses_pred, poses_gt) / .update_no_det(obj_ids) / .result() / .pprint_objs_str(...)

The O(P^2) nearest-neighbour distances of ADD-S run on the GPU (``suo_pose_errors``); the AUC itself is a sort and
a prefix sum over a few thousand scalars and stays host numpy exactly as in the reference.  No CPU fallback for the
distances: without the HIP library / a GPU ``EvalMeter`` raises.
"""
from __future__ import annotations

import ctypes as C
from collections import defaultdict

import numpy as np

from . import _lib


def compute_auc_posecnn(errors):
    """Area under the accuracy-vs-threshold curve up to 10 cm, PoseCNN style.  ``errors`` are mm (inf = missed).

    Follows eval_meter.py:22-45 including its numeric types: a list is taken as float32, scaled to metres in
    float32, and the recall axis is those float32 values widened to float64.  (One deliberate difference: a
    single-element list works here; the reference raises on it because np.squeeze leaves a 0-d scalar.)"""
    e = np.array(errors, dtype=np.float32) if isinstance(errors, list) else np.array(errors)
    e = np.atleast_1d(np.squeeze(e))
    e = e * e.dtype.type(1e-3) if e.dtype == np.float32 else 1e-3 * e
    n = e.shape[0]
    if n == 0:
        return 0
    rec = np.sort(e[e <= 0.1]).astype(np.float64)        # beyond 10 cm (and inf / nan) never counts
    if rec.size == 0:
        return 0
    prec = np.arange(1, rec.size + 1, dtype=np.float64) / n
    mrec = np.concatenate(([0.0], rec, [0.1]))
    mpre = np.maximum.accumulate(np.concatenate(([0.0], prec, [prec[-1]])))
    step = np.flatnonzero(mrec[1:] != mrec[:-1]) + 1
    return ((mrec[step] - mrec[step - 1]) * mpre[step]).sum() * 10


class AverageMeter:
    """Running mean with per-update weights (eval_meter.py:47-64)."""

    def __init__(self):
        self.avg = 0
        self.n = 0

    def update(self, x, k=1):
        self.n += k
        self.avg = ((self.n - k) * self.avg + x) / self.n

    def average(self):
        return self.avg


class AddAucMeter:
    """Per-class error lists -> AUC (eval_meter.py:66-95).  ``obj_avg``: mean of per-object AUCs (DeepIM / CosyPose)
    instead of one AUC over all errors (PoseCNN)."""

    def __init__(self, obj_avg=False):
        self.err_map = defaultdict(list)
        self.obj_avg = obj_avg

    def update(self, obj_ids, errs):
        for obj_id, err in zip(obj_ids, errs):
            self.err_map[obj_id].append(err)

    def average(self):
        assert len(self.err_map) > 0, "Called AucMeter.average without feeding any data!"
        auc_map = {obj_id: compute_auc_posecnn(errs) for obj_id, errs in self.err_map.items()}
        if self.obj_avg:
            return sum(auc_map.values()) / len(auc_map), auc_map
        everything = [e for errs in self.err_map.values() for e in errs]
        return compute_auc_posecnn(everything), auc_map


def _points_numpy(p):
    if hasattr(p, "detach"):
        p = p.detach().cpu().numpy()
    return np.ascontiguousarray(p, np.float32).reshape(-1, 3)


class EvalMeter:
    """ADD, ADD-S and ADD(-S) meters over a mesh database ``{obj_id: {"points": [P,3] mm, "is_symmetric": bool, ...}}``
    (lib/utils/mesh_database.py:34-40).  The point clouds are uploaded once; each ``update`` is one kernel set."""

    def __init__(self, mesh_db, sample_n_points=None, d=0.1, seed=0):
        self.mesh_db = mesh_db
        self.d = d
        self.sample_n_points = sample_n_points
        self.lib = _lib.lib()
        _lib.require_gpu()
        key = "points"
        if sample_n_points is not None:
            # eval_meter.py:102-111.  The reference's tensor branch of sample_pts returns indices instead of points
            # (eval_meter.py:14-15); here a seeded subset of the points themselves is stored.
            assert type(sample_n_points) == int
            rng = np.random.default_rng(seed)
            key = "points_sampled"
            for obj_id in self.mesh_db.keys():
                pts = _points_numpy(self.mesh_db[obj_id]["points"])
                assert sample_n_points <= pts.shape[0], f"Not enough points in mesh to sample {sample_n_points} points"
                have = self.mesh_db[obj_id].get("points_sampled")
                if have is None or _points_numpy(have).shape[0] != sample_n_points:
                    self.mesh_db[obj_id]["points_sampled"] = pts[rng.choice(pts.shape[0], size=sample_n_points, replace=False)]
        self._index = {obj_id: i for i, obj_id in enumerate(self.mesh_db.keys())}
        clouds = [_points_numpy(self.mesh_db[obj_id][key]) for obj_id in self.mesh_db.keys()]
        n_pts = np.array([c.shape[0] for c in clouds], np.int32)
        allpts = np.ascontiguousarray(np.concatenate(clouds, 0))
        h = C.c_void_p()
        _lib.check(self.lib.suo_mesh_db_create(len(clouds), n_pts.ctypes.data, allpts.ctypes.data, C.byref(h)), "suo_mesh_db_create")
        self._h = h
        self.add_meter = AddAucMeter(obj_avg=True)
        self.adds_meter = AddAucMeter(obj_avg=True)
        self.add_maybe_s_meter = AddAucMeter(obj_avg=True)

    def close(self):
        if getattr(self, "_h", None) is not None:
            self.lib.suo_mesh_db_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pose_errors(self, obj_ids, poses_pred, poses_gt):
        """(ADD[n], ADD-S[n]) in mesh units for n (object, predicted pose, ground-truth pose) triples."""
        n = len(obj_ids)
        idx = np.array([self._index[o] for o in obj_ids], np.int32)

        def pack(T):
            if hasattr(T, "detach"):
                T = T.detach().cpu().numpy()
            T = np.asarray(T, np.float64).reshape(n, -1, 4)[:, :3, :]
            return np.ascontiguousarray(T.astype(np.float32)).reshape(n, 12)

        Tp, Tg = pack(poses_pred), pack(poses_gt)
        add, adds = np.zeros(n, np.float32), np.zeros(n, np.float32)
        _lib.check(self.lib.suo_pose_errors(self._h, n, idx.ctypes.data, Tp.ctypes.data, Tg.ctypes.data, add.ctypes.data, adds.ctypes.data),
                   "suo_pose_errors")
        return add, adds

    def update(self, obj_ids, poses_pred, poses_gt):
        """eval_meter.py:119-155: ADD for every object, ADD-S for every object, ADD(-S) picks by ``is_symmetric``."""
        obj_ids = [int(o) if isinstance(o, (np.integer,)) else o for o in obj_ids]
        add, adds = self.pose_errors(obj_ids, poses_pred, poses_gt)
        sym = np.array([bool(self.mesh_db[o]["is_symmetric"]) for o in obj_ids])
        self.add_meter.update(obj_ids, add.tolist())
        self.adds_meter.update(obj_ids, adds.tolist())
        self.add_maybe_s_meter.update(obj_ids, np.where(sym, adds, add).tolist())

    def update_no_det(self, obj_ids):
        """Ground-truth objects that were not detected count as infinite error (eval_meter.py:158-162)."""
        miss = [np.inf for _ in obj_ids]
        for m in (self.add_meter, self.adds_meter, self.add_maybe_s_meter):
            m.update(obj_ids, miss)

    def result(self):
        return {
            "AUC of ADD": self.add_meter.average(),
            "AUC of ADD-S": self.adds_meter.average(),
            "AUC of ADD(-S)": self.add_maybe_s_meter.average(),
        }

    def pprint_objs_str(self, gt_obj_map):
        """LaTeX-style per-object table, same text as eval_meter.py:172-207.  ``gt_obj_map``: obj_id -> printed name."""
        result = self.result()
        cols = ["AUC of ADD", "AUC of ADD-S"]

        def row(name, cells, end):
            assert len(str(name)) <= 22, f"String {name} is too long for width (22)"
            return f"{str(name):<22}& " + "& ".join(f"{c:<15}" for c in cells) + end

        lines = row("", cols, "\\\\\n")
        for obj_id in sorted(gt_obj_map.keys()):
            lines += row(gt_obj_map[obj_id], [f"{100 * result[k][1].get(obj_id, 0):.1f}" for k in cols], "\\\\\n")
        lines += row("Mean", [f"{100 * result[k][0]:.1f}" for k in cols], "\n\n")
        lines += f'AUC of ADD(-S): {100 * result["AUC of ADD(-S)"][0]:.1f}\n'
        return lines

    def pprint_objs(self, gt_obj_map):
        bar = "=" * 59
        print(bar)
        print(self.pprint_objs_str(gt_obj_map))
        print(bar)
        print("\n\n")

    def pprint(self):
        for k, v in self.result().items():
            print(f"{k}: {v[0]}")

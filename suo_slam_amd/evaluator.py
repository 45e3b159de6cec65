"""Evaluation harness around the hot path: BOP tree in -> process_view per frame ->
ADD(-S) AUC table + BOP-format results CSV out.

Host mirror of the reference's ``Evaluator`` (evaluate.py:50-393) without its visualisation and GUI parts: same
constructor keywords that affect results, same per-dataset thresholds (evaluate.py:56-74), same scene / view loop
(single-view ``nviews=1``, SfM ``nviews>1``, SLAM ``nviews<0``), same detection selection (ground-truth boxes or saved
PoseCNN / Pix2Pose detections), same bookkeeping of missed detections, same CSV line format
(``scene,im,obj,score,R(9),t(3),-1``).  The T-LESS VSD step (an external bop_toolkit subprocess, evaluate.py:323-336,
row N4) is handed off exactly as the reference does it: ``run`` returns the command line, environment and working directory
(``bop_eval_command``) and starts it when ``run_bop_eval=True`` and the toolkit is mounted; the CSV it consumes is pinned
against the toolkit's own ``inout.load_bop_results`` (tests/golden/make_bop_results_golden.py).

Everything numeric happens in ``ObjectSLAM`` (HIP network, PnP, bundle adjustment) and ``EvalMeter`` (HIP ADD-S).
"""
from __future__ import annotations

import os
from time import time

import numpy as np

from . import bop, detections
from .eval_meter import EvalMeter
from .geometry import invert_SE3
from .object_slam import ObjectSLAM

YCBV_CLASSES = dict(zip(range(1, 22), (
    "002_master_chef_can", "003_cracker_box", "004_sugar_box", "005_tomato_soup_can", "006_mustard_bottle", "007_tuna_fish_can", "008_pudding_box",
    "009_gelatin_box", "010_potted_meat_can", "011_banana", "019_pitcher_base", "021_bleach_cleanser", "024_bowl", "025_mug", "035_power_drill",
    "036_wood_block", "037_scissors", "040_large_marker", "051_large_clamp", "052_extra_large_clamp", "061_foam_brick")))
TLESS_CLASSES = {i: str(i) for i in range(1, 31)}

# evaluate.py:56-74
_SETTINGS = {
    "ycbv": dict(models="models_bop-compat_eval", split="test", do_add=True, kp_var_thresh=0.2, bbox_thresh=0.9, manual_kp_std=0.01,
                 opt_init_with_outliers=False),
    "tless": dict(models="models_eval", split="test_primesense", do_add=False, kp_var_thresh=0.5, bbox_thresh=1.0, manual_kp_std=0.1,
                  opt_init_with_outliers=True),
}


def bop_csv_line(scene_id, view_id, obj_id, score, T_OtoC):
    """One line of a BOP results file (evaluate.py:277-282): rotation row-major, translation in mm, time = -1."""
    R = " ".join(str(v) for v in np.asarray(T_OtoC)[:3, :3].reshape(-1).tolist())
    t = " ".join(str(v) for v in np.asarray(T_OtoC)[:3, 3].reshape(-1).tolist())
    return f"{scene_id},{view_id},{obj_id},{score},{R},{t},-1\n"


def bop_eval_command(csv_path, outdir, targets_filename, repo_root="."):
    """The T-LESS VSD hand-off of evaluate.py:323-336 (row N4): the exact subprocess the reference starts after writing the
    CSV -- bop_toolkit's ``scripts/eval_siso.py`` with the python renderer, run from ``thirdparty/bop_toolkit/`` with
    PYTHONPATH = that directory and BOP_PATH = <cwd>/data/bop_datasets/.  Returns (argv, env additions, cwd); the toolkit
    itself (renderer, VSD) is third-party and stays a subprocess here as it is there."""
    toolkit = os.path.join(repo_root, "thirdparty/bop_toolkit/")
    argv = ["python", "scripts/eval_siso.py", "--renderer_type", "python", "--result_filename", os.path.realpath(csv_path),
            "--results_path", "", "--eval_path", os.path.realpath(outdir), "--targets_filename", targets_filename]
    env = {"PYTHONPATH": os.path.realpath(toolkit), "BOP_PATH": os.path.join(os.path.abspath(repo_root), "data/bop_datasets/")}
    return argv, env, toolkit


class Evaluator:
    def __init__(self, dataset, data_root, chkpt_path, nviews=1, no_network_cov=False, detection_type="saved", debug_gt_kp=False,
                 gt_cam_pose=False, no_prior_det=False, debug_saved_only=False, give_all_prior=False, out_dir=None, state_dict=None,
                 do_add=None, seed=666, verbose=False, repo_root=".", run_bop_eval=False, frames_per_call=1):
        """``dataset``: "ycbv" | "tless"; ``data_root``: the dataset directory of the BOP tree.  ``out_dir`` defaults to
        the checkpoint's directory like the reference.  ``do_add`` overrides the per-dataset default (the reference
        evaluates ADD only on YCB-V).  ``frames_per_call`` > 1 (single-view evaluation, nviews == 1, only): that many reference views go through
        ONE network call and ONE geometry launch (ObjectSLAM.process_views_single) -- the views of evaluate.py's loop are independent there
        (reset / process_view / collect_results per view, evaluate.py:338-395); results are the per-view loop's (geometry bit for bit on the same network
        outputs; the shared network call agrees with per-view calls to the network's tolerance: ObjectSLAM.process_views_single)."""
        cfg = _SETTINGS[dataset]
        self.frames_per_call = int(frames_per_call) if nviews == 1 else 1
        self.model_path = out_dir if out_dir is not None else os.path.dirname(chkpt_path or ".")
        self.do_add = cfg["do_add"] if do_add is None else do_add
        if debug_gt_kp:
            detection_type = "gt"                                     # evaluate.py:378-379
        self.dataset = bop.BopDataset(data_root, cfg["split"], bop_dset=dataset, ignore_symmetry=True)
        self.mesh_db = bop.load_mesh_db(os.path.join(data_root, cfg["models"]))
        self.debug_saved_only = debug_saved_only
        self.nviews, self.detection_type, self.debug_gt_kp, self.gt_cam_pose = nviews, detection_type, debug_gt_kp, gt_cam_pose
        self.verbose = verbose
        self.repo_root, self.run_bop_eval = repo_root, run_bop_eval    # where thirdparty/bop_toolkit lives; start eval_siso.py after a T-LESS run
        self._rng = np.random.RandomState(seed)                       # evaluate.py:386 seeds numpy with 666
        if not debug_saved_only:
            self.object_slam = ObjectSLAM(chkpt_path, self.mesh_db, no_network_cov=no_network_cov, no_prior_det=no_prior_det,
                                          debug_gt_kp=debug_gt_kp, sfm_mode=nviews > 0, single_view_mode=nviews == 1,
                                          kp_var_thresh=cfg["kp_var_thresh"], bbox_thresh=cfg["bbox_thresh"], bbox_inflate=0.0,
                                          manual_kp_std=cfg["manual_kp_std"], opt_init_with_outliers=cfg["opt_init_with_outliers"],
                                          give_all_prior=give_all_prior, state_dict=state_dict, max_crops=16 * max(1, self.frames_per_call))
        self.saved_detections, self.saved_detections_map = None, {}
        if detection_type == "saved":
            load = detections.load_posecnn_results if dataset == "ycbv" else detections.load_pix2pose_results
            self.saved_detections = load(self.dataset.bop_root)
            self.saved_detections_map = detections.build_detection_map(self.saved_detections, self.dataset.targets)

    def method_name(self):
        """Directory / file stem of the results (evaluate.py:150-163)."""
        m = f"pkpnet-epoch={self.object_slam.model_epoch}-nviews={self.nviews}-det={self.detection_type}"
        for flag, tag in ((self.debug_gt_kp, "-GT-KP"), (self.gt_cam_pose, "-GT-CAM-POSE"), (self.object_slam.give_all_prior, "-ALL-PRIOR"),
                          (self.object_slam.no_network_cov, "-NO-COV"), (self.object_slam.no_prior_det, "-NO-PRIOR-DET")):
            if flag:
                m += tag
        return m + f"_{self.dataset.bop_dset}-{self.dataset.split}"

    def _log(self, *a):
        if self.verbose:
            print(*a)

    # ---- one reference view (+ companions in SfM mode) through the hot path: evaluate.py:338-393 ----
    def _view_args(self, scene_id, view_id, first_view):
        """What evaluate.py:341-393 hands process_view for one view: (args tuple, keyword dict), or None when the view has no detections."""
        gt_ids = self.dataset.obj_ids(scene_id, view_id)
        if "gt" in self.detection_type:
            obj_ids = gt_ids
        else:
            obj_ids = [o for o in self.saved_detections_map.get(scene_id, {}).get(view_id, {}).keys() if o in gt_ids]
            assert len(obj_ids) == len(set(obj_ids)), "Duplicates in detections?"
            if len(obj_ids) == 0:
                self._log(f"WARNING no detections for scene {scene_id} view {view_id}")
                return None
        sample = self.dataset.get_raw(scene_id, view_id, obj_ids)
        if "gt" in self.detection_type:
            bboxes = sample["bboxes"].numpy()
        else:
            bboxes = [self.saved_detections["bboxes"][self.saved_detections_map[scene_id][view_id][o]] for o in obj_ids]
        cam_pose = None
        if self.gt_cam_pose:
            first = -1 if self.nviews < 0 else first_view
            cam_pose = self._to4(self.dataset.get_cam_pose(scene_id, view_id)) @ invert_SE3(self._to4(self.dataset.get_cam_pose(scene_id, first)))
        img = (255 * sample["img"].numpy().transpose((1, 2, 0))).astype(np.uint8)
        return ((view_id, img, sample["K"].numpy(), np.array(obj_ids, dtype=int), np.array(bboxes), sample["model_kps"].numpy(),
                 sample["kp_model_masks"].numpy(), sample["kp_masks"].numpy()),
                {"uv_gt": sample["kp_uvs"].numpy() if self.debug_gt_kp else None, "cam_pose": cam_pose})

    def _run_slam(self, scene_id, views_to_proc):
        if self.nviews > 0:
            self.object_slam.reset()
        else:
            assert len(views_to_proc) == 1
        for view_id in views_to_proc:
            a = self._view_args(scene_id, view_id, views_to_proc[0])
            if a is not None:
                self.object_slam.process_view(*a[0], **a[1])
        return self.object_slam.collect_results(last_only=self.nviews < 0, no_viz=True)

    @staticmethod
    def _to4(T):
        T = np.asarray(T, np.float64)
        if T.shape == (4, 4):
            return T
        out = np.eye(4)
        out[:3, :] = T[:3, :]
        return out

    def run(self):
        """Returns ``{"method", "csv_path", "summary_path", "result" (EvalMeter.result() or None), "saved_result",
        "num_views", "num_cam_poses_found", "seconds"}``."""
        t_start = time()
        ds = self.dataset
        meter = saved_meter = None
        if self.saved_detections is not None and self.do_add:
            saved_meter = EvalMeter(self.mesh_db)
        csv_lines, num, num_cam_poses_found = [], 0, 0
        if not self.debug_saved_only:
            if self.do_add:
                meter = EvalMeter(self.mesh_db)
            method = self.method_name()
            outdir = os.path.join(self.model_path, method)
            os.makedirs(outdir, exist_ok=True)
        for scene_id in ds.scene_ids():
            view_ids = ds.view_ids(scene_id)
            if not self.debug_saved_only and self.nviews < 0:
                self.object_slam.reset()
            scene_results = []
            batched = not self.debug_saved_only and self.frames_per_call > 1 and not self.debug_gt_kp and not self.gt_cam_pose
            pending = []                                              # (view_id, gt_obj_ids, process_view arguments) waiting for their shared call

            inflight = []                                             # batches submitted and not yet collected (at most two), oldest first

            def collect_one():
                batch = inflight.pop(0)
                for (vid, gt_ids, _), res in zip(batch, self.object_slam.collect_views_single()):
                    if len(res):
                        scene_results.append((vid, res[vid]["poses"], gt_ids))

            def flush(last=False):
                # a batch is SUBMITTED as soon as it is full and COLLECTED when the next one is on the device (ObjectSLAM.submit_views_single /
                # collect_views_single): the host's bookkeeping of batch i and the preparation of batch i + 1 run under the device work of i + 1
                if pending:
                    batch = list(pending)
                    pending.clear()
                    if self.object_slam.single_views_take_the_device_chain([p[2] for p in batch]):
                        self.object_slam.submit_views_single([p[2] for p in batch])
                        inflight.append(batch)
                        if len(inflight) == 2:
                            collect_one()
                    else:
                        while inflight:
                            collect_one()
                        for (vid, gt_ids, _), res in zip(batch, self.object_slam.process_views_single([p[2] for p in batch])):
                            if len(res):
                                scene_results.append((vid, res[vid]["poses"], gt_ids))
                if last:
                    while inflight:
                        collect_one()
            for j, view_id in enumerate(view_ids):
                gt_obj_ids = ds.obj_ids(scene_id, view_id)
                if batched:
                    a = self._view_args(scene_id, view_id, view_id)
                    if a is not None:
                        pending.append((view_id, gt_obj_ids, a[0]))
                        if len(pending) == self.frames_per_call:
                            flush()
                elif not self.debug_saved_only:
                    views = [view_id]
                    if self.nviews > 1:                               # SfM: nviews-1 random companions (evaluate.py:194-197)
                        views += self._rng.choice(view_ids[:j] + view_ids[j + 1:], size=self.nviews - 1, replace=False).tolist()
                    results = self._run_slam(scene_id, views)
                    if len(results) == 0:
                        continue
                    scene_results.append((view_id, results[view_id]["poses"] if self.nviews > 0 else None, gt_obj_ids))
                if saved_meter is not None:
                    for o in gt_obj_ids:
                        idx = self.saved_detections_map.get(scene_id, {}).get(view_id, {}).get(o)
                        if idx is not None:
                            saved_meter.update([o], np.asarray(self.saved_detections["poses"][idx])[None, ...], ds.get_obj_pose(scene_id, view_id, o)[None, ...])
                        else:
                            saved_meter.update_no_det([o])
            flush(last=True)
            if self.debug_saved_only:
                continue
            final = self.object_slam.collect_results(no_viz=True, final=True) if self.nviews < 0 else None
            for view_id, pred_poses, gt_obj_ids in scene_results:
                num += 1
                if self.nviews < 0:
                    if view_id not in final:
                        if meter is not None:
                            meter.update_no_det(list(gt_obj_ids))
                        continue
                    num_cam_poses_found += 1
                    pred_poses = final[view_id]["poses"]
                for o in gt_obj_ids:
                    res = pred_poses.get(o)
                    if res is not None and res["T_OtoC"] is not None:
                        if meter is not None:
                            meter.update([o], res["T_OtoC"][None, ...], ds.get_obj_pose(scene_id, view_id, o)[None, ...])
                        if ds.is_target(scene_id, view_id, o):
                            csv_lines.append(bop_csv_line(scene_id, view_id, o, res["score"], res["T_OtoC"]))
                    else:
                        self._log(f"NOTE: Could not obtain object pose for object {o}")
                        if meter is not None:                         # the reference updates unconditionally; without ADD there is no meter
                            meter.update_no_det([o])
        out = {"method": None, "csv_path": None, "summary_path": None, "result": None, "saved_result": None, "num_views": num,
               "num_cam_poses_found": num_cam_poses_found,
               # a silent demotion of the network to the bf16x3 form must show in the reported numbers: calls re-issued, and the form it ended on
               "fp16_range_reissues": int(getattr(self.object_slam, "fp16_range_reissues", 0)),
               "matrix_pipe_at_end": (None if self.object_slam.model is None else {0: "f32", 1: "bf16x3", 2: "f16x2"}[self.object_slam.model.pipe()])}
        gt_obj_map = YCBV_CLASSES if ds.bop_dset == "ycbv" else TLESS_CLASSES
        if saved_meter is not None:
            out["saved_result"] = saved_meter.result()
            self._log(saved_meter.pprint_objs_str(gt_obj_map))
            saved_meter.close()
        if not self.debug_saved_only:
            out["method"] = method
            out["summary_path"] = os.path.join(outdir, "summary.txt")
            with open(out["summary_path"], "w") as f:
                if meter is not None:
                    f.write(meter.pprint_objs_str(gt_obj_map))
                if num > 0:
                    for s in (f"NOTE: {100 * num_cam_poses_found / num:.1f}% of camera poses found!", self.object_slam.get_tracking_strtime(),
                              self.object_slam.get_global_opt_strtime(), f"Average keypoint stdev: {self.object_slam.avg_std_meter.average()}"):
                        f.write("\n" + s + "\n")
            out["csv_path"] = os.path.join(outdir, method + ".csv")
            with open(out["csv_path"], "w") as f:
                f.writelines(csv_lines)
            if meter is not None:
                out["result"] = meter.result()
                meter.close()
            if ds.bop_dset == "tless":                             # evaluate.py:323-336: VSD recall via bop_toolkit (N4)
                argv, env, cwd = bop_eval_command(out["csv_path"], outdir, ds.targets_filename, self.repo_root)
                out["bop_eval"] = {"argv": argv, "env": env, "cwd": cwd, "returncode": None}
                if self.run_bop_eval:
                    import subprocess
                    assert os.path.exists(env["PYTHONPATH"]), "thirdparty/bop_toolkit is not mounted"
                    out["bop_eval"]["returncode"] = subprocess.call(argv, env={**os.environ, **env}, cwd=cwd)
        out["seconds"] = time() - t_start
        return out

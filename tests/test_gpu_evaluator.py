"""The evaluation harness end to end on the GPU: BOP tree -> BopDataset -> ObjectSLAM.process_view
(HIP network / PnP / bundle adjustment) -> EvalMeter (HIP ADD-S) -> summary + BOP CSV."""
import os

import numpy as np
import pytest

from oracle import eval_oracle as EO
from suo_slam_amd import bop, evaluator
from suo_slam_amd import eval_meter as EM
from tests import bop_tree

pytestmark = pytest.mark.gpu


def _parse_csv(path):
    rows = []
    for ln in open(path).read().strip().split("\n"):
        if not ln:
            continue
        s, v, o, score, R, t, tm = ln.split(",")
        rows.append((int(s), int(v), int(o), float(score), np.array(R.split(), float).reshape(3, 3), np.array(t.split(), float), tm))
    return rows


def test_single_view_ground_truth_keypoints_recovers_the_poses(tmp_path):
    """The reference's --debug_gt_kp mode (object_slam.py:1129-1131): projected model keypoints + 0.01 NDC noise drive
    PnP + LM; every object with >= 4 keypoints inside its box must come back within millimetres, the table says so,
    and the CSV carries one well-formed line per target."""
    desc = bop_tree.build(str(tmp_path), dset="ycbv", seed=21, n_scenes=2, n_views=3)
    ev = evaluator.Evaluator("ycbv", desc["data_root"], None, nviews=1, debug_gt_kp=True, out_dir=str(tmp_path / "out"))
    out = ev.run()
    ds = ev.dataset
    assert out["method"] == "pkpnet-epoch=-1-nviews=1-det=gt-GT-KP-NO-COV_ycbv-test"
    assert os.path.exists(out["summary_path"]) and os.path.exists(out["csv_path"])
    n_views = sum(len(ds.view_ids(s)) for s in ds.scene_ids())
    assert out["num_views"] == n_views
    rows = _parse_csv(out["csv_path"])
    seen, good = set(), 0
    for s, v, o, score, R, t, tm in rows:
        assert tm == "-1" and score >= 1 and (s, v, o) not in seen
        seen.add((s, v, o))
        gt = ds.get_obj_pose(s, v, o)
        assert abs(np.linalg.det(R) - 1) < 1e-6
        if np.linalg.norm(t - gt[:3, 3]) < 0.02 * gt[2, 3] and np.linalg.norm(R - gt[:3, :3]) < 0.1:
            good += 1
    # objects whose box holds >= 4 keypoints are solvable; the builder's deliberately tiny box is not
    solvable = 0
    for s in ds.scene_ids():
        for v in ds.view_ids(s):
            sample = ds.get_all_obj(s, v)
            solvable += int((sample["kp_masks"].numpy().sum(1) >= 4).sum())
    assert len(rows) >= solvable - 1 and good >= len(rows) - 1, (len(rows), solvable, good)
    res = out["result"]
    assert res["AUC of ADD-S"][0] >= res["AUC of ADD"][0] - 1e-9
    per_obj = res["AUC of ADD(-S)"][1]
    assert max(per_obj.values()) > 0.8
    txt = open(out["summary_path"]).read()
    assert "AUC of ADD(-S):" in txt and "TIMING: Tracking time" in txt and "% of camera poses found" in txt


def test_saved_detections_meter_and_full_network_path(tmp_path):
    """detection_type='saved' with random network weights: the whole path runs (RoIAlign -> hourglass -> decode ->
    masks -> PnP -> LM) on the saved boxes; the saved-detection meter (PoseCNN poses vs ground truth) is checked
    against the oracle."""
    from suo_slam_amd import weights
    desc = bop_tree.build(str(tmp_path), dset="ycbv", seed=22, n_scenes=1, n_views=2)
    reader = bop.BopDataset(desc["data_root"], desc["split"], bop_dset="ycbv", ignore_symmetry=True)
    bop_tree.write_saved_detections(str(tmp_path), desc, reader, seed=3, trans_noise_mm=4.0)
    sd = weights.make_random_state_dict(seed=0, logit_gain=8.0)
    ev = evaluator.Evaluator("ycbv", desc["data_root"], None, nviews=1, detection_type="saved", out_dir=str(tmp_path / "out"), state_dict=sd)
    out = ev.run()
    assert out["method"].startswith("pkpnet-epoch=-1-nviews=1-det=saved") and os.path.exists(out["csv_path"])
    # oracle replay of the saved-detection meter
    det, dmap, mesh = ev.saved_detections, ev.saved_detections_map, ev.mesh_db
    errs = {}
    for s in reader.scene_ids():
        for v in reader.view_ids(s):
            for o in reader.obj_ids(s, v):
                idx = dmap.get(s, {}).get(v, {}).get(o)
                if idx is None:
                    errs.setdefault(o, []).append(np.inf)
                else:
                    add, adds = EO.pose_errors(mesh[o]["points"], det["poses"][idx], reader.get_obj_pose(s, v, o))
                    errs.setdefault(o, []).append(float(adds if mesh[o]["is_symmetric"] else add))
    # (the oracle, like the reference, cannot take a single-element list; the product's host AUC handles those)
    want = np.mean([float(EO.compute_auc_posecnn(e)) if len(e) > 1 else float(EM.compute_auc_posecnn(e)) for e in errs.values()])
    got = out["saved_result"]["AUC of ADD(-S)"][0]
    assert abs(got - want) < 2e-4, (got, want)
    assert np.isfinite(got) and 0.3 < got <= 1.0                   # 4 mm noise, 1 in 5 detections missing


def test_slam_mode_over_a_consistent_sequence(tmp_path):
    """nviews=-1: one pass over a geometrically consistent 12-view sequence with ground-truth keypoints; camera poses
    are found for every view and the final (globally optimised) object poses match ground truth."""
    desc = bop_tree.build_sequence(str(tmp_path), seed=5, n_views=12, n_objs=5)
    ev = evaluator.Evaluator("ycbv", desc["data_root"], None, nviews=-1, debug_gt_kp=True, out_dir=str(tmp_path / "out"))
    out = ev.run()
    assert out["num_views"] == 12 and out["num_cam_poses_found"] == 12
    rows = _parse_csv(out["csv_path"])
    assert len(rows) >= 12 * 5 - 5
    ds = ev.dataset
    bad = 0
    for s, v, o, score, R, t, tm in rows:
        gt = ds.get_obj_pose(s, v, o)
        if not (np.linalg.norm(t - gt[:3, 3]) < 0.02 * gt[2, 3] and np.linalg.norm(R - gt[:3, :3]) < 0.1):
            bad += 1
    assert bad <= 3, bad
    assert out["result"]["AUC of ADD-S"][0] > 0.7

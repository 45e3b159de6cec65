"""Host-side rows against vectors produced by the reference's own modules (tests/golden/make_host_golden.py imports
lib/utils/eval_meter.py and lib/utils/utils.py).  CPU only.

  N1  oracle/eval_oracle.py AND the product's host AUC code (suo_slam_amd/eval_meter.py) vs the reference's
      compute_auc_posecnn / AddAucMeter / EvalMeter outputs
  a9  fix_K_for_bbox_ndc
  a25 make_prior_kp_input index logic (window rectangles exact; patch values to the documented 1.2e-2 of the ring)
  N3  saved-detection loaders
"""
import os

import numpy as np
import pytest

from oracle import eval_oracle as EO
from suo_slam_amd import detections as D
from suo_slam_amd import eval_meter as EM
from suo_slam_amd import geometry as geo
from suo_slam_amd import object_slam as OS

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "host_golden.npz"))


# ---- N1 ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl", [EO.compute_auc_posecnn, EM.compute_auc_posecnn], ids=["oracle", "product"])
def test_auc_matches_reference(impl):
    for i in range(int(GOLD["auc_n"])):
        got = impl(GOLD[f"auc_in_{i}"].tolist())
        assert abs(float(got) - float(GOLD[f"auc_out_{i}"])) <= 1e-12, (i, got, GOLD[f"auc_out_{i}"])


def test_auc_edge_cases_of_the_product():
    assert EM.compute_auc_posecnn([np.inf, np.inf]) == 0
    assert EM.compute_auc_posecnn([]) == 0
    # one perfect detection: the curve is 1 from 0 to 10 cm; the reference cannot evaluate a single-element list
    assert abs(EM.compute_auc_posecnn([0.0]) - 1.0) < 1e-12
    # the VOC-style rectangle rule credits the first recall step with the precision reached AFTER it
    assert abs(EM.compute_auc_posecnn([50.0, 50.0]) - float(EO.compute_auc_posecnn([50.0, 50.0]))) < 1e-15
    assert abs(EM.compute_auc_posecnn([50.0]) - 1.0) < 1e-12
    # ndarray input keeps its dtype (float64 path), list input is float32 like the reference
    e = np.array([10.0, 20.0, 30.0, 250.0])
    assert abs(EM.compute_auc_posecnn(e) - EO.compute_auc_posecnn(e)) < 1e-15


def test_add_auc_meter_matches_reference():
    ids, errs = GOLD["aam_ids"].tolist(), GOLD["aam_errs"].tolist()
    for flag in (True, False):
        want_total = float(GOLD[f"aam_total_{int(flag)}"])
        want_per = {int(k): v for k, v in GOLD[f"aam_per_{int(flag)}"]}
        tot, per = EO.auc_meter_average(ids, errs, flag)
        assert abs(tot - want_total) < 1e-12 and all(abs(per[k] - want_per[k]) < 1e-12 for k in want_per)
        m = EM.AddAucMeter(obj_avg=flag)
        m.update(ids[:50], errs[:50])
        m.update(ids[50:], errs[50:])
        tot, per = m.average()
        assert abs(tot - want_total) < 1e-12 and set(per) == set(want_per)
        assert all(abs(per[k] - want_per[k]) < 1e-12 for k in want_per)


def test_oracle_pose_errors_match_reference_eval_meter():
    """ADD / ADD-S of the 40 recorded updates.  fp32 evaluation order differs (BLAS matmul vs explicit sums), so the
    bound is a few ulp of the ~1 m coordinates: 3e-4 mm absolute + 1e-5 relative."""
    ids, pred, gt = GOLD["em_ids"], GOLD["em_pred"], GOLD["em_gt"]
    sym = {int(k): bool(v) for k, v in GOLD["em_sym"]}
    seen = {k: 0 for k in sym}
    for k, oid in enumerate(ids.tolist()):
        add, adds = EO.pose_errors(GOLD[f"em_pts_{oid}"], pred[k], gt[k])
        # the reference's per-object lists interleave inf entries from update_no_det: skip them
        ref_add = GOLD[f"em_add_errs_{oid}"]
        ref_adds = GOLD[f"em_adds_errs_{oid}"]
        ref_ms = GOLD[f"em_addms_errs_{oid}"]
        while not np.isfinite(ref_add[seen[oid]]):
            seen[oid] += 1
        j = seen[oid]
        seen[oid] += 1
        assert abs(add - ref_add[j]) <= 3e-4 + 1e-5 * ref_add[j], (k, add, ref_add[j])
        assert abs(adds - ref_adds[j]) <= 3e-4 + 1e-5 * ref_adds[j], (k, adds, ref_adds[j])
        assert ref_ms[j] == (ref_adds[j] if sym[oid] else ref_add[j])
        assert adds <= add + 1e-6


# ---- a9 ----------------------------------------------------------------------------------------------
def test_fix_K_for_bbox_ndc_matches_reference():
    for K, bb, want in zip(GOLD["fixk_K"], GOLD["fixk_bbox"], GOLD["fixk_out"]):
        got = geo.fix_K_for_bbox_ndc(K, bb)
        assert np.allclose(got, want, rtol=1e-13, atol=1e-13)


# ---- a25 ---------------------------------------------------------------------------------------------
def _rects(x):
    r = np.full((x.shape[0], 6), -1, np.int32)
    for c in range(x.shape[0]):
        ys, xs = np.nonzero(x[c])
        if len(ys):
            my, mx = np.unravel_index(np.argmax(x[c]), x[c].shape)
            r[c] = [ys.min(), ys.max() + 1, xs.min(), xs.max() + 1, my, mx]
    return r


def _patch_values_match(got, gold):
    """The fixture's patch comes from a cv2 stand-in WITHOUT OpenCV's border reflection, stored as float16: inside the patch the values agree
    to float16 resolution; the outermost ring (<= 1.2e-2 of the peak) carries the doubled BORDER_REFLECT_101 samples, which
    tests/test_oracle_cnn.py pins against F.conv2d + reflect padding to 1e-6."""
    d = np.abs(got - gold)
    assert d[gold > 1.2e-2].max() < 5e-4
    assert d.max() < 1.3e-2


def test_prior_stamp_windows_match_reference_index_logic():
    got = OS.make_prior_kp_input(GOLD["prior_kp"], GOLD["prior_mask"], (256, 256), ndc=True)
    assert np.array_equal(_rects(got), GOLD["prior_ndc_rect"])
    # values: float16 fixture + the doubled outer ring of BORDER_REFLECT_101 (max 2 * 5.7e-3), see object_slam._gaussian_patch
    _patch_values_match(got, GOLD["prior_ndc"].astype(np.float32))
    px = GOLD["prior_px"]
    got = OS.make_prior_kp_input(px, np.ones(len(px), bool), (480, 640), ndc=False)
    assert np.array_equal(_rects(got), GOLD["prior_px_rect"])
    _patch_values_match(got, GOLD["prior_px_out"].astype(np.float32))


# ---- N3 ----------------------------------------------------------------------------------------------
def _bop_root(tmp_path):
    (tmp_path / "saved_detections").mkdir()
    (tmp_path / "ycbv").mkdir()
    (tmp_path / "saved_detections" / "ycbv_posecnn.pkl").write_bytes(GOLD["det_posecnn_pkl"].tobytes())
    (tmp_path / "saved_detections" / "tless_pix2pose_retinanet_siso_top1.pkl").write_bytes(GOLD["det_pix2pose_pkl"].tobytes())
    (tmp_path / "ycbv" / "offsets.txt").write_bytes(GOLD["det_offsets_txt"].tobytes())
    return str(tmp_path)


@pytest.mark.parametrize("tag,loader", [("posecnn", D.load_posecnn_results), ("pix2pose", D.load_pix2pose_results)])
def test_saved_detection_loaders_match_reference(tmp_path, tag, loader):
    d = loader(_bop_root(tmp_path))
    assert set(d) == {"scene_ids", "view_ids", "scores", "obj_ids", "poses", "bboxes"}
    for k in ("scene_ids", "view_ids", "scores", "obj_ids"):
        assert np.array_equal(np.array(d[k], np.float64), GOLD[f"det_{tag}_{k}"]), k
    assert np.array_equal(np.array(d["bboxes"], np.float64), GOLD[f"det_{tag}_bboxes"])
    poses = np.array([np.asarray(p)[:3, :4] for p in d["poses"]])
    assert np.allclose(poses, GOLD[f"det_{tag}_poses"], rtol=0, atol=1e-9)


def test_detection_map_filters_by_targets_and_rejects_duplicates(tmp_path):
    d = D.load_posecnn_results(_bop_root(tmp_path))
    m = D.build_detection_map(d)
    n = sum(len(v) for s in m.values() for v in s.values())
    assert n == len(d["obj_ids"])
    for i, (s, v, o) in enumerate(zip(d["scene_ids"], d["view_ids"], d["obj_ids"])):
        assert m[s][v][o] == i
    s0, v0, o0 = d["scene_ids"][0], d["view_ids"][0], d["obj_ids"][0]
    m = D.build_detection_map(d, targets={s0: {v0: [o0]}})
    assert m[s0][v0] == {o0: 0} and sum(len(v) for s in m.values() for v in s.values()) == 1
    dup = {k: list(v) + [v[0]] for k, v in d.items()}
    with pytest.raises(AssertionError):
        D.build_detection_map(dup)

"""BASELINE configs[3] on the GPU: the T-LESS configuration of the reference's evaluate.py (:66-76 -- kp_var_thresh 0.5,
bbox_thresh 1.0, manual_kp_std 0.1, opt_init_with_outliers=True, split test_primesense, no ADD meter) with saved Pix2Pose /
RetinaNet detections (boxes stored (y1,x1,y2,x2) and swapped, lib/utils/utils.py:557-561), on 8- and 10-keypoint classes
(kp_configs/tless_kp_config.csv).  Every call the run makes into libsuo_hip.so is recorded and replayed through the oracle
on the same inputs (tests/replay.py): network + masks vs the torch-CPU oracle, PnP and LM vs the C oracles."""
import os

import numpy as np
import pytest

from suo_slam_amd import bop, evaluator, kp_config
from tests import bop_tree, replay

pytestmark = pytest.mark.gpu


def _confident_weights():
    """Random weights whose keypoint classifier says "visible" (bias +4): with the stock random classifier nothing passes
    kp_mask > 0.3 and the geometry half of the path would never run on network output."""
    from suo_slam_amd import weights
    sd = weights.make_random_state_dict(seed=0, logit_gain=8.0)
    sd["classifier.2.bias"] = (np.asarray(sd["classifier.2.bias"]) + 4.0).astype(np.float32)
    return sd


def test_tless_saved_detections_single_view_replays_against_the_oracle(tmp_path):
    desc = bop_tree.build(str(tmp_path), dset="tless", seed=31, n_scenes=2, n_views=2)
    reader = bop.BopDataset(desc["data_root"], desc["split"], bop_dset="tless", ignore_symmetry=True)
    bop_tree.write_saved_detections_pix2pose(str(tmp_path), desc, reader, seed=4)
    sd = _confident_weights()
    ev = evaluator.Evaluator("tless", desc["data_root"], None, nviews=1, detection_type="saved", out_dir=str(tmp_path / "out"), state_dict=sd)
    slam = ev.object_slam
    # the per-dataset settings of evaluate.py:66-76 reached the driver
    assert (slam.kp_var_thresh, slam.bbox_thresh, slam.manual_kp_std, slam.opt_init_with_outliers) == (0.5, 1.0, 0.1, True)
    assert ev.do_add is False and ev.dataset.split == "test_primesense"
    with replay.record() as rec:
        out = ev.run()
    assert out["result"] is None and out["saved_result"] is None            # no ADD meter on T-LESS (evaluate.py:72)
    assert out["method"] == "pkpnet-epoch=-1-nviews=1-det=saved_tless-test_primesense" and os.path.exists(out["csv_path"])
    # the VSD hand-off of evaluate.py:323-336 is prepared (not started: bop_toolkit is not mounted here)
    be = out["bop_eval"]
    assert be["argv"][:4] == ["python", "scripts/eval_siso.py", "--renderer_type", "python"] and be["returncode"] is None
    assert be["argv"][be["argv"].index("--result_filename") + 1] == os.path.realpath(out["csv_path"])
    assert be["argv"][be["argv"].index("--targets_filename") + 1].endswith("all_target_tless.json")
    # --- boxes: the column swap happened, and the network saw exactly the saved boxes ------------------------------
    det = ev.saved_detections
    n_frames = 0
    for c in rec.forward:
        n_frames += 1
        for b in c["boxes"]:
            assert any(np.array_equal(b, np.asarray(sb, np.float32)) for sb in det["bboxes"])
            assert b[2] - b[0] > 5 and b[3] - b[1] > 5 and b[2] <= 660 and b[3] <= 500
    assert n_frames >= 4
    # --- classes: 8 keypoints for box-like, 10 for cylinder-like objects, nothing else ever reaches PnP -------------
    seen = set()
    for m in rec.masks:
        assert (m["bt"], m["vt"]) == (1.0, 0.5)
        for row, mm in zip(m["out"], m["mm"]):
            n_model = int(np.count_nonzero(mm))
            assert n_model in (8, 10) and not np.any(row & ~mm.astype(bool))
            seen.add(n_model)
    assert seen == {8, 10}
    # --- replay through the oracle ----------------------------------------------------------------------------------
    assert replay.check_network(rec, sd) >= 4
    # single-view frames run the device chain (csrc/frame_geom.hip): every launch against the PnP / LM oracles on its own device inputs
    assert len(rec.pnp) == 0 and len(rec.ba) == 0 and len(rec.chain) >= 4
    n_pnp, n_lm = replay.check_chain(rec)
    assert n_pnp >= 6 and n_lm >= 3, (n_pnp, n_lm)
    for c in rec.chain:                                                      # single-view graphs: rounds [10,10,40,40], predicted covariances
        assert tuple(c["its"]) == (10, 10, 40, 40) and c["use_cov"] and c["do_lm"] and len(c["frame_first"]) == 2
    # --- CSV: one line per target with a pose, BOP format ------------------------------------------------------------
    lines = [ln for ln in open(out["csv_path"]).read().strip().split("\n") if ln]
    for ln in lines:
        s, v, o, score, R, t, tm = ln.split(",")
        assert ev.dataset.is_target(int(s), int(v), int(o)) and tm == "-1" and len(R.split()) == 9 and len(t.split()) == 3


def test_tless_slam_mode_starts_current_view_rounds_with_outliers(tmp_path):
    """nviews=-1 on a consistent T-LESS-shaped sequence with ground-truth keypoints: the current-view adjustments carry
    opt_init_with_outliers (evaluate.py:75 -> object_slam.py:848-852: every edge starts at level 0), the global ones do not;
    all of them replay against the oracle, and the objects come back."""
    desc = bop_tree.build_sequence(str(tmp_path), seed=6, n_views=11, n_objs=5, dset="tless")
    ev = evaluator.Evaluator("tless", desc["data_root"], None, nviews=-1, debug_gt_kp=True, out_dir=str(tmp_path / "out"))
    with replay.record() as rec:
        out = ev.run()
    assert out["num_views"] == 11 and out["num_cam_poses_found"] == 11
    curr = [b for b in rec.ba if b["obj_fixed"].all()]
    glob = [b for b in rec.ba if not b["obj_fixed"].any()]
    assert len(curr) == 11 and len(glob) >= 2 and len(curr) + len(glob) == len(rec.ba)
    assert all(b["init_with_outliers"] and tuple(b["its"]) == (10, 10, 10, 10) for b in curr)
    assert all((not b["init_with_outliers"]) and tuple(b["its"]) == (10, 10, 40, 40) for b in glob)
    assert replay.check_pnp(rec) >= 40
    assert replay.check_ba(rec) == len(rec.ba)
    ds = ev.dataset
    good = n = 0
    for ln in open(out["csv_path"]).read().strip().split("\n"):
        s, v, o, score, R, t, tm = ln.split(",")
        gt = ds.get_obj_pose(int(s), int(v), int(o))
        n += 1
        good += int(np.linalg.norm(np.array(t.split(), float) - gt[:3, 3]) < 0.03 * gt[2, 3])
    assert n >= 11 * 5 - 6 and good >= n - 4, (n, good)
    assert {len(kp_config.kp_list_of("tless", o)) for o in next(iter(desc["scenes"].values()))[1]} <= {8, 10}


def test_tless_frame_shape_with_large_boxes_matches_the_oracle_end_to_end():
    """BASELINE configs[3]'s frame shape through the whole network: a 720 x 540 frame whose eight detections are five boxes <= 256 px, two of 256-512 px and one
    > 512 px (suo_slam_amd.synthetic.make_frame_tless: the sizes saved T-LESS detections produce) -- the fused RoIAlign + stem launch takes torchvision's adaptive
    1, 2 and 3 samples per bin and axis (pkpnet.py:93) -- against the CPU oracle on the same inputs, at the network tolerance (1e-5)."""
    import torch
    from oracle import cnn_oracle as O
    from suo_slam_amd import synthetic as S
    from suo_slam_amd import weights
    from suo_slam_amd.pkpnet import PkpNet
    fr = S.make_frame_tless(np.random.default_rng(77))
    boxes = fr["boxes"].astype(np.float32)
    side = np.maximum(boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1])
    assert fr["image"].shape == (540, 720, 3) and (side <= 256).any() and ((side > 256) & (side <= 512)).any() and (side > 512).any()
    sd = weights.make_random_state_dict(seed=0, logit_gain=8.0)
    net = PkpNet(state_dict=sd, max_crops=8)
    out = net(fr["image"], [torch.from_numpy(boxes)], None)
    ref = O.pkpnet_forward(fr["image"], boxes, None, sd)
    lg, lr = out["prob_logits"].cpu().numpy(), ref["prob_logits"].numpy()
    assert np.abs(lg - lr).max() / np.abs(lr).max() < 1e-5
    np.testing.assert_allclose(out["uv"].cpu().numpy(), ref["uv"].numpy(), atol=1e-5, rtol=0)
    np.testing.assert_allclose(out["cov"].cpu().numpy(), ref["cov"].numpy(), atol=1e-5, rtol=0)
    np.testing.assert_allclose(out["kp_mask"].cpu().numpy(), ref["kp_mask"].numpy(), atol=1e-5, rtol=0)


def test_schedule_bytes_accounts_every_launch_of_a_call():
    """suo_net_schedule_bytes (bench.py's roofline_all.whole_call): a dry run of the launch schedule -- nothing runs, the next forward is unaffected -- whose per-kind
    sums behave like the schedule: decode reads every logit once, a call of 2L crops moves about twice the bytes of one of L (the weights are counted once per
    launch; the kernels chosen change with the launch size), the prior pass stages 48 channels where the prior-less pass stages none, and the launch count is that of the kernel trace (tools/profile_round.sh)."""
    import torch
    from suo_slam_amd import weights
    from suo_slam_amd.pkpnet import PkpNet
    sd = weights.make_random_state_dict(seed=0, logit_gain=8.0)
    net = PkpNet(state_dict=sd, max_crops=32)
    rng = np.random.default_rng(5)
    img = (rng.uniform(0, 1, (480, 640, 3)) * 255).astype(np.uint8)
    boxes = np.array([[100, 80, 300, 290], [350, 100, 600, 400]], np.float32)
    before = net(img, [torch.from_numpy(boxes)], None)["prob_logits"].clone()
    b8, b16, b32 = net.schedule_bytes(8), net.schedule_bytes(16, 2), net.schedule_bytes(32, 4)
    assert torch.equal(net(img, [torch.from_numpy(boxes)], None)["prob_logits"], before)
    for b in (b8, b16, b32):
        assert all(b[k] > 0 for k in PkpNet.SCHEDULE_KINDS if k != "one_launch_blocks") and abs(sum(b[k] for k in PkpNet.SCHEDULE_KINDS) - b["total"]) < 1.0
    assert b8["decode_classifier"] >= 4.0 * 8 * 41 * 4096
    # (not exactly "< 2x": a larger call leaves the one-launch blocks for per-layer launches, whose intermediate tensors travel)
    assert 1.5 * b8["total"] < b16["total"] < 2.3 * b8["total"] and 1.5 * b16["total"] < b32["total"] < 2.3 * b16["total"]
    # a crop's compulsory traffic is at least its 59 Residual blocks' inputs and outputs at 64x64 alone ... and far below one fp32 pass per layer over 169 MB of slabs
    assert 8 * 20e6 < b8["total"] < 8 * 400e6
    assert 60 <= b32["launches"] <= 400
    wp = net.schedule_bytes(8, 1, with_priors=True)
    assert wp["staging_stem"] > b8["staging_stem"] and wp["total"] > b8["total"]

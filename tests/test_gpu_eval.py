"""ADD / ADD-S meter on the GPU (csrc/eval.hip through the C ABI) vs the reference's recorded outputs and the oracle."""
import os

import numpy as np
import pytest

from oracle import eval_oracle as EO
from suo_slam_amd import eval_meter as EM

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "host_golden.npz"))
TOL_ABS, TOL_REL = 3e-4, 1e-5        # mm: a few fp32 ulp of ~1 m coordinates (the reference itself is fp32)


def _golden_mesh_db():
    sym = {int(k): bool(v) for k, v in GOLD["em_sym"]}
    return {oid: {"points": GOLD[f"em_pts_{oid}"], "is_symmetric": s, "diameter": 150.0} for oid, s in sym.items()}


def _rand_pose(rng, z=800.0):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, zq = q
    R = np.array([[1 - 2 * (y * y + zq * zq), 2 * (x * y - zq * w), 2 * (x * zq + y * w)],
                  [2 * (x * y + zq * w), 1 - 2 * (x * x + zq * zq), 2 * (y * zq - x * w)],
                  [2 * (x * zq - y * w), 2 * (y * zq + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.standard_normal(3) * 50 + [0, 0, z]
    return T


def test_eval_meter_replays_the_reference_session():
    """The 40 updates (+ missed detections) recorded from the reference's EvalMeter: per-update errors, the three
    AUCs, the per-object AUCs and the printed table."""
    meter = EM.EvalMeter(_golden_mesh_db())
    ids, pred, gt = GOLD["em_ids"].tolist(), GOLD["em_pred"], GOLD["em_gt"]
    for k, oid in enumerate(ids):
        meter.update([oid], pred[k][None], gt[k][None])
        if k % 9 == 4:
            meter.update_no_det([oid])
    for name, m in (("add", meter.add_meter), ("adds", meter.adds_meter), ("addms", meter.add_maybe_s_meter)):
        for oid in (1, 2, 3, 4):
            got, want = np.array(m.err_map[oid], np.float64), GOLD[f"em_{name}_errs_{oid}"]
            assert got.shape == want.shape and np.array_equal(np.isfinite(got), np.isfinite(want))
            f = np.isfinite(want)
            assert np.all(np.abs(got[f] - want[f]) <= TOL_ABS + TOL_REL * want[f]), (name, oid, np.abs(got[f] - want[f]).max())
    res = meter.result()
    for key, tag in (("AUC of ADD", "add"), ("AUC of ADD-S", "adds"), ("AUC of ADD(-S)", "addms")):
        assert abs(res[key][0] - float(GOLD[f"em_auc_{tag}"])) < 2e-5          # errors move by <=3e-4 mm of a 100 mm axis
        for oid, want in GOLD[f"em_auc_{tag}_per"]:
            assert abs(res[key][1][int(oid)] - want) < 2e-5
    assert meter.pprint_objs_str({1: "alpha", 2: "beta", 3: "gamma", 4: "delta", 5: "absent"}) == str(GOLD["em_table"])
    meter.close()


@pytest.mark.parametrize("P", [1, 5, 255, 257, 1024, 1025, 4099])
def test_pose_errors_vs_oracle_ragged_sizes(P):
    """Tile / split edges of the pair kernel: clouds smaller than a block, one over a block, one over an LDS tile."""
    rng = np.random.default_rng(P)
    pts = (rng.standard_normal((P, 3)) * [40, 25, 60]).astype(np.float32)
    meter = EM.EvalMeter({7: {"points": pts, "is_symmetric": True}})
    gt = _rand_pose(rng)
    for scale in (0.0, 0.01, 0.5, 3.0):
        pr = gt.copy()
        pr[:3, 3] += rng.standard_normal(3) * 20 * scale
        pr[:3, :3] = _rand_pose(rng)[:3, :3] if scale > 1 else pr[:3, :3]
        add, adds = meter.pose_errors([7], pr[None], gt[None])
        o_add, o_adds = EO.pose_errors(pts, pr, gt)
        assert abs(add[0] - o_add) <= TOL_ABS + TOL_REL * o_add
        assert abs(adds[0] - o_adds) <= TOL_ABS + TOL_REL * o_adds
        if scale == 0.0:
            assert add[0] == 0.0 and adds[0] == 0.0
    meter.close()


def test_pose_errors_batched_mixed_models_equals_one_by_one():
    rng = np.random.default_rng(3)
    db = {o: {"points": (rng.standard_normal((n, 3)) * 50).astype(np.float32), "is_symmetric": bool(o % 2)} for o, n in ((1, 300), (2, 2000), (3, 77), (4, 1500))}
    meter = EM.EvalMeter(db)
    ids = [int(rng.integers(1, 5)) for _ in range(23)]
    gt = np.stack([_rand_pose(rng) for _ in ids])
    pr = gt.copy()
    pr[:, :3, 3] += rng.standard_normal((len(ids), 3)) * 8
    add, adds = meter.pose_errors(ids, pr, gt)
    for k, o in enumerate(ids):
        a1, s1 = meter.pose_errors([o], pr[k][None], gt[k][None])
        assert a1[0] == add[k] and s1[0] == adds[k]                  # bit-identical: min / sums are order-independent here
    # a pure translation: ADD equals its length exactly-ish, ADD-S can only be smaller
    d = np.linalg.norm(pr[:, :3, 3] - gt[:, :3, 3], axis=1)
    assert np.allclose(add, d, rtol=1e-5, atol=3e-4) and np.all(adds <= add + 1e-6)
    meter.close()


def test_large_cloud_properties():
    """BOP-sized cloud (P = 40k, 1.6e9 pairs): identity pose -> 0; ADD-S is invariant to a permutation of the
    predicted cloud's symmetry (a 180-degree flip of a point-symmetric cloud) where ADD is not."""
    rng = np.random.default_rng(11)
    half = (rng.standard_normal((20000, 3)) * [40, 40, 90]).astype(np.float32)
    pts = np.concatenate([half, half * np.float32([-1, -1, 1])])       # symmetric under rotation by pi about z
    meter = EM.EvalMeter({1: {"points": pts, "is_symmetric": True}})
    gt = _rand_pose(rng)
    add, adds = meter.pose_errors([1], gt[None], gt[None])
    assert add[0] == 0 and adds[0] == 0
    flip = gt.copy()
    flip[:3, :3] = gt[:3, :3] @ np.diag([-1.0, -1.0, 1.0])
    add, adds = meter.pose_errors([1], flip[None], gt[None])
    assert add[0] > 30 and adds[0] < 1e-3
    meter.close()


def test_sampled_points_and_errors():
    rng = np.random.default_rng(5)
    db = {1: {"points": (rng.standard_normal((900, 3)) * 50).astype(np.float32), "is_symmetric": False},
          2: {"points": (rng.standard_normal((700, 3)) * 50).astype(np.float32), "is_symmetric": True}}
    meter = EM.EvalMeter(db, sample_n_points=256)
    assert db[1]["points_sampled"].shape == (256, 3)
    gt = np.stack([_rand_pose(rng), _rand_pose(rng)])
    pr = gt.copy()
    pr[:, :3, 3] += 5
    meter.update([1, 2], pr, gt)
    o = EO.pose_errors(db[2]["points_sampled"], pr[1], gt[1])
    assert abs(meter.adds_meter.err_map[2][0] - o[1]) <= TOL_ABS + TOL_REL * o[1]
    assert meter.add_maybe_s_meter.err_map[1][0] == meter.add_meter.err_map[1][0]
    assert meter.add_maybe_s_meter.err_map[2][0] == meter.adds_meter.err_map[2][0]
    with pytest.raises(KeyError):
        meter.pose_errors([9], pr[:1], gt[:1])
    meter.close()

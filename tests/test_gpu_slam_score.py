"""csrc/slam_score.hip: the chi-square inlier counts of __estimate_camera_pose (/root/reference/lib/object_slam.py:1032-1066) and
__maybe_reinit_objects (:648-681) on the device -- counts EXACT against the per-detection host rule (tests/host_scoring.py) and against
oracle/slam_rules.py's restatement; then the CPU suite's rule tests (random states, boundary states, the reference's golden states) re-run with
the kernel in place of the injected host scoring."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import slam_rules as R  # noqa: E402
from suo_slam_amd import slam_score as SC  # noqa: E402
from suo_slam_amd.object_slam import CHI2_2DOF_95  # noqa: E402
from tests import host_scoring as HS  # noqa: E402
from tests import test_host_logic as TH  # noqa: E402
from tests import test_slam_golden as TG  # noqa: E402
from tests import test_slam_rules as TR  # noqa: E402


class Owner:
    pass


def _cases(rng, with_cov):
    slam, fr = TH._slam_with_state(rng, noise=0.004)
    Ts, dets = [], []
    for rep in range(8):
        for k, o in enumerate(fr["obj_ids"]):
            d = dict(slam.detections[0][o])
            d.pop("_cache", None)
            d.pop("_slot", None)
            if not with_cov:
                d["cov_pred"] = None
            n = len(d["model_kp"])
            d["inliers"] = rng.uniform(size=n) < 0.7
            T = fr["T_OtoC"][k].copy()
            if rep == 1:
                T[:3, 3] += rng.normal(scale=3.0, size=3)            # a few keypoints drop out
            elif rep == 2:
                T[2, 3] = -T[2, 3]                                   # everything behind the camera
            elif rep == 3:
                T[2, 3] = 0.02                                       # some behind, some in front
            elif rep == 4:
                d["inliers"] = np.zeros(n, bool)
            elif rep == 5:                                           # empty detection
                d.update(model_kp=d["model_kp"][:0], uv_pred=d["uv_pred"][:0], inliers=np.zeros(0, bool),
                         cov_pred=None if d["cov_pred"] is None else d["cov_pred"][:0])
            elif rep == 6 and with_cov:                              # variances below the 1e-4 clamp (:669,:1054), correlated covariances
                c = np.array(d["cov_pred"], dtype=np.float32)
                c[::2, 0, 0] = 1e-6
                c[1::2, 1, 1] = 3e-5
                c[:, 0, 1] = c[:, 1, 0] = 0.3 * np.sqrt(c[:, 0, 0] * c[:, 1, 1])
                d["cov_pred"] = c
                T[:3, 3] += rng.normal(scale=0.5, size=3)
            elif rep == 7:                                           # all 41 keypoints
                m = 41
                d.update(model_kp=rng.uniform(-40, 40, (m, 3)), uv_pred=rng.uniform(-1, 1, (m, 2)), inliers=rng.uniform(size=m) < 0.5,
                         cov_pred=None if d["cov_pred"] is None else np.tile(np.eye(2, dtype=np.float32) * 0.5, (m, 1, 1)))
            Ts.append(T)
            dets.append(d)
    return Ts, dets


@pytest.mark.parametrize("with_cov", [True, False])
def test_counts_equal_the_per_detection_rule_and_the_oracle(with_cov):
    rng = np.random.default_rng(5)
    Ts, dets = _cases(rng, with_cov)
    own = Owner()
    for subset in (True, False):
        got = SC.chi2_counts(own, np.stack(Ts), dets, subset, 0.005, CHI2_2DOF_95)
        one = [HS._chi2_inliers(T, d, subset, 0.005) for T, d in zip(Ts, dets)]
        assert list(got) == one
        orc = [R._count_chi2_inliers(T, d["model_kp"][d["inliers"] if subset else slice(None)], d["uv_pred"][d["inliers"] if subset else slice(None)],
                                     None if d["cov_pred"] is None else d["cov_pred"][d["inliers"] if subset else slice(None)], d["K"], 0.005)
               for T, d in zip(Ts, dets)]
        assert list(got) == orc
    assert max(one) > 0 and min(one) == 0
    assert own._score_store.n_slots == len(dets)                      # each detection was written once, whatever the number of calls
    assert len(SC.chi2_counts(own, np.zeros((0, 4, 4)), [], True, 0.005, CHI2_2DOF_95)) == 0


def test_store_grows_and_keeps_old_rows():
    rng = np.random.default_rng(9)
    own = Owner()
    own._score_store = SC.DetectionStore(capacity=4)
    Ts, dets = _cases(rng, True)
    first = SC.chi2_counts(own, np.stack(Ts[:3]), dets[:3], False, 0.005, CHI2_2DOF_95)
    rest = SC.chi2_counts(own, np.stack(Ts), dets, False, 0.005, CHI2_2DOF_95)           # 4 -> 64 slots: reallocation with the first rows copied over
    assert list(rest[:3]) == list(first) and list(rest) == [HS._chi2_inliers(T, d, False, 0.005) for T, d in zip(Ts, dets)]
    d = dict(dets[0], uv_pred=dets[0]["uv_pred"] + 0.5)                                    # a NEW detection built from an old dict: not the old slot
    assert SC.chi2_counts(own, np.stack(Ts[:1]), [d], False, 0.005, CHI2_2DOF_95)[0] == HS._chi2_inliers(Ts[0], d, False, 0.005)


def test_store_is_a_soft_bounded_ring(monkeypatch):
    """A long sequence does not grow the store: a call that would take it past max_slots rows invalidates every tag, starts over at row 0 and writes the call's own
    detections again -- counts unchanged; a single call naming more distinct detections than the bound still gets its rows."""
    rng = np.random.default_rng(11)
    monkeypatch.setenv("SUO_SLAM_STORE_SLOTS", "12")
    own = Owner()
    Ts, dets = _cases(rng, True)
    want = [HS._chi2_inliers(T, d, False, 0.005) for T, d in zip(Ts, dets)]
    st = None
    for lo in range(0, len(dets) - 8, 4):                           # a sliding window of 8 detections, 4 new per call
        got = SC.chi2_counts(own, np.stack(Ts[lo:lo + 8]), dets[lo:lo + 8], False, 0.005, CHI2_2DOF_95)
        st = own._score_store
        assert list(got) == want[lo:lo + 8], lo
        assert st.n_slots <= 12
    assert st.recycled >= 1
    got = SC.chi2_counts(own, np.stack(Ts), dets, False, 0.005, CHI2_2DOF_95)            # one call beyond the bound: served in full
    assert list(got) == want and st.n_slots == len(dets)


def test_reset_starts_a_new_scene_in_the_same_store():
    """ObjectSLAM.reset() (evaluate.py:338: once per scene) rewinds the store: slots are reused, detections of the previous scene are not found by tag."""
    from suo_slam_amd.object_slam import ObjectSLAM
    rng = np.random.default_rng(4)
    Ts, dets = _cases(rng, True)
    slam = ObjectSLAM(None, {1: {"diameter": 100.0, "is_symmetric": False}}, debug_gt_kp=True)
    a = SC.chi2_counts(slam, np.stack(Ts[:8]), dets[:8], False, 0.005, CHI2_2DOF_95)
    st = slam._score_store
    assert st.n_slots == 8
    slam.reset()
    assert slam._score_store is st and st.n_slots == 0
    b = SC.chi2_counts(slam, np.stack(Ts[4:12]), dets[4:12], False, 0.005, CHI2_2DOF_95)      # 4 of them carry tags of the old scene
    assert st.n_slots == 8 and list(b) == [HS._chi2_inliers(T, d, False, 0.005) for T, d in zip(Ts[4:12], dets[4:12])] and list(a[4:]) == list(b[:4])


def test_nan_in_the_information_matrix_is_reported():
    rng = np.random.default_rng(2)
    Ts, dets = _cases(rng, True)
    d = dict(dets[0])
    d.pop("_slot", None)
    c = np.array(d["cov_pred"], dtype=np.float32)
    c[0] = np.nan
    d["cov_pred"] = c
    with pytest.raises(AssertionError, match="NaN in information matrix"):
        SC.chi2_counts(Owner(), np.stack(Ts[:1]), [d], False, 0.005, CHI2_2DOF_95)


def test_bad_slots_are_refused():
    from suo_slam_amd import _lib
    st = SC.DetectionStore(capacity=4)
    with pytest.raises(_lib.SuoError, match="slot"):
        st.counts(np.eye(4)[None], np.array([1000]), np.array([1], np.uint64), CHI2_2DOF_95, 1.0)


# ---- the CPU suite's rule tests, on the kernel ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("use_cov", [True, False])
def test_estimate_camera_pose_random_states_on_the_device(use_cov):
    TR.test_estimate_camera_pose_matches_the_restatement_on_random_states(use_cov)


@pytest.mark.parametrize("use_cov", [True, False])
def test_maybe_reinit_random_states_on_the_device(use_cov):
    TR.test_maybe_reinit_matches_the_restatement_on_random_states(use_cov)


@pytest.mark.parametrize("n_a,n_b,fires", [(9, 3, False), (10, 3, True), (2, 0, False), (3, 0, True), (3, 1, False), (4, 1, True), (0, 5, False)])
def test_rule_boundaries_on_the_device(n_a, n_b, fires):
    TR.test_reinit_rule_at_its_boundaries(n_a, n_b, fires)


def test_window_clamp_and_hypothesis_floor_on_the_device():
    TR.test_reinit_counts_only_the_last_fifteen_views_and_needs_two()
    TR.test_covariance_clamp_decides_the_inlier()
    TR.test_camera_pose_needs_four_hypothesis_inliers_and_first_best_wins()


def test_reference_golden_states_on_the_device():
    """tests/golden/slam_golden.npz: 258 states recorded from the reference's own lib/object_slam.py (counts, chosen hypotheses, re-initialised
    objects), incl. the states searched to sit on every rule boundary."""
    TG.test_estimate_camera_pose_equals_the_reference()
    TG.test_maybe_reinit_objects_equals_the_reference()

"""suo_pnp_replay (csrc/pnp.hip, REPLAY): the HIP RANSAC consuming the REFERENCE's own draw sequence (std::default_random_engine through
get4RandomInRange0, /root/reference/thirdparty/lambdatwist/pnp_ransac.cpp:161-232; the sequence itself is pinned bit for bit in tests/test_ref_sampler.py).
Under the same table the HIP kernel -- which evaluates 256 hypotheses at a time and replays the sequential accept rule as a scan -- must choose the SAME
hypothesis as the sequential loop of the oracle, stop after the same number of iterations, and hold the same consensus set; and the legacy
lambdatwist.pnp() in reference-sampler mode must walk the process-global stream exactly as consecutive reference calls would."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _objects(seed, n_obj, noise, outliers):
    from suo_slam_amd import geometry as geo
    from suo_slam_amd import synthetic as S
    fr = S.make_frame(np.random.default_rng(seed), n_obj, noise=noise, outlier_frac=outliers, with_image=False)
    out = []
    for o in range(n_obj):
        m = fr["model_kps_masks"][o]
        out.append((fr["model_kps"][o][m].astype(np.float64), geo.normalize_uv(fr["uv"][o][m].astype(np.float64), fr["K_bbox"][o])))
    return out


def _inliers(T, xs, ys, thr=1e-3):
    X = xs @ T[:3, :3].T + T[:3, 3]
    e = X[:, :2] / X[:, 2:3] - ys
    return (X[:, 2] > 0) & ((e ** 2).sum(1) < thr * thr)


@pytest.mark.parametrize("noise,outliers", [(0.002, 0.1), (0.01, 0.3), (0.03, 0.5)])
def test_replay_chooses_the_hypothesis_the_sequential_loop_chooses(noise, outliers):
    from oracle import geometry as G
    from suo_slam_amd import _lib, lambdatwist as lt
    _lib.require_gpu()
    stream = G.RefSampler()
    n_checked = 0
    for xs, ys in _objects(11, 8, noise, outliers):
        if len(xs) < 4:
            continue
        tab = stream.fork().get4(len(xs), lt.MAX_ITERATIONS)
        for refine in (False, True):
            To, best_o, its_o, win_o = G.pnp_with_draws(xs, ys, tab, 1e-3, refine=refine)
            Th, info = lt.pnp_replay(xs, ys, tab, 1e-3, refine=refine)
            assert (info["winner"], info["iterations"], info["best_inliers"]) == (win_o, its_o, best_o)
            assert np.abs(Th - To).max() < 1e-8
            if not refine and win_o >= 0:
                assert np.array_equal(_inliers(Th, xs, ys), _inliers(To, xs, ys)) and _inliers(Th, xs, ys).sum() == best_o
                assert np.array_equal(tab[info["winner"]], tab[win_o])     # the chosen 4-point sample
        stream.get4(len(xs), its_o)                                         # the next object continues the stream, as the next pnp call does
        n_checked += 1
    assert n_checked >= 6


def test_legacy_pnp_walks_the_reference_stream_call_after_call():
    from oracle import geometry as G
    from suo_slam_amd import _lib, lambdatwist as lt
    _lib.require_gpu()
    objs = [o for o in _objects(5, 10, 0.004, 0.2) if len(o[0]) >= 4]
    ref_stream = G.RefSampler()
    s = lt.set_reference_sampler(True)
    try:
        for xs, ys in objs:
            T = lt.pnp(xs, ys, 0.001)
            To, _, its, _ = G.pnp_with_draws(xs, ys, ref_stream.fork().get4(len(xs), 1000), 1e-3, refine=True)
            ref_stream.get4(len(xs), its)
            assert np.abs(T - To).max() < 1e-8 and s.state == ref_stream.state.value
    finally:
        lt.set_reference_sampler(False)
    # back on the counter-based sampler
    T = lt.pnp(*objs[0], 0.001)
    assert np.isfinite(T).all()


def test_replay_refuses_a_table_shorter_than_the_iteration_cap():
    from suo_slam_amd import _lib, lambdatwist as lt
    _lib.require_gpu()
    xs, ys = _objects(2, 1, 0.0, 0.0)[0]
    with pytest.raises(_lib.SuoError):
        lt.pnp_replay(xs, ys, np.zeros((999, 4), np.int32))

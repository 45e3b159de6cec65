"""Index-work parity of the RANSAC sampler.  The reference draws its 4-point samples from the process-global std::default_random_engine through
std::uniform_int_distribution and a std::set (/root/reference/thirdparty/lambdatwist/utils/random.h:65-116, pnp_ransac.cpp:161-183).  Both restatements --
the oracle's C (oracle/pnp_oracle.c: orc_ref_*) and the product's host-side table generator (suo_slam_amd/lambdatwist.py: ReferenceSampler, what feeds
suo_pnp_replay) -- must reproduce the reference's sequence BIT FOR BIT: against tests/golden/sampler_golden.npz (generated from the reference's header compiled
in the build container, tests/golden/make_sampler_golden.py) and, where oracle/_ref/librandom_ref.so is present, against that library directly."""
import os

import numpy as np
import pytest

from oracle import geometry as G
from suo_slam_amd import lambdatwist as lt

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler_golden.npz")


def test_both_restatements_reproduce_the_golden_stream():
    g = np.load(GOLD)
    o, p = G.RefSampler(), lt.ReferenceSampler()
    for m, want in zip(g["randui_ranges"], g["randui_draws"]):
        assert [o.randui(0, int(m) - 1) for _ in range(len(want))] == want.tolist()
        assert [p.randui(0, int(m) - 1) for _ in range(len(want))] == want.tolist()
    for n, want in zip(g["get4_counts"], g["get4_blocks"]):
        assert np.array_equal(p.table(int(n), 1000), want)                  # the table is drawn from a copy ...
        assert np.array_equal(o.get4(int(n), 1000), want)
        p.advance(int(n), 1000)                                             # ... the stream itself moves only here
        assert p.state == o.state.value
        assert np.all(np.diff(want, axis=1) > 0) and want.min() >= 0 and want.max() < n      # 4 distinct, ascending, in range


def test_restatement_equals_the_compiled_reference_header():
    R = G.ref_random()
    if R is None:
        pytest.skip("oracle/_ref/librandom_ref.so not built (`make -C oracle ref` needs /root/reference)")
    R.ref_rng_reset()
    o = G.RefSampler()
    rng = np.random.default_rng(0)
    for _ in range(40):                                                     # interleaved raw draws and samples on one continuing stream
        n = int(rng.integers(4, 64))
        k = int(rng.integers(1, 200))
        if rng.random() < 0.5:
            a = np.zeros(k, np.int32)
            R.ref_randui(n, k, a)
            assert [o.randui(0, n - 1) for _ in range(k)] == a.tolist()
        else:
            a = np.zeros((k, 4), np.int32)
            R.ref_get4(n, k, a.reshape(-1))
            assert np.array_equal(o.get4(n, k), a)


def test_oracle_ransac_under_a_draw_table():
    """orc_pnp_ransac_draws: the winner is a hypothesis of the table, the consensus is what that hypothesis counts, and a table that repeats the winning
    sample first wins at iteration 0 with the same pose."""
    from suo_slam_amd import geometry as geo
    from suo_slam_amd import synthetic as S
    fr = S.make_frame(np.random.default_rng(3), 3, noise=0.004, outlier_frac=0.25, with_image=False)
    s = G.RefSampler()
    for o in range(3):
        m = fr["model_kps_masks"][o]
        xs = fr["model_kps"][o][m].astype(np.float64)
        ys = geo.normalize_uv(fr["uv"][o][m].astype(np.float64), fr["K_bbox"][o])
        tab = s.fork().get4(len(xs), 1000)
        T, best, its, win = G.pnp_with_draws(xs, ys, tab, 1e-3, refine=False)
        assert 0 <= win < its <= 1000 and best >= 4
        s.get4(len(xs), its)                                                # the reference's stream after this call
        T2, best2, its2, win2 = G.pnp_with_draws(xs, ys, np.concatenate([tab[win:win + 1], tab]), 1e-3, refine=False)
        assert win2 == 0 and best2 == best and np.array_equal(T, T2)


def _libstdcxx_draw(state, n):
    """std::uniform_int_distribution<int>(0, n - 1) on a std::minstd_rand0 whose state is `state` -- libstdc++ itself, compiled here (None without g++)."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("g++") is None:
        return None
    src = ("#include <random>\n#include <sstream>\n#include <cstdio>\nint main(int c, char** v) { std::minstd_rand0 e; std::istringstream(v[1]) >> e;\n"
           "std::uniform_int_distribution<int> d(0, atoi(v[2]) - 1); printf(\"%d\\n\", d(e)); return 0; }\n")
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "d.cpp"), "w").write(src)
        exe = os.path.join(td, "d")
        subprocess.check_call(["g++", "-O1", "-o", exe, os.path.join(td, "d.cpp")])
        return int(subprocess.check_output([exe, str(state), str(n)]).split()[0])


@pytest.mark.parametrize("n", [5, 19])
def test_draws_on_the_scaling_boundary(n):
    """n divides the engine's range 2^31 - 3 = 5 * 19 * 22605091 for objects of 5 or 19 keypoints, so `scaling` is exact and a one-off range
    ((M - 3) / n instead of (M - 2) / n) shifts the bucket edges by one draw each.  States searched to land ON the edges: both restatements and libstdc++ agree."""
    M = 2147483647
    scaling = (M - 2) // n
    assert scaling * n == M - 2
    inv = pow(16807, -1, M)
    for edge in (1, 2, n - 1):
        for r in (edge * scaling - 1, edge * scaling):                      # the draw's value x - 1, last of bucket edge - 1 and first of bucket edge
            pre = ((r + 1) * inv) % M                                         # the engine state whose next step yields x = r + 1
            assert (pre * 16807) % M == r + 1
            p, o = lt.ReferenceSampler(), G.RefSampler()
            p.state = pre
            o.state.value = pre
            got_p, got_o = p.randui(0, n - 1), o.randui(0, n - 1)
            assert got_p == got_o == r // scaling
            ref = _libstdcxx_draw(pre, n)
            if ref is not None:
                assert got_p == ref
    # the state the advisor checked against g++: uniform_int_distribution(0, 4) returns 0
    p = lt.ReferenceSampler()
    p.state = 1584412847
    assert p.randui(0, 4) == 0

"""The reduced system's solve as the LM kernels run it (csrc/lm_device.h: wg_cholesky_solve -- block-6 Cholesky of one workgroup, trailing updates on fp64 MFMA tiles, the
right-hand side eliminated inside the factorisation) against numpy's LAPACK solve.  g2o solves this system with a sparse LL^T
(/root/reference/thirdparty/g2opy/g2o/core/block_solver.hpp:464-566); any backward-stable solve agrees with it to a few ulps times the condition number."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(A, b):
    from suo_slam_amd import _lib
    lib = _lib.lib()
    ns = len(b)
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros(ns)
    ok = C.c_int(-1)
    _lib.check(lib.suo_debug_cholesky_solve(A.ctypes.data, b.ctypes.data, ns, x.ctypes.data, C.addressof(ok)), "suo_debug_cholesky_solve")
    return x, ok.value


def _spd(rng, ns, cond):
    """A Schur-complement-like matrix: strong 6 x 6 diagonal blocks, dense coupling, the requested condition number."""
    Q, _ = np.linalg.qr(rng.standard_normal((ns, ns)))
    ev = np.geomspace(1.0, cond, ns)
    A = (Q * ev) @ Q.T
    return 0.5 * (A + A.T)


@pytest.mark.parametrize("ns", [6, 12, 18, 24, 30, 42, 48, 54, 66, 72, 84, 90, 96])
def test_solution_matches_lapack(ns):
    rng = np.random.default_rng(100 + ns)
    for cond in (1e2, 1e6):
        A = _spd(rng, ns, cond)
        b = rng.standard_normal(ns) * 10.0
        x, ok = _solve(A, b)
        assert ok == 1
        ref = np.linalg.solve(A, b)
        # backward stable: relative residual at rounding level, solution within cond * eps of LAPACK's
        assert np.linalg.norm(A @ x - b) <= 1e-13 * (np.linalg.norm(A, 2) * np.linalg.norm(x) + np.linalg.norm(b))
        assert np.linalg.norm(x - ref) <= 50 * cond * np.finfo(float).eps * np.linalg.norm(ref)


def test_only_the_lower_triangle_is_read():
    rng = np.random.default_rng(7)
    A = _spd(rng, 48, 1e3)
    b = rng.standard_normal(48)
    x0, _ = _solve(A, b)
    junk = np.tril(A) + np.triu(np.full((48, 48), np.nan), 1)
    x1, ok = _solve(junk, b)
    assert ok == 1 and np.array_equal(x0, x1)


def test_block_diagonal_system_is_solved_block_by_block():
    """No coupling between the objects (every camera fixed but one object each): the blocks' solutions are independent of their neighbours."""
    rng = np.random.default_rng(8)
    A = np.zeros((96, 96))
    for k in range(16):
        M = rng.standard_normal((6, 6))
        A[6 * k:6 * k + 6, 6 * k:6 * k + 6] = M @ M.T + 6 * np.eye(6)
    b = rng.standard_normal(96)
    x, ok = _solve(A, b)
    assert ok == 1
    np.testing.assert_allclose(x, np.linalg.solve(A, b), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("ns,bad_at", [(48, 0), (48, 29), (96, 95), (96, 50)])
def test_a_non_positive_pivot_is_reported(ns, bad_at):
    rng = np.random.default_rng(9)
    A = _spd(rng, ns, 1e2)
    A[bad_at, bad_at] = -abs(A[bad_at, bad_at])
    x, ok = _solve(A, rng.standard_normal(ns))
    assert ok == 0


def test_sizes_outside_the_lds_form_are_refused():
    from suo_slam_amd import _lib
    for ns in (0, 5, 102):
        x = np.zeros(max(ns, 1))
        ok = C.c_int(0)
        A = np.eye(max(ns, 1))
        assert _lib.lib().suo_debug_cholesky_solve(A.ctypes.data, x.ctypes.data, ns, x.ctypes.data, C.addressof(ok)) != 0

"""Host half of the two-term fp16 form (suo_slam_amd/csrc/f16x2.h): the weight packers' split, scale and layout -- no GPU needed (the packers are host code).
numpy's float16 conversion (round-to-nearest-even, subnormals kept, overflow to inf) is the independent implementation of the rounding."""
import numpy as np

from suo_slam_amd import _lib


def _pack(w):
    N, K = w.shape
    h = np.empty(2 * N * K, np.uint16)
    osc = np.empty(N, np.float32)
    _lib.check(_lib.lib().suo_pack_gemm_weight_f16x2(np.ascontiguousarray(w).ctypes.data, N, K, h.ctypes.data, osc.ctypes.data))
    # out[((ks * NB + nb) * 2 + plane) * 64 + lane][e]: n = nb*32 + (lane & 31), k = ks*16 + 8*(lane >> 5) + e
    p = h.view(np.float16).reshape(K // 16, N // 32, 2, 2, 32, 8)          # [ks][nb][plane][lane >> 5][lane & 31][e]
    planes = p.transpose(2, 1, 4, 0, 3, 5).reshape(2, N, K)                 # [plane][n][k]
    return planes, osc


def test_gemm_packer_splits_every_row_into_two_fp16_terms():
    rng = np.random.default_rng(0)
    N, K = 64, 96 + 32
    w = (rng.standard_normal((N, K)) / 11).astype(np.float32)
    w *= np.exp2(rng.integers(-40, 41, (N, 1))).astype(np.float32)         # rows of very different magnitude
    w[3] = 0
    w[4, 5] = 0
    w[7] = np.float32(2.0 ** -100)
    planes, osc = _pack(w)
    t = -np.log2(osc.astype(np.float64)) - 4                               # oscale = 2^-(t + 4)
    assert np.array_equal(t, np.round(t))
    ws = w.astype(np.float64) * np.exp2(t)[:, None]
    mx = np.abs(ws).max(1)
    live = np.abs(w).max(1) > 2.0 ** -80
    assert np.all((mx[live] >= 2.0 ** 12) & (mx[live] < 2.0 ** 13)) and t[3] == 0 and t[7] == 100      # (the shift is clamped: 2^-(t + 4) stays a normal fp32)
    hi = ws.astype(np.float32).astype(np.float16)                          # numpy: RNE, the same rounding
    lo = (ws.astype(np.float32) - hi.astype(np.float32)).astype(np.float16)
    assert np.array_equal(planes[0].view(np.uint16), hi.view(np.uint16))
    assert np.array_equal(planes[1].view(np.uint16), lo.view(np.uint16))
    # hi + lo reproduces the scaled weight to 2^-22 relative (or 2^-25 absolute where the residual is subnormal)
    err = np.abs(planes[0].astype(np.float64) + planes[1].astype(np.float64) - ws)
    assert np.all(err <= np.maximum(np.abs(ws) * 2.0 ** -22, 2.0 ** -25))
    assert np.all(np.isfinite(planes.astype(np.float32)))


def test_fp16_rounding_edge_cases_of_the_packer():
    """Ties to even, the normal / subnormal boundary, the largest subnormal carrying into the smallest normal: one row whose maximum pins the scale to 2^0."""
    K = 16
    vals = np.array([4096.0, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 2.0 ** -14, 2.0 ** -14 - 2.0 ** -25, 2.0 ** -24 * 1.5, 2.0 ** -24 * 2.5, 2.0 ** -25, 2.0 ** -26,
                     -(1.0 + 2.0 ** -11), 1023.5 * 2.0 ** -24, 0.1, -0.3, 1e-9, 2047.5, 4095.9], np.float32)
    w = np.zeros((32, K), np.float32)
    w[0] = vals
    planes, osc = _pack(w)
    assert osc[0] == np.float32(2.0 ** -4)                                 # max 4096 -> t = 0
    hi = vals.astype(np.float16)
    assert np.array_equal(planes[0][0].view(np.uint16), hi.view(np.uint16))
    lo = (vals - hi.astype(np.float32)).astype(np.float16)
    assert np.array_equal(planes[1][0].view(np.uint16), lo.view(np.uint16))

"""Thin test helpers that call the per-stage C-ABI entry points of libsuo_hip.so on torch CUDA tensors."""
import ctypes as C

import numpy as np
import torch

from suo_slam_amd import _lib


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda().contiguous()


def nhwc(x):  # [L,C,H,W] numpy/torch -> NHWC cuda
    return dev(np.ascontiguousarray(np.asarray(x).transpose(0, 2, 3, 1)))


def nchw(t):  # NHWC cuda -> numpy [L,C,H,W]
    return t.cpu().numpy().transpose(0, 3, 1, 2)


def pack_gemm(w, Np, Kp):
    w = np.ascontiguousarray(w, np.float32)
    out = np.empty(2 * Np * Kp, np.float32)
    _lib.check(_lib.lib().suo_pack_gemm_weight(w.ctypes.data, w.shape[0], w.shape[1], Np, Kp, out.ctypes.data), "pack_gemm")
    return out


def pack_conv(w, Np, Cp, CK):
    w = np.ascontiguousarray(w, np.float32)
    out = np.empty(2 * Np * ((Cp * w.shape[2] * w.shape[3] + 15) // 16 * 16), np.float32)     # K padded to 16 (CK = 4 / 8 stems)
    _lib.check(_lib.lib().suo_pack_conv_weight(w.ctypes.data, w.shape[0], w.shape[1], w.shape[2], Np, Cp, CK, out.ctypes.data), "pack_conv")
    return out


def conv1x1(a1, w1, bias, pro=None, a2=None, w2=None, res=None, relu=False, nchw_hw=0):
    """a1 [M,K1] cuda, w1 [N,K1] numpy, bias [N]; optional pro=(scale,shift), a2 [M,K2]/w2 [N,K2], res [M,N]."""
    M, K1 = a1.shape
    N = w1.shape[0]
    Np = (N + 63) // 64 * 64
    K2 = a2.shape[1] if a2 is not None else 0
    full = np.zeros((Np, K1 + K2), np.float32)
    full[:N, :K1] = w1
    if a2 is not None:
        full[:N, K1:] = w2
    wp = dev(pack_gemm(full, Np, K1 + K2))
    b = np.zeros(Np, np.float32)
    b[:N] = bias
    b = dev(b)
    if nchw_hw:
        out = torch.empty((M // nchw_hw, N, nchw_hw), device="cuda")
        ldo = 0
    else:
        out = torch.empty((M, N), device="cuda")
        ldo = N
    ps = dev(pro[0]) if pro is not None else None
    pt = dev(pro[1]) if pro is not None else None
    _lib.check(_lib.lib().suo_conv1x1(P(a1), a1.stride(0), K1, P(ps), P(pt), P(a2), a2.stride(0) if a2 is not None else 0, K2,
                                      P(wp), P(b), P(res), res.stride(0) if res is not None else 0, P(out), ldo, M, Np, N,
                                      int(relu), nchw_hw, S()), "suo_conv1x1")
    torch.cuda.synchronize()
    return out


def conv1x1_pool(a1, w1, bias, H, W, pro=None, a2=None, w2=None, res=None, relu=False, want_full=True):
    """conv1x1 with the 2x2 max-pool fused into its epilogue: returns (full [M,N] or None, pooled [M/4,N])."""
    M, K1 = a1.shape
    N = w1.shape[0]
    assert N % 128 == 0
    K2 = a2.shape[1] if a2 is not None else 0
    full = np.zeros((N, K1 + K2), np.float32)
    full[:, :K1] = w1
    if a2 is not None:
        full[:, K1:] = w2
    wp = dev(pack_gemm(full, N, K1 + K2))
    b = dev(np.ascontiguousarray(bias, np.float32))
    out = torch.empty((M, N), device="cuda") if want_full else None
    pooled = torch.empty((M // 4, N), device="cuda")
    ps = dev(pro[0]) if pro is not None else None
    pt = dev(pro[1]) if pro is not None else None
    _lib.check(_lib.lib().suo_conv1x1_pool(P(a1), a1.stride(0), K1, P(ps), P(pt), P(a2), a2.stride(0) if a2 is not None else 0, K2,
                                           P(wp), P(b), P(res), res.stride(0) if res is not None else 0, P(out), N, M, N,
                                           int(relu), H, W, P(pooled), S()), "suo_conv1x1_pool")
    torch.cuda.synchronize()
    return out, pooled


def conv_kxk(x_nhwc, w, bias, relu=False):
    L, H, W, C = x_nhwc.shape
    N, Cw, KS, _ = w.shape
    CK = 32 if KS == 3 else (C if C in (4, 8) else 16)           # image-only stems: one chunk of 4 (paired taps) or 8 channels
    Np = (N + 63) // 64 * 64
    assert C % CK == 0 and Cw <= C
    wp = dev(pack_conv(w, Np, C, CK))
    b = np.zeros(Np, np.float32)
    b[:N] = bias
    b = dev(b)
    OH, OW = (H, W) if KS == 3 else (H // 2, W // 2)
    out = torch.empty((L, OH, OW, Np), device="cuda")
    _lib.check(_lib.lib().suo_conv_kxk(KS, P(x_nhwc), L, H, W, C, P(wp), P(b), P(out), Np, int(relu), S()), "suo_conv_kxk")
    torch.cuda.synchronize()
    return out[..., :N]


def conv3x3_conv1x1_skip(x_nhwc, w2, b2, w3, b3, skip_nhwc):
    """Fused tail of a Residual block: x [L,H,W,128], w2 [128,128,3,3], w3 [256,128], skip [L,H,W,256] -> [L,H,W,256]."""
    L, H, W, C = x_nhwc.shape
    assert C == 128 and w2.shape == (128, 128, 3, 3) and w3.shape == (256, 128) and tuple(skip_nhwc.shape) == (L, H, W, 256)
    wp2 = dev(pack_conv(w2, 128, 128, 32))
    wp3 = dev(pack_gemm(np.ascontiguousarray(w3, np.float32), 256, 128))
    out = torch.empty((L, H, W, 256), device="cuda")
    b2d, b3d = dev(b2), dev(b3)                       # (named: a temporary would be freed -- and its block reused -- before the launch)
    _lib.check(_lib.lib().suo_conv3x3_conv1x1_skip(P(x_nhwc), L, H, W, P(wp2), P(b2d), P(wp3), P(b3d), P(skip_nhwc), P(out), S()),
               "suo_conv3x3_conv1x1_skip")
    torch.cuda.synchronize()
    return out


def conv3x3_wino(x_nhwc, w, bias, relu=False):
    """3x3 convolution in Winograd F(2x2,3x3) form: x [L,H,W,C], w [128,C,3,3] -> [L,H,W,128]."""
    L, H, W, C = x_nhwc.shape
    N = w.shape[0]
    assert N in (64, 128) and C % 16 == 0
    w = np.ascontiguousarray(w, np.float32)
    packed = np.empty(16 * N * C, np.float32)
    _lib.check(_lib.lib().suo_pack_wino_weight(w.ctypes.data, N, C, N, C, packed.ctypes.data), "pack_wino")
    wp, b = dev(packed), dev(bias)
    out = torch.empty((L, H, W, N), device="cuda")
    _lib.check(_lib.lib().suo_conv3x3_wino(P(x_nhwc), L, H, W, C, P(wp), P(b), P(out), N, int(relu), S()), "suo_conv3x3_wino")
    torch.cuda.synchronize()
    return out


def conv3x3_wino_conv1x1_skip(x_nhwc, w2, b2, w3, b3, skip_nhwc):
    """Fused tail of a Residual block with the 3x3 in Winograd form (what the network launches)."""
    L, H, W, C = x_nhwc.shape
    assert C == 128 and w2.shape == (128, 128, 3, 3) and w3.shape == (256, 128) and tuple(skip_nhwc.shape) == (L, H, W, 256)
    w2 = np.ascontiguousarray(w2, np.float32)
    packed = np.empty(16 * 128 * 128, np.float32)
    _lib.check(_lib.lib().suo_pack_wino_weight(w2.ctypes.data, 128, 128, 128, 128, packed.ctypes.data), "pack_wino")
    wq2 = dev(packed)
    wp3 = dev(pack_gemm(np.ascontiguousarray(w3, np.float32), 256, 128))
    out = torch.empty((L, H, W, 256), device="cuda")
    b2d, b3d = dev(b2), dev(b3)
    _lib.check(_lib.lib().suo_conv3x3_wino_conv1x1_skip(P(x_nhwc), L, H, W, P(wq2), P(b2d), P(wp3), P(b3d), P(skip_nhwc), P(out), S()),
               "suo_conv3x3_wino_conv1x1_skip")
    torch.cuda.synchronize()
    return out


def conv3x3_wino_conv1x1_skip_up(x_nhwc, w2, b2, w3, b3, skip_nhwc, up_nhwc):
    """The fused Winograd Residual tail with the Hourglass's up-sampled low branch added in its epilogue."""
    L, H, W, C = x_nhwc.shape
    w2 = np.ascontiguousarray(w2, np.float32)
    packed = np.empty(16 * 128 * 128, np.float32)
    _lib.check(_lib.lib().suo_pack_wino_weight(w2.ctypes.data, 128, 128, 128, 128, packed.ctypes.data), "pack_wino")
    wq2 = dev(packed)
    wp3 = dev(pack_gemm(np.ascontiguousarray(w3, np.float32), 256, 128))
    out = torch.empty((L, H, W, 256), device="cuda")
    b2d, b3d = dev(b2), dev(b3)
    _lib.check(_lib.lib().suo_conv3x3_wino_conv1x1_skip_up(P(x_nhwc), L, H, W, P(wq2), P(b2d), P(wp3), P(b3d), P(skip_nhwc), P(up_nhwc), P(out), S()),
               "suo_conv3x3_wino_conv1x1_skip_up")
    torch.cuda.synchronize()
    return out


def _pack_x3(w2, w3=None):
    lib = _lib.lib()
    w2 = np.ascontiguousarray(w2, np.float32)
    n = w2.shape[0]
    q = np.empty(3 * 16 * n * n, np.uint16)
    _lib.check(lib.suo_pack_wino_weight_bf16x3(w2.ctypes.data, n, n, q.ctypes.data), "pack_wino_x3")
    wq3 = torch.from_numpy(q.view(np.int16)).cuda()
    if w3 is None:
        return wq3
    w3 = np.ascontiguousarray(w3, np.float32)
    t = np.empty(3 * 256 * 128, np.uint16)
    _lib.check(lib.suo_pack_tail_weight_bf16x3(w3.ctypes.data, 256, 128, t.ctypes.data), "pack_tail_x3")
    return wq3, torch.from_numpy(t.view(np.int16)).cuda()


def conv3x3_wino_x3(x_nhwc, w, bias, relu=False):
    """csrc/conv_wino_x3.hip: the Winograd 3x3 convolution (128 -> 128) on the bf16 matrix pipe with 3-way split operands."""
    L, H, W, C = x_nhwc.shape
    assert C in (128, 64) and w.shape == (C, C, 3, 3)
    wq3, b = _pack_x3(w), dev(bias)
    out = torch.empty((L, H, W, C), device="cuda")
    _lib.check(_lib.lib().suo_conv3x3_wino_x3_n(P(x_nhwc), L, H, W, C, P(wq3), P(b), P(out), int(relu), S()), "suo_conv3x3_wino_x3_n")
    torch.cuda.synchronize()
    return out


def conv3x3_wino_x3_conv1x1_skip_up(x_nhwc, w2, b2, w3, b3, skip_nhwc, up_nhwc=None, tail_x3=True):
    """The fused Residual tail of csrc/conv_wino_x3.hip: conv3 on the bf16 pipe as well (tail_x3, what the network launches) or on the fp32 pipe."""
    L, H, W, C = x_nhwc.shape
    wq3, w3x = _pack_x3(w2, w3)
    wp3 = w3x if tail_x3 else dev(pack_gemm(np.ascontiguousarray(w3, np.float32), 256, 128))
    out = torch.empty((L, H, W, 256), device="cuda")
    b2d, b3d = dev(b2), dev(b3)
    _lib.check(_lib.lib().suo_conv3x3_wino_x3_conv1x1_skip_up(P(x_nhwc), L, H, W, P(wq3), P(b2d), P(wp3), int(tail_x3), P(b3d), P(skip_nhwc), P(up_nhwc), P(out), S()),
               "suo_conv3x3_wino_x3_conv1x1_skip_up")
    torch.cuda.synchronize()
    return out


# ---- the two-term fp16 form (csrc/f16x2.h): every helper returns (out, range_flag) -- flag 1 = an activation left fp16's range, `out` is invalid -----------
def _flag():
    return torch.zeros(1, dtype=torch.int32, device="cuda")


def pack_gemm_f16x2(w):
    """-> (planes int16 cuda [2*N*K], oscale cuda [N], oscale numpy)"""
    w = np.ascontiguousarray(w, np.float32)
    N, K = w.shape
    h = np.empty(2 * N * K, np.uint16)
    osc = np.empty(N, np.float32)
    _lib.check(_lib.lib().suo_pack_gemm_weight_f16x2(w.ctypes.data, N, K, h.ctypes.data, osc.ctypes.data), "pack_gemm_f16x2")
    return torch.from_numpy(h.view(np.int16)).cuda(), dev(osc), osc


def conv1x1_f16x2(a1, w1, bias, pro=None, a2=None, w2=None, res=None, relu=False, pool_hw=None):
    """csrc/gemm_bf16x3.hip with NP = 2.  a1 [M,K1] cuda, w1 [N,K1]; optional pro=(scale,shift), a2 [M,K2] / w2 [N,K2], res [M,N], pool_hw=(H,W) -> also the pooled result."""
    M, K1 = a1.shape
    N = w1.shape[0]
    K2 = a2.shape[1] if a2 is not None else 0
    full = np.concatenate([w1, w2], 1) if a2 is not None else w1
    w16, osc, _ = pack_gemm_f16x2(full)
    b = dev(bias)
    out = torch.empty((M, N), device="cuda")
    flag = _flag()
    ps, pt = (dev(pro[0]), dev(pro[1])) if pro is not None else (None, None)
    lib = _lib.lib()
    if pool_hw is None:
        _lib.check(lib.suo_conv1x1_f16x2_ex(P(a1), a1.stride(0), K1, P(ps), P(pt), P(a2), a2.stride(0) if a2 is not None else 0, K2, P(w16), P(osc), P(b), P(res),
                                            res.stride(0) if res is not None else 0, P(out), N, M, N, int(relu), P(flag), S()), "suo_conv1x1_f16x2_ex")
        torch.cuda.synchronize()
        return out, int(flag.item())
    pooled = torch.empty((M // 4, N), device="cuda")
    _lib.check(lib.suo_conv1x1_f16x2_pool(P(a1), a1.stride(0), K1, P(ps), P(pt), P(a2), a2.stride(0) if a2 is not None else 0, K2, P(w16), P(osc), P(b), P(res),
                                          res.stride(0) if res is not None else 0, P(out), N, M, N, int(relu), pool_hw[0], pool_hw[1], P(pooled), P(flag), S()),
               "suo_conv1x1_f16x2_pool")
    torch.cuda.synchronize()
    return (out, pooled), int(flag.item())


def _pack_f16x2(w2, w3=None):
    lib = _lib.lib()
    w2 = np.ascontiguousarray(w2, np.float32)
    n = w2.shape[0]
    q = np.empty(2 * 16 * n * n, np.uint16)
    o2 = np.empty(n, np.float32)
    _lib.check(lib.suo_pack_wino_weight_f16x2(w2.ctypes.data, n, n, q.ctypes.data, o2.ctypes.data), "pack_wino_f16x2")
    wq = torch.from_numpy(q.view(np.int16)).cuda()
    if w3 is None:
        return wq, dev(o2)
    w3 = np.ascontiguousarray(w3, np.float32)
    t = np.empty(2 * 256 * 128, np.uint16)
    o3 = np.empty(256, np.float32)
    _lib.check(lib.suo_pack_tail_weight_f16x2(w3.ctypes.data, 256, 128, t.ctypes.data, o3.ctypes.data), "pack_tail_f16x2")
    return wq, dev(o2), torch.from_numpy(t.view(np.int16)).cuda(), dev(o3)


def conv3x3_wino_f16x2(x_nhwc, w, bias, relu=False):
    """csrc/conv_wino_x3.hip with NP = 2: the Winograd 3x3 convolution (128 -> 128 / 64 -> 64) on two fp16 terms per operand."""
    L, H, W, C = x_nhwc.shape
    assert C in (128, 64) and w.shape == (C, C, 3, 3)
    wq, osc = _pack_f16x2(w)
    b = dev(bias)
    out = torch.empty((L, H, W, C), device="cuda")
    flag = _flag()
    _lib.check(_lib.lib().suo_conv3x3_wino_f16x2_n(P(x_nhwc), L, H, W, C, P(wq), P(osc), P(b), P(out), int(relu), P(flag), S()), "suo_conv3x3_wino_f16x2_n")
    torch.cuda.synchronize()
    return out, int(flag.item())


def conv3x3_wino_f16x2_conv1x1_skip_up(x_nhwc, w2, b2, w3, b3, skip_nhwc, up_nhwc=None):
    """The fused Residual tail on two fp16 terms per operand (what the network launches by default)."""
    L, H, W, C = x_nhwc.shape
    wq, o2, w3p, o3 = _pack_f16x2(w2, w3)
    out = torch.empty((L, H, W, 256), device="cuda")
    b2d, b3d = dev(b2), dev(b3)
    flag = _flag()
    _lib.check(_lib.lib().suo_conv3x3_wino_f16x2_conv1x1_skip_up(P(x_nhwc), L, H, W, P(wq), P(o2), P(b2d), P(w3p), P(o3), P(b3d), P(skip_nhwc), P(up_nhwc), P(out),
                                                                 P(flag), S()), "suo_conv3x3_wino_f16x2_conv1x1_skip_up")
    torch.cuda.synchronize()
    return out, int(flag.item())


def res_block(x_nhwc, pro, w1, b1, w2, b2, w3, b3, up_nhwc=None, pool_in=False):
    """csrc/res_small.hip: a whole 256 -> 256 Residual block in one launch.  x [L,H,W,256] (pool_in: [L,2H,2W,256]); pro = (scale, shift) [256];
    w1 [128,256], w2 [128,128,3,3], w3 [256,128] with their BatchNorms already folded; up [L,H/2,W/2,256] or None."""
    lib = _lib.lib()
    L, H, W, Cc = x_nhwc.shape
    if pool_in:
        H, W = H // 2, W // 2
    assert Cc == 256 and w1.shape == (128, 256) and w2.shape == (128, 128, 3, 3) and w3.shape == (256, 128)
    w1, w2, w3 = (np.ascontiguousarray(t, np.float32) for t in (w1, w2, w3))
    p1, p2, p3 = np.empty(128 * 256, np.float32), np.empty(128 * 128 * 9, np.float32), np.empty(256 * 128, np.float32)
    _lib.check(lib.suo_pack_res_block(w1.ctypes.data, w2.ctypes.data, None, w3.ctypes.data, p1.ctypes.data, p2.ctypes.data, p3.ctypes.data), "suo_pack_res_block")
    d = [dev(t) for t in (pro[0], pro[1], p1, b1, p2, b2, p3, b3)]
    out = torch.empty((L, H, W, 256), device="cuda")
    _lib.check(lib.suo_res_block(P(x_nhwc), L, H, W, int(pool_in), P(d[0]), P(d[1]), P(d[2]), P(d[3]), P(d[4]), P(d[5]), P(d[6]), P(d[7]), P(up_nhwc), P(out), S()),
               "suo_res_block")
    torch.cuda.synchronize()
    return out


def res_block_x3(x_nhwc, pro, w1, b1, w2, b2, w3, b3, up_nhwc=None, pool_in=False):
    """csrc/res_small_x3.hip: the one-launch Residual block on the bf16 matrix pipe (3-way split operands); arguments as res_block."""
    lib = _lib.lib()
    L, H, W, Cc = x_nhwc.shape
    if pool_in:
        H, W = H // 2, W // 2
    w1, w2, w3 = (np.ascontiguousarray(t, np.float32) for t in (w1, w2, w3))
    p1, p2, p3 = np.empty(3 * 128 * 256, np.uint16), np.empty(3 * 128 * 128 * 9, np.uint16), np.empty(3 * 256 * 128, np.uint16)
    _lib.check(lib.suo_pack_res_block_bf16x3(w1.ctypes.data, w2.ctypes.data, None, w3.ctypes.data, p1.ctypes.data, p2.ctypes.data, p3.ctypes.data), "suo_pack_res_block_bf16x3")
    dw = [torch.from_numpy(t.view(np.int16)).cuda() for t in (p1, p2, p3)]
    d = [dev(t) for t in (pro[0], pro[1], b1, b2, b3)]
    out = torch.empty((L, H, W, 256), device="cuda")
    _lib.check(lib.suo_res_block_bf16x3(P(x_nhwc), L, H, W, int(pool_in), P(d[0]), P(d[1]), P(dw[0]), P(d[2]), P(dw[1]), P(d[3]), P(dw[2]), P(d[4]), P(up_nhwc), P(out), S()),
               "suo_res_block_bf16x3")
    torch.cuda.synchronize()
    return out


def conv3x3_wino_f16x2_tail_next(x_nhwc, w2, b2, w3, b3, skip_nhwc, up_nhwc, n_pro, n_w1, n_b1):
    """The fused Residual tail on two fp16 terms WITH the next block's conv1 (csrc/conv_wino_x3.hip, NEXT): returns (out [L,H,W,256], next_mid1 [L,H,W,128], flag)."""
    L, H, W, C = x_nhwc.shape
    wq, o2, w3p, o3 = _pack_f16x2(w2, w3)
    w1h, osc1, _ = pack_gemm_f16x2(n_w1)
    out = torch.empty((L, H, W, 256), device="cuda")
    nxt = torch.full((L, H, W, 128), -7.0, device="cuda")
    b2d, b3d, ns, nt, nb = dev(b2), dev(b3), dev(n_pro[0]), dev(n_pro[1]), dev(n_b1)
    flag = _flag()
    _lib.check(_lib.lib().suo_conv3x3_wino_f16x2_conv1x1_skip_up_next(P(x_nhwc), L, H, W, P(wq), P(o2), P(b2d), P(w3p), P(o3), P(b3d), P(skip_nhwc), P(up_nhwc), P(out), P(ns), P(nt),
                                                                      P(w1h), P(osc1), P(nb), P(nxt), P(flag), S()), "suo_conv3x3_wino_f16x2_conv1x1_skip_up_next")
    torch.cuda.synchronize()
    return out, nxt, int(flag.item())


def res_block_f16x2(x_nhwc, pro, w1, b1, w2, b2, w3, b3, up_nhwc=None, pool_in=False):
    """csrc/res_small_x3.hip with NP = 2: the one-launch Residual block on two fp16 terms per operand; arguments as res_block.  Returns (out, range_flag)."""
    lib = _lib.lib()
    L, H, W, Cc = x_nhwc.shape
    if pool_in:
        H, W = H // 2, W // 2
    w1, w2, w3 = (np.ascontiguousarray(t, np.float32) for t in (w1, w2, w3))
    p1, p2, p3 = np.empty(2 * 128 * 256, np.uint16), np.empty(2 * 128 * 128 * 9, np.uint16), np.empty(2 * 256 * 128, np.uint16)
    o1, o2, o3 = np.empty(128, np.float32), np.empty(128, np.float32), np.empty(256, np.float32)
    _lib.check(lib.suo_pack_res_block_f16x2(w1.ctypes.data, w2.ctypes.data, None, w3.ctypes.data, p1.ctypes.data, p2.ctypes.data, p3.ctypes.data, o1.ctypes.data,
                                            o2.ctypes.data, o3.ctypes.data), "suo_pack_res_block_f16x2")
    dw = [torch.from_numpy(t.view(np.int16)).cuda() for t in (p1, p2, p3)]
    do = [dev(t) for t in (o1, o2, o3)]
    d = [dev(t) for t in (pro[0], pro[1], b1, b2, b3)]
    out = torch.empty((L, H, W, 256), device="cuda")
    flag = _flag()
    _lib.check(lib.suo_res_block_f16x2(P(x_nhwc), L, H, W, int(pool_in), P(d[0]), P(d[1]), P(dw[0]), P(do[0]), P(d[2]), P(dw[1]), P(do[1]), P(d[3]), P(dw[2]), P(do[2]), P(d[4]),
                                       P(up_nhwc), P(out), P(flag), S()), "suo_res_block_f16x2")
    torch.cuda.synchronize()
    return out, int(flag.item())


def conv1x1_chain_head_f16x2(a, w1, b1, w2, b2, n_valid, hw):
    """csrc/gemm_bf16x3.hip: gemm_chain_head_kernel.  a [M,256] cuda; w1 [256,256], b1 [256]; w2 [64,256] (rows >= n_valid zero), b2 [64] -> (NCHW [M/hw, n_valid, hw], flag)."""
    M = a.shape[0]
    w1h, o1, _ = pack_gemm_f16x2(w1)
    w2h, o2, _ = pack_gemm_f16x2(w2)
    b1d, b2d = dev(b1), dev(b2)
    out = torch.full((M // hw, n_valid, hw), float("nan"), device="cuda")
    flag = _flag()
    _lib.check(_lib.lib().suo_conv1x1_chain_head_f16x2(P(a), a.stride(0), M, P(w1h), P(o1), P(b1d), P(w2h), P(o2), P(b2d), P(out), n_valid, hw, P(flag), S()),
               "suo_conv1x1_chain_head_f16x2")
    torch.cuda.synchronize()
    return out, int(flag.item())

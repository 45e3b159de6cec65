"""Legs of bench.py after the timed region: per-kernel rooflines under HIP events, the whole call's HBM accounting, the other matrix pipes."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

from .common import (BF16_MFMA_PEAK_TF, FP32_MFMA_PEAK_TF, HBM_PEAK_GBPS, ROOT, _timed, committed_pmc, committed_traffic, dominant_kernel_name, matrix_pipe, pack_conv,
                     pack_gemm, wino_bf16x3_enabled)


def conv_roofline(L, iters=30):
    """Live HIP-event timing of the dominant kernel at the launch shape of the timed region: the tail of a 256 -> 256 Residual block
    at 64x64 in ONE launch -- conv2 (3x3, 128 -> 128, Winograd F(2x2,3x3)) + ReLU, conv3 (1x1, 128 -> 256) + skip (8 launches per network
    call, about a third of its kernel time).  What the network launches (csrc/net.hip):
      * default: wino3x3_x3_kernel<true,*,true> (csrc/conv_wino_x3.hip) -- every product on the BF16 matrix pipe, both operands split into
        three bf16 terms, 6 of the 9 cross terms accumulated in fp32 (fp32 accuracy: tests/test_gpu_cnn.py).  `achieved` / `frac` count
        the bf16 FLOPs the kernel EXECUTES (6 MFMAs of 32x32x16 per component / k-step) against the dense bf16 MFMA peak; the fp32-equivalent
        rates (what an fp32 kernel would have to sustain for the same launch time) are beside it: `f32_equivalent_executed_tflops` (Winograd-
        counted, / 157.3 = `f32_equivalent_over_f32_peak`) and the reference-counted `algorithmic_tflops`;
      * SUO_WINO_BF16X3=0: wino3x3_kernel<true> (csrc/conv_wino.hip) on the fp32 pipe: `achieved` / `frac` = executed fp32 FLOPs (16 products
        per 2x2 tile and channel pair where the direct form issues 36) against the fp32 MFMA peak.
    The other kernel of the pair, the 3x3 alone and the direct forms are timed in the same process under `same_process`."""
    import torch
    from suo_slam_amd import _lib
    rng = np.random.default_rng(0)
    x = torch.rand((L, 64, 64, 128), device="cuda") - 0.5
    skip = torch.rand((L, 64, 64, 256), device="cuda") - 0.5
    w2 = (rng.standard_normal((128, 128, 3, 3)) / 34.0).astype(np.float32)
    w3 = (rng.standard_normal((256, 128)) / 11.0).astype(np.float32)
    lib = _lib.lib()
    wq = np.empty(16 * 128 * 128, np.float32)
    _lib.check(lib.suo_pack_wino_weight(np.ascontiguousarray(w2).ctypes.data, 128, 128, 128, 128, wq.ctypes.data), "pack_wino")
    wq2 = torch.from_numpy(wq).cuda()
    wq3h = np.empty(3 * 16 * 128 * 128, np.uint16)
    _lib.check(lib.suo_pack_wino_weight_bf16x3(np.ascontiguousarray(w2).ctypes.data, 128, 128, wq3h.ctypes.data), "pack_wino_x3")
    wq3 = torch.from_numpy(wq3h.view(np.int16)).cuda()
    w3xh = np.empty(3 * 256 * 128, np.uint16)
    _lib.check(lib.suo_pack_tail_weight_bf16x3(np.ascontiguousarray(w3).ctypes.data, 256, 128, w3xh.ctypes.data), "pack_tail_x3")
    w3x = torch.from_numpy(w3xh.view(np.int16)).cuda()
    wq16h, o2h, w3p16h, o3h = np.empty(2 * 16 * 128 * 128, np.uint16), np.empty(128, np.float32), np.empty(2 * 256 * 128, np.uint16), np.empty(256, np.float32)
    _lib.check(lib.suo_pack_wino_weight_f16x2(np.ascontiguousarray(w2).ctypes.data, 128, 128, wq16h.ctypes.data, o2h.ctypes.data), "pack_wino_f16x2")
    _lib.check(lib.suo_pack_tail_weight_f16x2(np.ascontiguousarray(w3).ctypes.data, 256, 128, w3p16h.ctypes.data, o3h.ctypes.data), "pack_tail_f16x2")
    wq16, o2, w3p16, o3 = torch.from_numpy(wq16h.view(np.int16)).cuda(), torch.from_numpy(o2h).cuda(), torch.from_numpy(w3p16h.view(np.int16)).cuda(), torch.from_numpy(o3h).cuda()
    rflag = torch.zeros(1, dtype=torch.int32, device="cuda")
    wp2 = torch.from_numpy(pack_conv(w2, 128, 128, 32)).cuda()
    wp3 = torch.from_numpy(pack_gemm(w3, 256, 128)).cuda()
    b2 = torch.zeros(128, device="cuda")
    b3 = torch.zeros(256, device="cuda")
    mid = torch.empty((L, 64, 64, 128), device="cuda")
    out = torch.empty((L, 64, 64, 256), device="cuda")
    st = torch.cuda.current_stream()
    s = C.c_void_p(st.cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def f16_fused():
        _lib.check(lib.suo_conv3x3_wino_f16x2_conv1x1_skip_up(P(x), L, 64, 64, P(wq16), P(o2), P(b2), P(w3p16), P(o3), P(b3), P(skip), None, P(out), P(rflag), s),
                   "suo_conv3x3_wino_f16x2_conv1x1_skip_up")

    def f16_plain():
        _lib.check(lib.suo_conv3x3_wino_f16x2_n(P(x), L, 64, 64, 128, P(wq16), P(o2), P(b2), P(mid), 1, P(rflag), s), "suo_conv3x3_wino_f16x2_n")

    def x3_fused():
        _lib.check(lib.suo_conv3x3_wino_x3_conv1x1_skip_up(P(x), L, 64, 64, P(wq3), P(b2), P(w3x), 1, P(b3), P(skip), None, P(out), s), "suo_conv3x3_wino_x3_conv1x1_skip_up")

    def x3_plain():
        _lib.check(lib.suo_conv3x3_wino_x3(P(x), L, 64, 64, P(wq3), P(b2), P(mid), 1, s), "suo_conv3x3_wino_x3")

    def wino_fused():
        _lib.check(lib.suo_conv3x3_wino_conv1x1_skip(P(x), L, 64, 64, P(wq2), P(b2), P(wp3), P(b3), P(skip), P(out), s), "suo_conv3x3_wino_conv1x1_skip")

    def wino_plain():
        _lib.check(lib.suo_conv3x3_wino(P(x), L, 64, 64, 128, P(wq2), P(b2), P(mid), 128, 1, s), "suo_conv3x3_wino")

    def direct_fused():
        _lib.check(lib.suo_conv3x3_conv1x1_skip(P(x), L, 64, 64, P(wp2), P(b2), P(wp3), P(b3), P(skip), P(out), s), "suo_conv3x3_conv1x1_skip")

    def direct_plain():
        _lib.check(lib.suo_conv_kxk(3, P(x), L, 64, 64, 128, P(wp2), P(b2), P(mid), 128, 1, s), "suo_conv_kxk")
    us_h, us_hp = (_timed(f, st, iters) for f in (f16_fused, f16_plain))
    us_x, us_xp, us_w, us_wp, us_df, us_dp = (_timed(f, st, iters) for f in (x3_fused, x3_plain, wino_fused, wino_plain, direct_fused, direct_plain))
    px = float(L) * 64 * 64
    flop3, flop1 = 2.0 * px * 128 * 128 * 9, 2.0 * px * 128 * 256
    flop = flop3 + flop1
    flop_exec = flop3 / 2.25 + flop1                               # 16 products per 2x2 tile and channel pair instead of 36
    flop_exec_bf16 = 6.0 * flop_exec                               # every product as 6 bf16 cross terms
    flop_exec_f16 = 3.0 * flop_exec                                # ... as 3 fp16 cross terms
    tf = lambda f, t: round(f / (t * 1e-6) / 1e12, 2) if t == t else None  # noqa: E731
    fr = lambda f, t, pk=FP32_MFMA_PEAK_TF: round(f / (t * 1e-6) / 1e12 / pk, 4) if t == t else None  # noqa: E731
    f32_entry = {"avg_launch_us": round(us_w, 2), "frac": fr(flop_exec, us_w), "achieved_tflops": tf(flop_exec, us_w), "peak": FP32_MFMA_PEAK_TF,
                 "algorithmic_tflops": tf(flop, us_w), "algorithmic_over_peak": fr(flop, us_w)}
    x3_entry = {"avg_launch_us": round(us_x, 2), "frac": fr(flop_exec_bf16, us_x, BF16_MFMA_PEAK_TF), "achieved_tflops": tf(flop_exec_bf16, us_x),
                "peak": BF16_MFMA_PEAK_TF, "f32_equivalent_executed_tflops": tf(flop_exec, us_x), "f32_equivalent_over_f32_peak": fr(flop_exec, us_x),
                "algorithmic_tflops": tf(flop, us_x)}
    f16_entry = {"avg_launch_us": round(us_h, 2), "frac": fr(flop_exec_f16, us_h, BF16_MFMA_PEAK_TF), "achieved_tflops": tf(flop_exec_f16, us_h),
                 "peak": BF16_MFMA_PEAK_TF, "f32_equivalent_executed_tflops": tf(flop_exec, us_h), "f32_equivalent_over_f32_peak": fr(flop_exec, us_h),
                 "algorithmic_tflops": tf(flop, us_h)}
    same = {"wino3x3_x3_kernel<false,false,false,4,2> (f16x2, 3x3 alone)": {"avg_launch_us": round(us_hp, 2), "frac": fr(3.0 * flop3 / 2.25, us_hp, BF16_MFMA_PEAK_TF),
                                                                            "f32_equivalent_over_f32_peak": fr(flop3 / 2.25, us_hp), "algorithmic_tflops": tf(flop3, us_hp)},
            "wino3x3_kernel<false> (fp32 pipe, 3x3 alone)": {"avg_launch_us": round(us_wp, 2), "frac": fr(flop3 / 2.25, us_wp), "algorithmic_tflops": tf(flop3, us_wp)},
            "wino3x3_x3_kernel<false> (bf16x3, 3x3 alone)": {"avg_launch_us": round(us_xp, 2), "frac": fr(6.0 * flop3 / 2.25, us_xp, BF16_MFMA_PEAK_TF),
                                                             "f32_equivalent_over_f32_peak": fr(flop3 / 2.25, us_xp), "algorithmic_tflops": tf(flop3, us_xp)},
            "convk_kernel<3,1,32,8,16,2,2,2,2,true> (direct, fused tail)": {"avg_launch_us": round(us_df, 2) if us_df == us_df else None, "frac": fr(flop, us_df)},
            "convk_kernel<3,1,32,8,16,2,2,2,2,false> (direct 3x3 alone)": {"avg_launch_us": round(us_dp, 2), "frac": fr(flop3, us_dp)}}
    # `achieved` / `frac` follow SURVEY.md 8(d): ALGORITHMIC FLOPs of the launch (the direct-form count the reference's hooks give:
    # 2 px (128*128*9 + 128*256)) over the launch's duration, against the dense peak of the pipe the kernel RUNS on.  Beside it, so that the
    # number cannot be misread: the FLOPs the kernel executes on that pipe over the same peak (`executed_frac`: Winograd issues 16 of 36
    # products, the bf16x3 form six MFMAs per product block), the fp32-equivalent rate over the fp32 peak, and from the committed PMC pass of
    # this very launch shape the share of cycles the matrix pipe was busy and the shader clock under this kernel's load (the peaks are quoted
    # at 2.4 GHz; `*_at_measured_clock` rescale them to what the chip actually ran).
    rec = committed_pmc("pmc_dominant_conv.json", L, dominant_kernel_name())
    clock = rec.get("shader_clock_ghz") if rec else None
    pmc = {"traffic": rec.get("hbm_bytes_per_launch") if rec else None, "traffic_source": "profiles/pmc_dominant_conv.json (rocprofv3 --pmc, tools/profile_round.sh)" if rec else None,
           "traffic_over_algorithmic_bytes": round(rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes"], 3) if rec else None,
           "mfma_busy": round(rec["mfma_util"], 4) if rec else None, "shader_clock_ghz": clock, "pmc_pass_avg_launch_us": rec.get("pmc_pass_avg_launch_us") if rec else None}
    at_clock = lambda v: round(v * 2.4 / clock, 4) if (clock and v is not None) else None  # noqa: E731
    abytes = 4.0 * px * (128 + 256 + 256) + 4.0 * (128 * 128 * 16 + 128 * 256)
    common = dict(pmc, algorithmic_flop_per_launch=flop, algorithmic_bytes_per_launch=abytes,
                  flop_basis="SURVEY.md 8(d): algorithmic FLOPs 2*px*(128*128*9 + 128*256) per launch / avg launch duration / dense peak of the pipe the kernel runs on",
                  bytes_basis="compulsory bytes of the launch: 4*px*(128 in + 256 skip + 256 out) + the weights once")

    def bound_entry(us, executed_flop, pipe_peak_tf, mfma_fields):
        """The roof that binds: the launch's intensity -- FLOPs it EXECUTES on its matrix pipe per algorithmic byte -- against the pipe's ridge (dense peak / 8 TB/s).
        Below the ridge the launch is HBM-bound: `achieved` = algorithmic bytes / duration in GB/s against the 8 TB/s spec (6.3 TB/s is what a copy reaches);
        the matrix-pipe accounting of rounds 1-5 stays beside it under `mfma`."""
        ridge = pipe_peak_tf * 1e12 / (HBM_PEAK_GBPS * 1e9)
        intensity, intensity_alg = executed_flop / abytes, flop / abytes
        if intensity < ridge:
            gbps = abytes / us / 1e3
            return dict(bound="hbm", unit="GB/s", achieved=round(gbps, 1), peak=HBM_PEAK_GBPS, frac=round(gbps / HBM_PEAK_GBPS, 4), frac_of_achievable_6300=round(gbps / 6300.0, 4),
                        intensity_flop_per_byte={"executed": round(intensity, 1), "algorithmic": round(intensity_alg, 1), "ridge_of_the_pipe": round(ridge, 1)},
                        mfma=dict(mfma_fields, unit="TFLOP/s", peak=pipe_peak_tf))
        return dict(mfma_fields, bound="mfma", unit="TFLOP/s", peak=pipe_peak_tf,
                    intensity_flop_per_byte={"executed": round(intensity, 1), "algorithmic": round(intensity_alg, 1), "ridge_of_the_pipe": round(ridge, 1)},
                    hbm_gbps_algorithmic=round(abytes / us / 1e3, 1))
    shape = "fused Residual tail: 3x3 128->128 (Winograd F(2x2,3x3)) + ReLU, 1x1 128->256 + skip @64x64, %d crops/launch" % L
    if matrix_pipe() == "f16x2":
        same["wino3x3_kernel<true> (fp32 pipe, SUO_WINO_BF16X3=0)"] = f32_entry
        same["wino3x3_x3_kernel<true,false,true,4,3> (bf16x3, SUO_F16X2=0)"] = x3_entry
        frac = fr(flop, us_h, BF16_MFMA_PEAK_TF)
        mf = dict(achieved=tf(flop, us_h), frac=frac, frac_at_measured_clock=at_clock(frac), executed_flop_per_launch=flop_exec_f16, executed_tflops=f16_entry["achieved_tflops"],
                  executed_frac=f16_entry["frac"], executed_frac_at_measured_clock=at_clock(f16_entry["frac"]),
                  executed_flop_basis="fp16 FLOPs issued to the MFMA pipe: 3 * (2*px*128*128*9/2.25 (Winograd 3x3) + 2*px*128*256 (1x1)); dense fp16 peak = the bf16 one",
                  f32_equivalent_executed_tflops=f16_entry["f32_equivalent_executed_tflops"], f32_equivalent_over_f32_peak=f16_entry["f32_equivalent_over_f32_peak"],
                  algorithmic_over_f32_peak=fr(flop, us_h))
        return dict(common, **bound_entry(us_h, flop_exec_f16, BF16_MFMA_PEAK_TF, mf), kernel="wino3x3_x3_kernel<true,false,true,4,2> " + shape,
                    dtype="f32 (2 x fp16 split products: 3 cross terms, fp32 accumulate)", pipe="fp16 MFMA", avg_launch_us=f16_entry["avg_launch_us"], same_process=same)
    same["wino3x3_x3_kernel<true,false,true,4,2> (f16x2, default)"] = f16_entry
    if wino_bf16x3_enabled():
        same["wino3x3_kernel<true> (fp32 pipe, SUO_WINO_BF16X3=0)"] = f32_entry
        frac = fr(flop, us_x, BF16_MFMA_PEAK_TF)
        mf = dict(achieved=tf(flop, us_x), frac=frac, frac_at_measured_clock=at_clock(frac), executed_flop_per_launch=flop_exec_bf16, executed_tflops=x3_entry["achieved_tflops"],
                  executed_frac=x3_entry["frac"], executed_frac_at_measured_clock=at_clock(x3_entry["frac"]),
                  executed_flop_basis="bf16 FLOPs issued to the MFMA pipe: 6 * (2*px*128*128*9/2.25 (Winograd 3x3) + 2*px*128*256 (1x1))",
                  f32_equivalent_executed_tflops=x3_entry["f32_equivalent_executed_tflops"], f32_equivalent_over_f32_peak=x3_entry["f32_equivalent_over_f32_peak"],
                  algorithmic_over_f32_peak=fr(flop, us_x))
        return dict(common, **bound_entry(us_x, flop_exec_bf16, BF16_MFMA_PEAK_TF, mf), kernel="wino3x3_x3_kernel<true,false,true> " + shape,
                    dtype="f32 (3 x bf16 split products: 6 cross terms, fp32 accumulate)", pipe="bf16 MFMA", avg_launch_us=x3_entry["avg_launch_us"], same_process=same)
    same["wino3x3_x3_kernel<true,false,true> (bf16x3, default)"] = x3_entry
    frac = fr(flop, us_w)
    mf = dict(achieved=tf(flop, us_w), frac=frac, frac_at_measured_clock=at_clock(frac), executed_flop_per_launch=flop_exec, executed_tflops=tf(flop_exec, us_w),
              executed_frac=fr(flop_exec, us_w), executed_frac_at_measured_clock=at_clock(fr(flop_exec, us_w)),
              executed_flop_basis="fp32 FLOPs issued to the MFMA pipe: 2*px*128*128*9/2.25 (Winograd 3x3) + 2*px*128*256 (1x1)")
    return dict(common, **bound_entry(us_w, flop_exec, FP32_MFMA_PEAK_TF, mf), kernel="wino3x3_kernel<true> " + shape, dtype="f32", pipe="fp32 MFMA", avg_launch_us=round(us_w, 2),
                same_process=same)


def gemm_roofline(L, iters=30):
    """The largest 1x1 convolution of a network call: conv1 of a 256 -> 256 Residual block at 64x64 -- BN + ReLU prologue (the
    pre-activation, layers/Residual.py:22-24), K = 256 -> N = 128, M = L * 4096 pixels.  2*M*N*K FLOPs against 4*(M*K + M*N) bytes = 42 FLOP/B.
    What the network launches (csrc/net.hip: residual): by default gemm_bf16x3_kernel (csrc/gemm_bf16x3.hip: bf16 matrix pipe, both operands
    split into three bf16 terms, 6 cross terms, fp32 accumulate) -- `achieved` / `frac` = executed bf16 FLOPs (6 x 2*M*N*K) against the dense
    bf16 peak, with the fp32-equivalent rate beside it; with SUO_WINO_BF16X3=0 the persistent fp32 GEMM (gemm_persist_kernel) against the
    fp32 MFMA peak.  The other form is timed in the same process."""
    import torch
    from suo_slam_amd import _lib
    rng = np.random.default_rng(1)
    M, K, N = L * 4096, 256, 128
    a = torch.rand((M, K), device="cuda") - 0.5
    out = torch.empty((M, N), device="cuda")
    w = (rng.standard_normal((N, K)) / 16.0).astype(np.float32)
    wp = torch.from_numpy(pack_gemm(w, N, K)).cuda()
    lib = _lib.lib()
    w3 = np.empty(3 * N * K, np.uint16)
    _lib.check(lib.suo_pack_gemm_weight_bf16x3(w.ctypes.data, N, K, w3.ctypes.data), "pack_bf16x3")
    w3d = torch.from_numpy(w3.view(np.int16)).cuda()
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, K).astype(np.float32)).cuda()
    sh = torch.from_numpy((rng.standard_normal(K) * 0.1).astype(np.float32)).cuda()
    b = torch.zeros(N, device="cuda")
    st = torch.cuda.current_stream()
    s = C.c_void_p(st.cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def gemm():
        _lib.check(lib.suo_conv1x1(P(a), K, K, P(sc), P(sh), None, 0, 0, P(wp), P(b), None, 0, P(out), N, M, N, N, 1, 0, s), "suo_conv1x1")

    def gemm_x3():
        _lib.check(lib.suo_conv1x1_bf16x3(P(a), K, K, P(sc), P(sh), P(w3d), P(b), P(out), N, M, N, 1, s), "suo_conv1x1_bf16x3")
    w16h, osch = np.empty(2 * N * K, np.uint16), np.empty(N, np.float32)
    _lib.check(lib.suo_pack_gemm_weight_f16x2(w.ctypes.data, N, K, w16h.ctypes.data, osch.ctypes.data), "pack_f16x2")
    w16d, oscd = torch.from_numpy(w16h.view(np.int16)).cuda(), torch.from_numpy(osch).cuda()
    rflag = torch.zeros(1, dtype=torch.int32, device="cuda")

    def gemm_f16():
        _lib.check(lib.suo_conv1x1_f16x2_ex(P(a), K, K, P(sc), P(sh), None, 0, 0, P(w16d), P(oscd), P(b), None, 0, P(out), N, M, N, 1, P(rflag), s), "suo_conv1x1_f16x2_ex")
    us, us3, us16 = _timed(gemm, st, iters), _timed(gemm_x3, st, iters), _timed(gemm_f16, st, iters)
    flop = 2.0 * M * N * K
    tf = lambda f, t: round(f / (t * 1e-6) / 1e12, 2) if t == t else None  # noqa: E731
    shape = "1x1 conv K256->N128 with BN+ReLU prologue, + ReLU, M = %d pixels (%d crops @64x64)" % (M, L)
    f32 = {"kernel": "gemm_persist_kernel: " + shape, "avg_launch_us": round(us, 2), "achieved_tflops": tf(flop, us), "peak": FP32_MFMA_PEAK_TF,
           "frac": round(flop / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF, 4)}
    x3 = {"kernel": "gemm_bf16x3_kernel: " + shape, "avg_launch_us": round(us3, 2), "achieved_tflops": tf(6.0 * flop, us3), "peak": BF16_MFMA_PEAK_TF,
          "frac": round(6.0 * flop / (us3 * 1e-6) / 1e12 / BF16_MFMA_PEAK_TF, 4), "f32_equivalent_tflops": tf(flop, us3),
          "f32_equivalent_over_f32_peak": round(flop / (us3 * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF, 4)}
    abytes = 4.0 * (M * K + M * N) + 4.0 * N * K
    # 42.7 FLOP per byte: below the ridge of either split form (six / three MFMAs per product block: 2500 / 6 / 6.3 TB/s = 66, 2500 / 3 / 6.3 = 132 FLOP/B) --
    # the split-form launches are HBM-bound: `achieved` = algorithmic bytes / time against the HBM peak (8 TB/s spec; ~6.3 achievable), the matrix-pipe rates beside it
    hbm = lambda t: {"bound": "hbm", "unit": "GB/s", "achieved": round(abytes / t / 1e3, 1), "peak": HBM_PEAK_GBPS, "frac": round(abytes / t / 1e3 / HBM_PEAK_GBPS, 4),  # noqa: E731
                     "frac_of_achievable_6300": round(abytes / t / 1e3 / 6300.0, 4), "flop_per_launch": flop, "algorithmic_bytes_per_launch": abytes,
                     "intensity_flop_per_byte": round(flop / abytes, 1)}
    f16 = {"kernel": "gemm_bf16x3_kernel<...,NP=2>: " + shape, "avg_launch_us": round(us16, 2), "executed_tflops": tf(3.0 * flop, us16),
           "executed_over_fp16_peak": round(3.0 * flop / (us16 * 1e-6) / 1e12 / BF16_MFMA_PEAK_TF, 4), "f32_equivalent_tflops": tf(flop, us16),
           "f32_equivalent_over_f32_peak": round(flop / (us16 * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF, 4), "hbm_gbps_algorithmic": round(abytes / us16 / 1e3, 1)}
    common = {"bound": "mfma", "unit": "TFLOP/s", "flop_per_launch": flop, "algorithmic_bytes_per_launch": abytes}
    if matrix_pipe() == "f16x2":
        x3["hbm_gbps_algorithmic"] = round(abytes / us3 / 1e3, 1)
        return dict(hbm(us16), kernel=f16["kernel"], dtype="f32 as 2 x fp16 (3 cross terms, fp32 accumulate)", avg_launch_us=f16["avg_launch_us"],
                    executed_flop_per_launch=3.0 * flop, executed_tflops=f16["executed_tflops"], executed_over_fp16_peak=f16["executed_over_fp16_peak"],
                    f32_equivalent_tflops=f16["f32_equivalent_tflops"], f32_equivalent_over_f32_peak=f16["f32_equivalent_over_f32_peak"],
                    traffic=committed_traffic("pmc_gemm.json", L, "gemm_bf16x3_kernel"), same_process={"bf16x3 (SUO_F16X2=0)": x3, "fp32 pipe (SUO_WINO_BF16X3=0)": f32})
    if wino_bf16x3_enabled():
        return dict(hbm(us3), kernel=x3["kernel"], dtype="f32 as 3 x bf16 (6 cross terms, fp32 accumulate)", avg_launch_us=x3["avg_launch_us"],
                    executed_flop_per_launch=6.0 * flop, executed_tflops=x3["achieved_tflops"], executed_over_bf16_peak=x3["frac"],
                    f32_equivalent_tflops=x3["f32_equivalent_tflops"], f32_equivalent_over_f32_peak=x3["f32_equivalent_over_f32_peak"],
                    traffic=committed_traffic("pmc_gemm.json", L, "gemm_bf16x3_kernel"), same_process={"f16x2 (default)": f16, "fp32 pipe (SUO_WINO_BF16X3=0)": f32})
    if False:
        return dict(common, kernel=x3["kernel"], dtype="f32 as 3 x bf16 (6 cross terms, fp32 accumulate)", achieved=x3["achieved_tflops"], peak=BF16_MFMA_PEAK_TF,
                    frac=x3["frac"], avg_launch_us=x3["avg_launch_us"], executed_flop_per_launch=6.0 * flop, f32_equivalent_tflops=x3["f32_equivalent_tflops"],
                    f32_equivalent_over_f32_peak=x3["f32_equivalent_over_f32_peak"], traffic=committed_traffic("pmc_gemm.json", L, "gemm_bf16x3_kernel"),
                    hbm_gbps_algorithmic=round(common["algorithmic_bytes_per_launch"] / us3 / 1e3, 1), same_process={"fp32 pipe (SUO_WINO_BF16X3=0)": f32})
    return dict(common, kernel=f32["kernel"], dtype="f32", achieved=f32["achieved_tflops"], peak=FP32_MFMA_PEAK_TF, frac=f32["frac"], avg_launch_us=f32["avg_launch_us"],
                traffic=committed_traffic("pmc_gemm.json", L, "gemm_persist_kernel"), same_process={"bf16x3 (default)": x3})


def bf16x3_leg(L, iters=30):
    """Accuracy of the bf16x3 form next to its speed: the largest 1x1 convolution of a call (as gemm_roofline) on the bf16 matrix pipe -- both
    operands split into three bf16 terms, 6 of the 9 cross products accumulated in fp32 (csrc/gemm_bf16x3.hip, what the network launches
    for conv1 of its Residual blocks) -- against the fp32 MFMA kernel: error of both against fp64 on the same inputs, and time."""
    import torch
    from suo_slam_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(1)
    M, K, N = L * 4096, 256, 128
    a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).cuda()
    w = (rng.standard_normal((N, K)) / 16.0).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, K).astype(np.float32), (rng.standard_normal(K) * 0.1).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    wp = torch.from_numpy(pack_gemm(w, N, K)).cuda()
    w3 = np.empty(3 * N * K, np.uint16)
    _lib.check(lib.suo_pack_gemm_weight_bf16x3(w.ctypes.data, N, K, w3.ctypes.data), "pack_bf16x3")
    w3d = torch.from_numpy(w3.view(np.int16)).cuda()
    scd, shd, bd = torch.from_numpy(sc).cuda(), torch.from_numpy(sh).cuda(), torch.from_numpy(b).cuda()
    o32, o3 = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
    st = torch.cuda.current_stream()
    s = C.c_void_p(st.cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    f32 = lambda: _lib.check(lib.suo_conv1x1(P(a), K, K, P(scd), P(shd), None, 0, 0, P(wp), P(bd), None, 0, P(o32), N, M, N, N, 1, 0, s), "suo_conv1x1")  # noqa: E731
    x3 = lambda: _lib.check(lib.suo_conv1x1_bf16x3(P(a), K, K, P(scd), P(shd), P(w3d), P(bd), P(o3), N, M, N, 1, s), "suo_conv1x1_bf16x3")  # noqa: E731
    us32, us3 = _timed(f32, st, iters), _timed(x3, st, iters)
    rows = slice(0, 4096)
    pre = np.maximum(a[rows].cpu().numpy() * sc + sh, 0).astype(np.float32).astype(np.float64)      # the prologue is float32 in both kernels
    ref = np.maximum(pre @ w.astype(np.float64).T + b, 0)
    e32, e3 = np.abs(o32[rows].cpu().numpy() - ref).max(), np.abs(o3[rows].cpu().numpy() - ref).max()
    flop = 2.0 * M * N * K
    return {"kernel": "gemm_bf16x3_kernel vs gemm_persist_kernel: 1x1 conv K256->N128, BN+ReLU prologue, + ReLU, M = %d" % M, "dtype": "f32 via bf16x3",
            "f32_mfma_us": round(us32, 1), "bf16x3_us": round(us3, 1), "speedup": round(us32 / us3, 3),
            "bf16x3_tflops_f32_equivalent": round(flop / us3 / 1e6, 1), "bf16x3_over_f32_mfma_peak": round(flop / us3 / 1e6 / FP32_MFMA_PEAK_TF, 3),
            "max_abs_err_vs_fp64": {"f32_mfma": float(f"{e32:.3e}"), "bf16x3": float(f"{e3:.3e}")}, "output_range": round(float(np.abs(ref).max()), 3),
            "algorithmic_bytes_per_launch": 4.0 * (M * K + M * N), "note": "fp32 accuracy holds (tests/test_gpu_cnn.py); see DESIGN.md section 4"}


def fp32_pipe_leg(args, L):
    """Why the line says dtype "f32" although most products are formed on the bf16 matrix pipe: the SAME benchmark with every product on the fp32
    matrix pipe (SUO_WINO_BF16X3=0, read when the network is built: a child process, 4 timed steps), and the dominant kernel of both forms
    against fp64 on the same inputs -- measured here, by whoever runs this file."""
    import torch
    import torch.nn.functional as Fn
    from suo_slam_amd import _lib
    out = {}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-legs", "--steps", "4", "--warmup", "2", "--objects", str(args.objects), "--frames-per-step",
           str(args.frames_per_step), "--depth", str(args.depth)] + (["--frames-from-host"] if args.frames_from_host else [])
    for tag, env_add in (("fp32_pipe", {"SUO_WINO_BF16X3": "0"}), ("bf16x3", {"SUO_F16X2": "0"})):      # the same timed region on the other two forms, a child process each
        if tag == "bf16x3" and matrix_pipe() != "f16x2":
            continue
        r = subprocess.run(cmd, env=dict(os.environ, **env_add), capture_output=True, text=True, timeout=400)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and line:
            j = json.loads(line[-1])
            out["frames_per_s_" + tag] = j["value"]
            out["ms_per_step_" + tag] = j["ms_per_step"]
            out["steps"] = j["steps"]
        else:
            out["error_" + tag] = (r.stderr or r.stdout)[-300:]
    # the dominant kernel (fused Residual tail @64x64) of both forms against fp64: 2 crops, same inputs
    lib = _lib.lib()
    rng = np.random.default_rng(3)
    Lc = 2
    x = rng.standard_normal((Lc, 64, 64, 128)).astype(np.float32)
    skip = rng.standard_normal((Lc, 64, 64, 256)).astype(np.float32)
    w2 = (rng.standard_normal((128, 128, 3, 3)) / 34.0).astype(np.float32)
    w3 = (rng.standard_normal((256, 128)) / 11.0).astype(np.float32)
    b2 = (rng.standard_normal(128) * 0.3).astype(np.float32)
    b3 = rng.standard_normal(256).astype(np.float32)
    wq = np.empty(16 * 128 * 128, np.float32)
    _lib.check(lib.suo_pack_wino_weight(w2.ctypes.data, 128, 128, 128, 128, wq.ctypes.data), "pack_wino")
    wq3h = np.empty(3 * 16 * 128 * 128, np.uint16)
    _lib.check(lib.suo_pack_wino_weight_bf16x3(w2.ctypes.data, 128, 128, wq3h.ctypes.data), "pack_wino_x3")
    w3xh = np.empty(3 * 256 * 128, np.uint16)
    _lib.check(lib.suo_pack_tail_weight_bf16x3(w3.ctypes.data, 256, 128, w3xh.ctypes.data), "pack_tail_x3")
    wq16h, o2h, w3p16h, o3h = np.empty(2 * 16 * 128 * 128, np.uint16), np.empty(128, np.float32), np.empty(2 * 256 * 128, np.uint16), np.empty(256, np.float32)
    _lib.check(lib.suo_pack_wino_weight_f16x2(w2.ctypes.data, 128, 128, wq16h.ctypes.data, o2h.ctypes.data), "pack_wino_f16x2")
    _lib.check(lib.suo_pack_tail_weight_f16x2(w3.ctypes.data, 256, 128, w3p16h.ctypes.data, o3h.ctypes.data), "pack_tail_f16x2")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    xd, sd, wqd, wq3d, w3xd, wp3d, b2d, b3d = d(x), d(skip), d(wq), d(wq3h.view(np.int16)), d(w3xh.view(np.int16)), d(pack_gemm(w3, 256, 128)), d(b2), d(b3)
    o32, o3 = torch.empty((Lc, 64, 64, 256), device="cuda"), torch.empty((Lc, 64, 64, 256), device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(lib.suo_conv3x3_wino_conv1x1_skip(P(xd), Lc, 64, 64, P(wqd), P(b2d), P(wp3d), P(b3d), P(sd), P(o32), s), "suo_conv3x3_wino_conv1x1_skip")
    _lib.check(lib.suo_conv3x3_wino_x3_conv1x1_skip_up(P(xd), Lc, 64, 64, P(wq3d), P(b2d), P(w3xd), 1, P(b3d), P(sd), None, P(o3), s), "suo_conv3x3_wino_x3_conv1x1_skip_up")
    o16, rflag = torch.empty((Lc, 64, 64, 256), device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    wq16d, o2d, w3p16d, o3d = d(wq16h.view(np.int16)), d(o2h), d(w3p16h.view(np.int16)), d(o3h)
    _lib.check(lib.suo_conv3x3_wino_f16x2_conv1x1_skip_up(P(xd), Lc, 64, 64, P(wq16d), P(o2d), P(b2d), P(w3p16d), P(o3d), P(b3d), P(sd), None, P(o16), P(rflag), s),
               "suo_conv3x3_wino_f16x2_conv1x1_skip_up")
    torch.cuda.synchronize()
    xm = torch.from_numpy(x).permute(0, 3, 1, 2).double()
    m = Fn.relu(Fn.conv2d(xm, torch.from_numpy(w2).double(), torch.from_numpy(b2).double(), padding=1))
    ref = (Fn.conv2d(m, torch.from_numpy(w3).double()[:, :, None, None], torch.from_numpy(b3).double())).permute(0, 2, 3, 1).numpy() + skip
    e32, e3 = float(np.abs(o32.cpu().numpy() - ref).max()), float(np.abs(o3.cpu().numpy() - ref).max())
    e16 = float(np.abs(o16.cpu().numpy() - ref).max())
    out["dominant_kernel_max_abs_err_vs_fp64"] = {"fp32_pipe (wino3x3_kernel<true>)": float(f"{e32:.3e}"), "bf16x3 (wino3x3_x3_kernel<true,false,true,4,3>)": float(f"{e3:.3e}"),
                                                  "f16x2 (wino3x3_x3_kernel<true,false,true,4,2>, the default)": float(f"{e16:.3e}"), "f16x2_range_flag": int(rflag.item()),
                                                  "output_range": round(float(np.abs(ref).max()), 3), "crops": Lc}
    return out


def latency_roofline(L=8, iters=50):
    """The dominant kernel of the reference's call shape (one frame = 8 crops per network call): the same fused Winograd tail at
    256 tiles -- one workgroup per CU, a quarter of the chip's wave slots."""
    r = conv_roofline(L, iters)
    keep = ("bound", "kernel", "dtype", "pipe", "achieved", "peak", "unit", "frac", "avg_launch_us", "mfma", "intensity_flop_per_byte", "flop_basis")
    out = {k: r[k] for k in keep if k in r}
    out["same_process"] = {k: v for k, v in r["same_process"].items() if k.startswith("wino3x3_kernel<true>") or k.startswith("wino3x3_x3_kernel<true")}
    return out


def whole_call(net_bytes, ms_per_step, steps_overlap):
    """The roof that binds the CALL: the algorithmic HBM bytes of every launch of one network call (suo_net_schedule_bytes: operands read once, results written once,
    weights once, summed over the schedule the network really runs at this crop count) over the measured time of a step, against 6.3 TB/s (what a copy kernel reaches
    on this part) and the 8 TB/s spec.  The step time is that of the timed region (`steps_overlap` steps in flight, geometry included), so the rate is a lower bound
    on what the network's launches sustain."""
    total = net_bytes["total"]
    gbps = total / (ms_per_step * 1e-3) / 1e9
    return {"algorithmic_bytes_per_call": total, "by_kind": {k: v for k, v in net_bytes.items() if k not in ("total", "launches")}, "launches_per_call": net_bytes["launches"],
            "ms_per_step": ms_per_step, "steps_in_flight": steps_overlap, "bound": "hbm", "unit": "GB/s", "achieved": round(gbps, 1), "peak_achievable": 6300.0,
            "frac_of_achievable_6300": round(gbps / 6300.0, 4), "peak": HBM_PEAK_GBPS, "frac": round(gbps / HBM_PEAK_GBPS, 4),
            "basis": "sum over the call's launches of (inputs + outputs + weights) bytes, fp32 activations / ms_per_step of the timed region"}

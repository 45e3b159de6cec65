"""What the legs of bench.py share: the workload's constants, the synthetic frame pool, the matrix-pipe switches, committed PMC summaries, HIP-event timing."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GFLOP_PER_CROP = 31.495          # conv FLOPs, hook-counted on the reference module
# Without priors (this workload: single-view frames, lib/object_slam.py:1094-1097 feeds zeros) 41 of the stem's 44 input
# channels are structural zeros and their MACs are never issued (csrc/net.hip: stem_img_): 2*128*128*64*49*41 per crop.
GFLOP_SKIPPED_PER_CROP = 2 * 128 * 128 * 64 * 49 * 41 / 1e9
FP32_MFMA_PEAK_TF = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md chip table
BF16_MFMA_PEAK_TF = 2500.0       # dense, same table (the fp32 pipe is 1/16 of it); fp16 runs at the bf16 rate
HBM_PEAK_GBPS = 8000.0           # HBM3E spec, same table (~6.3 TB/s achievable)
BBOX_THRESH, KP_VAR_THRESH = 1.0, 0.5      # evaluate.py:66-74 (the T-LESS pair): with random weights the YCB-V pair masks everything


def winograd_saved_gflop_per_crop(crops_per_call):
    """MACs the Winograd F(2x2,3x3) form does not execute (csrc/conv_wino.hip): the 3x3 convolution of a Residual block (128 -> 128,
    or 64 -> 64 in r1 / r4) runs in that form when its launch has enough tiles of 8 x 16 pixels (csrc/net.hip: 32 on the fp16 pipe, 256 on the
    others) and is not taken by the one-launch block kernels (maps of <= 32 pixels a side up to 768 tiles of 4 x 8: direct products), at 16
    instead of 36 products per 2x2 tile.  Such convolutions per crop (hg.py:7-58, 2 stacks): 128 channels -- 9 at 64x64 (r5, up1 and the
    post-hourglass blocks), 12 at 32x32, 12 at 16x16; 64 channels -- r1 at 128x128, r4 at 64x64."""
    min_tiles = 32 if matrix_pipe() == "f16x2" else 256
    saved = 0.0
    for hw, count, ch in ((64, 9, 128), (32, 12, 128), (16, 12, 128), (128, 1, 64), (64, 1, 64)):
        tiles = crops_per_call * (hw // 8) * (hw // 16)
        one_launch = hw <= 32 and crops_per_call * (hw // 4) * (hw // 8) <= 768
        if tiles >= min_tiles and not one_launch:
            saved += count * 2.0 * hw * hw * ch * ch * 9 * (1 - 1 / 2.25) / 1e9
    return saved


N_OBJ = 8


def make_pool(rng, n, L):
    """Synthetic frames: pixels, boxes, class masks, model keypoints, diameters (what the dataset hands process_view) plus the
    ground truth the pose check needs.  Nothing derived from them is precomputed."""
    from suo_slam_amd import synthetic as S
    return [S.make_frame(rng, L, noise=0.01, outlier_frac=0.05) for _ in range(n)]


def confident_state_dict():
    """Seeded random weights whose validity head says yes (bias + 4): the decode / mask / compaction path then hands real,
    data-dependent keypoint sets to PnP and LM (tests/test_gpu_sixteen_objects.py uses the same construction)."""
    from suo_slam_amd import weights
    sd = weights.make_random_state_dict(0, 8.0)
    sd["classifier.2.bias"] = (np.asarray(sd["classifier.2.bias"]) + 4.0).astype(np.float32)
    return sd


def pack_conv(w, Np, Cp, CK):
    from suo_slam_amd import _lib
    w = np.ascontiguousarray(w, np.float32)
    out = np.empty(2 * Np * ((Cp * w.shape[2] * w.shape[3] + 15) // 16 * 16), np.float32)
    _lib.check(_lib.lib().suo_pack_conv_weight(w.ctypes.data, w.shape[0], w.shape[1], w.shape[2], Np, Cp, CK, out.ctypes.data), "pack_conv")
    return out


def pack_gemm(w, Np, Kp):
    from suo_slam_amd import _lib
    w = np.ascontiguousarray(w, np.float32)
    out = np.empty(2 * Np * Kp, np.float32)
    _lib.check(_lib.lib().suo_pack_gemm_weight(w.ctypes.data, w.shape[0], w.shape[1], Np, Kp, out.ctypes.data), "pack_gemm")
    return out


def committed_traffic(name, L, kernel_prefix):
    """HBM bytes per launch from a committed PMC summary (tools/profile_round.sh -> tools/pmc_to_json.py), or None when the summary is
    for another launch shape / kernel."""
    pmc = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(pmc):
        return None
    rec = json.load(open(pmc))
    if rec.get("crops_per_launch") == L and rec.get("kernel", "").replace(" ", "").startswith(kernel_prefix):
        return rec.get("hbm_bytes_per_launch")
    return None


def committed_pmc(name, L, kernel_prefix):
    """The committed PMC summary itself (tools/profile_round.sh -> tools/pmc_to_json.py) when it is for this launch shape / kernel."""
    pmc = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(pmc):
        return None
    rec = json.load(open(pmc))
    if rec.get("crops_per_launch") == L and rec.get("kernel", "").replace(" ", "").startswith(kernel_prefix):
        return rec
    return None


def wino_bf16x3_enabled():
    """csrc/net.hip: the Residual blocks' 3x3 convolution + fused tail run on the bf16 matrix pipe with 3-way split operands unless SUO_WINO_BF16X3=0."""
    return os.environ.get("SUO_WINO_BF16X3", "1") not in ("0", "")


def matrix_pipe():
    """csrc/net.hip, read when a network is built: "f32" (SUO_WINO_BF16X3=0), "bf16x3" (SUO_F16X2=0: three bf16 terms per operand, six MFMAs per product block)
    or "f16x2" (default: two fp16 terms, three MFMAs, range-guarded -- csrc/f16x2.h)."""
    if not wino_bf16x3_enabled():
        return "f32"
    return "bf16x3" if os.environ.get("SUO_F16X2", "1") in ("0", "") else "f16x2"


# the arithmetic the path computes in (the line's `dtype`): tensors and accumulation are fp32 in every form; what differs is how a product is formed
DTYPE = {"f16x2": "f32 (2 x fp16 split products)", "bf16x3": "f32 (3 x bf16 split products)", "f32": "f32"}
DTYPE_NOTE = {
    "f16x2": ("fp32 tensors and fp32 accuracy end to end; the Residual blocks' 3x3 + tail and the large 1x1 convolutions form their products on the fp16 matrix "
              "pipe from operands split into two fp16 terms (hi*lo + lo*hi + hi*hi, fp32 accumulate; operands scaled into fp16's range by exact powers of two, "
              "a range guard re-issues a call that leaves it on the bf16x3 form): suo_slam_amd/csrc/f16x2.h, DESIGN.md section 4"),
    "bf16x3": ("fp32 tensors and fp32 accuracy end to end; the Residual blocks' 3x3 + tail and the large 1x1 convolutions form their products on "
               "the bf16 matrix pipe from operands split into three bf16 terms (6 cross terms, fp32 accumulate; SUO_F16X2=0): DESIGN.md section 4"),
    "f32": "fp32 MFMA throughout (SUO_WINO_BF16X3=0)"}


def dominant_kernel_name():
    # (template arguments FUSE, UP, TX3, NT, NP, NEXT, W8: the four-wave fused tail of a batched launch)
    return {"f16x2": "wino3x3_x3_kernel<true,false,true,4,2,false,false>", "bf16x3": "wino3x3_x3_kernel<true,false,true,4,3,false,false>", "f32": "wino3x3_kernel<true"}[matrix_pipe()]


def dominant_kernel_traffic(L):
    return committed_traffic("pmc_dominant_conv.json", L, dominant_kernel_name())


def _timed(f, st, iters):
    import torch
    try:
        f()
    except Exception:
        return float("nan")
    for _ in range(10):                                      # (the first launches of a kernel in a process run 5-25 % slow)
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        f()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

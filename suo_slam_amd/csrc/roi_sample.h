// RoIAlign(aligned=False, sampling_ratio=-1, spatial_scale=1) sampling shared by the crop kernel (csrc/misc.hip: roi_align_concat_kernel) and the
// fused stem (csrc/stem_x3.hip): the restated torchvision arithmetic, one function so that both stage the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace suo {

// FMT 0: uint8 HWC (scaled by 1/255 here = object_slam.py:1092 fused); FMT 1: float32 CHW planar.
// LUT (FMT 0): the taps' u / 255.0f read from 256 floats in LDS that roi_fill_lut computed with THIS expression on the device -- the same bits
// without an IEEE division per tap (12 per sampled pixel).
__device__ __forceinline__ void roi_fill_lut(float* lut, int tid) {          // lut[0..255] by the first 256 threads of a workgroup; the caller synchronises
    if (tid < 256) lut[tid] = (float)(uint8_t)tid / 255.0f;
}

// One bilinear sample in two halves, so that a caller can put the loads of several samples in flight before it consumes any (csrc/stem_x3.hip):
// roi_taps_issue computes the four taps' weights and REQUESTS their 12 values; roi_taps_value turns them into the sample (same operation order as
// the restated torchvision kernel: ((w00*v00 + w01*v01) + w10*v10) + w11*v11, added to a zero accumulator).
template <int FMT>
struct RoiTaps {
    float w[4];
    uint8_t u[4][3];       // FMT 0
    float f[4][3];         // FMT 1
    bool in;               // false: the sample lies outside the frame (contributes 0, nothing was loaded)
};

template <int FMT>
__device__ __forceinline__ void roi_taps_issue(const void* __restrict__ img, int H, int W, float y, float x, RoiTaps<FMT>& t) {
    t.in = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (!t.in) return;
    y = fmaxf(y, 0.f);
    x = fmaxf(x, 0.f);
    int y0 = (int)y, x0 = (int)x, y1, x1;
    if (y0 >= H - 1) { y0 = y1 = H - 1; y = (float)y0; } else y1 = y0 + 1;
    if (x0 >= W - 1) { x0 = x1 = W - 1; x = (float)x0; } else x1 = x0 + 1;
    const float ly = y - (float)y0, lx = x - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    t.w[0] = hy * hx; t.w[1] = hy * lx; t.w[2] = ly * hx; t.w[3] = ly * lx;
    const size_t o[4] = {(size_t)y0 * W + x0, (size_t)y0 * W + x1, (size_t)y1 * W + x0, (size_t)y1 * W + x1};
    if (FMT == 0) {
        const uint8_t* b = (const uint8_t*)img;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) t.u[k][c] = b[o[k] * 3 + c];
    } else {
        const float* b = (const float*)img;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) t.f[k][c] = b[(size_t)c * H * W + o[k]];
    }
}

template <int FMT, bool LUT>
__device__ __forceinline__ void roi_taps_value(const RoiTaps<FMT>& t, const float* lut, float acc[3]) {
    if (!t.in) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = FMT == 0 ? (LUT ? lut[t.u[k][c]] : (float)t.u[k][c] / 255.0f) : t.f[k][c];
        acc[c] += __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(t.w[0], v[0]), __fmul_rn(t.w[1], v[1])), __fmul_rn(t.w[2], v[2])), __fmul_rn(t.w[3], v[3]));
    }
}

template <int FMT, bool LUT = false>
__device__ __forceinline__ void bilinear3(const void* __restrict__ img, int H, int W, float y, float x, float acc[3], const float* lut = nullptr) {
    RoiTaps<FMT> t;
    roi_taps_issue<FMT>(img, H, W, y, x, t);
    roi_taps_value<FMT, LUT>(t, lut, acc);
}

// geometry of a box's RoIAlign bins
struct RoiBins { float x1, y1, bin_w, bin_h; int gw, gh; };
__device__ __forceinline__ RoiBins roi_bins(float x1, float y1, float x2, float y2) {
    const float roi_w = fmaxf(x2 - x1, 1.0f), roi_h = fmaxf(y2 - y1, 1.0f);
    return RoiBins{x1, y1, roi_w / 256.0f, roi_h / 256.0f, (int)ceilf(roi_w / 256.0f), (int)ceilf(roi_h / 256.0f)};
}
// the one sample of bin (ph, pw) when gh == gw == 1 (boxes of <= 256 px): t / 1.0f == t, so the general path's divisions drop out bit for bit
__device__ __forceinline__ void roi_single_sample_pos(const RoiBins& b, int ph, int pw, float& y, float& x) {
    y = __fadd_rn(__fadd_rn(b.y1, __fmul_rn((float)ph, b.bin_h)), __fmul_rn(0.5f, b.bin_h));
    x = __fadd_rn(__fadd_rn(b.x1, __fmul_rn((float)pw, b.bin_w)), __fmul_rn(0.5f, b.bin_w));
}


// value of crop pixel (ph, pw) of the 256 x 256 RoIAlign of box (x1, y1, x2, y2): the average of ceil(roi / 256)^2 bilinear samples per bin
template <int FMT, bool LUT = false>
__device__ __forceinline__ void roi_sample(const void* __restrict__ img, int H, int W, float x1, float y1, float x2, float y2, int ph, int pw, float out[3],
                                           const float* lut = nullptr) {
    const RoiBins b = roi_bins(x1, y1, x2, y2);
    const float bin_h = b.bin_h, bin_w = b.bin_w;
    const int gh = b.gh, gw = b.gw;
    float acc[3] = {0.f, 0.f, 0.f};
    if (gh == 1 && gw == 1) {
        float y, x;
        roi_single_sample_pos(b, ph, pw, y, x);
        bilinear3<FMT, LUT>(img, H, W, y, x, acc, lut);
        out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
        return;
    }
    for (int iy = 0; iy < gh; ++iy) {
        const float y = __fadd_rn(__fadd_rn(y1, __fmul_rn((float)ph, bin_h)), __fmul_rn((float)iy + 0.5f, bin_h) / (float)gh);
        for (int ix = 0; ix < gw; ++ix) {
            const float x = __fadd_rn(__fadd_rn(x1, __fmul_rn((float)pw, bin_w)), __fmul_rn((float)ix + 0.5f, bin_w) / (float)gw);
            bilinear3<FMT, LUT>(img, H, W, y, x, acc, lut);
        }
    }
    const float cnt = (float)(gh * gw);
    out[0] = acc[0] / cnt; out[1] = acc[1] / cnt; out[2] = acc[2] / cnt;
}

}  // namespace suo

// RoIAlign(aligned=False, sampling_ratio=-1, spatial_scale=1) sampling shared by the crop kernel (csrc/misc.hip: roi_align_concat_kernel) and the
// fused stem (csrc/stem_x3.hip): the restated torchvision arithmetic, one function so that both stage the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace suo {

// FMT 0: uint8 HWC (scaled by 1/255 here = object_slam.py:1092 fused); FMT 1: float32 CHW planar
template <int FMT>
__device__ __forceinline__ float pix(const void* __restrict__ img, int H, int W, int y, int x, int c) {
    if (FMT == 0) return (float)((const uint8_t*)img)[((size_t)y * W + x) * 3 + c] / 255.0f;
    return ((const float*)img)[((size_t)c * H + y) * W + x];
}

template <int FMT>
__device__ __forceinline__ void bilinear3(const void* __restrict__ img, int H, int W, float y, float x, float acc[3]) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return;
    y = fmaxf(y, 0.f);
    x = fmaxf(x, 0.f);
    int y0 = (int)y, x0 = (int)x, y1, x1;
    if (y0 >= H - 1) { y0 = y1 = H - 1; y = (float)y0; } else y1 = y0 + 1;
    if (x0 >= W - 1) { x0 = x1 = W - 1; x = (float)x0; } else x1 = x0 + 1;
    const float ly = y - (float)y0, lx = x - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // same operation order as the restated torchvision kernel: ((w00*v00 + w01*v01) + w10*v10) + w11*v11
        const float v00 = pix<FMT>(img, H, W, y0, x0, c), v01 = pix<FMT>(img, H, W, y0, x1, c);
        const float v10 = pix<FMT>(img, H, W, y1, x0, c), v11 = pix<FMT>(img, H, W, y1, x1, c);
        acc[c] += __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w00, v00), __fmul_rn(w01, v01)), __fmul_rn(w10, v10)), __fmul_rn(w11, v11));
    }
}


// value of crop pixel (ph, pw) of the 256 x 256 RoIAlign of box (x1, y1, x2, y2): the average of ceil(roi / 256)^2 bilinear samples per bin
template <int FMT>
__device__ __forceinline__ void roi_sample(const void* __restrict__ img, int H, int W, float x1, float y1, float x2, float y2, int ph, int pw, float out[3]) {
    const float roi_w = fmaxf(x2 - x1, 1.0f), roi_h = fmaxf(y2 - y1, 1.0f);
    const float bin_h = roi_h / 256.0f, bin_w = roi_w / 256.0f;
    const int gh = (int)ceilf(roi_h / 256.0f), gw = (int)ceilf(roi_w / 256.0f);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int iy = 0; iy < gh; ++iy) {
        const float y = __fadd_rn(__fadd_rn(y1, __fmul_rn((float)ph, bin_h)), __fmul_rn((float)iy + 0.5f, bin_h) / (float)gh);
        for (int ix = 0; ix < gw; ++ix) {
            const float x = __fadd_rn(__fadd_rn(x1, __fmul_rn((float)pw, bin_w)), __fmul_rn((float)ix + 0.5f, bin_w) / (float)gw);
            bilinear3<FMT>(img, H, W, y, x, acc);
        }
    }
    const float cnt = (float)(gh * gw);
    out[0] = acc[0] / cnt; out[1] = acc[1] / cnt; out[2] = acc[2] / cnt;
}

}  // namespace suo

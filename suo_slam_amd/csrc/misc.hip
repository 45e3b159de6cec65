// HBM-bound kernels of the keypoint-CNN path: RoI crop + prior concat, 2x2 max-pool, nearest 2x
// up-sample + add, heat-map decode (softmax -> soft-argmax -> 2x2 covariance), validity head and
// the boolean keypoint masks.  NHWC fp32 activations; 16-byte vector accesses; wave64 reductions.
//
// Reference ops replaced:
//   image prep  img/255, HWC->CHW                 /root/reference/lib/object_slam.py:1092
//   torchvision.ops.roi_align + torch.cat         /root/reference/lib/models/pkpnet.py:93-101
//   nn.MaxPool2d(2,2)                             /root/reference/lib/models/hg.py:16,71
//   F.interpolate(scale_factor=2) + add           /root/reference/lib/models/hg.py:56-58
//   spatial_softmax / mesh_grid / post_process_kp /root/reference/lib/models/pkpnet.py:13-63
//   classifier (ReLU, Linear, sigmoid)            /root/reference/lib/models/pkpnet.py:74-78,116-118
//   keypoint mask logic                           /root/reference/lib/object_slam.py:1100-1115
#include <math.h>

#include <mutex>

#include <algorithm>

#include "roi_sample.h"
#include "suo_internal.h"

namespace suo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------------------------------------
// 2x2 max-pool, NHWC.  One thread per output float4.
// Index arithmetic in 32 bits, with shifts when the extents are powers of two (every map of this network).  Both kernels
// are HBM streams: 5.0 TB/s (max-pool of the 128x128x128 map) and 4.8 TB/s (up-sample + add at 64x64x256) at 128 crops.
struct Pow2Dims { unsigned c_mask, c_shift, w_mask, w_shift, h_mask, h_shift; int pow2; };
static Pow2Dims pow2_dims(int H, int W, int C4) {
    auto lg = [](int v) { int s = 0; while ((1 << s) < v) ++s; return s; };
    auto p2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    Pow2Dims d;
    d.pow2 = p2(H) && p2(W) && p2(C4);
    d.c_mask = C4 - 1; d.c_shift = lg(C4); d.w_mask = W - 1; d.w_shift = lg(W); d.h_mask = H - 1; d.h_shift = lg(H);
    return d;
}
__device__ __forceinline__ void split_index(unsigned i, const Pow2Dims& d, int H, int W, int C4, int& l, int& y, int& x, int& c) {
    if (d.pow2) {
        c = i & d.c_mask; unsigned p = i >> d.c_shift;
        x = p & d.w_mask; p >>= d.w_shift;
        y = p & d.h_mask; l = p >> d.h_shift;
    } else {
        c = i % (unsigned)C4; unsigned p = i / (unsigned)C4;
        x = p % (unsigned)W; p /= (unsigned)W;
        y = p % (unsigned)H; l = p / (unsigned)H;
    }
}

__global__ void maxpool2_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int L, int OH, int OW, int C4, Pow2Dims dims) {
    const unsigned total = (unsigned)L * OH * OW * C4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int l, oy, ox, c;
        split_index(i, dims, OH, OW, C4, l, oy, ox, c);
        const size_t W = (size_t)OW * 2;
        const size_t base = (((size_t)l * OH * 2 + oy * 2) * W + ox * 2) * C4 + c;
        const f32x4 a = in[base], b = in[base + C4], d = in[base + W * C4], e = in[base + W * C4 + C4];
        f32x4 r;
#pragma unroll
        for (int t = 0; t < 4; ++t) r[t] = fmaxf(fmaxf(a[t], b[t]), fmaxf(d[t], e[t]));
        out[i] = r;
    }
}

int launch_maxpool2(const float* in, float* out, int L, int H, int W, int C, hipStream_t s) {
    if ((H | W) & 1 || (C & 3)) { suo_set_error("maxpool2: bad shape"); return SUO_ERR_ARG; }
    const size_t total = (size_t)L * (H / 2) * (W / 2) * (C / 4);
    if (total >= (1ull << 31)) { suo_set_error("maxpool2: tensor too large for 32-bit indexing"); return SUO_ERR_ARG; }
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(maxpool2_kernel, dim3(blocks), dim3(256), 0, s, (const f32x4*)in, (f32x4*)out, L, H / 2, W / 2, C / 4,
                       pow2_dims(H / 2, W / 2, C / 4));
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

// ------------------------------------------------------------------------------------------------
// out[l,y,x,:] = up1[l,y,x,:] + low[l,y/2,x/2,:]   (nearest 2x up-sample + add), out is [L,H,W,C]
__global__ void upsample2_add_kernel(const f32x4* __restrict__ up1, const f32x4* __restrict__ low,
                                     f32x4* __restrict__ out, int L, int H, int W, int C4, Pow2Dims dims) {
    const unsigned total = (unsigned)L * H * W * C4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int l, y, x, c;
        split_index(i, dims, H, W, C4, l, y, x, c);
        const size_t li = (((size_t)l * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) * C4 + c;
        out[i] = up1[i] + low[li];
    }
}

int launch_upsample2_add(const float* up1, const float* low, float* out, int L, int H, int W, int C, hipStream_t s) {
    if ((H | W) & 1 || (C & 3)) { suo_set_error("upsample2_add: bad shape"); return SUO_ERR_ARG; }
    const size_t total = (size_t)L * H * W * (C / 4);
    if (total >= (1ull << 31)) { suo_set_error("upsample2_add: tensor too large for 32-bit indexing"); return SUO_ERR_ARG; }
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(upsample2_add_kernel, dim3(blocks), dim3(256), 0, s, (const f32x4*)up1, (const f32x4*)low,
                       (f32x4*)out, L, H, W, C / 4, pow2_dims(H, W, C / 4));
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

// ------------------------------------------------------------------------------------------------
// RoIAlign(aligned=False, sampling_ratio=-1, spatial_scale=1) of a uint8 HWC image (values /255)
// to 256x256, concatenated with the 41 prior heat-maps (NCHW, or NULL = zeros), written NHWC with
// the 44 channels padded to 48.  One thread per output pixel.  The sampling itself: csrc/roi_sample.h (shared with the fused stem).

// ---- prior heat-maps rendered on the device ------------------------------------------------------------------------
// make_prior_kp_input / draw_gaussian_2d / gaussian_2d (/root/reference/lib/utils/utils.py:356-411): per valid keypoint a
// 91 x 91 max-normalised Gaussian (cv2.GaussianBlur of an impulse: sigma = 0.3*((91-1)*0.5-1)+0.8 = 14, BORDER_REFLECT_101
// doubles the outermost ring) is PASTED (assigned) with its upper-left corner at pt - 45, pt = round-half-even of the
// keypoint's pixel position; the paste window is [pt-45, pt+45) clipped to the image, i.e. the last row / column of the
// patch is never used.  patch[i][j] = w[i]*w[j] with w[i] = exp(-(i-45)^2 / (2 sigma^2)) * (i == 0 || i == 90 ? 2 : 1).
// The reference renders on the host and uploads 41*256*256 floats per crop (10.7 MB); here the stamp is evaluated while
// the crop is staged and the dense tensor never exists.
constexpr int PRIOR_HALF = 45, PRIOR_SIZE = 2 * PRIOR_HALF + 1;
__constant__ double c_prior_w[PRIOR_SIZE];

static int ensure_prior_table() {
    static int rc = -1;
    static std::once_flag once;
    std::call_once(once, [] {
        double w[PRIOR_SIZE];
        const double sigma = 0.3 * ((PRIOR_SIZE - 1) * 0.5 - 1) + 0.8;
        for (int i = 0; i < PRIOR_SIZE; ++i) {
            const double d = (double)i - (PRIOR_SIZE - 1) / 2;
            w[i] = exp(-(d * d) / (2 * sigma * sigma)) * ((i == 0 || i == PRIOR_SIZE - 1) ? 2.0 : 1.0);
        }
        rc = hipMemcpyToSymbol(HIP_SYMBOL(c_prior_w), w, sizeof(w)) == hipSuccess ? SUO_OK : SUO_ERR_HIP;
    });
    return rc;
}

// upper-left corner of the paste window of keypoint (u, v) in NDC on an S x S map; false if the channel stays zero
__device__ __forceinline__ bool prior_corner(float uf, float vf, bool on, int S, int& ulx, int& uly) {
    if (!on || !isfinite(uf) || !isfinite(vf)) return false;
    const double u = fmin(fmax((double)uf, -1.0), 1.0) * S / 2 + S / 2 - 0.5;
    const double v = S - 0.5 - (fmin(fmax((double)vf, -1.0), 1.0) * S / 2 + S / 2);
    ulx = (int)rint(u) - PRIOR_HALF;           // rint: round half to even, like Python's round()
    uly = (int)rint(v) - PRIOR_HALF;
    return true;
}
__device__ __forceinline__ float prior_value(int x, int y, int ulx, int uly) {
    const int dx = x - ulx, dy = y - uly;
    if (dx < 0 || dy < 0 || dx >= 2 * PRIOR_HALF || dy >= 2 * PRIOR_HALF) return 0.f;
    return (float)(c_prior_w[dy] * c_prior_w[dx]);
}

// dense [L,41,256,256] rendering (the tensor the reference builds) -- parity / interoperability entry point
__global__ __launch_bounds__(256) void render_priors_kernel(const float* __restrict__ uv, const uint8_t* __restrict__ mask, float* __restrict__ out) {
    const int lk = blockIdx.y;                                   // crop * 41 + keypoint
    const int p = blockIdx.x * 256 + threadIdx.x;
    int ulx = 0, uly = 0;
    const bool on = prior_corner(uv[lk * 2], uv[lk * 2 + 1], mask[lk] != 0, CROP, ulx, uly);
    out[(size_t)lk * CROP * CROP + p] = on ? prior_value(p & (CROP - 1), p >> 8, ulx, uly) : 0.f;
}

int launch_render_priors(const float* uv, const uint8_t* mask, int L, float* out, hipStream_t s) {
    if (L <= 0 || !uv || !mask || !out) { suo_set_error("render_priors: bad argument"); return SUO_ERR_ARG; }
    int rc = ensure_prior_table();
    if (rc) { suo_set_error("render_priors: table upload failed"); return rc; }
    hipLaunchKernelGGL(render_priors_kernel, dim3(CROP * CROP / 256, L * NUM_KP), dim3(256), 0, s, uv, mask, out);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

template <int FMT, int OUT_C>
__global__ void roi_align_concat_kernel(const void* __restrict__ img0, int H, int W, const float* __restrict__ boxes,
                                        const int* __restrict__ box_img, const float* __restrict__ priors,
                                        const float* __restrict__ prior_uv, const uint8_t* __restrict__ prior_mask, float* __restrict__ out) {
    const int l = blockIdx.y;
    __shared__ int s_ul[NUM_KP][2];
    __shared__ unsigned char s_on[NUM_KP];
    if (OUT_C >= 3 + NUM_KP && prior_uv) {                       // block-uniform: the 41 paste windows of this crop
        if (threadIdx.x < NUM_KP) {
            const int k = threadIdx.x, lk = l * NUM_KP + k;
            int ulx = 0, uly = 0;
            s_on[k] = prior_corner(prior_uv[lk * 2], prior_uv[lk * 2 + 1], prior_mask[lk] != 0, CROP, ulx, uly) ? 1 : 0;
            s_ul[k][0] = ulx; s_ul[k][1] = uly;
        }
        __syncthreads();
    }
    // several frames per launch: crop l samples image box_img[l] of a contiguous [B,H,W,3] (or [B,3,H,W]) stack
    const size_t img_elems = (size_t)H * W * 3;
    const void* img = box_img ? (FMT == 0 ? (const void*)((const uint8_t*)img0 + box_img[l] * img_elems)
                                          : (const void*)((const float*)img0 + box_img[l] * img_elems)) : img0;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;   // 0 .. 65535
    const int ph = p >> 8, pw = p & 255;
    float smp[3];
    roi_sample<FMT>(img, H, W, boxes[l * 4 + 0], boxes[l * 4 + 1], boxes[l * 4 + 2], boxes[l * 4 + 3], ph, pw, smp);      // csrc/roi_sample.h
    float* o = out + ((size_t)l * CROP * CROP + p) * OUT_C;
    f32x4 v[OUT_C / 4];
#pragma unroll
    for (int i = 0; i < OUT_C / 4; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    v[0][0] = smp[0];
    v[0][1] = smp[1];
    v[0][2] = smp[2];
    if (OUT_C >= 3 + NUM_KP && priors) {
        const float* pr = priors + (size_t)l * NUM_KP * CROP * CROP + p;
#pragma unroll
        for (int k = 0; k < NUM_KP; ++k) v[(3 + k) >> 2][(3 + k) & 3] = pr[(size_t)k * CROP * CROP];
    } else if (OUT_C >= 3 + NUM_KP && prior_uv) {
#pragma unroll
        for (int k = 0; k < NUM_KP; ++k)
            v[(3 + k) >> 2][(3 + k) & 3] = s_on[k] ? prior_value(pw, ph, s_ul[k][0], s_ul[k][1]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < OUT_C / 4; ++i) ((f32x4*)o)[i] = v[i];
}

// out_c = IN_C: [L,256,256,48] = image + priors (zeros when priors == NULL) + pad;  out_c = IMG_C: [L,256,256,16] = image + pad,
// for the prior-less single-view pass whose 41 prior channels are structural zeros (lib/object_slam.py:1094-1097)
// priors: dense [L,41,256,256] heat-maps, OR prior_uv [L,41,2] + prior_mask [L,41]: keypoints whose heat-maps are rendered here
int launch_roi_align_concat(const void* img, int fmt, int H, int W, const float* boxes, const int* box_img, int L, int out_c,
                            const float* priors, const float* prior_uv, const uint8_t* prior_mask, float* out, hipStream_t s) {
    if (L <= 0 || H <= 1 || W <= 1) { suo_set_error("roi_align: bad shape"); return SUO_ERR_ARG; }
    if ((out_c != IN_C && out_c != IMG_C) || (out_c == IMG_C && (priors || prior_uv))) { suo_set_error("roi_align: bad channel count %d", out_c); return SUO_ERR_ARG; }
    if ((priors && prior_uv) || (prior_uv && !prior_mask)) { suo_set_error("roi_align: give dense priors OR prior keypoints + mask"); return SUO_ERR_ARG; }
    if (prior_uv) { int rc = ensure_prior_table(); if (rc) { suo_set_error("roi_align: prior table upload failed"); return rc; } }
    const dim3 grid(CROP * CROP / 256, L), block(256);
    if (fmt == 0 && out_c == IN_C) hipLaunchKernelGGL((roi_align_concat_kernel<0, IN_C>), grid, block, 0, s, img, H, W, boxes, box_img, priors, prior_uv, prior_mask, out);
    else if (fmt == 0) hipLaunchKernelGGL((roi_align_concat_kernel<0, IMG_C>), grid, block, 0, s, img, H, W, boxes, box_img, priors, prior_uv, prior_mask, out);
    else if (fmt == 1 && out_c == IN_C) hipLaunchKernelGGL((roi_align_concat_kernel<1, IN_C>), grid, block, 0, s, img, H, W, boxes, box_img, priors, prior_uv, prior_mask, out);
    else if (fmt == 1) hipLaunchKernelGGL((roi_align_concat_kernel<1, IMG_C>), grid, block, 0, s, img, H, W, boxes, box_img, priors, prior_uv, prior_mask, out);
    else { suo_set_error("roi_align: unknown image format %d", fmt); return SUO_ERR_ARG; }
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

// ------------------------------------------------------------------------------------------------
// Heat-map decode: one wave per (crop, keypoint) 64x64 heat-map held entirely in registers
// (16 float4 per lane).  Three in-register passes: max; exp / sum / first moments; centred second
// moments (two-pass covariance exactly like post_process_kp, no E[x^2]-mu^2 cancellation).
// Axis convention: u = sum p * r[row],  v = sum p * (-r[col]),  r[i] = (i+0.5)/32 - 1.
// Optional diagnostics (NULL = not computed): argmax_idx = flat index h*64+w of the FIRST maximum of the heat-map, as torch.argmax
// over the flattened map returns it (the reference itself has no hard argmax; this one is the bit-exact integer
// keypoint output north_star asks for); prob = the soft-max itself, the reference's ret["prob"] (pkpnet.py:111).
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ logits, int n_maps, float* __restrict__ uv,
                                                     float* __restrict__ cov, float* __restrict__ mean_logit,
                                                     int* __restrict__ argmax_idx, float* __restrict__ prob) {
    const int lane = threadIdx.x & 63;
    const int map = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (map >= n_maps) return;
    const f32x4* src = (const f32x4*)(logits + (size_t)map * HEAT * HEAT);
    f32x4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = src[j * 64 + lane];   // element index e = (j*64+lane)*4 + t
    float mx = -INFINITY, raw = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) { mx = fmaxf(mx, v[j][t]); raw += v[j][t]; }
    mx = wave_max(mx);
    raw = wave_sum(raw);
    if (argmax_idx) {                     // (uniform branch) smallest flat index holding the maximum: a wave-wide min over the lanes' own
        int idx = HEAT * HEAT;
#pragma unroll
        for (int j = 15; j >= 0; --j)
#pragma unroll
            for (int t = 3; t >= 0; --t) idx = (v[j][t] == mx) ? (j * 64 + lane) * 4 + t : idx;
        idx = -(int)wave_max((float)-idx);          // indices < 2^24 are exact in fp32
        if (lane == 0) argmax_idx[map] = idx;
    }
    // row = e / 64 = j*4 + lane/16 ; col = (lane & 15)*4 + t
    const float colbase = (float)((lane & 15) * 4);
    float s0 = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float rx = ((float)(j * 4 + (lane >> 4)) + 0.5f) / 32.0f - 1.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float e = expf(v[j][t] - mx);
            v[j][t] = e;
            const float ry = -((colbase + (float)t + 0.5f) / 32.0f - 1.0f);
            s0 += e;
            sx = fmaf(e, rx, sx);
            sy = fmaf(e, ry, sy);
        }
    }
    s0 = wave_sum(s0);
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    const float inv = 1.0f / s0;
    const float mu_x = sx * inv, mu_y = sy * inv;
    if (prob) {
        f32x4* dst = (f32x4*)(prob + (size_t)map * HEAT * HEAT);
#pragma unroll
        for (int j = 0; j < 16; ++j) dst[j * 64 + lane] = v[j] * inv;
    }
    float cxx = 0.f, cxy = 0.f, cyy = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float dx = (((float)(j * 4 + (lane >> 4)) + 0.5f) / 32.0f - 1.0f) - mu_x;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float dy = -((colbase + (float)t + 0.5f) / 32.0f - 1.0f) - mu_y;
            const float e = v[j][t];
            cxx = fmaf(e * dx, dx, cxx);
            cxy = fmaf(e * dx, dy, cxy);
            cyy = fmaf(e * dy, dy, cyy);
        }
    }
    cxx = wave_sum(cxx) * inv;
    cxy = wave_sum(cxy) * inv;
    cyy = wave_sum(cyy) * inv;
    if (lane == 0) {
        uv[map * 2 + 0] = mu_x;
        uv[map * 2 + 1] = mu_y;
        cov[map * 4 + 0] = cxx;
        cov[map * 4 + 1] = cxy;
        cov[map * 4 + 2] = cxy;
        cov[map * 4 + 3] = cyy;
        mean_logit[map] = raw * (1.0f / (HEAT * HEAT));
    }
}

int launch_decode(const float* logits, int L, float* uv, float* cov, float* mean_logit, int* argmax_idx, float* prob, hipStream_t s) {
    if (L <= 0) { suo_set_error("decode: L<=0"); return SUO_ERR_ARG; }
    const int n_maps = L * NUM_KP;
    hipLaunchKernelGGL(decode_kernel, dim3((n_maps + 3) / 4), dim3(256), 0, s, logits, n_maps, uv, cov, mean_logit, argmax_idx, prob);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

// ------------------------------------------------------------------------------------------------
// validity head: sigmoid(W * relu(mean_logit) + b); one wave per crop, lane = output keypoint
__global__ void classifier_kernel(const float* __restrict__ mean_logit, const float* __restrict__ Wc,
                                  const float* __restrict__ bc, float* __restrict__ kp_logit, float* __restrict__ kp_prob) {
    const int l = blockIdx.x, k = threadIdx.x;
    __shared__ float m[NUM_KP];
    if (k < NUM_KP) m[k] = fmaxf(mean_logit[l * NUM_KP + k], 0.f);
    __syncthreads();
    if (k < NUM_KP) {
        float a = 0.f;
        for (int j = 0; j < NUM_KP; ++j) a = fmaf(Wc[k * NUM_KP + j], m[j], a);
        a += bc[k];
        if (kp_logit) kp_logit[l * NUM_KP + k] = a;
        kp_prob[l * NUM_KP + k] = 1.0f / (1.0f + expf(-a));
    }
}

int launch_classifier(const float* mean_logit, const float* Wc, const float* bc, int L, float* kp_logit, float* kp_prob,
                      hipStream_t s) {
    hipLaunchKernelGGL(classifier_kernel, dim3(L), dim3(64), 0, s, mean_logit, Wc, bc, kp_logit, kp_prob);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

// ------------------------------------------------------------------------------------------------
// object_slam.py:1100-1115 on device: boolean keypoint masks (bit-exact target)
__global__ void kp_masks_kernel(const float* __restrict__ uv, const float* __restrict__ cov, const float* __restrict__ kp_prob,
                                const uint8_t* __restrict__ model_mask, int n, float bbox_thresh, float two_var,
                                uint8_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = uv[i * 2], v = uv[i * 2 + 1];
    bool m = (kp_prob[i] > 0.3f) && (model_mask ? model_mask[i] != 0 : true);
    m = m && (fminf(u, v) > -bbox_thresh) && (fmaxf(u, v) < bbox_thresh);
    const float sx = sqrtf(cov[i * 4 + 0]), sy = sqrtf(cov[i * 4 + 3]);
    m = m && (sx < two_var) && (sy < two_var);
    out[i] = m ? 1 : 0;
}

int launch_kp_masks(const float* uv, const float* cov, const float* kp_prob, const uint8_t* model_mask, int L,
                    float bbox_thresh, float kp_var_thresh, uint8_t* out_mask, hipStream_t s) {
    const int n = L * NUM_KP;
    hipLaunchKernelGGL(kp_masks_kernel, dim3((n + 255) / 256), dim3(256), 0, s, uv, cov, kp_prob, model_mask, n,
                       bbox_thresh, 2.0f * kp_var_thresh, out_mask);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

// ------------------------------------------------------------------------------------------------
// Host -> device upload by a kernel: the source is PINNED host memory (hipHostMalloc / torch pin_memory: mapped into the device's
// address space), read with coalesced 16-byte loads over the host link and written to HBM.  Stream-ordered like any kernel.  Why
// not hipMemcpyAsync: on this stack an asynchronous H2D copy queued in front of kernels makes the NEXT host-side wait on that
// stream take 10-20 ms (measured, tools/time_frame_chain.py: copy + 2.5 ms of kernels + synchronize = 13-25 ms, the kernels
// themselves unchanged) -- fatal at one frame per call.  0.92 MB (one 640x480 frame): ~25 us.
__global__ __launch_bounds__(256) void upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, const uint8_t* __restrict__ src8,
                                                     uint8_t* __restrict__ dst8, size_t tail0, size_t bytes) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
    for (size_t i = tail0 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < bytes; i += stride) dst8[i] = src8[i];
}

int launch_upload(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s) {
    if (bytes == 0) return SUO_OK;
    if (((uintptr_t)dst_dev | (uintptr_t)src_host) & 15) { suo_set_error("upload: pointers must be 16-byte aligned"); return SUO_ERR_ARG; }
    const size_t n16 = bytes / 16;
    const int blocks = (int)std::min<size_t>(1024, (n16 + 255) / 256 + 1);
    hipLaunchKernelGGL(upload_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)src_host, (uint4*)dst_dev, n16, (const uint8_t*)src_host,
                       (uint8_t*)dst_dev, n16 * 16, bytes);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo

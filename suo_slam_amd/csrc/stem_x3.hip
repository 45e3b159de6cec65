// The stem of the prior-less pass -- RoIAlign of the frame + conv 7x7 / stride 2 over the 3 image channels + BN + ReLU (lib/models/pkpnet.py:93,
// lib/models/hg.py:67-69,96-98; priors are zeros in every single-view pass, lib/object_slam.py:1094-1097) -- in ONE launch on the bf16 matrix
// pipe at fp32 accuracy (3-way operand split, csrc/bf16x3.h).  It replaces roi_align_concat_kernel<*, 4> + convk_kernel<7,2,4,...>: the staged
// [L,256,256,4] tensor (268 MB at 256 crops) is never written, and the 147-term dot products leave the fp32 pipe (864 us at 256 crops).
//
// Workgroup = 8 x 16 output pixels of one crop x all 64 channels.  The 21 x 37 input pixels it needs are SAMPLED from the frame while staging
// (csrc/roi_sample.h: the crop kernel's own arithmetic, so the values are the ones roi_align_concat_kernel would have written; outside the crop:
// the convolution's zero padding), split into three bf16 terms and stored de-interleaved by column parity, 4 channels (3 + a zero) per entry:
// with stride 2 the seven taps of an output column are entries [ox, ox + 3] of the EVEN buffer (kx = 0, 2, 4, 6) and [ox, ox + 2] of the ODD
// one (kx = 1, 3, 5) -- contiguous, 8-byte aligned runs.  K is therefore walked as 7 rows x {even, odd} = 14 steps of 16 (21 of every 32
// products real); A fragments are two ds_read_b64 per plane, B fragments host-split planes straight from L2.  Four waves as 2 x 2: wave
// (wm, wn) owns 64 pixels x 32 channels.  Epilogue through an LDS patch, 16-byte stores.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bf16x3.h"
#include "f16x2.h"
#include "buffer_ops.h"
#include "roi_sample.h"
#include "suo_internal.h"

namespace suo {

typedef float sx_f32x4 __attribute__((ext_vector_type(4)));
typedef float sx_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned sx_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned sx_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sx_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sx_f16x8 __attribute__((ext_vector_type(8)));

constexpr int SX_STEPS = 14, SX_N = 64;

// host: W[64][Cw][7][7] (the first 3 input channels; times out_scale[n]: bn1 folded) -> [step = ky * 2 + h][n-tile][plane][lane][8 bf16],
//   k = 8 (lane >> 5) + e  ->  entry j = k / 4, channel c = k % 4, tap kx = 2 j + h   (zero for c == 3 and for the odd buffer's 4th entry)
void pack_stem_weight_bf16x3(const float* W, int Cw, const float* out_scale, uint16_t* out) {
    memset(out, 0, (size_t)SX_STEPS * 2 * 3 * 64 * 8 * sizeof(uint16_t));
    for (int ky = 0; ky < 7; ++ky)
        for (int h = 0; h < 2; ++h)
            for (int n = 0; n < SX_N; ++n)
                for (int k = 0; k < 16; ++k) {
                    const int j = k / 4, c = k % 4, kx = 2 * j + h;
                    if (c == 3 || kx > 6) continue;
                    const int s = ky * 2 + h, nb = n / 32, lane = (k / 8) * 32 + (n % 32), e = k % 8;
                    const float sc = out_scale ? out_scale[n] : 1.f;
                    uint16_t t[3];
                    s3_split_host(W[(((size_t)n * Cw + c) * 7 + ky) * 7 + kx] * sc, t);
                    for (int p = 0; p < 3; ++p) out[((((size_t)(s * 2 + nb) * 3 + p) * 64) + lane) * 8 + e] = t[p];
                }
}

// host, two fp16 planes (csrc/f16x2.h): the same layout with 2 planes, output channel n times 2^t_n; oscale_out[n] = 2^-(t_n + S2_XSHIFT)
void pack_stem_weight_f16x2(const float* W, int Cw, const float* out_scale, uint16_t* out, float* oscale_out) {
    memset(out, 0, (size_t)SX_STEPS * 2 * 2 * 64 * 8 * sizeof(uint16_t));
    for (int n = 0; n < SX_N; ++n) {
        const float sc = out_scale ? out_scale[n] : 1.f;
        float mx = 0.f;
        for (int c = 0; c < 3; ++c)
            for (int t = 0; t < 49; ++t) mx = fmaxf(mx, fabsf(W[((size_t)n * Cw + c) * 49 + t] * sc));
        const int sh = s2_row_shift(mx);
        oscale_out[n] = ldexpf(1.f, -(sh + S2_XSHIFT));
        for (int ky = 0; ky < 7; ++ky)
            for (int h = 0; h < 2; ++h)
                for (int k = 0; k < 16; ++k) {
                    const int j = k / 4, c = k % 4, kx = 2 * j + h;
                    if (c == 3 || kx > 6) continue;
                    const int st = ky * 2 + h, nb = n / 32, lane = (k / 8) * 32 + (n % 32), e = k % 8;
                    uint16_t t[2];
                    s2_split_host(ldexpf(W[(((size_t)n * Cw + c) * 7 + ky) * 7 + kx] * sc, sh), t);
                    for (int p = 0; p < 2; ++p) out[((((size_t)(st * 2 + nb) * 2 + p) * 64) + lane) * 8 + e] = t[p];
                }
    }
}

#ifdef SUO_SX_PROF      // tools/build_variant.sh sxprof -DSUO_SX_PROF: per-workgroup phase stamps (s_memtime) + hardware ids, dumped by the launcher (tools/stem_phases.py)
__device__ long long sx_prof[65536 * 6];
#define SX_T(i) do { if (tid == 0 && blockIdx.x < 65536) sx_prof[blockIdx.x * 6 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define SX_T(i) do { } while (0)
#endif

__device__ __forceinline__ int sx_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

struct StemArgs {
    const void* img; int fmt, H, W;                     // frame(s): FMT 0 uint8 HWC, FMT 1 float32 CHW
    const float* boxes; const int* box_img; int L;      // [L,4] xyxy; optional frame index per crop
    const uint16_t* Wx; const float* bias;              // pack_stem_weight_bf16x3 planes, folded bias [64]
    float* out;                                         // [L,128,128,64]
    const float* osc; unsigned* range_flag;             // NP = 2: per-channel factors 2^-(t_n + S2_XSHIFT), range-guard flag (a float frame may hold anything)
    // NEXT (NP = 2 only): the first Residual block's conv1 on the tile while it is in the CU -- relu(bn(out)) x W1 (64 -> 64, BatchNorm folded) + b1, ReLU -> n_out [L,128,128,64]
    const float* n_scale; const float* n_shift; const uint16_t* n_W1; const float* n_osc1; const float* n_b1; float* n_out;
};

// NP = operand planes: 3 = three bf16 terms (six MFMAs per product block), 2 = two fp16 terms (three; csrc/f16x2.h: samples times 2^S2_XSHIFT, weight rows times 2^t_n)
// NEXT: the stem's output has two readers, r1's conv1 (1x1, 64 -> 64 behind a BatchNorm + ReLU) and r1's skip convolution; conv1 is computed here, on the tile in the epilogue
// patch (one 32-pixel m-tile per wave, K = 64 in four k-steps, both 32-channel n-tiles) -- its launch (a 1.07 GB read at 256 crops) is gone.  Same products in the same order as
// gemm_bf16x3_kernel<NP = 2> on the stored tensor: bit-identical (tests/test_gpu_stem.py).
template <int FMT, int NP = 3, bool NEXT = false>
__global__ __launch_bounds__(256) void stem_x3_kernel(const StemArgs a) {
    static_assert(!NEXT || NP == 2, "the next block's conv1 rides on the fp16 form");
    constexpr int TH = 8, TW = 16, IR = 2 * TH + 5, IC = 2 * TW + 5;          // 21 x 37 input pixels
    constexpr int HALF_B = 160, ROW_B = 2 * HALF_B, PLANE_B = IR * ROW_B;    // bytes: 20 entries of 8 per half row
    constexpr int PP = SX_N + 4;                                             // epilogue patch pitch (floats)
    constexpr int LDS_B = TH * TW * PP * 4 > NP * PLANE_B ? TH * TW * PP * 4 : NP * PLANE_B;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_B];       // the staged planes, then (dead by then) the epilogue patch: 34 KB, 4 workgroups per CU
    unsigned char* A3 = lds;
    float* P = (float*)lds;
    __shared__ float lut[256];                                              // u / 255.0f for the uint8 frame (csrc/roi_sample.h)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);          // XCD-aware tile order (csrc/conv.hip)
    const int l = bid >> 7, t = bid & 127;                                   // 16 x 8 tiles per crop
    const int oy0 = (t >> 3) * TH, ox0 = (t & 7) * TW;
    SX_T(0);
#ifdef SUO_SX_PROF
    if (tid == 0 && blockIdx.x < 65536) sx_prof[blockIdx.x * 6 + 5] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);
#endif
    const __amdgpu_buffer_rsrc_t w_srd = make_srd(a.Wx, (size_t)SX_STEPS * 2 * NP * 1024);
    const int wv = lane * 16;
    constexpr int R = 4;
    sx_u32x4 ring[R][NP];
    float gmax = 0.f;                                                       // NP = 2: largest scaled sample this lane staged (range guard)
    auto loadw = [&](int s, sx_u32x4 (&b)[NP]) {
        const int k = s < SX_STEPS ? s : SX_STEPS - 1;
#pragma unroll
        for (int p = 0; p < NP; ++p) b[p] = __builtin_bit_cast(sx_u32x4, buf_load(w_srd, wv + p * 1024, (k * 2 + wn) * NP * 1024));
    };
    // ---- stage: 21 rows x 2 halves x 20 entries; slot -> input pixel (r, xl = 2 j + h) ---------------------------------------------------
    if (FMT == 0) {
        roi_fill_lut(lut, tid);
        __syncthreads();
    }
    {
        const size_t img_elems = (size_t)a.H * a.W * 3;
        const void* img = a.box_img ? (FMT == 0 ? (const void*)((const uint8_t*)a.img + a.box_img[l] * img_elems)
                                                : (const void*)((const float*)a.img + a.box_img[l] * img_elems)) : a.img;
        const float x1 = a.boxes[l * 4 + 0], y1 = a.boxes[l * 4 + 1], x2 = a.boxes[l * 4 + 2], y2 = a.boxes[l * 4 + 3];
        // slot -> (row r, half h, entry j): input pixel (r, xl = 2 j + h) of the tile, crop pixel (py, px); entries beyond the 37th column and
        // pixels outside the crop (the convolution's zero padding) are written as zeros
        auto slot_geom = [&](int slot, int& r, int& h, int& j, int& py, int& px) {
            r = slot / 40;
            const int q = slot - r * 40;
            h = q / 20; j = q - h * 20;
            const int xl = 2 * j + h;
            py = 2 * oy0 - 3 + r; px = 2 * ox0 - 3 + xl;
            return xl < IC && py >= 0 && py < CROP && px >= 0 && px < CROP;
        };
        auto store3 = [&](int r, int h, int j, float c0, float c1, float c2) {
            unsigned char* d = A3 + r * ROW_B + h * HALF_B + j * 8;
            if constexpr (NP == 2) {
                c0 *= S2_XSCALE; c1 *= S2_XSCALE; c2 *= S2_XSCALE;
                gmax = fmaxf(s2_track(gmax, c0, c1), fabsf(c2));
                const unsigned h0 = s2_pack_rn(c0, c1), h1 = s2_pack_rn(c2, 0.f);
                *(sx_u32x2*)d = sx_u32x2{h0, h1};
                *(sx_u32x2*)(d + PLANE_B) = sx_u32x2{s2_lo_pack(c0, c1, h0), s2_lo_pack(c2, 0.f, h1)};
                return;
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned q0 = s3_pack_rn(c0, c1), q1 = s3_pack_rn(c2, 0.f);
                *(sx_u32x2*)(d + p * PLANE_B) = sx_u32x2{q0, q1};
                if (p < 2) { c0 -= s3_lo(q0); c1 -= s3_hi(q0); c2 -= s3_lo(q1); }
            }
        };
        constexpr int NSLOT = IR * 40, NIT = (NSLOT + 255) / 256;
        const RoiBins bins = roi_bins(x1, y1, x2, y2);
        if (bins.gh == 1 && bins.gw == 1) {
            // boxes of <= 256 px (one sample per bin): the taps of ALL of this thread's slots are requested before the first is consumed --
            // one memory latency per tile instead of one per slot
            RoiTaps<FMT> tp[NIT];
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                int r, h, j, py, px;
                const int slot = tid + 256 * i;
                tp[i].in = false;
                if (slot < NSLOT && slot_geom(slot, r, h, j, py, px)) {
                    float y, x;
                    roi_single_sample_pos(bins, py, px, y, x);
                    roi_taps_issue<FMT>(img, a.H, a.W, y, x, tp[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                int r, h, j, py, px;
                const int slot = tid + 256 * i;
                if (slot < NSLOT) {
                    slot_geom(slot, r, h, j, py, px);
                    float v[3] = {0.f, 0.f, 0.f};
                    roi_taps_value<FMT, FMT == 0>(tp[i], lut, v);
                    store3(r, h, j, v[0], v[1], v[2]);
                }
            }
        } else {
            for (int slot = tid; slot < NSLOT; slot += 256) {
                int r, h, j, py, px;
                const bool inside = slot_geom(slot, r, h, j, py, px);
                float v[3] = {0.f, 0.f, 0.f};
                if (inside) roi_sample<FMT, FMT == 0>(img, a.H, a.W, x1, y1, x2, y2, py, px, v, lut);
                store3(r, h, j, v[0], v[1], v[2]);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < R - 1; ++s) loadw(s, ring[s]);          // (after the staging: its tap registers are free again; the barrier hides the L2 latency)
    __syncthreads();
    SX_T(1);

    sx_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    constexpr int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};      // the six cross terms, smallest first
    constexpr int UI[3] = {0, 1, 0}, UJ[3] = {1, 0, 0};                       // NP = 2: hi lo, lo hi, hi hi
    // this lane's pixels: m-tile i of the wave = tile pixels 64 wm + 32 i + (lane & 31)
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = 64 * wm + 32 * i + (lane & 31), oyl = p >> 4, oxl = p & 15;
        abase[i] = (2 * oyl) * ROW_B + oxl * 8 + (lane >> 5) * 16;
    }
#pragma unroll
    for (int s = 0; s < SX_STEPS; ++s) {                        // s = ky * 2 + h
        loadw(s + R - 1, ring[(s + R - 1) % R]);
        sx_bf16x8 af[2][NP];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const unsigned char* src = A3 + p * PLANE_B + abase[i] + (s >> 1) * ROW_B + (s & 1) * HALF_B;
                const sx_u32x2 lo = *(const sx_u32x2*)src, hi = *(const sx_u32x2*)(src + 8);      // (8-byte aligned runs: two ds_read_b64)
                af[i][p] = __builtin_bit_cast(sx_bf16x8, sx_u32x4{lo[0], lo[1], hi[0], hi[1]});
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < (NP == 2 ? 3 : 6); ++tt)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (NP == 2)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sx_f16x8, af[i][UI[tt]]), __builtin_bit_cast(sx_f16x8, ring[s % R][UJ[tt]]), acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][TI[tt] % NP], __builtin_bit_cast(sx_bf16x8, ring[s % R][TJ[tt] % NP]), acc[i], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    // relu(acc + bias) -> patch [pixel][64], then 16-byte stores
    SX_T(2);
    __syncthreads();                                            // (every wave has read its last A fragment: the patch overwrites the planes)
    {
        const int col = 32 * wn + (lane & 31);
        const float b = a.bias[col];
        const float oc = NP == 2 ? a.osc[col] : 1.f;                        // back to scale: an exact power of two per channel
        if constexpr (NP == 2) s2_raise(a.range_flag, gmax);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) P[(64 * wm + 32 * i + sx_acc_row(r, lane)) * PP + col] = NP == 2 ? fmaxf(fmaf(acc[i][r], oc, b), 0.f) : fmaxf(acc[i][r] + b, 0.f);
    }
    __syncthreads();
    SX_T(3);
    {
        float* o = a.out + (size_t)l * 128 * 128 * SX_N;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = tid + 256 * k, p = idx >> 4, q = idx & 15;
            const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
            *(sx_f32x4*)(o + ((size_t)oy * 128 + ox) * SX_N + 4 * q) = *(const sx_f32x4*)&P[p * PP + 4 * q];
        }
    }
    SX_T(4);
    if constexpr (NEXT) {
        // A fragments straight out of the patch (fp32 -> prologue -> split in registers); wave w owns pixels [32 w, 32 w + 32) and all 64 output channels
        constexpr int UI[3] = {0, 1, 0}, UJ[3] = {1, 0, 0};
        const __amdgpu_buffer_rsrc_t n_srd = make_srd(a.n_W1, (size_t)64 * 64 * 2 * sizeof(uint16_t));
        sx_u32x4 nb[4][2][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) nb[ks][n][pl] = __builtin_bit_cast(sx_u32x4, buf_load(n_srd, wv + pl * 1024, ((ks * 2 + n) * 2) * 1024));
        sx_f32x16 acc2[2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[n][r] = 0.f;
        float gmax2 = 0.f;
        const float* prow = P + (32 * w + (lane & 31)) * PP + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c0 = 16 * ks + 8 * (lane >> 5);
            float v[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const sx_f32x4 x = *(const sx_f32x4*)(prow + 16 * ks + 4 * q);
                const sx_f32x4 sc = *(const sx_f32x4*)(a.n_scale + c0 + 4 * q), sh = *(const sx_f32x4*)(a.n_shift + c0 + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * q + j] = fmaxf(fmaf(x[j], sc[j] * S2_XSCALE, sh[j] * S2_XSCALE), 0.f);      // (the GEMM's prologue, scale folded in the same way)
            }
            sx_u32x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gmax2 = s2_track(gmax2, v[2 * j], v[2 * j + 1]);
                hi[j] = s2_pack_rn(v[2 * j], v[2 * j + 1]);
                lo[j] = s2_lo_pack(v[2 * j], v[2 * j + 1], hi[j]);
            }
            const sx_f16x8 af[2] = {__builtin_bit_cast(sx_f16x8, hi), __builtin_bit_cast(sx_f16x8, lo)};
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[UI[tt]], __builtin_bit_cast(sx_f16x8, nb[ks][n][UJ[tt]]), acc2[n], 0, 0, 0);
        }
        s2_raise(a.range_flag, gmax2);
        __syncthreads();                                        // every wave has read its rows of the patch (and the first output's stores have read theirs)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = 32 * n + (lane & 31);
            const float oc = a.n_osc1[col], b = a.n_b1[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) P[(32 * w + sx_acc_row(r, lane)) * PP + col] = fmaxf(fmaf(acc2[n][r], oc, b), 0.f);
        }
        __syncthreads();
        float* o2 = a.n_out + (size_t)l * 128 * 128 * SX_N;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = tid + 256 * k, p = idx >> 4, q = idx & 15;
            const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
            *(sx_f32x4*)(o2 + ((size_t)oy * 128 + ox) * SX_N + 4 * q) = *(const sx_f32x4*)&P[p * PP + 4 * q];
        }
    }
}

// frame(s) + boxes -> stem output [L,128,128,64] (prior-less pass); Wx = pack_stem_weight_bf16x3, bias = bn1-folded conv bias
int launch_stem_x3(const void* img, int fmt, int H, int W, const float* boxes, const int* box_img, int L, const uint16_t* Wx, const float* bias,
                   float* out, hipStream_t s, const float* osc, unsigned* range_flag, const StemNext* next) {
    if (L <= 0 || H <= 1 || W <= 1 || !img || !boxes || !Wx || !bias || !out || ((osc == nullptr) != (range_flag == nullptr))) { suo_set_error("stem_x3: bad arguments"); return SUO_ERR_ARG; }
    if (next && (!osc || !next->scale || !next->shift || !next->W1 || !next->osc1 || !next->b1 || !next->out)) { suo_set_error("stem_x3: the next block's conv1 needs the fp16 form and all of its operands"); return SUO_ERR_ARG; }
    StemArgs a = {img, fmt, H, W, boxes, box_img, L, Wx, bias, out, osc, range_flag, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (next) { a.n_scale = next->scale; a.n_shift = next->shift; a.n_W1 = next->W1; a.n_osc1 = next->osc1; a.n_b1 = next->b1; a.n_out = next->out; }
    if (fmt != 0 && fmt != 1) { suo_set_error("stem_x3: unknown image format %d", fmt); return SUO_ERR_ARG; }
    if (osc && next) {
        if (fmt == 0) hipLaunchKernelGGL((stem_x3_kernel<0, 2, true>), dim3(L * 128), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((stem_x3_kernel<1, 2, true>), dim3(L * 128), dim3(256), 0, s, a);
    } else if (osc) {                                   // Wx = pack_stem_weight_f16x2 planes
        if (fmt == 0) hipLaunchKernelGGL((stem_x3_kernel<0, 2>), dim3(L * 128), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((stem_x3_kernel<1, 2>), dim3(L * 128), dim3(256), 0, s, a);
    } else if (fmt == 0) hipLaunchKernelGGL((stem_x3_kernel<0, 3>), dim3(L * 128), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((stem_x3_kernel<1, 3>), dim3(L * 128), dim3(256), 0, s, a);
    SUO_HIP_CHECK(hipGetLastError());
#ifdef SUO_SX_PROF
    {
        static int n_dump = 0;
        if (getenv("SUO_SX_PROF_OUT") && ++n_dump == 8) {
            SUO_HIP_CHECK(hipStreamSynchronize(s));
            const int n = L * 128 < 65536 ? L * 128 : 65536;
            long long* h = (long long*)malloc((size_t)n * 6 * 8);
            SUO_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(sx_prof), (size_t)n * 6 * 8));
            FILE* f = fopen(getenv("SUO_SX_PROF_OUT"), "wb");
            if (f) { fwrite(h, 8, (size_t)n * 6, f); fclose(f); }
            free(h);
        }
    }
#endif
    return SUO_OK;
}

}  // namespace suo

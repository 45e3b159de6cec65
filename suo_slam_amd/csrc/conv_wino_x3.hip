// The Winograd F(2x2,3x3) convolution of csrc/conv_wino.hip with its element-wise products on the BF16 matrix pipe at fp32 accuracy (3-way
// operand split, 6 of 9 cross terms, fp32 accumulate: see csrc/gemm_bf16x3.hip for the arithmetic) -- what the network launches for the
// Residual blocks' 3x3 convolutions and fused tails (csrc/net.hip; SUO_WINO_BF16X3=0 keeps the fp32-pipe kernels).
//
//   Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A
// Per 16-channel chunk and Winograd component the product V_xi,nu (32 tiles x 16 channels) * U_xi,nu (16 channels x 32 outputs) is
// EIGHT v_mfma_f32_32x32x2_f32 (8 x 64 = 512 pipe cycles) in the fp32 kernel and SIX v_mfma_f32_32x32x16_bf16 (6 x 32 = 192 cycles)
// here: U is split on the host (fp64 transform, rounded once to fp32, then three round-to-nearest bf16 terms: csrc/bf16x3.h), V = B^T d B is computed in fp32
// exactly as before and split by the transforming thread on its way to LDS (three bf16 planes, 49 KB).  Same workgroup tile (8 x 16
// output pixels = 32 Winograd tiles x 128 output channels, 4 waves, wave w owns output channels [32 w, 32 w + 32)), same halo
// staging, same two-step output transform (rows xi = 0, 3 accumulate straight into Z, rows 1, 2 through a scratch accumulator and 16
// vector additions per chunk), same epilogues -- the accumulator layout of the bf16 MFMA is that of the fp32 one, so the fused
// Residual tail (conv3 1x1 + skip [+ up-sampled addend]) could be taken over unchanged (TX3 = false: conv3 on the fp32 pipe); the network uses
// TX3 = true, conv3 on the bf16 pipe as well (two passes of 64 pixels: see the tail's comment).
#include <string.h>

#include "bf16x3.h"
#include "f16x2.h"
#include "buffer_ops.h"
#include "suo_internal.h"
#include "tune.h"

namespace suo {

typedef float x_f32x4 __attribute__((ext_vector_type(4)));
typedef float x_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned x_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned x_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 x_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 x_f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ x_f32x16 x_mfma32(float a, float b, x_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int x_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// (Tried and dropped, tools/bench_wino_x3.py: forcing the transform / fold additions into v_pk_add_f32 (-30 % VALU instructions) or into
// plain v_add_f32 changes nothing measurable; inline-asm VALU on MFMA results is unsafe -- hipcc's hazard recogniser does not see through it.)
__device__ __forceinline__ x_f32x4 x_sub4(x_f32x4 a, x_f32x4 b) { return a - b; }
__device__ __forceinline__ x_f32x4 x_add4(x_f32x4 a, x_f32x4 b) { return a + b; }
__device__ __forceinline__ void x_add16(x_f32x16& z, const x_f32x16& t) { z += t; }
__device__ __forceinline__ void x_sub16(x_f32x16& z, const x_f32x16& t) { z -= t; }

constexpr int X_CK = 16, X_PKH = 20, X_TH = 8, X_TW = 16, X_IH = 10, X_IW = 18, X_NPIX = X_IH * X_IW;

// host: U[n][c][comp] = (G g G^T)[xi][nu] in fp64 (BN scale folded in), rounded once to fp32, split into three bf16 terms (csrc/bf16x3.h);
//   Up3[chunk][comp][nb][plane][lane][e] = term `plane` of +-U[nb*32 + (lane&31)][chunk*16 + 8*(lane>>5) + e][comp]     (B operand of
//   v_mfma_f32_32x32x16_bf16: lane l supplies k = 8 (l >> 5) .. + 7 of column l & 31); the components of row xi = 3 are stored negated
void pack_wino_weight_bf16x3(const float* W, int N, int C, int Np, int Cp, const float* out_scale, uint16_t* out) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int NB = Np / 32;
    memset(out, 0, (size_t)Np * Cp * 16 * 3 * sizeof(uint16_t));
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float* g = W + ((size_t)n * C + c) * 9;
            const double sc = out_scale ? (double)out_scale[n] : 1.0;
            double Gg[4][3], U[4][4];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) Gg[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) U[i][j] = (Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2]) * sc;
            const int chunk = c / X_CK, cc = c % X_CK, nb = n / 32, lane = (cc / 8) * 32 + (n % 32), e = cc % 8;
            for (int comp = 0; comp < 16; ++comp) {
                uint16_t t[3];
                s3_split_host((float)(comp >= 12 ? -U[comp >> 2][comp & 3] : U[comp >> 2][comp & 3]), t);
                for (int p = 0; p < 3; ++p) out[((((size_t)(chunk * 16 + comp) * NB + nb) * 3 + p) * 64 + lane) * 8 + e] = t[p];
            }
        }
}

// host: conv3 weight W3[N2][K] (1x1) -> three bf16 terms (csrc/bf16x3.h) in B-operand order of v_mfma_f32_32x32x16_bf16:
//   W3x[(ks * NB + nb) * 3 + plane][lane][e] = term `plane` of W3[nb*32 + (lane&31)][ks*16 + 8*(lane>>5) + e]
void pack_tail_weight_bf16x3(const float* W3, int N2, int K, uint16_t* out) {
    const int NB = N2 / 32;
    for (int n = 0; n < N2; ++n)
        for (int k = 0; k < K; ++k) {
            const int ks = k / 16, kk = k % 16, lane = (kk / 8) * 32 + (n % 32), e = kk % 8, nb = n / 32;
            uint16_t t[3];
            s3_split_host(W3[(size_t)n * K + k], t);
            for (int p = 0; p < 3; ++p) out[((((size_t)(ks * NB + nb) * 3 + p) * 64 + lane) * 8) + e] = t[p];
        }
}

// host, two-term fp16 form (csrc/f16x2.h): the same U, every output channel n times 2^t_n (max over its C x 16 entries in [2^12, 2^13)), two planes;
// oscale_out[n] = 2^-(t_n + S2_XSHIFT) for the kernel's epilogue
void pack_wino_weight_f16x2(const float* W, int N, int C, int Np, int Cp, const float* out_scale, uint16_t* out, float* oscale_out) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int NB = Np / 32;
    memset(out, 0, (size_t)Np * Cp * 16 * 2 * sizeof(uint16_t));
    for (int n = 0; n < Np; ++n) oscale_out[n] = ldexpf(1.f, -S2_XSHIFT);
    float* Un = (float*)malloc((size_t)C * 16 * sizeof(float));
    for (int n = 0; n < N; ++n) {
        float mx = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* g = W + ((size_t)n * C + c) * 9;
            const double sc = out_scale ? (double)out_scale[n] : 1.0;
            double Gg[4][3];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) Gg[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const float u = (float)((Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2]) * sc);      // rounded once to fp32, as the bf16 form's
                    Un[c * 16 + i * 4 + j] = i == 3 ? -u : u;                                                        // row xi = 3 stored negated
                    mx = fmaxf(mx, fabsf(u));
                }
        }
        const int t = s2_row_shift(mx);
        oscale_out[n] = ldexpf(1.f, -(t + S2_XSHIFT));
        for (int c = 0; c < C; ++c) {
            const int chunk = c / X_CK, cc = c % X_CK, nb = n / 32, lane = (cc / 8) * 32 + (n % 32), e = cc % 8;
            for (int comp = 0; comp < 16; ++comp) {
                uint16_t h[2];
                s2_split_host(ldexpf(Un[c * 16 + comp], t), h);
                for (int p = 0; p < 2; ++p) out[((((size_t)(chunk * 16 + comp) * NB + nb) * 2 + p) * 64 + lane) * 8 + e] = h[p];
            }
        }
    }
    free(Un);
}

// host: conv3 weight of the fused tail, two fp16 planes: the layout of pack_gemm_weight_f16x2 (csrc/gemm_bf16x3.hip)
void pack_tail_weight_f16x2(const float* W3, int N2, int K, uint16_t* out, float* oscale_out) { pack_gemm_weight_f16x2(W3, N2, K, out, oscale_out); }

// NT = output channels / 32: 4 (128 channels: wave w owns n-tile w and all 16 components) or 2 (64 -> 64 channels: wave (wc, wn) owns n-tile wn and
// the 8 components of rows xi = 2 wc, 2 wc + 1 -- one accumulator each, no fold; the two halves meet through LDS before the epilogue, as in
// csrc/conv_wino.hip).
// NP = operand planes: 3 = three bf16 terms, six MFMAs per product block (csrc/bf16x3.h); 2 = two fp16 terms, three MFMAs (csrc/f16x2.h: the staged input times
// 2^S2_XSHIFT, U's rows times 2^t_n, accumulators back to scale by a.oscale / a.oscale3 in the epilogues, a.range_flag raised beyond fp16's range)
// NEXT (fp16 form of the fused tail only): the block's output never comes back for the NEXT block's conv1 -- relu(bn_next(out2)) of the tile is split into LDS
// right where out2 is stored and multiplied by the next block's W1 (256 -> 128) here: the 256-channel tensor is written once and not re-read by a GEMM launch
// (csrc/net.hip: residual(..., next)).  Same products in the same order as gemm_bf16x3_kernel<NP = 2> forms them: bit-identical to the separate launch.
// W8 (round 6; 128 -> 128 channels, fp16 form): EIGHT waves on the tile for launches of at most one workgroup per CU (a one-frame call, the 2-7 crops of a SLAM
// pass).  There a workgroup's latency is the launch's duration -- 50 us for the fused tail whether 64 or 256 workgroups run, against 64 us per round of two co-resident
// ones in a 256-crop launch: stalls, not throughput -- so the tile's work is spread over twice the waves: wave (wn, wc) owns n-tile wn = w & 3 and the eight components
// of rows xi = 2 wc, 2 wc + 1 (wc = w >> 2) -- one accumulator each, half the products per wave, NO fold -- exactly the 64-channel form's scheme, the halves meeting
// through LDS after the K loop; a thread of the transform computes ONE row of B^T d B (four components); the tail's conv3 gives every wave one 32-channel n-tile of
// the 256.  Each output element is the same sum of the same products in the same order, except that Y = (R0 + R1) + R2 / (R3' - R2) + R1 pairs its rows as the
// 64-channel form does, where the four-wave form folds per chunk: results agree to fp32 rounding of that last addition, not bit for bit.
template <bool FUSE, bool UP = false, bool TX3 = false, int NT = 4, int NP = 3, bool NEXT = false, bool W8 = false>
__global__ __launch_bounds__(W8 ? 512 : 256) __attribute__((amdgpu_waves_per_eu(2))) void wino3x3_x3_kernel(const ConvArgs a) {
    static_assert(NT == 4 || (NT == 2 && !FUSE), "64-channel form: plain convolution only");
    static_assert(!W8 || (NT == 4 && NP == 2 && !NEXT && (TX3 || !FUSE)), "eight waves: the fp16 form of the 128-channel kernel, without the next block's conv1");
    constexpr int NTHR = W8 ? 512 : 256;
    static_assert(!NEXT || (FUSE && TX3 && NP == 2), "the next block's conv1 rides on the fp16 tail");
    static_assert(NP == 3 || !FUSE || TX3, "the fp16 form's tail runs on the fp16 pipe");
    // one LDS array: halo double buffer (fp32) | V (three bf16 planes); the fused tail re-uses ALL of it for the conv2 tile
    constexpr int HSZ = X_NPIX * X_PKH;                       // floats per halo buffer
    constexpr int VROW = 16;                                  // bf16 per (tile, chunk) row = 32 bytes; the two 16-byte halves swap where bits 2 and 3 of the tile differ (4-7, 8-11, 20-23, 24-27):
                                                              // conflict-free A-fragment ds_read_b128, 2-way instead of 4-way on the transform's ds_write_b64
    constexpr int VPL = 16 * 32 * VROW;                       // bf16 per plane
    constexpr int VFLOATS = NP * VPL / 2;
    // (W8: one workgroup per CU anyway -- the tail's eight transposition patches get their own room behind the operand planes)
    constexpr int TAILF = NP * 8 * (64 * 32 + 32) / 4;        // floats of the tail's operand planes (NP planes x 8 k-steps x (64 pixels x 32 bytes + 32))
    constexpr int SFLOATS = W8 ? (2 * HSZ + VFLOATS > TAILF + 8 * 32 * 36 ? 2 * HSZ + VFLOATS : TAILF + 8 * 32 * 36) : 2 * HSZ + VFLOATS;
    __shared__ __attribute__((aligned(16))) float S[SFLOATS];
    float (*Hin)[HSZ] = reinterpret_cast<float (*)[HSZ]>(&S[0]);
    uint16_t* V = reinterpret_cast<uint16_t*>(&S[2 * HSZ]);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (a.OW + X_TW - 1) / X_TW, tiles_y = (a.OH + X_TH - 1) / X_TH;
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);          // XCD-aware tile order (csrc/conv.hip)
    const int l = bid / (tiles_x * tiles_y);
    bid -= l * tiles_x * tiles_y;
    const int ty0 = bid / tiles_x, tx0 = bid - ty0 * tiles_x;
    const int oy0 = ty0 * X_TH, ox0 = tx0 * X_TW;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int nch = a.C / X_CK;
    const float* in_l = a.in + (size_t)l * a.H * a.W * a.C;
    const __amdgpu_buffer_rsrc_t in_srd = make_srd(in_l, (size_t)a.H * a.W * a.C * sizeof(float));
    const __amdgpu_buffer_rsrc_t w_srd = make_srd(a.Wp, (size_t)a.N * a.C * 16 * NP * sizeof(uint16_t));
    const __amdgpu_buffer_rsrc_t out_srd = make_srd(a.out + (size_t)l * a.OH * a.OW * a.N, (size_t)a.OH * a.OW * a.N * sizeof(float));

    // ---- halo staging (as csrc/conv_wino.hip): 180 pixels x 4 float4 per chunk over 256 threads -----------------------------
    constexpr int NF4 = X_NPIX * 4, NLD = (NF4 + NTHR - 1) / NTHR;
    int avoff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * NTHR;
        const int pix = idx >> 2, cc = idx & 3;
        const int py = pix / X_IW, px = pix - py * X_IW;
        const int iy = iy0 + py, ix = ix0 + px;
        const bool ok = idx < NF4 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        avoff[i] = ok ? ((iy * a.W + ix) * a.C + cc * 4) * 4 : BUF_OOB;
    }
    x_f32x4 areg[NLD];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) areg[i] = buf_load(in_srd, avoff[i], c * X_CK * 4);
    };
    float dmax = 0.f;                                         // NP = 2: largest scaled input magnitude this lane staged (range guard)
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * NTHR;
            if constexpr (NP == 2) {                              // the activation scale (exact) and the guard's running max ride on the staging copy
                areg[i] *= S2_XSCALE;
                dmax = s2_track(s2_track(dmax, areg[i][0], areg[i][1]), areg[i][2], areg[i][3]);
            }
            if (idx < NF4) *(x_f32x4*)&Hin[buf][(idx >> 2) * X_PKH + (idx & 3) * 4] = areg[i];
        }
    };
    // ---- weights: Up3[(chunk * 16 + comp)][nb][plane][lane][8 bf16]: 3 KB per (component, n-tile), three 16-byte loads per lane -----
    // (128 output channels: NB = 4 n-tiles; the plane offset rides in the instruction's immediate field, one scalar addition per component.
    // The group offset is the instruction's SCALAR offset, which raw buffer loads do not range-check: every prefetch must name a group
    // inside Wq3 -- the loop below re-requests the last chunk's first pair instead of running past the end.)
    const int wvoff = lane * 16;
    const int wn = NT == 4 ? (W8 ? (w & 3) : w) : (w & 1), wc = NT == 4 ? (W8 ? (w >> 2) : 0) : (w >> 1);      // n-tile, component half
    constexpr bool HALF = NT == 2 || W8;                        // the wave owns half of the components, one accumulator each
    constexpr int NPAIR = HALF ? 4 : 8;                         // component pairs per chunk and wave
    const int wsbase = wn * NP * 1024;
    auto bload = [&](int gc, x_u32x4 (&b)[NP]) {              // gc = chunk * 16 + comp
        const int g = gc;
#pragma unroll
        for (int p = 0; p < NP; ++p) b[p] = __builtin_bit_cast(x_u32x4, buf_load(w_srd, wvoff + p * 1024, g * (NT * NP * 1024) + wsbase));
    };
    // ---- transform: thread = (tile tt, channel quad tq, half th) as in csrc/conv_wino.hip; every result vector is split on its way to LDS ----
    const int tt = tid & 31, tq = (tid >> 5) & 3, th = (tid >> 7) & 1;
    const int tv = __builtin_amdgcn_readfirstlane(tid >> 8);   // W8: which of the half's two rows this thread computes (0: first - third, 1: second +- other)
    const int t_ty = tt >> 3, t_tx = tt & 7;
    const int hbase = ((2 * t_ty + th) * X_IW + 2 * t_tx) * X_PKH + tq * 4;
    // V element offset of (component 0, tile tt, channels 4 tq ..): row tt, half (tq >> 1) swapped where bits 2, 3 of tt differ, 4 bf16 = 8 bytes
    const int vbase = tt * VROW + ((((tq >> 1) ^ (((tt >> 2) ^ (tt >> 3)) & 1)) * 8) + (tq & 1) * 4);
    auto vstore = [&](int comp, x_f32x4 v) {
        if constexpr (NP == 2) {                                  // hi = rn16(v), lo = rn16(v - hi) (the residual is exact)
            const unsigned h0 = s2_pack_rn(v[0], v[1]), h1 = s2_pack_rn(v[2], v[3]);
            *(x_u32x2*)&V[comp * 32 * VROW + vbase] = x_u32x2{h0, h1};
            const unsigned l0 = s2_lo_pack(v[0], v[1], h0), l1 = s2_lo_pack(v[2], v[3], h1);
            *(x_u32x2*)&V[VPL + comp * 32 * VROW + vbase] = x_u32x2{l0, l1};
            return;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned q0 = s3_pack_rn(v[0], v[1]), q1 = s3_pack_rn(v[2], v[3]);       // round-to-nearest terms (csrc/bf16x3.h); p == 2: exact
            *(x_u32x2*)&V[p * VPL + comp * 32 * VROW + vbase] = x_u32x2{q0, q1};
            if (p < 2) {
                v = x_sub4(v, x_f32x4{s3_lo(q0), s3_hi(q0), s3_lo(q1), s3_hi(q1)});        // exact residual
            }
        }
    };
    // rows of B^T d:  half 0 (input rows 0,1,2): xi0 = r0 - r2, xi1 = r1 + r2;  half 1 (rows 1,2,3): xi3 = r1 - r3, xi2 = r2 - r1
    // = (first - third, second +- other) with `other` = third / first row of the half's three: the row is picked by address, the sign is a scalar
    const int hother = th ? 0 : 2 * X_IW * X_PKH;
    const float hsign = th ? -1.f : 1.f;
    auto transform = [&](int buf) {
        const float* hs = &Hin[buf][hbase];
        if constexpr (W8) {                                      // one row per thread: two of the four halo rows, four components
            const int xi = tv == 0 ? (th ? 3 : 0) : (th ? 2 : 1);
            x_f32x4 e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (tv == 0) {
                    e[c] = x_sub4(*(const x_f32x4*)(hs + c * X_PKH), *(const x_f32x4*)(hs + (2 * X_IW + c) * X_PKH));
                } else {
                    const x_f32x4 l1 = *(const x_f32x4*)(hs + (X_IW + c) * X_PKH), lx = *(const x_f32x4*)(hs + hother + c * X_PKH);
                    e[c] = x_f32x4{__builtin_fmaf(lx[0], hsign, l1[0]), __builtin_fmaf(lx[1], hsign, l1[1]), __builtin_fmaf(lx[2], hsign, l1[2]), __builtin_fmaf(lx[3], hsign, l1[3])};
                }
            }
            vstore(xi * 4 + 0, x_sub4(e[0], e[2]));
            vstore(xi * 4 + 1, x_add4(e[1], e[2]));
            vstore(xi * 4 + 2, x_sub4(e[2], e[1]));
            vstore(xi * 4 + 3, x_sub4(e[1], e[3]));
            return;
        }
        x_f32x4 L0[4], L1[4], L2[4], Lx[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            L0[c] = *(const x_f32x4*)(hs + c * X_PKH);
            L1[c] = *(const x_f32x4*)(hs + (X_IW + c) * X_PKH);
            L2[c] = *(const x_f32x4*)(hs + (2 * X_IW + c) * X_PKH);
            Lx[c] = *(const x_f32x4*)(hs + hother + c * X_PKH);
        }
        x_f32x4 eA[4], eB[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            eA[c] = x_sub4(L0[c], L2[c]);
            eB[c] = x_f32x4{__builtin_fmaf(Lx[c][0], hsign, L1[c][0]), __builtin_fmaf(Lx[c][1], hsign, L1[c][1]), __builtin_fmaf(Lx[c][2], hsign, L1[c][2]),
                            __builtin_fmaf(Lx[c][3], hsign, L1[c][3])};                      // (a product by +-1 is exact: with or without contraction the same value)
        }
        const int xiA = th ? 3 : 0, xiB = th ? 2 : 1;
        // columns: nu0 = c0 - c2, nu1 = c1 + c2, nu2 = c2 - c1, nu3 = c1 - c3
        vstore(xiA * 4 + 0, x_sub4(eA[0], eA[2]));
        vstore(xiA * 4 + 1, x_add4(eA[1], eA[2]));
        vstore(xiA * 4 + 2, x_sub4(eA[2], eA[1]));
        vstore(xiA * 4 + 3, x_sub4(eA[1], eA[3]));
        vstore(xiB * 4 + 0, x_sub4(eB[0], eB[2]));
        vstore(xiB * 4 + 1, x_add4(eB[1], eB[2]));
        vstore(xiB * 4 + 2, x_sub4(eB[2], eB[1]));
        vstore(xiB * 4 + 3, x_sub4(eB[1], eB[3]));
    };

    x_f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    x_f32x16 out[4], Z[8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[p][r] = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) Z[p][r] = 0.f;
    // the six cross terms of a component, smallest first
    auto mac6 = [&](x_f32x16& acc, const x_bf16x8 (&f)[NP], const x_u32x4 (&bw)[NP], bool from_zero) {
        if constexpr (NP == 2) {                                  // hi lo, lo hi, hi hi
            constexpr int UI[3] = {0, 1, 0}, UJ[3] = {1, 0, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x_f16x8, f[UI[t]]), __builtin_bit_cast(x_f16x8, bw[UJ[t]]), (t == 0 && from_zero) ? zero16 : acc, 0, 0, 0);
            return;
        }
        constexpr int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[TI[t] % NP], __builtin_bit_cast(x_bf16x8, bw[TJ[t] % NP]), (t == 0 && from_zero) ? zero16 : acc, 0, 0, 0);
    };
    // Consumption order of the components, in pairs: pairs 0-3 = (xi 0, xi 3) of nu = pair, accumulated straight into their Z; pairs 4-7 =
    // (xi 1, xi 2) of nu = pair - 4, through two scratch accumulators that are folded into Z[0][nu], Z[1][nu].  The weights of pair p + 1
    // are requested before pair p's MFMAs.  Measured alternatives, all slower (tools/bench_wino_x3.py, 256 crops, plain / fused, us):
    // this schedule 920 / 1620; the two accumulation chains of a pair interleaved 1030 / 1800; one component at a time with the A
    // fragments one and the weights one / three components ahead 943 / 1690 and 1052 / 1825.
    auto comp_of = [&](int pair, int which) -> int {
        if (!HALF) return pair < 4 ? (which ? 12 + pair : pair) : (which ? 8 + (pair - 4) : 4 + (pair - 4));
        return wc * 8 + 2 * pair + which;                     // (64-channel form / eight waves: the wave's own half, in order)
    };
    // A operand of component comp: tile = lane & 31, channels 8 (lane >> 5) .. + 7 (the half, swapped as the rows were written)
    const int afoff = (lane & 31) * VROW + (((lane >> 5) ^ (((lane >> 2) ^ (lane >> 3)) & 1)) * 8);
    x_u32x4 bring[2][2][NP];                                  // [slot][which][plane]
    gload(0);
    bload(comp_of(0, 0), bring[0][0]);
    bload(comp_of(0, 1), bring[0][1]);
    sstore(0);
    __syncthreads();

#ifdef SUO_WX3_PROF
    long long pt[5] = {0, 0, 0, 0, 0}, p0 = clock64();
#define XPROF(i) do { const long long _t = clock64(); pt[i] += _t - p0; p0 = _t; } while (0)
#else
#define XPROF(i) do { } while (0)
#endif
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < nch;
        if (more) gload(c + 1);
        transform(buf);
        XPROF(0);
        __syncthreads();
        XPROF(1);
        if (more) sstore(buf ^ 1);
        XPROF(2);
#pragma unroll
        for (int pair = 0; pair < NPAIR; ++pair) {
            const int slot = pair & 1;
            {   // next pair (of this chunk, or pair 0 of the next one; after the last chunk pair 0 of that chunk again -- never consumed, but in range)
                const int np = pair + 1 < NPAIR ? pair + 1 : 0, nc = pair + 1 < NPAIR ? c : (more ? c + 1 : c);
                bload(nc * 16 + comp_of(np, 0), bring[slot ^ 1][0]);
                bload(nc * 16 + comp_of(np, 1), bring[slot ^ 1][1]);
            }
            const int ca = comp_of(pair, 0), cb = comp_of(pair, 1);
            x_bf16x8 afa[NP], afb[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                afa[p] = *(const x_bf16x8*)&V[p * VPL + ca * 32 * VROW + afoff];
                afb[p] = *(const x_bf16x8*)&V[p * VPL + cb * 32 * VROW + afoff];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (HALF) {                                       // Z[local component]
                mac6(Z[2 * pair], afa, bring[slot][0], false);
                mac6(Z[2 * pair + 1], afb, bring[slot][1], false);
            } else if (pair < 4) {                            // accumulate into their own Z, no scratch, no additions
                mac6(Z[pair], afa, bring[slot][0], false);
                mac6(Z[4 + pair], afb, bring[slot][1], false);
            } else {
                x_f32x16 ta, tb;
                mac6(ta, afa, bring[slot][0], true);
                mac6(tb, afb, bring[slot][1], true);
                x_add16(Z[pair - 4], ta); x_add16(Z[pair - 4], tb); x_add16(Z[pair], ta); x_sub16(Z[pair], tb);      // Z[0][nu] += M1 + M2, Z[1][nu] += M1 - M2
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        XPROF(3);
        __syncthreads();
        XPROF(4);
    }
#ifdef SUO_WX3_PROF
    if (blockIdx.x == 1000 && (tid & 63) == 0) printf("wave %d cycles: transform %lld  barrier1 %lld  sstore %lld  mfma+fold %lld  barrier2 %lld\n", w, pt[0], pt[1], pt[2], pt[3], pt[4]);
#endif
    if constexpr (NP == 2) s2_raise(a.range_flag, 4.f * dmax);    // |B^T d B| <= 4 max |d|: conservative by at most 4x, never late
    // second step of the output transform: R[h][j] = sum_nu A^T[j][nu] Z[4 h + nu].  128-channel form: h = i, R = Y; 64-channel form: h = the wave's local row
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        out[2 * h] = Z[4 * h] + Z[4 * h + 1] + Z[4 * h + 2];
        out[2 * h + 1] = Z[4 * h + 1] - Z[4 * h + 2] - Z[4 * h + 3];
    }
    // W8: the two component halves of an n-tile meet (as in the 64-channel form below): wave wc = 0 holds R of rows xi = 0, 1 (out[0..1], out[2..3]), wave wc = 1 of
    // rows 2, 3 (row 3 negated in the weights).  Y[0][j] = (R0 + R1) + R2, Y[1][j] = (R3' - R2) + R1: wave wc keeps output row i = wc and hands its MIDDLE row to the
    // partner through the V area ([wave][register][lane]: 32 KB per column j, two rounds)
    x_f32x16 mine[2];
    if constexpr (W8) {
        static_assert(VFLOATS >= 8 * 16 * 64, "the exchange area is the V planes");
        float* X = &S[2 * HSZ];                                // (the last chunk's closing barrier has passed: V is free)
        const int pw = (1 - wc) * 4 + wn;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) X[(w * 16 + r) * 64 + lane] = wc == 0 ? out[2 + q][r] : out[q][r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[q][r] = (wc == 0 ? out[q][r] + out[2 + q][r] : out[2 + q][r] - out[q][r]) + X[(pw * 16 + r) * 64 + lane];
            __syncthreads();
        }
    }

    if constexpr (FUSE && TX3) {
        // ---- tail of the Residual block (layers/Residual.py:27-35) on the bf16 pipe as well: out2 = W3 relu(conv2 + bias2) + bias3 + skip [+ up],
        // conv3 = 1x1, 128 -> 256.  The conv2 tile (128 pixels x 128 channels, in `out`: wave w holds channels [32 w, 32 w + 32) as 4 output
        // positions x 32 Winograd tiles) is the A operand; split into three bf16 planes it is 96 KB, so it goes through LDS in two halves of
        // 64 pixels: pass h = the output positions of pixel row parity h (image rows oy0 + 2 ty + h).  Per pass: every lane splits its 32
        // values and stores the bf16 terms (round-to-nearest of the value and of its two exact residuals: ds_write_b16_d16_hi) into
        // AP[plane][k-step][pixel m = 16 ty + x][16 channels] (A-operand order: 32-byte rows, the 16-byte halves swapped for pixels 8-15 of
        // every 16); barrier; wave w computes the 64 pixels x output channels [64 w, 64 w + 64) (2 x 2 accumulators, 8 k-steps x 24 MFMAs),
        // then its epilogue (transposition through a wave-private patch, + bias3 + skip [+ up], 16-byte stores).
        constexpr int KS_STRIDE = 64 * 32 + 32;                 // bytes per k-step image (+ 32: the two k-steps a wave stores to hit different banks)
        constexpr int PL_STRIDE = 8 * KS_STRIDE;
        static_assert(NP * PL_STRIDE == TAILF * 4 && NP * PL_STRIDE + (W8 ? 8 : 4) * 32 * 36 * 4 <= SFLOATS * 4, "tail staging must fit the workgroup's LDS");
        constexpr int NJ = W8 ? 1 : 2;                          // n-tiles of conv3 per wave: W8 -> the one tile w of eight; else tiles w and 4 + w
        unsigned char* AP = reinterpret_cast<unsigned char*>(&S[0]);
        float* T = &S[NP * PL_STRIDE / 4] + w * (32 * 36);
        const __amdgpu_buffer_rsrc_t w3_srd = make_srd(a.W3p, (size_t)a.N * a.N2 * NP * sizeof(uint16_t));
        const size_t crop2 = (size_t)a.OH * a.OW * a.N2;
        const __amdgpu_buffer_rsrc_t r_srd = make_srd(a.R + (size_t)l * crop2, crop2 * sizeof(float));
        const __amdgpu_buffer_rsrc_t o2_srd = make_srd(a.out2 + (size_t)l * crop2, crop2 * sizeof(float));
        constexpr bool has_up = UP;
        const __amdgpu_buffer_rsrc_t up_srd = make_srd(has_up ? a.up + (size_t)l * (crop2 / 4) : a.R, has_up ? crop2 / 4 * sizeof(float) : 0);
        // NP = 2: the conv2 accumulator carries 2^(t_n + S2_XSHIFT); conv3's operand is 2^S2_XSHIFT relu(conv2 + b2) = relu(acc 2^-t_n + 2^S2_XSHIFT b2): one fma
        const float b2v = NP == 2 ? a.bias[wn * 32 + (lane & 31)] * S2_XSCALE : a.bias[wn * 32 + (lane & 31)];
        const float c2v = NP == 2 ? a.oscale[wn * 32 + (lane & 31)] * S2_XSCALE : 1.f;
        float tmax = 0.f;
        // store address of the lane's channel n = 32 w + (lane & 31): k-step n >> 4, half (n >> 3) & 1 (swapped for pixels 8-15: = lane >> 5, see m below),
        // element n & 7; pixel m = 16 (r >> 2) + 2 (r & 3) + 8 (lane >> 5) + pj for accumulator row r (tile (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
        const int sbase = (2 * wn + ((lane >> 4) & 1)) * KS_STRIDE + (lane >> 5) * (8 * 32) + ((((lane >> 3) & 1) ^ (lane >> 5)) * 16) + (lane & 7) * 2;
        // A operand: pixel m = 32 i + (lane & 31), channels 8 (lane >> 5) .. + 7 of the k-step
        const int aoff = (lane & 31) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) * 16);
        // weights: W3x[(ks * 8 + nb) * 3 + plane][lane][8 bf16], 1 KB each (pack_tail_weight_bf16x3); the wave's n-tiles are w and 4 + w (output channels
        // [32 w, 32 w + 32) and [128 + 32 w, ...): the four waves' first tiles are channels 0-127 in order, which is the K order the NEXT conv1 consumes them in)
        const int w3voff = lane * 16;
        auto b3load = [&](int ks, x_u32x4 (&b)[2][NP]) {
            const int k = ks < 8 ? ks : 7;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p) b[j][p] = __builtin_bit_cast(x_u32x4, buf_load(w3_srd, w3voff, (((k * 8 + (W8 ? w : 4 * j + w)) * NP) + p) * 1024));
        };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            x_u32x4 b3[2][2][NP];
            b3load(0, b3[0]);
            if (!W8 || wc == h)                               // (W8: the pass's pixel rows are held by the waves of component half h)
#pragma unroll
            for (int pj = 0; pj < 2; ++pj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float cv = W8 ? mine[pj][r] : out[2 * h + pj][r];
                    float v = NP == 2 ? fmaxf(fmaf(cv, c2v, b2v), 0.f) : fmaxf(cv + b2v, 0.f);
                    const int off = sbase + (16 * (r >> 2) + 2 * (r & 3) + pj) * 32;
                    if constexpr (NP == 2) {
                        tmax = fmaxf(tmax, v);                            // (v >= 0)
                        const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                        *reinterpret_cast<_Float16*>(AP + off) = hi;
                        *reinterpret_cast<_Float16*>(AP + PL_STRIDE + off) = lo;
                        continue;
                    }
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const unsigned q = s3_pack_rn(v, v);              // both halves = rn(v): the store takes the upper one (ds_write_b16_d16_hi)
                        *reinterpret_cast<uint16_t*>(AP + p * PL_STRIDE + off) = (uint16_t)(q >> 16);
                        if (p < 2) v -= s3_hi(q);
                    }
                }
            __syncthreads();
            x_f32x16 acc2[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                b3load(ks + 1, b3[(ks + 1) & 1]);
                x_bf16x8 af[2][NP];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < NP; ++p) af[i][p] = *(const x_bf16x8*)(AP + p * PL_STRIDE + ks * KS_STRIDE + i * 1024 + aoff);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) mac6(acc2[i][j], af[i], b3[ks & 1][j], false);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (NEXT) __syncthreads();              // every wave is past its conv3 reads of AP: the NEXT conv1's operand planes overwrite it
            x_f32x16 acc1[2];                                 // NEXT: conv1 of the next block, 64 pixels x channels [32 w, 32 w + 32)
            if constexpr (NEXT) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc1[i] = zero16;
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {                    // j = the wave's n-tile: output channels [128 j + 32 w, + 32) (W8: [32 w, + 32)); NEXT: round j = conv1's K half j
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int col = (W8 ? w : 4 * j + w) * 32 + (lane & 7) * 4;
                    const x_f32x4 bv = *(const x_f32x4*)(a.bias3 + col);
                    x_f32x4 osc3 = x_f32x4{1.f, 1.f, 1.f, 1.f};
                    if constexpr (NP == 2) osc3 = *(const x_f32x4*)(a.oscale3 + col);
                    x_f32x4 nsc, nsh;
                    if constexpr (NEXT) { nsc = *(const x_f32x4*)(a.n_scale + col) * S2_XSCALE; nsh = *(const x_f32x4*)(a.n_shift + col) * S2_XSCALE; }
                    int off[4];
                    x_f32x4 rv[4], uv[UP ? 4 : 1];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int m = 32 * i + (lane >> 3) + 8 * k;
                        const int oy = oy0 + 2 * (m >> 4) + h, ox = ox0 + (m & 15);
                        const bool in = oy < a.OH && ox < a.OW;
                        off[k] = in ? ((oy * a.OW + ox) * a.N2 + col) * 4 : BUF_OOB;
                        rv[k] = buf_load(r_srd, off[k], 0);
                        if constexpr (UP && !NEXT) uv[k] = buf_load(up_srd, in ? (((oy >> 1) * (a.OW >> 1) + (ox >> 1)) * a.N2 + col) * 4 : BUF_OOB, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) T[x_acc_row(r, lane) * 36 + (lane & 31)] = acc2[i][j][r];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if constexpr (UP && NEXT) {           // (requested here, one at a time: with the next block's accumulators live there is no room for four)
                            const int m = 32 * i + (lane >> 3) + 8 * k;
                            const int oy = oy0 + 2 * (m >> 4) + h, ox = ox0 + (m & 15);
                            uv[0] = buf_load(up_srd, (oy < a.OH && ox < a.OW) ? (((oy >> 1) * (a.OW >> 1) + (ox >> 1)) * a.N2 + col) * 4 : BUF_OOB, 0);
                        }
                        x_f32x4 o = *(const x_f32x4*)&T[((lane >> 3) + 8 * k) * 36 + (lane & 7) * 4];
                        if constexpr (NP == 2) o *= osc3;             // back to scale (an exact power of two per column)
                        o = (o + bv) + rv[k];
                        if constexpr (UP) o += uv[NEXT ? 0 : k];
                        buf_store(o, o2_srd, off[k]);
                        if constexpr (NEXT) {
                            // the next block's operand: 2^S2_XSHIFT relu(bn_next(o)) of this pixel's 4 channels (local channel 32 w + 4 (lane & 7) of K half j) as
                            // two fp16 planes in A-operand order -- k-step 2 w + ((lane & 7) >> 2), pixel m, 16-byte half ((lane & 3) >> 1) swapped for pixels 8-15
                            const int m = 32 * i + (lane >> 3) + 8 * k;
                            float xn[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) xn[c] = fmaxf(fmaf(o[c], nsc[c], nsh[c]), 0.f);
                            tmax = fmaxf(fmaxf(tmax, fmaxf(xn[0], xn[1])), fmaxf(xn[2], xn[3]));
                            const unsigned h0 = s2_pack_rn(xn[0], xn[1]), h1 = s2_pack_rn(xn[2], xn[3]);
                            const unsigned l0 = s2_lo_pack(xn[0], xn[1], h0), l1 = s2_lo_pack(xn[2], xn[3], h1);
                            unsigned char* d = AP + (2 * w + ((lane & 7) >> 2)) * KS_STRIDE + m * 32 + (((((lane & 3) >> 1) ^ ((m >> 3) & 1))) * 16) + (lane & 1) * 8;
                            *reinterpret_cast<x_u32x2*>(d) = x_u32x2{h0, h1};
                            *reinterpret_cast<x_u32x2*>(d + PL_STRIDE) = x_u32x2{l0, l1};
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if constexpr (NEXT) {
                    // K half j of conv1 (channels [128 j, 128 j + 128) = k-steps 8 j .. 8 j + 7, ascending: the order of gemm_bf16x3_kernel): wave w -> n-tile w
                    const __amdgpu_buffer_rsrc_t n1_srd = make_srd(a.n_W1, (size_t)128 * 256 * 2 * sizeof(uint16_t));
                    auto n1load = [&](int q, x_u32x4 (&b)[2]) {
                        const int ksg = 8 * j + (q < 8 ? q : 7);
#pragma unroll
                        for (int p = 0; p < 2; ++p) b[p] = __builtin_bit_cast(x_u32x4, buf_load(n1_srd, w3voff + p * 1024, ((ksg * 4 + w) * 2) * 1024));
                    };
                    constexpr int NR = 4;                     // weight k-steps in flight: a k-step is only 6 MFMAs (192 cycles), an L2 round trip three times that
                    x_u32x4 nb[NR][2];
#pragma unroll
                    for (int q = 0; q < NR - 1; ++q) n1load(q, nb[q]);
                    __syncthreads();                          // the four waves' planes of this K half are complete
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        n1load(q + NR - 1, nb[(q + NR - 1) % NR]);
                        x_bf16x8 af[2][2];
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int p = 0; p < 2; ++p) af[i][p] = *(const x_bf16x8*)(AP + p * PL_STRIDE + q * KS_STRIDE + i * 1024 + aoff);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 2; ++i) mac6(acc1[i], af[i], nb[q % NR], false);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __syncthreads();                          // every wave has read the planes: the next K half / the next pass's conv2 tile may overwrite them
                }
            }
            if constexpr (NEXT) {
                // conv1's epilogue as gemm_bf16x3_kernel's: acc * oscale + bias, ReLU, 16-byte stores of the wave's 32 channels of the 64 pixels
                const int col = w * 32 + (lane & 7) * 4;
                const x_f32x4 b1 = *(const x_f32x4*)(a.n_b1 + col), o1 = *(const x_f32x4*)(a.n_osc1 + col);
                const size_t cropn = (size_t)a.OH * a.OW * 128;
                const __amdgpu_buffer_rsrc_t n_srd = make_srd(a.n_out + (size_t)l * cropn, cropn * sizeof(float));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) T[x_acc_row(r, lane) * 36 + (lane & 31)] = acc1[i][r];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int m = 32 * i + (lane >> 3) + 8 * k;
                        const int oy = oy0 + 2 * (m >> 4) + h, ox = ox0 + (m & 15);
                        x_f32x4 o = *(const x_f32x4*)&T[((lane >> 3) + 8 * k) * 36 + (lane & 7) * 4];
                        o *= o1;
                        o += b1;
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], 0.f);
                        buf_store(o, n_srd, (oy < a.OH && ox < a.OW) ? ((oy * a.OW + ox) * 128 + col) * 4 : BUF_OOB);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (h == 0 && !NEXT) __syncthreads();             // every wave is done with the A planes of pass 0 (NEXT: the K half's closing barrier)
        }
        if constexpr (NP == 2) s2_raise(a.range_flag, tmax);
        return;
    }

    if constexpr (FUSE && !TX3) {
        // ---- tail of the Residual block (layers/Residual.py:27-35) exactly as in csrc/conv_wino.hip: conv3 (1x1, 128 -> 256) + bias + skip
        // on relu(conv2 + bias2), on the fp32 pipe: the conv2 tile goes through LDS (pitch 132) as the A operand, 2 x 4 accumulators
        constexpr int MP = 132;
        static_assert(128 * MP <= 2 * HSZ + VFLOATS, "the conv2 tile must fit the workgroup's LDS");
        float* M2 = &S[0];
        const int wm = w >> 1, wn = w & 1;
        const int NB2 = a.N2 >> 5;
        const __amdgpu_buffer_rsrc_t w3_srd = make_srd(a.W3p, (size_t)a.N * a.N2 * sizeof(float));
        const size_t crop2 = (size_t)a.OH * a.OW * a.N2;
        const __amdgpu_buffer_rsrc_t r_srd = make_srd(a.R + (size_t)l * crop2, crop2 * sizeof(float));
        const __amdgpu_buffer_rsrc_t o2_srd = make_srd(a.out2 + (size_t)l * crop2, crop2 * sizeof(float));
        constexpr bool has_up = UP;
        const __amdgpu_buffer_rsrc_t up_srd = make_srd(has_up ? a.up + (size_t)l * (crop2 / 4) : a.R, has_up ? crop2 / 4 * sizeof(float) : 0);
        const float b2v = a.bias[w * 32 + (lane & 31)];
        const int w3voff = lane * 16;
        auto b3load = [&](int q, x_f32x4(&b)[4]) {
            const int kg = q < 16 ? q : 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = buf_load(w3_srd, w3voff, (kg * NB2 + wn * 4 + j) * 1024);
        };
        constexpr int R3 = 3;
        x_f32x4 b3[R3][4];
#pragma unroll
        for (int r = 0; r < R3 - 1; ++r) b3load(r, b3[r]);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = x_acc_row(r, lane);
                const int pix = (2 * (t >> 3) + (p >> 1)) * X_TW + 2 * (t & 7) + (p & 1);
                M2[pix * MP + w * 32 + (lane & 31)] = fmaxf(out[p][r] + b2v, 0.f);
            }
        __syncthreads();
        x_f32x16 acc2[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
        const float* ms = M2 + ((wm * 2) * 32 + (lane & 31)) * MP + (lane >> 5) * 4;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            b3load(q + R3 - 1, b3[(q + R3 - 1) % R3]);
            __builtin_amdgcn_sched_barrier(0);
            x_f32x4 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const x_f32x4*)(ms + i * 32 * MP + q * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc2[i][j] = x_mfma32(af[i][t], b3[q % R3][j][t], acc2[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* T = M2 + w * (32 * 36);
                const int col = (wn * 4 + j) * 32 + (lane & 7) * 4;
                const x_f32x4 bv = *(const x_f32x4*)(a.bias3 + col);
                int off[4];
                x_f32x4 rv[4], uv[UP ? 4 : 1];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pp = (wm * 2 + i) * 32 + (lane >> 3) + 8 * k;
                    const int oy = oy0 + pp / X_TW, ox = ox0 + pp % X_TW;
                    const bool in = oy < a.OH && ox < a.OW;
                    off[k] = in ? ((oy * a.OW + ox) * a.N2 + col) * 4 : BUF_OOB;
                    rv[k] = buf_load(r_srd, off[k], 0);
                    if constexpr (UP) uv[k] = buf_load(up_srd, in ? (((oy >> 1) * (a.OW >> 1) + (ox >> 1)) * a.N2 + col) * 4 : BUF_OOB, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) T[x_acc_row(r, lane) * 36 + (lane & 31)] = acc2[i][j][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    x_f32x4 o = (*(const x_f32x4*)&T[((lane >> 3) + 8 * k) * 36 + (lane & 7) * 4] + bv) + rv[k];
                    if constexpr (UP) o += uv[k];
                    buf_store(o, o2_srd, off[k]);
                }
                __builtin_amdgcn_wave_barrier();
            }
        return;
    }

    if constexpr (NT == 2 || W8) {
        // the two component halves of an n-tile meet (W8: they already have, above).  Wave wc = 0 holds rows xi = 0, 1 (out[0..1] = R of row 0, out[2..3] = R of row 1), wave wc = 1
        // rows xi = 2, 3 (row 3 negated in the weights).  Y[0][j] = R0 + R1 + R2, Y[1][j] = R1 - R2 + R3': wave wc keeps output row i = wc and
        // hands its MIDDLE row (1 or 2) to the partner through LDS ([position][register][lane]: conflict-free).
        if constexpr (NT == 2) {
            float* X = &S[2 * HSZ];                            // (the last chunk's closing barrier has passed: V is free)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) X[((w * 2 + q) * 16 + r) * 64 + lane] = wc == 0 ? out[2 + q][r] : out[q][r];
            __syncthreads();
            const int pw = (1 - wc) * 2 + wn;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    mine[q][r] = (wc == 0 ? out[q][r] + out[2 + q][r] : out[2 + q][r] - out[q][r]) + X[((pw * 2 + q) * 16 + r) * 64 + lane];
            __syncthreads();                                  // the exchange area becomes the transposition patches
        }
        float* T = &S[2 * HSZ] + w * (32 * 36);
        const int col = wn * 32 + (lane & 7) * 4;
        const x_f32x4 bv = *(const x_f32x4*)(a.bias + col);
        x_f32x4 osc = x_f32x4{1.f, 1.f, 1.f, 1.f};
        if constexpr (NP == 2) osc = *(const x_f32x4*)(a.oscale + col);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = 2 * wc + q, pi = p >> 1, pj = p & 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) T[x_acc_row(r, lane) * 36 + (lane & 31)] = mine[q][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = (lane >> 3) + 8 * k;
                const int oy = oy0 + 2 * (t >> 3) + pi, ox = ox0 + 2 * (t & 7) + pj;
                x_f32x4 o = *(const x_f32x4*)&T[t * 36 + (lane & 7) * 4];
                if constexpr (NP == 2) o *= osc;
                o += bv;
                if (a.relu) {
#pragma unroll
                    for (int z = 0; z < 4; ++z) o[z] = fmaxf(o[z], 0.f);
                }
                buf_store(o, out_srd, (oy < a.OH && ox < a.OW) ? ((oy * a.OW + ox) * a.N + col) * 4 : BUF_OOB);
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }

    // ---- epilogue (plain convolution): per output position transpose the 32 tiles x 32 channels through a wave-private patch -> 16-byte stores
    float* T = &S[2 * HSZ] + w * (32 * 36);
    const int col = w * 32 + (lane & 7) * 4;
    const x_f32x4 bv = *(const x_f32x4*)(a.bias + col);
    x_f32x4 osc = x_f32x4{1.f, 1.f, 1.f, 1.f};
    if constexpr (NP == 2) osc = *(const x_f32x4*)(a.oscale + col);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int pi = p >> 1, pj = p & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) T[x_acc_row(r, lane) * 36 + (lane & 31)] = out[p][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = (lane >> 3) + 8 * k;
            const int oy = oy0 + 2 * (t >> 3) + pi, ox = ox0 + 2 * (t & 7) + pj;
            x_f32x4 o = *(const x_f32x4*)&T[t * 36 + (lane & 7) * 4];
            if constexpr (NP == 2) o *= osc;
            o += bv;
            if (a.relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = fmaxf(o[q], 0.f);
            }
            buf_store(o, out_srd, (oy < a.OH && ox < a.OW) ? ((oy * a.OW + ox) * a.N + col) * 4 : BUF_OOB);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Launches of at most one workgroup per CU (<= 256 tiles: a one-frame call, a SLAM pass) take the eight-wave form of the fp16 kernel (W8): there the
// workgroup's latency IS the launch's duration.  SUO_WINO_W8=0 (supported switch, include/suo_hip.h; read per launch): the four-wave form everywhere (A/B);
// SUO_WINO_W8_TILES (tuning builds): largest such launch.
bool conv3x3_wino_f16x2_w8(long tiles) {
    static const long upto = SUO_TUNE("SUO_WINO_W8_TILES", 256);
    return tiles > 0 && tiles <= upto && env_switch("SUO_WINO_W8", 1) != 0;
}

// a.Wp = weights packed by pack_wino_weight_bf16x3 / _f16x2 (uint16 under a float pointer); 128 -> 128 or 64 -> 64 channels
template <int NP>
static int launch_wino_split(const ConvArgs& a, hipStream_t s) {
    if (NP == 2 && (!a.oscale || !a.range_flag)) { suo_set_error("conv3x3_wino_f16x2: oscale / range_flag missing"); return SUO_ERR_ARG; }
    if (a.OH == a.H && a.OW == a.W && a.N == 64 && a.C == 64) {
        const int tiles64 = ((a.OW + X_TW - 1) / X_TW) * ((a.OH + X_TH - 1) / X_TH) * a.L;
        hipLaunchKernelGGL((wino3x3_x3_kernel<false, false, false, 2, NP>), dim3(tiles64), dim3(256), 0, s, a);
        SUO_HIP_CHECK(hipGetLastError());
        return SUO_OK;
    }
    if (a.OH != a.H || a.OW != a.W || a.N != 128 || a.C != 128) {
        suo_set_error("conv3x3_wino_x3: unsupported shape H=%d W=%d C=%d N=%d", a.H, a.W, a.C, a.N);
        return SUO_ERR_ARG;
    }
    const int tiles = ((a.OW + X_TW - 1) / X_TW) * ((a.OH + X_TH - 1) / X_TH) * a.L;
    if (NP == 2 && conv3x3_wino_f16x2_w8(tiles)) hipLaunchKernelGGL((wino3x3_x3_kernel<false, false, false, 4, 2, false, true>), dim3(tiles), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((wino3x3_x3_kernel<false, false, false, 4, NP>), dim3(tiles), dim3(256), 0, s, a);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}
int launch_conv3x3_wino_x3(const ConvArgs& a, hipStream_t s) { return launch_wino_split<3>(a, s); }
int launch_conv3x3_wino_f16x2(const ConvArgs& a, hipStream_t s) { return launch_wino_split<2>(a, s); }

// the fused Residual tail on two fp16 planes: a.Wp = pack_wino_weight_f16x2, a.W3p = pack_tail_weight_f16x2, a.oscale / a.oscale3 their factors
int launch_conv3x3_wino_f16x2_fused(const ConvArgs& a, hipStream_t s) {
    if (a.OH != a.H || a.OW != a.W || a.N != 128 || a.C != 128 || a.N2 != 256 || !a.W3p || !a.bias3 || !a.R || !a.out2 || !a.oscale || !a.oscale3 || !a.range_flag) {
        suo_set_error("conv3x3_wino_f16x2_fused: unsupported shape C=%d N=%d N2=%d", a.C, a.N, a.N2);
        return SUO_ERR_ARG;
    }
    const int tiles = ((a.OW + X_TW - 1) / X_TW) * ((a.OH + X_TH - 1) / X_TH) * a.L;
    if (a.n_W1) {                                      // ... with the next block's conv1 on the tile (all of n_* set)
        if (!a.n_scale || !a.n_shift || !a.n_osc1 || !a.n_b1 || !a.n_out) { suo_set_error("conv3x3_wino_f16x2_fused: incomplete next-block arguments"); return SUO_ERR_ARG; }
        // (with an up-sampled addend the kernel would need 14-23 registers more than the 256 a two-workgroup-per-CU kernel has: it spilled and measured
        //  SLOWER than the two launches, profiles/REJECTED.md -- not built; csrc/net.hip does not ask for it)
        if (a.up) { suo_set_error("conv3x3_wino_f16x2_fused: the next block's conv1 cannot ride on a tail with an up-sampled addend"); return SUO_ERR_ARG; }
        hipLaunchKernelGGL((wino3x3_x3_kernel<true, false, true, 4, 2, true>), dim3(tiles), dim3(256), 0, s, a);
    } else if (conv3x3_wino_f16x2_w8(tiles)) {
        if (a.up) hipLaunchKernelGGL((wino3x3_x3_kernel<true, true, true, 4, 2, false, true>), dim3(tiles), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((wino3x3_x3_kernel<true, false, true, 4, 2, false, true>), dim3(tiles), dim3(512), 0, s, a);
    } else if (a.up) hipLaunchKernelGGL((wino3x3_x3_kernel<true, true, true, 4, 2>), dim3(tiles), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wino3x3_x3_kernel<true, false, true, 4, 2>), dim3(tiles), dim3(256), 0, s, a);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

int launch_conv3x3_wino_x3_fused(const ConvArgs& a, hipStream_t s) {
    if (a.OH != a.H || a.OW != a.W || a.N != 128 || a.C != 128 || a.N2 != 256 || !a.W3p || !a.bias3 || !a.R || !a.out2) {
        suo_set_error("conv3x3_wino_x3_fused: unsupported shape C=%d N=%d N2=%d", a.C, a.N, a.N2);
        return SUO_ERR_ARG;
    }
    const int tiles = ((a.OW + X_TW - 1) / X_TW) * ((a.OH + X_TH - 1) / X_TH) * a.L;
    if (a.w3_bf16x3) {                                 // W3p packed by pack_tail_weight_bf16x3: the tail on the bf16 pipe too
        if (a.up) hipLaunchKernelGGL((wino3x3_x3_kernel<true, true, true>), dim3(tiles), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wino3x3_x3_kernel<true, false, true>), dim3(tiles), dim3(256), 0, s, a);
    } else {
        if (a.up) hipLaunchKernelGGL((wino3x3_x3_kernel<true, true, false>), dim3(tiles), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wino3x3_x3_kernel<true, false, false>), dim3(tiles), dim3(256), 0, s, a);
    }
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo

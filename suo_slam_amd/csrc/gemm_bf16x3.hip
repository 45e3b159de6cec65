// fp32-accurate 1x1 convolution on the bf16 matrix pipe by 3-way operand splitting: what the network launches for its 1x1 convolutions with
// 128 / 256 output channels at >= 4096 pixels on the fp16 pipe, >= 32768 on the bf16x3 pipe (csrc/net.hip: x3_min_rows, gemm_maybe_pooled; SUO_WINO_BF16X3=0 keeps the fp32-pipe kernels).
//
// gfx950 runs fp32 MFMAs at the vector rate (157 TFLOP/s) and bf16 MFMAs 16x faster (2.5 PFLOP/s dense).  An fp32 number is exactly
// the sum of three bf16 numbers (8 significand bits each: x0 = rn(x), x1 = rn(x - x0), x2 = x - x0 - x1, every subtraction exact: csrc/bf16x3.h),
// a bf16 x bf16 product is exact in fp32, and v_mfma_f32_32x32x16_bf16 accumulates in fp32.  So
//     x * w  =  sum over i + j <= 2 of x_i * w_j   +   O(2^-24 |x w|), either sign  (6 of the 9 cross terms)
// costs 6 bf16 MFMAs of K = 16 (6 x 32 = 192 cycles per SIMD) where the fp32 form needs 8 MFMAs of K = 2 (8 x 64 = 512 cycles): 2.67x
// fewer matrix-pipe cycles per MAC at fp32 accuracy -- the only lever above the fp32-MFMA roof.
//   out[M, N] = [relu]( [relu(A1 * scale + shift) or A1] W1^T + A2 W2^T + bias + R )  [and / or its 2x2 max-pool]
// Kernel layout and the measurements behind it: the comment at gemm_bf16x3_kernel and DESIGN.md section 4 ("Measured, bf16 pipe").
#include <stdlib.h>
#include <string.h>

#include "bf16x3.h"
#include "f16x2.h"
#include "buffer_ops.h"
#include "suo_internal.h"
#include "tune.h"

namespace suo {

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 x3_f16x8 __attribute__((ext_vector_type(8)));
typedef float x3_f32x16 __attribute__((ext_vector_type(16)));
typedef float x3_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned x3_u32x4 __attribute__((ext_vector_type(4)));

#ifndef SUO_X3_F16_WAVES
#define SUO_X3_F16_WAVES 3      // waves per SIMD the fp16 form is compiled for: 3 = <= 168 registers, THREE workgroups per CU (with a 2-slot weight ring: 152-160
#endif                          // registers, no spill).  Measured against 2 (184-192 registers, 4 slots) at 256 crops, us: conv1 394 -> 365, lin 703 -> 623, re-injection 796 -> 748
#ifndef SUO_X3_BK
#define SUO_X3_BK 16
#endif
// k-step per barrier: 16 (one MFMA k-group, 24 MFMAs between barriers) or 32 (tools/build_variant.sh bk32 -DSUO_X3_BK=32: 48 MFMAs between barriers,
// 128-byte row segments per request, 61 KB of LDS).  Measured equal within 3 % on every shape of the network (tools/bench_gemm_x3_shapes.py), as is
// the depth of the request ring (1 or 3 steps ahead): the kernel's time follows its MFMA count -- the call runs at the package power cap, see DESIGN 4.2
constexpr int X3_BN = 128, X3_BK = SUO_X3_BK, X3_GH = X3_BK / 16, X3_PITCH = X3_BK + 8;      // LDS row pitch in bf16 (48 / 80 bytes: conflict-free 16-byte fragment reads)
constexpr int X3_LPR = X3_BK / 4, X3_RPP = 256 / X3_LPR;                           // staging: lanes per row (a float4 each), rows per pass
constexpr int X3_ASLOTS = 64 / X3_BK;                                              // activation steps in flight (K a multiple of 64: slots and stages are compile-time indices)
#ifndef SUO_X3_BSLOTS
#define SUO_X3_BSLOTS 4
#endif
#ifndef SUO_X3_F16_BSLOTS
#define SUO_X3_F16_BSLOTS 2
#endif
constexpr int X3_BSLOTS = SUO_X3_BSLOTS;                                                       // weight k-groups in flight (a ring over 16-wide groups)
static_assert(X3_BK == 16 || X3_BK == 32, "k-step");

// host: W[N][K] fp32 -> B-operand order of v_mfma_f32_32x32x16_bf16, split like the device does (csrc/bf16x3.h: round-to-nearest terms):
//   out[((ks * NB + nb) * 3 + plane) * 64 + lane][e] = term `plane` of W[nb*32 + (lane&31)][ks*16 + 8*(lane>>5) + e]
void pack_gemm_weight_bf16x3(const float* W, int N, int K, uint16_t* out) {
    const int NB = N / 32;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const int ks = k / 16, kk = k % 16, lane = (kk / 8) * 32 + (n % 32), e = kk % 8, nb = n / 32;
            uint16_t t[3];
            s3_split_host(W[(size_t)n * K + k], t);
            for (int p = 0; p < 3; ++p) out[((((size_t)(ks * NB + nb) * 3 + p) * 64 + lane) * 8) + e] = t[p];
        }
}

// host: the two-term fp16 form (csrc/f16x2.h).  Row n is scaled by 2^t_n (max_k |W[n][k]| 2^t_n in [2^12, 2^13)) before the split; the kernel's epilogue
// multiplies by oscale[n] = 2^-(t_n + S2_XSHIFT) (the activations enter times 2^S2_XSHIFT).  Same B-operand order with two planes:
//   out[((ks * NB + nb) * 2 + plane) * 64 + lane][e] = term `plane` of 2^t_n W[n = nb*32 + (lane&31)][ks*16 + 8*(lane>>5) + e]
void pack_gemm_weight_f16x2(const float* W, int N, int K, uint16_t* out, float* oscale_out) {
    const int NB = N / 32;
    for (int n = 0; n < N; ++n) {
        float mx = 0.f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(W[(size_t)n * K + k]));
        const int t = s2_row_shift(mx);
        oscale_out[n] = ldexpf(1.f, -(t + S2_XSHIFT));
        for (int k = 0; k < K; ++k) {
            const int ks = k / 16, kk = k % 16, lane = (kk / 8) * 32 + (n % 32), e = kk % 8, nb = n / 32;
            uint16_t h[2];
            s2_split_host(ldexpf(W[(size_t)n * K + k], t), h);
            for (int p = 0; p < 2; ++p) out[((((size_t)(ks * NB + nb) * 2 + p) * 64 + lane) * 8) + e] = h[p];
        }
    }
}

__device__ __forceinline__ int x3_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Workgroup = 128 rows x 128 columns, four waves as 2 x 2 (64 x 64 each = 2 x 2 accumulators), one 16-wide k-step per barrier.
//   * activations: global fp32 (four adjacent lanes fetch the 64 contiguous bytes a row contributes to a k-step) -> registers -> prologue ->
//     round-to-nearest split (csrc/bf16x3.h) -> three bf16 planes in LDS (two stages) -> A fragments by ds_read_b128;
//   * weights: host-split, B-operand order, straight from L2 into registers (the two waves of a column half read the same lines: L1);
//   * vmcnt retires in order, so a request can only be waited for once everything issued before it has landed: both streams are requested
//     a fixed number of k-steps ahead (an activation load from HBM issued just before a "nearer" weight load would stall that one);
//   * epilogue: accumulators transposed through a wave-private LDS patch, bias (+ ReLU) and stores on 16-byte vectors.
// POOL: the result's 2x2 max-pool (nn.MaxPool2d(2, 2)) written as well (or only: g.out may be null).  The tile is then two image rows x 64 columns
// (row r of the tile = pixel (y0 + (r >> 6), x0 + (r & 63))): the horizontal maximum is a lane exchange (rows j, j + 1 sit 8 lanes apart), the vertical
// one meets through LDS (image row 0 belongs to the waves wm = 0, row 1 to wm = 1).
// NCB: 32-column blocks per wave -- 2: the 128-column tile; 1: a 64-column tile (the 64-channel 1x1 convolutions of the first Residual blocks)
// RB: 32-row blocks per wave -- 2: 128-row tiles; 1: 64-row tiles, for launches that would otherwise be fewer tiles than two per CU (the one-frame call)
// NP: operand planes -- 3: three bf16 terms, six MFMAs per product block (csrc/bf16x3.h); 2: two fp16 terms, three MFMAs (csrc/f16x2.h: activations enter times
//     2^S2_XSHIFT, the weights' rows times 2^t_n, the epilogue multiplies by g.oscale[n]; g.range_flag is raised when an activation leaves fp16's range)
template <bool PRO, bool DUAL, bool RES, bool POOL, int NCB = 2, int RB = 2, int NP = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NP == 2 ? SUO_X3_F16_WAVES : 2))) void gemm_bf16x3_kernel(const GemmArgs g, const uint16_t* __restrict__ Wp) {
    static_assert(RB == 2 || !POOL, "the pooled epilogue is laid out for 128-row tiles");
    constexpr int X3_BM = 64 * RB, X3_NR = X3_BM / X3_RPP;                    // rows of the tile; staging passes
    static_assert(X3_NR >= 1, "64-row tiles need the 16-wide k-step");
    constexpr int PLANE = X3_BM * X3_PITCH;                                   // bf16 elements of one plane
    constexpr int EPI_BYTES = 4 * 32 * 36 * 4 + (POOL ? 16384 : 0);           // the epilogue's four transposition patches (+ the pooled rows' exchange)
    constexpr int SFL = NP * PLANE >= EPI_BYTES / 4 ? NP * PLANE : EPI_BYTES / 4;      // uint16 per stage: two fp16 planes can be smaller than the patches
    __shared__ __attribute__((aligned(16))) uint16_t S[2][SFL];               // [stage][plane][row][pitch] = 36 864 bytes at NP = 3 (the epilogue re-uses it)
    __shared__ __attribute__((aligned(16))) float P[2][512];                  // prologue scale / shift (K1 <= 512)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int M = g.M, K = g.K1 + (DUAL ? g.K2 : 0), ldo = g.ldo;
    if (PRO) {
        // (NP = 2: the activation scale rides in the prologue -- fmaf(x, 16 a, 16 b) = 16 fmaf(x, a, b) exactly, and relu commutes with it)
        for (int k = tid; k < g.K1; k += 256) { P[0][k] = NP == 2 ? g.pro_scale[k] * S2_XSCALE : g.pro_scale[k]; P[1][k] = NP == 2 ? g.pro_shift[k] * S2_XSCALE : g.pro_shift[k]; }
    }
    // tile = (row tile, column tile of 128); the column tiles of a row tile are neighbours (the second one finds the activations in L2)
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);          // XCD-aware tile order
    static_assert(NCB == 2 || !POOL, "the pooled epilogue is laid out for 128-column tiles");
    constexpr int BN = 64 * NCB;                                              // columns of the tile
    const int ntn = g.N / BN, NBT = g.N >> 5;
    const int tn = bid % ntn, mt = bid / ntn, m0 = mt * X3_BM;
    // pixel (row of the operands) of tile row r
    int pool_base = 0;                                                        // POOL: pixel of tile row 0; tile row r -> pool_base + (r >> 6) * W + (r & 63)
    if (POOL) {
        const int tpr = g.pool_W >> 6, tpc = (g.pool_H >> 1) * tpr;           // tiles per image row pair, per crop
        const int l = mt / tpc, rr = mt - l * tpc, yp = rr / tpr, xb = rr - yp * tpr;
        pool_base = (l * g.pool_H + 2 * yp) * g.pool_W + 64 * xb;
    }
    auto pixel_of = [&](int r) -> int { return POOL ? pool_base + (r >> 6) * g.pool_W + (r & 63) : m0 + r; };
    const int ns1 = g.K1 / X3_BK, nsteps = K / X3_BK;
    const __amdgpu_buffer_rsrc_t a1_srd = make_srd(g.A1, (size_t)M * g.lda1 * sizeof(float));
    const __amdgpu_buffer_rsrc_t a2_srd = make_srd(DUAL ? g.A2 : g.A1, DUAL ? (size_t)M * g.lda2 * sizeof(float) : 0);
    const __amdgpu_buffer_rsrc_t w_srd = make_srd(Wp, (size_t)g.N * K * NP * sizeof(uint16_t));
    // staging roles per k-step: rows tid / LPR + RPP i, the 4 floats at k = 4 (tid % LPR): the LPR lanes of a row fetch its 64 / 128 contiguous bytes
    const int ar = tid / X3_LPR, aq = tid % X3_LPR;
    int avoff1[X3_NR], avoff2[X3_NR];
    bool rok[X3_NR];
#pragma unroll
    for (int i = 0; i < X3_NR; ++i) {
        const int row = pixel_of(X3_RPP * i + ar);
        rok[i] = row < M;
        avoff1[i] = row < M ? (row * g.lda1 + 4 * aq) * 4 : BUF_OOB;          // rows past M read zeros (and the prologue's result is zeroed below)
        avoff2[i] = DUAL && row < M ? (row * g.lda2 + 4 * aq) * 4 : BUF_OOB;
    }
    const int wvoff = lane * 16;
    x3_f32x4 araw[X3_ASLOTS][X3_NR];
    constexpr int BSL = NP == 2 ? SUO_X3_F16_BSLOTS : X3_BSLOTS;          // weight k-groups in flight
    x3_u32x4 braw[BSL][NCB][NP];
    auto requestA = [&](int ks, int slot) {
        if (!DUAL || ks < ns1) {
#pragma unroll
            for (int i = 0; i < X3_NR; ++i) araw[slot][i] = buf_load(a1_srd, avoff1[i], ks * X3_BK * 4);
        } else {                                                              // second K segment (conv4 on the block's input)
#pragma unroll
            for (int i = 0; i < X3_NR; ++i) araw[slot][i] = buf_load(a2_srd, avoff2[i], (ks - ns1) * X3_BK * 4);
        }
    };
    auto requestB = [&](int kg, int slot) {                                   // kg: 16-wide k-group
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                braw[slot][cb][p] = __builtin_bit_cast(x3_u32x4, buf_load(w_srd, wvoff + p * 1024, ((kg * NBT + 2 * NCB * tn + NCB * wn + cb) * NP) * 1024));
    };
    float xmax = 0.f;                                                         // NP = 2: largest scaled magnitude this lane split (range guard)
    auto split_store = [&](int ks, int slot, int stage) {
        uint16_t* As = &S[stage][0];
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        x3_f32x4 sc, sh;
        if (PRO) { sc = *(const x3_f32x4*)&P[0][ks * X3_BK + 4 * aq]; sh = *(const x3_f32x4*)&P[1][ks * X3_BK + 4 * aq]; }      // (single K segment only: checked by the launcher)
#pragma unroll
        for (int i = 0; i < X3_NR; ++i) {
            float x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = araw[slot][i][t];
            if (PRO) {
#pragma unroll
                for (int t = 0; t < 4; ++t) x[t] = rok[i] ? fmaxf(fmaf(x[t], sc[t], sh[t]), 0.f) : 0.f;
            }
            if constexpr (NP == 3) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const unsigned q0 = s3_pack_rn(x[0], x[1]), q1 = s3_pack_rn(x[2], x[3]);      // (p == 2: the conversion is exact)
                    *(u32x2*)&As[p * PLANE + (X3_RPP * i + ar) * X3_PITCH + 4 * aq] = u32x2{q0, q1};
                    if (p < 2) { x[0] -= s3_lo(q0); x[1] -= s3_hi(q0); x[2] -= s3_lo(q1); x[3] -= s3_hi(q1); }      // exact residuals
                }
            } else {
                if (!PRO) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) x[t] *= S2_XSCALE;            // (exact; with a prologue the scale is in sc / sh)
                }
                xmax = s2_track(s2_track(xmax, x[0], x[1]), x[2], x[3]);
                const unsigned h0 = s2_pack_rn(x[0], x[1]), h1 = s2_pack_rn(x[2], x[3]);
                *(u32x2*)&As[(X3_RPP * i + ar) * X3_PITCH + 4 * aq] = u32x2{h0, h1};
                const unsigned l0 = s2_lo_pack(x[0], x[1], h0), l1 = s2_lo_pack(x[2], x[3], h1);      // exact residuals, rounded once
                *(u32x2*)&As[PLANE + (X3_RPP * i + ar) * X3_PITCH + 4 * aq] = u32x2{l0, l1};
            }
        }
    };
    x3_f32x16 acc[RB][NCB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int ngroups = nsteps * X3_GH;
    // prologue: the first ASLOTS activation steps and BSLOTS weight groups in flight, step 0 split into stage 0
#pragma unroll
    for (int u = 0; u < X3_ASLOTS; ++u) requestA(u < nsteps ? u : nsteps - 1, u);
#pragma unroll
    for (int u = 0; u < BSL; ++u) requestB(u < ngroups ? u : ngroups - 1, u);
    __syncthreads();                                                          // (scale / shift staged)
    split_store(0, 0, 0);
    constexpr int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};      // six cross terms, smallest first
    constexpr int UI[3] = {0, 1, 0}, UJ[3] = {1, 0, 0};                       // NP = 2: hi lo, lo hi, hi hi
    const int ko = 8 * (lane >> 5);
    for (int ks0 = 0; ks0 < nsteps; ks0 += X3_ASLOTS) {
#pragma unroll
        for (int u = 0; u < X3_ASLOTS; ++u) {
            const int ks = ks0 + u;
            __syncthreads();                                                  // stage u & 1 complete; every wave is past its reads of the other stage
            // this step's A fragments are requested first: their LDS latency passes under the split of the next step's activations
            const uint16_t* As = &S[u & 1][0];
            x3_bf16x8 af[X3_GH][RB][NP];
#pragma unroll
            for (int h = 0; h < X3_GH; ++h)
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) af[h][rb][p] = *(const x3_bf16x8*)&As[p * PLANE + (32 * RB * wm + 32 * rb + (lane & 31)) * X3_PITCH + ko + 16 * h];
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < nsteps) split_store(ks + 1, (u + 1) % X3_ASLOTS, (u + 1) & 1);
            requestA(ks + X3_ASLOTS < nsteps ? ks + X3_ASLOTS : nsteps - 1, u);
#pragma unroll
            for (int h = 0; h < X3_GH; ++h) {
                // weights of this k-group out of their ring slot, then the slot's next request
                const int slot = (u * X3_GH + h) % BSL, kg = ks * X3_GH + h;
                x3_u32x4 bw[NCB][NP];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int p = 0; p < NP; ++p) bw[cb][p] = braw[slot][cb][p];
                requestB(kg + BSL < ngroups ? kg + BSL : ngroups - 1, slot);
#pragma unroll
                for (int t = 0; t < (NP == 3 ? 6 : 3); ++t)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb) {
                            if constexpr (NP == 3)
                                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[h][rb][TI[t]], __builtin_bit_cast(x3_bf16x8, bw[cb][TJ[t]]), acc[rb][cb], 0, 0, 0);
                            else
                                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x3_f16x8, af[h][rb][UI[t]]), __builtin_bit_cast(x3_f16x8, bw[cb][UJ[t]]), acc[rb][cb], 0, 0, 0);
                        }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();                                                          // the stages are free: the epilogue's patches live there
    if constexpr (NP == 2) s2_raise(g.range_flag, xmax);
    float* T = reinterpret_cast<float*>(&S[0][0]) + w * (32 * 36);
    x3_f32x4* PX = reinterpret_cast<x3_f32x4*>(reinterpret_cast<float*>(&S[0][0]) + 4 * 32 * 36);      // POOL: 16 KB behind the four patches
    x3_f32x4 hold[POOL ? 2 : 1][POOL ? 2 : 1][POOL ? 4 : 1];
    const __amdgpu_buffer_rsrc_t o_srd = make_srd((POOL && !g.out) ? (float*)g.pool_out : g.out, (POOL && !g.out) ? 0 : (size_t)M * ldo * sizeof(float));
    const __amdgpu_buffer_rsrc_t r_srd = make_srd(RES ? g.R : g.out, RES ? (size_t)M * g.ldr * sizeof(float) : 0);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int col = BN * tn + 32 * NCB * wn + 32 * cb + (lane & 7) * 4;
            x3_f32x4 bv = x3_f32x4{0.f, 0.f, 0.f, 0.f};
            if (g.bias) bv = *(const x3_f32x4*)(g.bias + col);
            x3_f32x4 osc = x3_f32x4{1.f, 1.f, 1.f, 1.f};
            if constexpr (NP == 2) osc = *(const x3_f32x4*)(g.oscale + col);
            x3_f32x4 rv[RES ? 4 : 1];
            if (RES) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = pixel_of(32 * RB * wm + 32 * rb + (lane >> 3) + 8 * k);
                    rv[k] = buf_load(r_srd, row < M ? (row * g.ldr + col) * 4 : BUF_OOB, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) T[x3_acc_row(r, lane) * 36 + (lane & 31)] = acc[rb][cb][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = pixel_of(32 * RB * wm + 32 * rb + (lane >> 3) + 8 * k);
                x3_f32x4 o = *(const x3_f32x4*)&T[((lane >> 3) + 8 * k) * 36 + (lane & 7) * 4];
                if constexpr (NP == 2) o *= osc;                              // back to scale: an exact power of two per column
                o += bv;
                if (RES) o += rv[k];                                          // (bias, then the residual: the order of the fp32 kernels)
                if (g.relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = fmaxf(o[q], 0.f);
                }
                if (!POOL || g.out) buf_store(o, o_srd, row < M ? (row * ldo + col) * 4 : BUF_OOB);
                if (POOL) {
                    x3_f32x4 hm;                                              // max over image columns j, j ^ 1 (8 lanes apart within a row of 16: row_ror:8)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        hm[q] = fmaxf(o[q], __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(o[q]), 0x128, 0xf, 0xf, false)));
                    if (wm == 0) hold[rb][cb][k] = hm;
                    else if (!(lane & 8)) PX[(((wn * 2 + rb) * 2 + cb) * 4 + k) * 32 + (lane >> 4) * 8 + (lane & 7)] = hm;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    if (POOL) {
        __syncthreads();
        if (wm == 0 && !(lane & 8)) {
            const __amdgpu_buffer_rsrc_t p_srd = make_srd(g.pool_out, (size_t)(M >> 2) * ldo * sizeof(float));
            const int pw = g.pool_W >> 1;
            const int prow0 = (pool_base / g.pool_W >> 1) * pw + ((pool_base % g.pool_W) >> 1);      // pooled pixel of tile row 0 (l * H / 2 + yp folded: H even)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int j = 32 * rb + (lane >> 3) + 8 * k;          // even: the pair (j, j + 1)
                        const int col = 128 * tn + 64 * wn + 32 * cb + (lane & 7) * 4;
                        const x3_f32x4 other = PX[(((wn * 2 + rb) * 2 + cb) * 4 + k) * 32 + (lane >> 4) * 8 + (lane & 7)];
                        x3_f32x4 v;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = fmaxf(hold[rb][cb][k][q], other[q]);
                        buf_store(v, p_srd, ((prow0 + (j >> 1)) * ldo + col) * 4);
                    }
        }
    }
}

// Two 1x1 convolutions in ONE launch where the tensor between them has a single reader: the last stack's  logits = tmpOut(relu(bn(lin(x))))  (hg.py:106-111; 256 -> 256 -> 41
// channels at 64x64).  Separately the 256-channel tensor is written (1 GB at 256 crops) and read back by the second launch; here a workgroup keeps its 64 rows of it in LDS
// as the two fp16 planes the second product wants anyway.  Two-term fp16 form only (csrc/f16x2.h).
//   stage 1, per 128-column half tn = 0, 1: exactly gemm_bf16x3_kernel<false, false, false, false, 2, 1, 2>'s loop and epilogue arithmetic (same k order, same term order:
//            y1 is bit-identical to that launch's output); relu(y1) times 16, split, into Y[plane][64 rows][128 columns] (16-byte chunks XOR-swizzled by the row);
//   stage 2: acc2 (64 rows x 64 columns, one 32 x 32 block per wave) += Y-half (K = 128, eight k-steps) x W2 -- over the two halves in ascending k: bit-identical to the
//            fp16 GEMM launched on the stored tensor;  epilogue: per-channel rescale + bias, NCHW store of the first n_valid channels.
// LDS 18 + 32 KB, three workgroups per CU like the plain fp16 GEMM.
struct GemmChainArgs {
    const float* A; int lda, M;                           // [M, 256] rows
    const float* bias1; const float* osc1;                // [256]: stage 1 (BatchNorm folded), its per-column factors
    const float* bias2; const float* osc2;                // [64]
    float* out; int n_valid, hw;                          // out[(row / hw) * n_valid * hw + ch * hw + row % hw], ch < n_valid
    unsigned* range_flag;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void gemm_chain_head_kernel(const GemmChainArgs g, const uint16_t* __restrict__ W1, const uint16_t* __restrict__ W2) {
    static_assert(X3_BK == 16, "written for the 16-wide k-step");
    constexpr int NP = 2, NCB = 2, BM = 64, K1 = 256, N1 = 256, NBT1 = N1 / 32, NB2 = 2;
    constexpr int PLANE = BM * X3_PITCH;
    constexpr int SFL = 4 * 32 * 36 * 4 / 4;                                  // uint16 per stage: the two stages together hold the epilogue's four transposition patches (18 KB)
    static_assert(NP * PLANE <= SFL, "stage");
    __shared__ __attribute__((aligned(16))) uint16_t S[2][SFL];
    __shared__ __attribute__((aligned(16))) uint16_t Y[NP][BM * 128];         // relu(y1) * 16 of the current half: [plane][row][16 chunks of 8, chunk ^ (row & 15)]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int M = g.M;
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);          // XCD-aware tile order
    const int m0 = bid * BM;
    const __amdgpu_buffer_rsrc_t a_srd = make_srd(g.A, (size_t)M * g.lda * sizeof(float));
    const __amdgpu_buffer_rsrc_t w1_srd = make_srd(W1, (size_t)N1 * K1 * NP * sizeof(uint16_t));
    const __amdgpu_buffer_rsrc_t w2_srd = make_srd(W2, (size_t)64 * N1 * NP * sizeof(uint16_t));
    const int ar = tid / X3_LPR, aq = tid % X3_LPR;                           // staging: row ar, the 4 floats at k = 4 aq of a k-step
    const int arow = m0 + ar;
    const int avoff = arow < M ? (arow * g.lda + 4 * aq) * 4 : BUF_OOB;
    const int wvoff = lane * 16;
    constexpr int BSL = SUO_X3_F16_BSLOTS, nsteps = K1 / X3_BK;
    constexpr int UI[3] = {0, 1, 0}, UJ[3] = {1, 0, 0};                       // hi lo, lo hi, hi hi
    const int ko = 8 * (lane >> 5);
    float xmax = 0.f;
    x3_f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    for (int tn = 0; tn < 2; ++tn) {
        x3_f32x4 araw[X3_ASLOTS];
        x3_u32x4 braw[BSL][NCB][NP];
        auto requestA = [&](int ks, int slot) { araw[slot] = buf_load(a_srd, avoff, ks * X3_BK * 4); };
        auto requestB = [&](int kg, int slot) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    braw[slot][cb][p] = __builtin_bit_cast(x3_u32x4, buf_load(w1_srd, wvoff + p * 1024, ((kg * NBT1 + 2 * NCB * tn + NCB * wn + cb) * NP) * 1024));
        };
        auto split_store = [&](int slot, int stage) {
            uint16_t* As = &S[stage][0];
            float x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = araw[slot][t] * S2_XSCALE;
            xmax = s2_track(s2_track(xmax, x[0], x[1]), x[2], x[3]);
            const unsigned h0 = s2_pack_rn(x[0], x[1]), h1 = s2_pack_rn(x[2], x[3]);
            *(u32x2*)&As[ar * X3_PITCH + 4 * aq] = u32x2{h0, h1};
            const unsigned l0 = s2_lo_pack(x[0], x[1], h0), l1 = s2_lo_pack(x[2], x[3], h1);
            *(u32x2*)&As[PLANE + ar * X3_PITCH + 4 * aq] = u32x2{l0, l1};
        };
        x3_f32x16 acc[NCB];
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int u = 0; u < X3_ASLOTS; ++u) requestA(u, u);
#pragma unroll
        for (int u = 0; u < BSL; ++u) requestB(u, u);
        split_store(0, 0);
        for (int ks0 = 0; ks0 < nsteps; ks0 += X3_ASLOTS) {
#pragma unroll
            for (int u = 0; u < X3_ASLOTS; ++u) {
                const int ks = ks0 + u;
                __syncthreads();
                const uint16_t* As = &S[u & 1][0];
                x3_f16x8 af[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) af[p] = *(const x3_f16x8*)&As[p * PLANE + (32 * wm + (lane & 31)) * X3_PITCH + ko];
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < nsteps) split_store((u + 1) % X3_ASLOTS, (u + 1) & 1);
                requestA(ks + X3_ASLOTS < nsteps ? ks + X3_ASLOTS : nsteps - 1, u);
                const int slot = u % BSL;
                x3_u32x4 bw[NCB][NP];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int p = 0; p < NP; ++p) bw[cb][p] = braw[slot][cb][p];
                requestB(ks + BSL < nsteps ? ks + BSL : nsteps - 1, slot);
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[UI[t]], __builtin_bit_cast(x3_f16x8, bw[cb][UJ[t]]), acc[cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                                      // the stages are free (the epilogue's patches live there); the previous half's Y has been consumed
        // the second product's weights for this half (k-steps 8 tn .. 8 tn + 7 of W2, n-tile wn), requested now: they arrive under the epilogue
        x3_u32x4 b2[8][NP];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int p = 0; p < NP; ++p) b2[kk][p] = __builtin_bit_cast(x3_u32x4, buf_load(w2_srd, wvoff + p * 1024, (((8 * tn + kk) * NB2 + wn) * NP) * 1024));
        float* T = reinterpret_cast<float*>(&S[0][0]) + w * (32 * 36);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int lcol = 32 * NCB * wn + 32 * cb + (lane & 7) * 4;       // column within the half
            const x3_f32x4 bv = *(const x3_f32x4*)(g.bias1 + 128 * tn + lcol), osc = *(const x3_f32x4*)(g.osc1 + 128 * tn + lcol);
#pragma unroll
            for (int r = 0; r < 16; ++r) T[x3_acc_row(r, lane) * 36 + (lane & 31)] = acc[cb][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = 32 * wm + (lane >> 3) + 8 * k;               // row of the tile
                x3_f32x4 o = *(const x3_f32x4*)&T[((lane >> 3) + 8 * k) * 36 + (lane & 7) * 4];
                o *= osc;
                o += bv;
                float y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) y[q] = fmaxf(o[q], 0.f) * S2_XSCALE;       // what the separate launch stores, times 2^S2_XSHIFT for the split
                xmax = s2_track(s2_track(xmax, y[0], y[1]), y[2], y[3]);
                const unsigned h0 = s2_pack_rn(y[0], y[1]), h1 = s2_pack_rn(y[2], y[3]);
                const unsigned l0 = s2_lo_pack(y[0], y[1], h0), l1 = s2_lo_pack(y[2], y[3], h1);
                const int at = row * 128 + (((lcol >> 3) ^ (row & 15)) << 3) + (lcol & 7);
                *(u32x2*)&Y[0][at] = u32x2{h0, h1};
                *(u32x2*)&Y[1][at] = u32x2{l0, l1};
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();                                                      // Y of this half complete
        {
            const int yrow = 32 * wm + (lane & 31);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int chunk = (2 * kk + (lane >> 5)) ^ (yrow & 15);
                x3_f16x8 af[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) af[p] = *(const x3_f16x8*)&Y[p][yrow * 128 + chunk * 8];
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[UI[t]], __builtin_bit_cast(x3_f16x8, b2[kk][UJ[t]]), acc2, 0, 0, 0);
            }
        }
        // (the next half's first barrier inside its k-loop comes after its first split_store into stage 0 / its reads of nothing of Y: Y is rewritten only in the next
        //  epilogue, behind the k-loop's barriers; the patches T of this epilogue are dead)
    }
    s2_raise(g.range_flag, xmax);
    // NCHW store straight from the accumulator: register group q holds 4 consecutive rows (pixels) of channel ch
    const int ch = 32 * wn + (lane & 31);
    if (ch < g.n_valid) {
        const float oc = g.osc2[ch], bb = g.bias2[ch];
        const __amdgpu_buffer_rsrc_t o_srd = make_srd(g.out, (size_t)(M / g.hw) * g.n_valid * g.hw * sizeof(float));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + 32 * wm + 8 * q + 4 * (lane >> 5);
            x3_f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaf(acc2[4 * q + j], oc, bb);
            const int img = row / g.hw, pix = row - img * g.hw;
            buf_store(o, o_srd, row < M ? ((img * g.n_valid + ch) * g.hw + pix) * 4 : BUF_OOB);
        }
    }
}

bool gemm_chain_head_takes(int M, int lda, int n_valid, int hw) {
    return M > 0 && M % 64 == 0 && lda % 4 == 0 && n_valid > 0 && n_valid <= 64 && hw > 0 && hw % 64 == 0 && M % hw == 0 && (size_t)M * lda * 4 < ((size_t)1 << 31);
}

// logits (NCHW, n_valid channels) = W2 relu(W1 A^T + bias1) + bias2 for A [M, 256]; W1h / W2h = pack_gemm_weight_f16x2 of [256][256] / [64][256] (rows beyond n_valid zero)
int launch_gemm_chain_head(const float* A, int lda, int M, const uint16_t* W1h, const float* osc1, const float* bias1, const uint16_t* W2h, const float* osc2, const float* bias2,
                           float* out, int n_valid, int hw, unsigned* range_flag, hipStream_t s) {
    if (!A || !W1h || !osc1 || !bias1 || !W2h || !osc2 || !bias2 || !out || !range_flag || !gemm_chain_head_takes(M, lda, n_valid, hw)) {
        suo_set_error("gemm_chain_head: unsupported arguments (M=%d lda=%d n_valid=%d hw=%d)", M, lda, n_valid, hw);
        return SUO_ERR_ARG;
    }
    GemmChainArgs g = {A, lda, M, bias1, osc1, bias2, osc2, out, n_valid, hw, range_flag};
    hipLaunchKernelGGL(gemm_chain_head_kernel, dim3(M / 64), dim3(256), 0, s, g, W1h, W2h);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

// out[M, N] = [relu]( [relu(A1 * scale + shift) or A1] W1^T + A2 W2^T + bias + R ): N a multiple of 128 (or 64, un-pooled), K1 and K2 multiples of 64 (K2 may be 0),
// the prologue only without a second segment; Wx3 = pack_gemm_weight_bf16x3 of the row-concatenated [W1 | W2] (N rows, K1 + K2 columns)
bool gemm_bf16x3_takes(const GemmArgs& g) {
    const size_t lim = (size_t)1 << 31;
    return g.N > 0 && (g.N % 128 == 0 || (g.N == 64 && !g.pool_out)) && g.n_valid == g.N && g.K1 > 0 && g.K1 % 64 == 0 && g.K2 % 64 == 0 && g.K1 <= 512 && !g.nchw_hw && (g.pool_out ? (g.pool_W % 64 == 0 && g.pool_H % 2 == 0 && g.M % (g.pool_H * g.pool_W) == 0) : g.out != nullptr) &&
           (!g.K2 || (!g.pro_scale && g.A2 && g.lda2 % 4 == 0)) && g.lda1 % 4 == 0 && g.ldo % 4 == 0 && (!g.R || g.ldr % 4 == 0) &&
           ((g.pro_scale == nullptr) == (g.pro_shift == nullptr)) && (size_t)g.M * g.lda1 * 4 < lim && (!g.out || (size_t)g.M * g.ldo * 4 < lim) &&
           (!g.K2 || (size_t)g.M * g.lda2 * 4 < lim) && (!g.R || (size_t)g.M * g.ldr * 4 < lim);
}

template <int NP>
static int launch_gemm_split(const GemmArgs& g, const uint16_t* Wx3, hipStream_t s) {
    if (!gemm_bf16x3_takes(g) || !Wx3 || (NP == 2 && (!g.oscale || !g.range_flag))) {
        suo_set_error("gemm_%s: unsupported shape (N=%d K=%d+%d M=%d)", NP == 3 ? "bf16x3" : "f16x2", g.N, g.K1, g.K2, g.M);
        return SUO_ERR_ARG;
    }
    const bool n64 = g.N % 128 != 0;                                          // 64 output channels: 64-column tiles
    // launches of fewer 128-row tiles than four per CU (the one-frame call: conv1 / lin at 64x64 for 8 crops = 256 / 512 tiles) take 64-row tiles: 2.206 -> 2.174 ms per network call, one frame in flight 2.857 -> 2.825 (tools/latency_ab.sh)
    static const long rb1_max_tiles = (long)SUO_TUNE("SUO_X3_ROWS64_MAX_TILES", 1023);
    const long tiles128 = (long)((g.M + 127) / 128) * (n64 ? 1 : g.N / 128);
    const bool rows64 = !n64 && !g.pool_out && X3_BK == 16 && tiles128 <= rb1_max_tiles;
    const int tiles = rows64 ? ((g.M + 63) / 64) * (g.N / 128) : (int)tiles128;
#define X3_LAUNCH(P_, D_, R_) do { if (g.pool_out) hipLaunchKernelGGL((gemm_bf16x3_kernel<P_, D_, R_, true, 2, 2, NP>), dim3(tiles), dim3(256), 0, s, g, Wx3); \
                                   else if (n64) hipLaunchKernelGGL((gemm_bf16x3_kernel<P_, D_, R_, false, 1, 2, NP>), dim3(tiles), dim3(256), 0, s, g, Wx3); \
                                   else if (rows64) hipLaunchKernelGGL((gemm_bf16x3_kernel<P_, D_, R_, false, 2, 1, NP>), dim3(tiles), dim3(256), 0, s, g, Wx3); \
                                   else hipLaunchKernelGGL((gemm_bf16x3_kernel<P_, D_, R_, false, 2, 2, NP>), dim3(tiles), dim3(256), 0, s, g, Wx3); } while (0)
    const bool res = g.R != nullptr;
    if (g.pro_scale) { if (res) X3_LAUNCH(true, false, true); else X3_LAUNCH(true, false, false); }
    else if (g.K2) { if (res) X3_LAUNCH(false, true, true); else X3_LAUNCH(false, true, false); }
    else { if (res) X3_LAUNCH(false, false, true); else X3_LAUNCH(false, false, false); }
#undef X3_LAUNCH
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

int launch_gemm_bf16x3_args(const GemmArgs& g, const uint16_t* Wx3, hipStream_t s) { return launch_gemm_split<3>(g, Wx3, s); }
// W16 = pack_gemm_weight_f16x2's planes; g.oscale its per-column factors, g.range_flag the guard flag (csrc/f16x2.h)
int launch_gemm_f16x2_args(const GemmArgs& g, const uint16_t* W16, hipStream_t s) { return launch_gemm_split<2>(g, W16, s); }

int launch_gemm_bf16x3(const float* A, int lda, int K, const float* pro_scale, const float* pro_shift, const uint16_t* Wp, const float* bias,
                       float* out, int ldo, int M, int N, int relu, hipStream_t s) {
    GemmArgs g = {};
    g.A1 = A; g.lda1 = lda; g.K1 = K; g.pro_scale = pro_scale; g.pro_shift = pro_shift; g.bias = bias; g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.n_valid = N;
    g.relu = relu;
    return launch_gemm_bf16x3_args(g, Wp, s);
}

}  // namespace suo

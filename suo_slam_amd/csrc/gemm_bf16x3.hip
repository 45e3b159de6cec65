// EXPERIMENTAL (not on the product path; bench.py reports it beside the fp32 line, never instead of it):
// fp32-accurate 1x1 convolution on the bf16 matrix pipe by 3-way operand splitting.
//
// gfx950 runs fp32 MFMAs at the vector rate (157 TFLOP/s) and bf16 MFMAs 16x faster (2.5 PFLOP/s dense).  An fp32 number is exactly
// the sum of three bf16 numbers (8 significand bits each: x0 = hi(x), x1 = hi(x - x0), x2 = hi(x - x0 - x1), every subtraction exact),
// a bf16 x bf16 product is exact in fp32, and v_mfma_f32_32x32x16_bf16 accumulates in fp32.  So
//     x * w  =  sum over i + j <= 2 of x_i * w_j   +   O(2^-24 |x w|)            (6 of the 9 cross terms)
// costs 6 bf16 MFMAs of K = 16 (6 x 32 = 192 cycles per SIMD) where the fp32 form needs 8 MFMAs of K = 2 (8 x 64 = 512 cycles): 2.67x
// fewer matrix-pipe cycles per MAC at fp32 accuracy -- the only lever above the fp32-MFMA roof.
//
// This file is the prototype VERDICT round 2 asked for, on the largest 1x1 shape of the network (Residual.conv1: BN + ReLU prologue,
// K = 256 -> N = 128, + ReLU):  out[M, N] = relu( relu(A * scale + shift) W^T + bias ).
//   * weights are split on the host (pack_gemm_weight_bf16x3) and packed per 32-wide K step as [plane][n][32] bf16;
//   * activations are split while they are staged: global fp32 -> registers (prefetched one K step ahead) -> prologue -> truncation
//     split (x & 0xffff0000 is the exact leading bf16 of x toward zero; the residuals keep the sign) -> three bf16 planes in LDS;
//   * workgroup tile 128 x 128, four waves as 2 x 2 (64 x 64 each = 2 x 2 accumulators), K step 32 = 2 MFMA k-steps; the six terms
//     of a k-step are issued smallest first;
//   * 61 KB of LDS and < 256 registers: two workgroups per CU, so one workgroup's split phase (VALU + LDS writes, which cannot
//     overlap its own MFMAs on this part) runs under the other's MFMA phase.
#include <stdlib.h>
#include <string.h>

#include "suo_internal.h"

namespace suo {

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef float x3_f32x16 __attribute__((ext_vector_type(16)));
typedef float x3_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned x3_u32x4 __attribute__((ext_vector_type(4)));

constexpr int X3_BM = 128, X3_BN = 128, X3_BK = 16, X3_PITCH = 24;      // LDS row pitch in bf16 (48 bytes: conflict-free 16-byte fragment reads)

// host: W[N][K] fp32 -> out[K/16][3][N][16] bf16 (as uint16), split by truncation like the device does
void pack_gemm_weight_bf16x3(const float* W, int N, int K, uint16_t* out) {
    for (int ks = 0; ks < K / X3_BK; ++ks)
        for (int n = 0; n < N; ++n)
            for (int kk = 0; kk < X3_BK; ++kk) {
                float x = W[(size_t)n * K + ks * X3_BK + kk];
                for (int p = 0; p < 3; ++p) {
                    uint32_t u;
                    memcpy(&u, &x, 4);
                    u &= 0xffff0000u;
                    float hi;
                    memcpy(&hi, &u, 4);
                    out[(((size_t)ks * 3 + p) * N + n) * X3_BK + kk] = (uint16_t)(u >> 16);
                    x -= hi;                                                   // exact
                }
            }
}

__device__ __forceinline__ unsigned x3_pack_hi(float a, float b) {          // the leading bf16 of a (low half) and of b (high half)
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float x3_hi(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_bf16x3_kernel(
    const float* __restrict__ A, int lda, int K, const float* __restrict__ pro_scale, const float* __restrict__ pro_shift,
    const uint16_t* __restrict__ Wp, const float* __restrict__ bias, float* __restrict__ out, int ldo, int M, int relu) {
    // two stages of {A planes, B planes}: stage s is written (split of step k) while the MFMAs of step k - 1 read stage s ^ 1
    constexpr int PLANE = X3_BM * X3_PITCH;                                   // bf16 elements of one plane
    __shared__ __attribute__((aligned(16))) uint16_t S[2][2][3 * PLANE];      // [stage][A / B][plane][row][pitch]  = 73 728 bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    // PERSISTENT workgroups walk the tile list (tile = 128 rows), two per CU: a tile's stores drain under the next tile's first steps
    const int ntiles = (M + X3_BM - 1) / X3_BM;
    const int nsteps = K / X3_BK;
    // staging roles per 16-wide K step: A -- rows tid / 4 and 64 + tid / 4, the 4 floats at k = 4 (tid & 3): four adjacent lanes read
    // the 64 contiguous bytes a row contributes to the step (with two lanes per row and 32 bytes each, every 64-byte segment was
    // requested by two separate instructions); B -- three 16-byte chunks (one per plane)
    const int ar = tid >> 2, aq = tid & 3;
    x3_f32x4 areg[4][2], sreg[4], hreg[4];
    x3_u32x4 breg[2][3];
    bool arow_ok[4][2] = {{false, false}, {false, false}, {false, false}, {false, false}};
    const int nwg = gridDim.x;
    auto tile_of = [&](int it) -> int {                                       // XCD-aware: workgroup b lives on XCD b % 8
        const int g = it * nwg + blockIdx.x;
        if (g >= ntiles) return ntiles;
        if ((nwg & 7) == 0 && (ntiles & 7) == 0) { const int per = ntiles >> 3; return (g & 7) * per + (g >> 3); }
        return g;
    };
    // activations come from HBM (1 GB per launch: the kernel is read-bound long before it is MFMA-bound), weights from L2: A is
    // prefetched FOUR steps ahead (4 register sets: 64 KB in flight per CU -- with two steps the chip had 8 MB in flight, about
    // half of what the HBM latency x bandwidth product asks for, and nothing overlapped), B two steps ahead
    auto gloadA = [&](int set, int tile, int ks) {
        const int m0 = tile * X3_BM;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + 64 * i + ar;
            arow_ok[set][i] = row < M;
#ifdef SUO_X3_EXP_NOLOAD
            areg[set][i] = x3_f32x4{(float)(row + ks), 1.f, 2.f, (float)aq};          // timing experiment: no HBM reads
#else
            areg[set][i] = *(const x3_f32x4*)(A + (size_t)(arow_ok[set][i] ? row : m0) * lda + ks * X3_BK + 4 * aq);
#endif
        }
        if (pro_scale) {
            sreg[set] = *(const x3_f32x4*)(pro_scale + ks * X3_BK + 4 * aq);
            hreg[set] = *(const x3_f32x4*)(pro_shift + ks * X3_BK + 4 * aq);
        }
    };
    auto gloadB = [&](int set, int ks) {
        const x3_u32x4* bp = (const x3_u32x4*)(Wp + (size_t)ks * 3 * X3_BN * X3_BK);      // [plane][n][16] bf16: 2 chunks per row
#pragma unroll
        for (int j = 0; j < 3; ++j) breg[set][j] = bp[tid + 256 * j];
    };
    auto sstore = [&](int set, int bset, int stage) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = areg[set][i][t];
                if (pro_scale) v = fmaxf(fmaf(v, sreg[set][t], hreg[set][t]), 0.f);
                x[4 * i + t] = arow_ok[set][i] ? v : 0.f;
            }
        uint16_t* As = &S[stage][0][0];
        uint16_t* Bs = &S[stage][1][0];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                *(u32x2*)&As[p * PLANE + (64 * i + ar) * X3_PITCH + 4 * aq] = u32x2{x3_pack_hi(x[4 * i], x[4 * i + 1]), x3_pack_hi(x[4 * i + 2], x[4 * i + 3])};
            }
            if (p < 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] -= x3_hi(x[e]);              // exact residual
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = tid + 256 * j;                                      // chunk = ((plane * 128 + n) * 2 + half)
            *(x3_u32x4*)&Bs[(c >> 1) * X3_PITCH + (c & 1) * 8] = breg[bset][j];
        }
    };

    int it = 0, tile = tile_of(0);
    if (tile >= ntiles) return;
    int next_tile = tile_of(1);
    // prologue: A steps 0..3 and B steps 0, 1 in flight (the step sequence runs on into the next tile)
    auto stepA = [&](int set, int q) {                                          // q = step number counted from the current tile's step 0
        if (q < nsteps) gloadA(set, tile, q);
        else if (next_tile < ntiles) gloadA(set, next_tile, q - nsteps);
    };
    auto stepB = [&](int set, int q) {
        if (q < nsteps || next_tile < ntiles) gloadB(set, q < nsteps ? q : q - nsteps);
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) stepA(u, u);
#pragma unroll
    for (int u = 0; u < 2; ++u) stepB(u, u);
    while (tile < ntiles) {
        const int m0 = tile * X3_BM;
        x3_f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int ks = 0; ks < nsteps; ks += 4) {
            // four steps per trip so that register sets and LDS stages are compile-time indices (K a multiple of 64)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#ifndef SUO_X3_EXP_NOSPLIT
                sstore(u, u & 1, u & 1);                                      // step ks + u: registers -> split -> LDS stage u & 1
#endif
                stepA(u, ks + u + 4);                                         // its register sets are free for later steps
                stepB(u & 1, ks + u + 2);
                __syncthreads();                                              // stage u complete; every wave is past its reads of stage u (two steps ago)
                const uint16_t* As = &S[u & 1][0][0];
                const uint16_t* Bs = &S[u & 1][1][0];
                x3_bf16x8 af[2][3], bf[2][3];
                const int ko = 8 * (lane >> 5);
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) af[rb][p] = *(const x3_bf16x8*)&As[p * PLANE + (64 * wm + 32 * rb + (lane & 31)) * X3_PITCH + ko];
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) bf[cb][p] = *(const x3_bf16x8*)&Bs[p * PLANE + (64 * wn + 32 * cb + (lane & 31)) * X3_PITCH + ko];
                }
                constexpr int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};      // six cross terms, smallest first
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
                            acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb][TI[t]], bf[cb][TJ[t]], acc[rb][cb], 0, 0, 0);
            }
        }
        // epilogue: bias (+ ReLU), straight from the accumulator layout (for a fixed register, 32 lanes hold 32 consecutive columns of one row)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int col = 64 * wn + 32 * cb + (lane & 31);
                const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 64 * wm + 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = acc[rb][cb][r] + bv;
                    if (relu) v = fmaxf(v, 0.f);
#ifdef SUO_X3_EXP_NOSTORE
                    if (row < M && v == 123456.f) out[(size_t)row * ldo + col] = v;      // timing experiment: (almost) no stores
#else
                    if (row < M) out[(size_t)row * ldo + col] = v;
#endif
                }
            }
        ++it;
        tile = next_tile;
        next_tile = tile_of(it + 1);
    }
}

int launch_gemm_bf16x3(const float* A, int lda, int K, const float* pro_scale, const float* pro_shift, const uint16_t* Wp, const float* bias,
                       float* out, int ldo, int M, int N, int relu, hipStream_t s) {
    if (N != X3_BN || K <= 0 || (K % 64) || M <= 0 || (lda % 4) || ((pro_scale == nullptr) != (pro_shift == nullptr))) {
        suo_set_error("gemm_bf16x3 (prototype): N must be 128 and K a multiple of 64 (N=%d K=%d)", N, K);
        return SUO_ERR_ARG;
    }
    const int tiles = (M + X3_BM - 1) / X3_BM;
    static const int wgs = getenv("SUO_X3_WGS") ? atoi(getenv("SUO_X3_WGS")) : 512;        // two resident workgroups per CU
    hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3(tiles < wgs ? tiles : wgs), dim3(256), 0, s, A, lda, K, pro_scale, pro_shift, Wp, bias, out, ldo, M, relu);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo

// The one-launch Residual block of csrc/res_small.hip with its products on the BF16 matrix pipe at fp32 accuracy (3-way operand split,
// csrc/bf16x3.h) -- what the network launches for the Hourglass blocks at 32x32 and 16x16 when a call holds few crops (one frame per
// call: lib/object_slam.py:1099).
//
// Why: at 8 crops a 32x32 level is 8192 pixels = 32 per CU, and the block is MFMA-bound on the fp32 pipe -- 70 k of the fp32 kernel's 88 k
// cycles are v_mfma_f32_16x16x4_f32 at 32 MAC / clock / SIMD (tools/prof_res_block.py).  Six v_mfma_f32_32x32x16_bf16 (192 pipe cycles
// per 32 x 32 x 16 block) replace sixteen of those (512 cycles): 23 k cycles of matrix work per workgroup.
//
// Workgroup = a 4 x 8 pixel tile of one crop, four waves (one per SIMD), 32 rows per MFMA:
//   1. x tile + halo (60 rows, 64 staged): relu(bn(x)) split into three bf16 planes in LDS (A-operand order: row-major, 16-byte k granules);
//   2. conv1 (256 -> 128): 64 rows x wave w's 32 channels, 16 k-steps x 12 MFMAs; relu(. + b1), zeros outside the map, split -> LDS planes;
//   3. conv2 (3x3): 32 rows, K = 9 taps x 128 channels = 72 k-steps x 6 MFMAs, the A rows of a tap picked per lane from the halo tile;
//      relu(. + b2), split -> LDS planes (over the dead x tile);
//   4. conv3 (128 -> 256): wave w's 64 channels, 8 k-steps x 12 MFMAs; patch -> + b3 + x [+ up] on 16-byte vectors.
// Weights: host-split planes in B-operand order ([k-step][32-channel tile][plane][lane][8 bf16], the layout of pack_gemm_weight_bf16x3),
// each wave streams only its own channels' 16-byte fragments from L2 through a static register ring: 1.28 MB per workgroup.
// LDS 153.6 KB, one workgroup per CU.  Accuracy: tests/test_gpu_res_block.py (fp64; never worse than 2x the fp32-pipe kernel of
// csrc/res_small.hip on the same inputs; per-element bound as tests/test_gpu_x3_accuracy.py).
#include <string.h>

#include "bf16x3.h"
#include "f16x2.h"
#include "buffer_ops.h"
#include "suo_internal.h"

namespace suo {

typedef float r3_f32x4 __attribute__((ext_vector_type(4)));
typedef float r3_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned r3_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned r3_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 r3_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 r3_f16x8 __attribute__((ext_vector_type(8)));

// host: W2[N = 128][C = 128][3][3] (times out_scale[n]) -> [tap][k-step][n-tile][plane][lane][8 bf16] with
//   term `plane` of W2[nb*32 + (lane&31)][16 ks + 8 (lane>>5) + e][tap]
void pack_res_conv3x3_bf16x3(const float* W, const float* out_scale, uint16_t* out) {
    constexpr int N = 128, C = 128, NB = N / 32;
    for (int tap = 0; tap < 9; ++tap)
        for (int n = 0; n < N; ++n)
            for (int c = 0; c < C; ++c) {
                const int ks = c / 16, cc = c % 16, lane = (cc / 8) * 32 + (n % 32), e = cc % 8, nb = n / 32;
                const float sc = out_scale ? out_scale[n] : 1.f;
                uint16_t t[3];
                s3_split_host(W[(((size_t)n * C + c) * 3 + tap / 3) * 3 + tap % 3] * sc, t);
                for (int p = 0; p < 3; ++p) out[(((((size_t)(tap * 8 + ks) * NB + nb) * 3 + p) * 64) + lane) * 8 + e] = t[p];
            }
}

// host, two fp16 planes (csrc/f16x2.h): output channel n times 2^t_n over its 9 x 128 entries; oscale_out[n] = 2^-(t_n + S2_XSHIFT)
void pack_res_conv3x3_f16x2(const float* W, const float* out_scale, uint16_t* out, float* oscale_out) {
    constexpr int N = 128, C = 128, NB = N / 32;
    for (int n = 0; n < N; ++n) {
        const float sc = out_scale ? out_scale[n] : 1.f;
        float mx = 0.f;
        for (int i = 0; i < C * 9; ++i) mx = fmaxf(mx, fabsf(W[(size_t)n * C * 9 + i] * sc));
        const int t = s2_row_shift(mx);
        oscale_out[n] = ldexpf(1.f, -(t + S2_XSHIFT));
        for (int tap = 0; tap < 9; ++tap)
            for (int c = 0; c < C; ++c) {
                const int ks = c / 16, cc = c % 16, lane = (cc / 8) * 32 + (n % 32), e = cc % 8, nb = n / 32;
                uint16_t h[2];
                s2_split_host(ldexpf(W[(((size_t)n * C + c) * 3 + tap / 3) * 3 + tap % 3] * sc, t), h);
                for (int p = 0; p < 2; ++p) out[(((((size_t)(tap * 8 + ks) * NB + nb) * 2 + p) * 64) + lane) * 8 + e] = h[p];
            }
    }
}

__device__ __forceinline__ int r3_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// rows of the 6 x 10 halo tile, interior first (as csrc/res_small.hip): [0, 32) the 4 x 8 output pixels, then top row, bottom row, left, right
__device__ __forceinline__ int r3_row(int hy, int hx) {
    constexpr int TH = 4, TW = 8, IW = 10, T = 32;
    if (hy >= 1 && hy <= TH && hx >= 1 && hx <= TW) return (hy - 1) * TW + (hx - 1);
    if (hy == 0) return T + hx;
    if (hy == TH + 1) return T + IW + hx;
    if (hx == 0) return T + 2 * IW + (hy - 1);
    return T + 2 * IW + TH + (hy - 1);
}
__device__ __forceinline__ bool r3_hyhx(int row, int& hy, int& hx) {
    constexpr int TH = 4, TW = 8, IW = 10, T = 32;
    if (row < T) { hy = row / TW + 1; hx = row % TW + 1; return true; }
    const int q = row - T;
    if (q < IW) { hy = 0; hx = q; return true; }
    if (q < 2 * IW) { hy = TH + 1; hx = q - IW; return true; }
    if (q < 2 * IW + TH) { hy = q - 2 * IW + 1; hx = 0; return true; }
    if (q < 2 * IW + 2 * TH) { hy = q - 2 * IW - TH + 1; hx = IW - 1; return true; }
    hy = hx = 0;
    return false;
}

#ifdef SUO_RS_PROF
#define R3_T(i) do { pt[i] = clock64(); } while (0)
#else
#define R3_T(i) do { } while (0)
#endif

// NP = operand planes: 3 = three bf16 terms, six MFMAs per product block; 2 = two fp16 terms, three MFMAs (csrc/f16x2.h: every activation tile times 2^S2_XSHIFT on its
// way into LDS, weight rows times 2^t_n, accumulators back to scale with a.osc1 / osc2 / osc3, a.range_flag raised beyond fp16's range) -- a third less weight
// traffic per workgroup (0.85 MB instead of 1.28), which is what bounds this kernel
template <bool POOL_IN, bool UP, int NP = 3>
__global__ __launch_bounds__(256) void res_block_x3_kernel(const ResBlockArgs a) {
#ifdef SUO_RS_PROF
    long long pt[10];
    R3_T(0);
#endif
    constexpr int TH = 4, TW = 8, T = 32, C = 256;
    constexpr int XPB = 528, XPL = 64 * XPB;                    // x tile: bytes per row (256 bf16 + 16), per plane (64 rows)
    constexpr int MPB = 272, MPL = 64 * MPB, M2PL = 32 * MPB;    // mid tiles: 128 bf16 + 16 per row; mid1 64 rows, mid2 32 rows per plane
    constexpr int PP = 260;                                     // output patch pitch (floats)
    static_assert(NP * M2PL <= NP * XPL && T * PP * 4 <= NP * MPL, "mid2 re-uses the x tile, the output patch the mid1 tile");
    __shared__ __attribute__((aligned(16))) unsigned char XA[NP * XPL];
    __shared__ __attribute__((aligned(16))) unsigned char M1[NP * MPL];
    unsigned char* M2 = XA;
    float* P3 = reinterpret_cast<float*>(M1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);          // XCD-aware tile order (csrc/conv.hip)
    const int l = bid / (tiles_x * tiles_y);
    bid -= l * tiles_x * tiles_y;
    const int ty0 = bid / tiles_x, tx0 = bid - ty0 * tiles_x;
    const int oy0 = ty0 * TH, ox0 = tx0 * TW;
    const int XW = POOL_IN ? 2 * W : W;
    const size_t xcrop = (size_t)(POOL_IN ? 4 : 1) * H * W * C, ocrop = (size_t)H * W * C;
    const __amdgpu_buffer_rsrc_t x_srd = make_srd(a.x + (size_t)l * xcrop, xcrop * sizeof(float));
    const __amdgpu_buffer_rsrc_t o_srd = make_srd(a.out + (size_t)l * ocrop, ocrop * sizeof(float));
    const __amdgpu_buffer_rsrc_t w1_srd = make_srd(a.W1, (size_t)128 * 256 * NP * sizeof(uint16_t));
    const __amdgpu_buffer_rsrc_t w2_srd = make_srd(a.W2, (size_t)128 * 128 * 9 * NP * sizeof(uint16_t));
    const __amdgpu_buffer_rsrc_t w3_srd = make_srd(a.W3, (size_t)256 * 128 * NP * sizeof(uint16_t));
    float gmax = 0.f;                                           // NP = 2: largest scaled activation this lane split (range guard)

    // ---- weight rings: [k-step][n-tile][plane][lane][16 bytes]; one k-step of one n-tile = 3 KB -------------------------------------
    constexpr int R1 = 4, R2 = 8, R3 = 4, NS1 = 16, NS2 = 72, NS3 = 8;
    const int wv = lane * 16;
    r3_u32x4 ring1[R1][NP], ring2[R2][NP], ring3[R3][2][NP];
    auto load1 = [&](int ks, r3_u32x4 (&b)[NP]) {               // conv1: 4 n-tiles, wave w -> tile w
        const int k = ks < NS1 ? ks : NS1 - 1;
#pragma unroll
        for (int p = 0; p < NP; ++p) b[p] = __builtin_bit_cast(r3_u32x4, buf_load(w1_srd, wv + p * 1024, (k * 4 + w) * NP * 1024));
    };
    auto load2 = [&](int ks, r3_u32x4 (&b)[NP]) {               // conv2: step = tap * 8 + k-step
        const int k = ks < NS2 ? ks : NS2 - 1;
#pragma unroll
        for (int p = 0; p < NP; ++p) b[p] = __builtin_bit_cast(r3_u32x4, buf_load(w2_srd, wv + p * 1024, (k * 4 + w) * NP * 1024));
    };
    auto load3 = [&](int ks, r3_u32x4 (&b)[2][NP]) {            // conv3: 8 n-tiles, wave w -> tiles 2 w, 2 w + 1
        const int k = ks < NS3 ? ks : NS3 - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p) b[j][p] = __builtin_bit_cast(r3_u32x4, buf_load(w3_srd, wv + p * 1024, (k * 8 + 2 * w + j) * NP * 1024));
    };
#pragma unroll
    for (int g = 0; g < R1 - 1; ++g) load1(g, ring1[g]);        // (first touch of the block's weights: under the x staging)

    // ---- 1. stage relu(bn(x)) of tile + halo as three bf16 planes: thread = (rows tid >> 6 + 4 i, channels 4 q .. 4 q + 3) -------------
    const int q = tid & 63;
    {
        r3_f32x4 sc = *(const r3_f32x4*)(a.pro_scale + 4 * q), sh = *(const r3_f32x4*)(a.pro_shift + 4 * q);
        if constexpr (NP == 2) { sc *= S2_XSCALE; sh *= S2_XSCALE; }      // fmaf(x, 16 a, 16 b) = 16 fmaf(x, a, b) exactly
        constexpr int NB = POOL_IN ? 4 : 16;                    // rows per thread in flight (every request before the first use)
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += NB) {
            r3_f32x4 v[NB][POOL_IN ? 4 : 1];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int row = (tid >> 6) + 4 * (i0 + u);
                int hy, hx;
                const bool real = r3_hyhx(row, hy, hx);
                const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                const bool ok = real && iy >= 0 && iy < H && ix >= 0 && ix < W;
                if (POOL_IN) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        v[u][s] = buf_load(x_srd, ok ? (((2 * iy + (s >> 1)) * XW + 2 * ix + (s & 1)) * C + 4 * q) * 4 : BUF_OOB, 0);
                } else {
                    v[u][0] = buf_load(x_srd, ok ? ((iy * XW + ix) * C + 4 * q) * 4 : BUF_OOB, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int row = (tid >> 6) + 4 * (i0 + u);
                r3_f32x4 x = v[u][0];
                if (POOL_IN) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) x[t] = fmaxf(fmaxf(v[u][0][t], v[u][1][t]), fmaxf(v[u][2][t], v[u][3][t]));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) x[t] = fmaxf(fmaf(x[t], sc[t], sh[t]), 0.f);
                if constexpr (NP == 2) {
                    gmax = s2_track(s2_track(gmax, x[0], x[1]), x[2], x[3]);
                    const unsigned h0 = s2_pack_rn(x[0], x[1]), h1 = s2_pack_rn(x[2], x[3]);
                    *(r3_u32x2*)(XA + row * XPB + q * 8) = r3_u32x2{h0, h1};
                    *(r3_u32x2*)(XA + XPL + row * XPB + q * 8) = r3_u32x2{s2_lo_pack(x[0], x[1], h0), s2_lo_pack(x[2], x[3], h1)};
                } else {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const unsigned q0 = s3_pack_rn(x[0], x[1]), q1 = s3_pack_rn(x[2], x[3]);
                    *(r3_u32x2*)(XA + p * XPL + row * XPB + q * 8) = r3_u32x2{q0, q1};
                    if (p < 2) { x[0] -= s3_lo(q0); x[1] -= s3_hi(q0); x[2] -= s3_lo(q1); x[3] -= s3_hi(q1); }
                }
                }
            }
        }
    }
    // which halo rows are pixels of the map: bit m * 16 + r for this lane's accumulator rows of m-tile m
    unsigned vmask = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int hy, hx;
            const bool real = r3_hyhx(m * 32 + r3_acc_row(r, lane), hy, hx);
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            if (real && iy >= 0 && iy < H && ix >= 0 && ix < W) vmask |= 1u << (m * 16 + r);
        }
    const bool ring_any = __builtin_amdgcn_readfirstlane((int)(__ballot((vmask >> 16) != 0) != 0ull)) != 0;      // (4x4 ... maps narrower than a tile: no ring pixel)
    R3_T(1);
    __syncthreads();
    R3_T(2);

    r3_f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    constexpr int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};      // the six cross terms, smallest first
    // two independent accumulation chains side by side, their MFMAs alternating (conv2: even / odd k-steps; conv3: the wave's two n-tiles)
    constexpr int UI[3] = {0, 1, 0}, UJ[3] = {1, 0, 0};                         // NP = 2: hi lo, lo hi, hi hi
    auto mm = [&](const r3_bf16x8 (&f)[NP], const r3_u32x4 (&bw)[NP], int t, r3_f32x16 acc) -> r3_f32x16 {
        if constexpr (NP == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(r3_f16x8, f[UI[t % 3]]), __builtin_bit_cast(r3_f16x8, bw[UJ[t % 3]]), acc, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[TI[t] % NP], __builtin_bit_cast(r3_bf16x8, bw[TJ[t] % NP]), acc, 0, 0, 0);
    };
    constexpr int NT = NP == 2 ? 3 : 6;
    auto mac6x2 = [&](r3_f32x16& accA, const r3_bf16x8 (&fA)[NP], const r3_u32x4 (&bA)[NP], r3_f32x16& accB, const r3_bf16x8 (&fB)[NP], const r3_u32x4 (&bB)[NP]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            accA = mm(fA, bA, t, accA);
            accB = mm(fB, bB, t, accB);
        }
    };
    auto mac6 = [&](r3_f32x16& acc, const r3_bf16x8 (&f)[NP], const r3_u32x4 (&bw)[NP]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc = mm(f, bw, t, acc);
    };

    // ---- 2. conv1: 64 rows x channels [32 w, 32 w + 32) ------------------------------------------------------------------------------
    r3_f32x16 acc1[2] = {zero16, zero16};
    {
        const unsigned char* xa = XA + lr * XPB + lk * 16;
#pragma unroll
        for (int ks = 0; ks < NS1; ++ks) {
            load1(ks + R1 - 1, ring1[(ks + R1 - 1) % R1]);
            r3_bf16x8 af[2][NP];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < NP; ++p) af[m][p] = *(const r3_bf16x8*)(xa + p * XPL + m * 32 * XPB + ks * 32);
            __builtin_amdgcn_sched_barrier(0);
            mac6(acc1[0], af[0], ring1[ks % R1]);                // (two chains side by side measured SLOWER here: 14.0 k vs 9.3 k cycles)
            if (ring_any) mac6(acc1[1], af[1], ring1[ks % R1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    R3_T(3);
#pragma unroll
    for (int g = 0; g < R2 - 2; ++g) load2(g, ring2[g]);        // conv2's first weights travel under the epilogue + barrier
    {   // relu(acc + b1) -> M1 planes (zeros outside the map: Conv2d(padding=1) pads conv2's INPUT)
        const int ch = 32 * w + lr;
        // NP = 2: the accumulator carries 2^(t_n + S2_XSHIFT); conv2's operand is 2^S2_XSHIFT relu(conv1 + b1) = relu(acc 2^-t_n + 2^S2_XSHIFT b1): one fma
        const float b1 = NP == 2 ? a.b1[ch] * S2_XSCALE : a.b1[ch];
        const float c1 = NP == 2 ? a.osc1[ch] * S2_XSCALE : 1.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = (vmask >> (m * 16 + r)) & 1u;
                float v = ok ? (NP == 2 ? fmaxf(fmaf(acc1[m][r], c1, b1), 0.f) : fmaxf(acc1[m][r] + b1, 0.f)) : 0.f;
                unsigned char* d = M1 + (m * 32 + r3_acc_row(r, lane)) * MPB + ch * 2;
                if constexpr (NP == 2) {
                    gmax = fmaxf(gmax, v);
                    const _Float16 hi = (_Float16)v;
                    *reinterpret_cast<_Float16*>(d) = hi;
                    *reinterpret_cast<_Float16*>(d + MPL) = (_Float16)(v - (float)hi);
                    continue;
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const unsigned qq = s3_pack_rn(v, v);
                    *reinterpret_cast<uint16_t*>(d + p * MPL) = (uint16_t)(qq >> 16);
                    if (p < 2) v -= s3_hi(qq);
                }
            }
    }
    __syncthreads();
    R3_T(4);

    // ---- 3. conv2 (3x3): 32 rows, tap -> 8 k-steps of 16 channels ---------------------------------------------------------------------
    r3_f32x16 acc2 = zero16, acc2b = zero16;                     // even / odd k-steps: two chains (summed below)
    {
        int arow[9];
        const int py = lr / TW, px = lr % TW;
#pragma unroll
        for (int t = 0; t < 9; ++t) arow[t] = r3_row(py + t / 3, px + t % 3) * MPB + lk * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int ks = 0; ks < 8; ks += 2) {                 // steps tap * 8 + ks, + 1 live in ring slots ks, ks + 1
                load2(tap * 8 + ks + R2 - 2, ring2[(ks + R2 - 2) % R2]);
                load2(tap * 8 + ks + R2 - 1, ring2[(ks + R2 - 1) % R2]);
                r3_bf16x8 af[2][NP];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int p = 0; p < NP; ++p) af[u][p] = *(const r3_bf16x8*)(M1 + p * MPL + arow[tap] + (ks + u) * 32);
                __builtin_amdgcn_sched_barrier(0);
                mac6x2(acc2, af[0], ring2[ks], acc2b, af[1], ring2[ks + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        acc2 += acc2b;
    }
    R3_T(5);
#pragma unroll
    for (int g = 0; g < R3 - 1; ++g) load3(g, ring3[g]);
    {   // relu(acc + b2) -> M2 planes (the x tile is dead: every wave is past conv1)
        const int ch = 32 * w + lr;
        const float b2 = NP == 2 ? a.b2[ch] * S2_XSCALE : a.b2[ch];
        const float c2 = NP == 2 ? a.osc2[ch] * S2_XSCALE : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = NP == 2 ? fmaxf(fmaf(acc2[r], c2, b2), 0.f) : fmaxf(acc2[r] + b2, 0.f);
            unsigned char* d = M2 + r3_acc_row(r, lane) * MPB + ch * 2;
            if constexpr (NP == 2) {
                gmax = fmaxf(gmax, v);
                const _Float16 hi = (_Float16)v;
                *reinterpret_cast<_Float16*>(d) = hi;
                *reinterpret_cast<_Float16*>(d + M2PL) = (_Float16)(v - (float)hi);
                continue;
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned qq = s3_pack_rn(v, v);
                *reinterpret_cast<uint16_t*>(d + p * M2PL) = (uint16_t)(qq >> 16);
                if (p < 2) v -= s3_hi(qq);
            }
        }
    }
    __syncthreads();
    R3_T(6);

    // ---- 4. conv3 (1x1, 128 -> 256): channels [64 w, 64 w + 64) -------------------------------------------------------------------------
    r3_f32x16 acc3[2] = {zero16, zero16};
    {
        const unsigned char* ma = M2 + lr * MPB + lk * 16;
#pragma unroll
        for (int ks = 0; ks < NS3; ++ks) {
            load3(ks + R3 - 1, ring3[(ks + R3 - 1) % R3]);
            r3_bf16x8 af[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) af[p] = *(const r3_bf16x8*)(ma + p * M2PL + ks * 32);
            __builtin_amdgcn_sched_barrier(0);
            mac6x2(acc3[0], af, ring3[ks % R3][0], acc3[1], af, ring3[ks % R3][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    R3_T(7);
    // accumulators -> patch [pixel][256] (mid1 is dead), then + b3 + x [+ up] and the stores on 16-byte vectors
    if constexpr (NP == 2) s2_raise(a.range_flag, gmax);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float c3 = NP == 2 ? a.osc3[(2 * w + j) * 32 + lr] : 1.f;       // back to scale: an exact power of two per column
#pragma unroll
        for (int r = 0; r < 16; ++r) P3[r3_acc_row(r, lane) * PP + (2 * w + j) * 32 + lr] = NP == 2 ? acc3[j][r] * c3 : acc3[j][r];
    }
    __syncthreads();
    {
        const r3_f32x4 b3 = *(const r3_f32x4*)(a.b3 + 4 * q);
        const size_t ucrop = (size_t)(H / 2) * (W / 2) * C;
        const __amdgpu_buffer_rsrc_t up_srd = make_srd(UP ? a.up + (size_t)l * ucrop : a.x, UP ? ucrop * sizeof(float) : 0);
#pragma unroll
        for (int i = 0; i < T / 4; ++i) {
            const int p = (tid >> 6) + 4 * i;
            const int oy = oy0 + p / TW, ox = ox0 + p % TW;
            const bool ok = oy < H && ox < W;
            r3_f32x4 xr;                                        // the skip: x itself (its 2x2 maximum when the pool is taken here)
            if (POOL_IN) {
                r3_f32x4 v[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] = buf_load(x_srd, ok ? (((2 * oy + (s >> 1)) * XW + 2 * ox + (s & 1)) * C + 4 * q) * 4 : BUF_OOB, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) xr[t] = fmaxf(fmaxf(v[0][t], v[1][t]), fmaxf(v[2][t], v[3][t]));
            } else {
                xr = buf_load(x_srd, ok ? ((oy * XW + ox) * C + 4 * q) * 4 : BUF_OOB, 0);
            }
            r3_f32x4 o = *(const r3_f32x4*)&P3[p * PP + 4 * q] + b3;
            o += xr;                                            // (bias, then the skip: the order of the per-layer kernels)
            if (UP) o += buf_load(up_srd, ok ? (((oy >> 1) * (W / 2) + (ox >> 1)) * C + 4 * q) * 4 : BUF_OOB, 0);
            buf_store(o, o_srd, ok ? ((oy * W + ox) * C + 4 * q) * 4 : BUF_OOB);
        }
    }
#ifdef SUO_RS_PROF
    R3_T(8);
    if (blockIdx.x == 0 && tid == 0)
        printf("res_block_x3 map %dx%d: stage x %lld  barrier %lld  conv1 %lld  epi1+barrier %lld  conv2 %lld  epi2+barrier %lld  conv3 %lld  patch+out %lld  total %lld cycles\n",
               H, W, pt[1] - pt[0], pt[2] - pt[1], pt[3] - pt[2], pt[4] - pt[3], pt[5] - pt[4], pt[6] - pt[5], pt[7] - pt[6], pt[8] - pt[7], pt[8] - pt[0]);
#endif
}

// ResBlockArgs with W1 / W2 / W3 = the uint16 planes of pack_gemm_weight_bf16x3(W1 [128][256]) / pack_res_conv3x3_bf16x3 / pack_gemm_weight_bf16x3(W3 [256][128])
// (NP = 2: of pack_gemm_weight_f16x2 / pack_res_conv3x3_f16x2 / pack_gemm_weight_f16x2, with their per-channel factors in osc1 / osc2 / osc3 and range_flag set)
template <int NP>
static int launch_res_block_split(const ResBlockArgs& a, hipStream_t s) {
    if (!res_block_takes(a) || (NP == 2 && (!a.osc1 || !a.osc2 || !a.osc3 || !a.range_flag))) {
        suo_set_error("res_block_%s: unsupported arguments (L=%d H=%d W=%d)", NP == 3 ? "x3" : "f16x2", a.L, a.H, a.W);
        return SUO_ERR_ARG;
    }
    const unsigned tiles = (unsigned)((long)a.L * ((a.H + 3) / 4) * ((a.W + 7) / 8));
    if (a.pool_in) { if (a.up) hipLaunchKernelGGL((res_block_x3_kernel<true, true, NP>), dim3(tiles), dim3(256), 0, s, a);
                     else hipLaunchKernelGGL((res_block_x3_kernel<true, false, NP>), dim3(tiles), dim3(256), 0, s, a); }
    else { if (a.up) hipLaunchKernelGGL((res_block_x3_kernel<false, true, NP>), dim3(tiles), dim3(256), 0, s, a);
           else hipLaunchKernelGGL((res_block_x3_kernel<false, false, NP>), dim3(tiles), dim3(256), 0, s, a); }
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}
int launch_res_block_x3(const ResBlockArgs& a, hipStream_t s) { return launch_res_block_split<3>(a, s); }
int launch_res_block_f16x2(const ResBlockArgs& a, hipStream_t s) { return launch_res_block_split<2>(a, s); }

}  // namespace suo

// One launch per Residual block on the small feature maps of a one-frame call (lib/models/layers/Residual.py:20-35; the 48 blocks of
// the two Hourglasses that run at 32x32 and below, lib/models/hg.py:37-58):
//
//     out = W3 relu(bn2(conv3x3(relu(bn1(W1 relu(bn(x))))))) + b3 + x   [+ up-sampled addend]        256 -> 128 -> 128 -> 256 channels
//
// At one frame per call (evaluate.py:389-393 -> lib/object_slam.py:1099: 8 crops) these maps are 8192 ... 128 pixels per launch.  The
// per-layer kernels (csrc/conv.hip, csrc/conv_small.hip) need three dependent launches per block there, each a few microseconds of
// ramp and drain around very little work (profiles/r03_latency_kernel_stats.txt: 108 launches of 5-12 us below 32x32, 36 of 10-26 us
// at 32x32 -- 1.3 of the network's 2.5 ms).  Here a workgroup owns a 4x4 (4x8 at 32x32) pixel tile of ONE crop and runs the whole
// block on it:
//   1. relu(bn(x)) of the tile + its 1-pixel halo -> LDS (the block's pre-activation, applied while staging);
//   2. conv1 (1x1, 256 -> 128, bn1 folded) on tile + halo: the halo is recomputed instead of exchanged (36 / 60 rows for 16 / 32
//      output pixels); relu(. + b1) -> LDS, halo pixels outside the map written as zeros (conv2's zero padding pads ITS input);
//   3. conv2 (3x3, 128 -> 128, bn2 folded) from that LDS tile; relu(. + b2) -> LDS;
//   4. conv3 (1x1, 128 -> 256) ; + b3 + x [+ up] on 16-byte vectors through an LDS patch.
// Optionally x is the 2x2 max-pool of a map of twice the size (the pool that precedes the first block of a level, hg.py:41), taken
// while staging -- no pool launch, no pooled tensor.
//
// fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32): rows = pixels, 16 per tile; four waves, one per SIMD, each owning a quarter of
// the output channels of every stage with ALL of K in one accumulator per 16x16 tile.  The weights of a wave's channels come
// straight from L2 in B-operand order through a static register ring (one 16-byte load per lane = 4 MFMA k-steps), 852 KB per
// workgroup -- the same bytes whatever the tile, which is what bounds the tile from below: at 128 MAC / clock / CU a 16-pixel tile is
// 34 k cycles of MFMA against 13 k of weight stream.
//
// Summation order.  Every accumulator adds its K terms in exactly the order of the per-layer fp32 kernels (gemm1x1_kernel /
// gemm_persist_kernel / convk_kernel<3,...>: 32-channel chunks ascending, 3x3: chunk -> tap -> channel; inside 8 channels the pairs
// (0,4) (1,5) (2,6) (3,7) of v_mfma_f32_32x32x2_f32) by giving MFMA j of a 16-channel group the channels rs_koff(j, 0..3) -- the
// 16x16x4 instruction adds its four blocks in order -- so this kernel is BIT-IDENTICAL to conv1x1 -> conv3x3 -> conv1x1 + skip
// launched separately (tests/test_gpu_res_block.py).
#include <string.h>

#include "buffer_ops.h"
#include "suo_internal.h"
#include "tune.h"

namespace suo {

typedef float rs_f32x4 __attribute__((ext_vector_type(4)));
typedef float rs_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ rs_f32x4 rs_mfma(float a, float b, rs_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// channel (inside its group of 16) that MFMA j = 0..3 of the group multiplies in lane block b = lane >> 4
__host__ __device__ constexpr int rs_koff(int j, int b) { return 8 * (j >> 1) + (((j & 1) * 4 + b) >> 1) + 4 * (((j & 1) * 4 + b) & 1); }
// ... and its inverse: position b * (4 * groups) + 4 * group + j of channel c in an LDS row of `groups` 16-channel groups
__device__ __forceinline__ int rs_pos(int c, int groups) {
    const int g = c >> 4, c8 = c & 7, i = (c8 & 3) * 2 + (c8 >> 2);
    return (i & 3) * (4 * groups) + 4 * g + 2 * ((c >> 3) & 1) + (i >> 2);
}

// ---- host: weights in B-operand order of v_mfma_f32_16x16x4_f32 with the k order above --------------------------------------------
//   out[((group * N/16 + nt) * 64 + lane) * 4 + j] = W[nt*16 + (lane&15)][16 group + rs_koff(j, lane >> 4)]
void pack_res16_gemm(const float* W, int N, int K, float* out) {
    const int NT = N / 16;
    for (int g = 0; g < K / 16; ++g)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j)
                    out[(((size_t)g * NT + nt) * 64 + lane) * 4 + j] = W[(size_t)(nt * 16 + (lane & 15)) * K + 16 * g + rs_koff(j, lane >> 4)];
}
// 3x3: W[N][C][3][3] (times out_scale[n]: bn2 folded like csrc/net.hip does for the per-layer kernels), groups in the order
// chunk (32 channels) -> tap -> half chunk:  group = (chunk * 9 + tap) * 2 + s,  channels 32 chunk + 16 s + rs_koff(j, lane >> 4)
void pack_res16_conv3x3(const float* W, int N, int C, const float* out_scale, float* out) {
    const int NT = N / 16;
    for (int ch = 0; ch < C / 32; ++ch)
        for (int tap = 0; tap < 9; ++tap)
            for (int s = 0; s < 2; ++s) {
                const int gi = (ch * 9 + tap) * 2 + s;
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j) {
                            const int n = nt * 16 + (lane & 15), c = 32 * ch + 16 * s + rs_koff(j, lane >> 4);
                            const float sc = out_scale ? out_scale[n] : 1.f;
                            out[(((size_t)gi * NT + nt) * 64 + lane) * 4 + j] = W[(((size_t)n * C + c) * 3 + tap / 3) * 3 + tap % 3] * sc;
                        }
            }
}

// ---- tile geometry: rows of the (TH + 2) x (TW + 2) halo tile, interior pixels first ----------------------------------------------
// rows [0, T): the TH x TW output pixels in raster order; rows [T, T + 2 IW + 2 TH): the ring (top row, bottom row, left column,
// right column).  The interior m-tiles are whole; the ring's m-tiles are skipped when all their pixels lie outside the map (4x4 maps).
template <int TH, int TW>
__device__ __forceinline__ int rs_row(int hy, int hx) {
    constexpr int IH = TH + 2, IW = TW + 2, T = TH * TW;
    if (hy >= 1 && hy <= TH && hx >= 1 && hx <= TW) return (hy - 1) * TW + (hx - 1);
    if (hy == 0) return T + hx;
    if (hy == IH - 1) return T + IW + hx;
    if (hx == 0) return T + 2 * IW + (hy - 1);
    return T + 2 * IW + TH + (hy - 1);
}
template <int TH, int TW>
__device__ __forceinline__ bool rs_hyhx(int row, int& hy, int& hx) {          // false: a padding row of the last m-tile
    constexpr int IH = TH + 2, IW = TW + 2, T = TH * TW;
    if (row < T) { hy = row / TW + 1; hx = row % TW + 1; return true; }
    const int q = row - T;
    if (q < IW) { hy = 0; hx = q; return true; }
    if (q < 2 * IW) { hy = IH - 1; hx = q - IW; return true; }
    if (q < 2 * IW + TH) { hy = q - 2 * IW + 1; hx = 0; return true; }
    if (q < 2 * IW + 2 * TH) { hy = q - 2 * IW - TH + 1; hx = IW - 1; return true; }
    hy = hx = 0;
    return false;
}

#ifdef SUO_RS_PROF                                              // tools/build_variant.sh rsprof -DSUO_RS_PROF: phase times of workgroup 0, wave 0
#define RS_T(i) do { pt[i] = clock64(); } while (0)
#else
#define RS_T(i) do { } while (0)
#endif

template <int TH, int TW, bool POOL_IN, bool UP>
__global__ __launch_bounds__(256) void res_block_kernel(const ResBlockArgs a) {
#ifdef SUO_RS_PROF
    long long pt[10];
    RS_T(0);
#endif
    constexpr int T = TH * TW, IW = TW + 2, NH = (TH + 2) * IW, MT1 = (NH + 15) / 16, MT2 = T / 16, MTI = MT2;
    constexpr int XP = 260, MP = 132, PP = 260;                  // LDS pitches (floats): x tile, mid tiles, output patch
    static_assert(T % 16 == 0 && T * PP <= MT1 * 16 * MP, "the output patch re-uses the mid1 tile");
    __shared__ __attribute__((aligned(16))) float Xs[MT1 * 16 * XP];      // relu(bn(x)) of tile + halo; later: relu(conv2) of the tile
    __shared__ __attribute__((aligned(16))) float M1[MT1 * 16 * MP];      // relu(conv1) of tile + halo; later: the output patch
    float* M2 = Xs;
    float* P3 = M1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lb = lane >> 4;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);          // XCD-aware tile order (csrc/conv.hip)
    const int l = bid / (tiles_x * tiles_y);
    bid -= l * tiles_x * tiles_y;
    const int ty0 = bid / tiles_x, tx0 = bid - ty0 * tiles_x;
    const int oy0 = ty0 * TH, ox0 = tx0 * TW;
    constexpr int C = 256;
    const int XH = POOL_IN ? 2 * H : H, XW = POOL_IN ? 2 * W : W;        // the tensor x lives in
    const size_t xcrop = (size_t)XH * XW * C, ocrop = (size_t)H * W * C;
    const __amdgpu_buffer_rsrc_t x_srd = make_srd(a.x + (size_t)l * xcrop, xcrop * sizeof(float));
    const __amdgpu_buffer_rsrc_t o_srd = make_srd(a.out + (size_t)l * ocrop, ocrop * sizeof(float));
    const __amdgpu_buffer_rsrc_t w1_srd = make_srd(a.W1, (size_t)128 * 256 * sizeof(float));
    const __amdgpu_buffer_rsrc_t w2_srd = make_srd(a.W2, (size_t)128 * 128 * 9 * sizeof(float));
    const __amdgpu_buffer_rsrc_t w3_srd = make_srd(a.W3, (size_t)256 * 128 * sizeof(float));

    // ---- weight rings: group gi of a stage lives in slot gi % R and is requested R - 1 groups before its MFMAs -----------------
    constexpr int R1 = 8, R2 = 6, R3 = 8, NG1 = 16, NG2 = 72, NG3 = 8;
    const int wv12 = (2 * w * 64 + lane) * 16;                  // conv1 / conv2: n-tiles 2 w, 2 w + 1 of 8 (8 KB per group)
    const int wv3 = (4 * w * 64 + lane) * 16;                   // conv3: n-tiles 4 w .. 4 w + 3 of 16 (16 KB per group)
    rs_f32x4 ring1[R1][2], ring2[R2][2], ring3[R3][4];
    auto load1 = [&](int g, rs_f32x4 (&b)[2]) {
        const int gc = g < NG1 ? g : NG1 - 1;
#pragma unroll
        for (int n = 0; n < 2; ++n) b[n] = buf_load(w1_srd, wv12 + n * 1024, gc * 8192);
    };
    auto load2 = [&](int g, rs_f32x4 (&b)[2]) {
        const int gc = g < NG2 ? g : NG2 - 1;
#pragma unroll
        for (int n = 0; n < 2; ++n) b[n] = buf_load(w2_srd, wv12 + n * 1024, gc * 8192);
    };
    auto load3 = [&](int g, rs_f32x4 (&b)[4]) {
        const int gc = g < NG3 ? g : NG3 - 1;
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = buf_load(w3_srd, wv3 + n * 1024, gc * 16384);
    };
#pragma unroll
    for (int g = 0; g < R1 - 1; ++g) load1(g, ring1[g]);        // (first touch of the block's weights: under the x staging)

    // ---- 1. stage relu(bn(x)) of tile + halo: thread = (row tid >> 6 + 4 i, channels 4 q .. 4 q + 3, q = tid & 63) -----------------
    const int q = tid & 63;
    {
        const rs_f32x4 sc = *(const rs_f32x4*)(a.pro_scale + 4 * q), sh = *(const rs_f32x4*)(a.pro_shift + 4 * q);
        // channels 4 q + t sit in group q >> 2, half (q >> 1) & 1; t = 0, 2 go to lane block q & 1, t = 1, 3 to block (q & 1) + 2 (rs_koff)
        const int pos0 = (q & 1) * 64 + (q >> 2) * 4 + 2 * ((q >> 1) & 1);
        constexpr int NB = POOL_IN ? 4 : MT1 * 4;               // rows per thread in flight (every request before the first use)
#pragma unroll
        for (int i0 = 0; i0 < MT1 * 4; i0 += NB) {
            rs_f32x4 v[NB][POOL_IN ? 4 : 1];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int row = (tid >> 6) + 4 * (i0 + u);
                int hy, hx;
                const bool real = rs_hyhx<TH, TW>(row, hy, hx);
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
                rs_f32x4 x = v[u][0];
                if (POOL_IN) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) x[t] = fmaxf(fmaxf(v[u][0][t], v[u][1][t]), fmaxf(v[u][2][t], v[u][3][t]));      // (the order of maxpool2_kernel)
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) x[t] = fmaxf(fmaf(x[t], sc[t], sh[t]), 0.f);
                float* d = &Xs[row * XP + pos0];
                *(rs_f32x2*)d = rs_f32x2{x[0], x[2]};
                *(rs_f32x2*)(d + 128) = rs_f32x2{x[1], x[3]};
            }
        }
    }
    // which rows of the halo tile are pixels of the map: per lane for the accumulator rows 4 lb + r of every m-tile (bit m * 4 + r),
    // per m-tile whether any row is (wave-uniform)
    unsigned vmask = 0;
    bool mt_any[MT1];
#pragma unroll
    for (int m = 0; m < MT1; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int hy, hx;
            const bool real = rs_hyhx<TH, TW>(m * 16 + 4 * lb + r, hy, hx);
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            if (real && iy >= 0 && iy < H && ix >= 0 && ix < W) vmask |= 1u << (m * 4 + r);
        }
        mt_any[m] = m < MTI || __builtin_amdgcn_readfirstlane((int)(__ballot(((vmask >> (m * 4)) & 15u) != 0) != 0ull)) != 0;
    }
    RS_T(1);
    __syncthreads();
    RS_T(2);

    // ---- 2. conv1: rows = tile + halo, wave w -> channels [32 w, 32 w + 32) ----------------------------------------------------------
    rs_f32x4 acc1[MT1][2];
#pragma unroll
    for (int m = 0; m < MT1; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc1[m][n] = rs_f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const float* xa = &Xs[lr * XP + lb * 64];
#pragma unroll
        for (int g = 0; g < NG1; ++g) {
            load1(g + R1 - 1, ring1[(g + R1 - 1) % R1]);
            rs_f32x4 af[MT1];
#pragma unroll
            for (int m = 0; m < MT1; ++m) af[m] = *(const rs_f32x4*)(xa + m * 16 * XP + g * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MT1; ++m)
                    if (mt_any[m]) {
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc1[m][n] = rs_mfma(af[m][j], ring1[g % R1][n][j], acc1[m][n]);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    RS_T(3);
#pragma unroll
    for (int g = 0; g < R2 - 1; ++g) load2(g, ring2[g]);        // conv2's first weights travel under the epilogue + barrier
    {   // relu(acc + b1) -> M1 (zeros outside the map: Conv2d(padding=1) pads conv2's INPUT)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int ch = (2 * w + n) * 16 + lr;
            const float b1 = a.b1[ch];
            const int pos = rs_pos(ch, 8);
#pragma unroll
            for (int m = 0; m < MT1; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = (vmask >> (m * 4 + r)) & 1u;
                    M1[(m * 16 + 4 * lb + r) * MP + pos] = ok ? fmaxf(acc1[m][n][r] + b1, 0.f) : 0.f;
                }
        }
    }
    __syncthreads();
    RS_T(4);

    // ---- 3. conv2 (3x3): rows = the tile's pixels, K = chunk (32 channels) -> tap -> 16 channels ------------------------------------
    rs_f32x4 acc2[MT2][2];
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc2[m][n] = rs_f32x4{0.f, 0.f, 0.f, 0.f};
    {
        int arow[MT2][9];
#pragma unroll
        for (int m = 0; m < MT2; ++m) {
            const int p = m * 16 + lr, py = p / TW, px = p % TW;
#pragma unroll
            for (int t = 0; t < 9; ++t) arow[m][t] = rs_row<TH, TW>(py + t / 3, px + t % 3) * MP + lb * 32;
        }
        for (int ch = 0; ch < 4; ++ch) {
#pragma unroll
            for (int u = 0; u < 18; ++u) {                      // (tap, half): group ch * 18 + u, ring slot u % 6 (18 = 3 * 6)
                const int g = ch * 18 + u;
                load2(g + R2 - 1, ring2[(u + R2 - 1) % R2]);
                rs_f32x4 af[MT2];
#pragma unroll
                for (int m = 0; m < MT2; ++m) af[m] = *(const rs_f32x4*)&M1[arow[m][u >> 1] + (ch * 2 + (u & 1)) * 4];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MT2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc2[m][n] = rs_mfma(af[m][j], ring2[u % R2][n][j], acc2[m][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    RS_T(5);
#pragma unroll
    for (int g = 0; g < R3 - 1; ++g) load3(g, ring3[g]);
    {   // relu(acc + b2) -> M2 (the x tile is dead: every wave is past conv1)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int ch = (2 * w + n) * 16 + lr;
            const float b2 = a.b2[ch];
            const int pos = rs_pos(ch, 8);
#pragma unroll
            for (int m = 0; m < MT2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) M2[(m * 16 + 4 * lb + r) * MP + pos] = fmaxf(acc2[m][n][r] + b2, 0.f);
        }
    }
    __syncthreads();
    RS_T(6);

    // ---- 4. conv3 (1x1, 128 -> 256): wave w -> channels [64 w, 64 w + 64) ------------------------------------------------------------
    rs_f32x4 acc3[MT2][4];
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc3[m][n] = rs_f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const float* ma = &M2[lr * MP + lb * 32];
#pragma unroll
        for (int g = 0; g < NG3; ++g) {
            load3(g + R3 - 1, ring3[(g + R3 - 1) % R3]);
            rs_f32x4 af[MT2];
#pragma unroll
            for (int m = 0; m < MT2; ++m) af[m] = *(const rs_f32x4*)(ma + m * 16 * MP + g * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MT2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc3[m][n] = rs_mfma(af[m][j], ring3[g % R3][n][j], acc3[m][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    RS_T(7);
    // accumulators -> patch [pixel][256] (mid1 is dead), then + b3 + x [+ up] and the stores on 16-byte vectors
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) P3[(m * 16 + 4 * lb + r) * PP + (4 * w + n) * 16 + lr] = acc3[m][n][r];
    __syncthreads();
    {
        const rs_f32x4 b3 = *(const rs_f32x4*)(a.b3 + 4 * q);
        const size_t ucrop = (size_t)(H / 2) * (W / 2) * C;
        const __amdgpu_buffer_rsrc_t up_srd = make_srd(UP ? a.up + (size_t)l * ucrop : a.x, UP ? ucrop * sizeof(float) : 0);
#pragma unroll
        for (int i = 0; i < T / 4; ++i) {
            const int p = (tid >> 6) + 4 * i;
            const int oy = oy0 + p / TW, ox = ox0 + p % TW;
            const bool ok = oy < H && ox < W;
            rs_f32x4 xr;                                        // the skip: x itself (its 2x2 maximum when the pool is taken here)
            if (POOL_IN) {
                rs_f32x4 v[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] = buf_load(x_srd, ok ? (((2 * oy + (s >> 1)) * XW + 2 * ox + (s & 1)) * C + 4 * q) * 4 : BUF_OOB, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) xr[t] = fmaxf(fmaxf(v[0][t], v[1][t]), fmaxf(v[2][t], v[3][t]));
            } else {
                xr = buf_load(x_srd, ok ? ((oy * XW + ox) * C + 4 * q) * 4 : BUF_OOB, 0);
            }
            rs_f32x4 o = *(const rs_f32x4*)&P3[p * PP + 4 * q] + b3;
            o += xr;                                            // (bias, then the skip: the order of the per-layer kernels)
            if (UP) o += buf_load(up_srd, ok ? (((oy >> 1) * (W / 2) + (ox >> 1)) * C + 4 * q) * 4 : BUF_OOB, 0);
            buf_store(o, o_srd, ok ? ((oy * W + ox) * C + 4 * q) * 4 : BUF_OOB);
        }
    }
#ifdef SUO_RS_PROF
    RS_T(8);
    if (blockIdx.x == 0 && tid == 0)
        printf("res_block %dx%d tile, map %dx%d: stage x %lld  barrier %lld  conv1 %lld  epi1+barrier %lld  conv2 %lld  epi2+barrier %lld  conv3 %lld  patch+out %lld  total %lld cycles\n",
               TH, TW, H, W, pt[1] - pt[0], pt[2] - pt[1], pt[3] - pt[2], pt[4] - pt[3], pt[5] - pt[4], pt[6] - pt[5], pt[7] - pt[6], pt[8] - pt[7], pt[8] - pt[0]);
#endif
}

bool res_block_takes(const ResBlockArgs& a) {
    const size_t lim = (size_t)1 << 31;
    return a.L > 0 && a.H > 0 && a.W > 0 && a.x && a.out && a.pro_scale && a.pro_shift && a.W1 && a.W2 && a.W3 && a.b1 && a.b2 && a.b3 &&
           (size_t)(a.pool_in ? 4 : 1) * a.H * a.W * 256 * 4 < lim && (!a.up || (a.H % 2 == 0 && a.W % 2 == 0));
}

// tile: 4 x 8 pixels when that still gives every CU a workgroup, else 4 x 4
int launch_res_block(const ResBlockArgs& a, hipStream_t s) {
    if (!res_block_takes(a)) { suo_set_error("res_block: unsupported arguments (L=%d H=%d W=%d)", a.L, a.H, a.W); return SUO_ERR_ARG; }
    static const int force = (int)SUO_TUNE("SUO_RES_TILE", 0);           // tuning aid: 16 / 32 pixels
    const long t32 = (long)a.L * ((a.H + 3) / 4) * ((a.W + 7) / 8);
    const bool big = force ? force == 32 : (a.W >= 8 && t32 >= 256);
#define RS_LAUNCH(TH_, TW_, tiles)                                                                                                         \
    do {                                                                                                                                   \
        if (a.pool_in) { if (a.up) hipLaunchKernelGGL((res_block_kernel<TH_, TW_, true, true>), dim3(tiles), dim3(256), 0, s, a);        \
                         else hipLaunchKernelGGL((res_block_kernel<TH_, TW_, true, false>), dim3(tiles), dim3(256), 0, s, a); }           \
        else { if (a.up) hipLaunchKernelGGL((res_block_kernel<TH_, TW_, false, true>), dim3(tiles), dim3(256), 0, s, a);                  \
               else hipLaunchKernelGGL((res_block_kernel<TH_, TW_, false, false>), dim3(tiles), dim3(256), 0, s, a); }                    \
    } while (0)
    if (big) RS_LAUNCH(4, 8, (unsigned)t32);
    else RS_LAUNCH(4, 4, (unsigned)((long)a.L * ((a.H + 3) / 4) * ((a.W + 3) / 4)));
#undef RS_LAUNCH
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo

// Latency-oriented convolution kernels for the small feature maps of the hourglass (16x16 and below at
// L = 8: 2048 ... 128 pixels per launch).  Same math as csrc/conv.hip (Residual.forward,
// /root/reference/lib/models/layers/Residual.py:20-35), different decomposition:
//
//   * at these sizes a launch is a dependent chain, not a throughput problem: a 32x32 tile with K = 1152 is
//     576 serial v_mfma_f32_32x32x2_f32 = 37k cycles on ONE wave while 250 CUs idle;
//   * so tiles are 16x16 (v_mfma_f32_16x16x4_f32, 8 cycles per k instead of 32) and the K dimension is SPLIT
//     across the waves of a workgroup (one 32-channel chunk per wave), partial tiles are reduced through LDS in
//     a fixed order, and every wave issues ALL its global loads (activations + weights) before its first MFMA;
//   * 3x3: each wave stages only its own channel chunk of the 6x6 halo into a wave-private LDS patch, so no
//     workgroup barrier sits between load and compute.
// Weights use the second half of the packed buffer ([K/16][N/16][64 lanes][4], see pack_gemm_weight).
#include "suo_internal.h"

namespace suo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------
// 1x1: one workgroup = one 16-pixel x 16-channel output tile, NW waves = NW K-slices
// ---------------------------------------------------------------------------------------------------------
template <int NW, int MAXG>
__global__ __launch_bounds__(NW * 64) void gemm_small_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float part[NW][256];
    const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
    const int K = a.K1 + a.K2;
    const int G = K >> 4;
    const int g0 = (G * ks) / NW, g1 = (G * (ks + 1)) / NW;
    const int NB16 = a.N >> 4, nb = n0 >> 4;
    const float* W16 = a.Wp + (size_t)a.N * K;
    const int row = m0 + (lane & 15), kq = (lane >> 4) * 4;
    const bool rok = row < a.M;

    f32x4 av[MAXG], bv[MAXG], sc[MAXG], sh[MAXG];
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi) {
        const int g = g0 + gi;
        av[gi] = f32x4{0.f, 0.f, 0.f, 0.f};
        bv[gi] = av[gi];
        sc[gi] = f32x4{1.f, 1.f, 1.f, 1.f};
        sh[gi] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g < g1) {
            const int k = g * 16 + kq;
            bv[gi] = *(const f32x4*)(W16 + ((size_t)(g * NB16 + nb) * 64 + lane) * 4);
            if (k < a.K1) {
                if (rok) av[gi] = *(const f32x4*)(a.A1 + (size_t)row * a.lda1 + k);
                if (a.pro_scale) { sc[gi] = *(const f32x4*)(a.pro_scale + k); sh[gi] = *(const f32x4*)(a.pro_shift + k); }
            } else if (rok) {
                av[gi] = *(const f32x4*)(a.A2 + (size_t)row * a.lda2 + (k - a.K1));
            }
        }
    }
    // residual for the epilogue wave: requested now, consumed after the reduction
    const int epx = lane >> 2, ec4 = lane & 3;
    const int erow = m0 + epx, ecol = n0 + ec4 * 4;
    f32x4 rv = {0.f, 0.f, 0.f, 0.f};
    if (ks == 0 && a.R && erow < a.M && ecol < a.n_valid) rv = *(const f32x4*)(a.R + (size_t)erow * a.ldr + ecol);

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi) {
        if (g0 + gi < g1) {
            f32x4 x = av[gi];
            if (a.pro_scale && (g0 + gi) * 16 < a.K1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) x[t] = fmaxf(fmaf(x[t], sc[gi][t], sh[gi][t]), 0.f);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (gi & 1) acc1 = mfma16(x[t], bv[gi][t], acc1);
                else acc0 = mfma16(x[t], bv[gi][t], acc0);
            }
        }
    }
    const f32x4 acc = acc0 + acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) part[ks][((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
    __syncthreads();
    if (ks == 0) {
        f32x4 o = *(const f32x4*)&part[0][epx * 16 + ec4 * 4];
#pragma unroll
        for (int w = 1; w < NW; ++w) o += *(const f32x4*)&part[w][epx * 16 + ec4 * 4];      // fixed order
        o = o + *(const f32x4*)(a.bias + ecol) + rv;
        if (a.relu) {
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = fmaxf(o[t], 0.f);
        }
        if (erow < a.M && ecol < a.n_valid) *(f32x4*)(a.out + (size_t)erow * a.ldo + ecol) = o;
    }
}

int launch_gemm_small(const GemmArgs& a, hipStream_t s) {
    const int G = (a.K1 + a.K2) >> 4;
    dim3 grid((a.M + 15) / 16, a.N / 16);
    if (G % 4 == 0 && G / 4 <= 5) {
        hipLaunchKernelGGL((gemm_small_kernel<4, 5>), grid, dim3(256), 0, s, a);
    } else if (G <= 20 * 2) {
        hipLaunchKernelGGL((gemm_small_kernel<8, 5>), grid, dim3(512), 0, s, a);
    } else {
        suo_set_error("gemm_small: K=%d too large", a.K1 + a.K2);
        return SUO_ERR_ARG;
    }
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 (stride 1, pad 1): one workgroup = a 4x4-pixel x 16-channel output tile, NW = C/32 waves (K-slices)
// ---------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void conv3x3_small_kernel(const ConvArgs a) {
    constexpr int PK = 36, HALO = 36;
    __shared__ __attribute__((aligned(16))) float As[NW][HALO * PK];
    __shared__ __attribute__((aligned(16))) float part[NW][256];
    const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
    const int tiles_x = (a.OW + 3) / 4, tiles_y = (a.OH + 3) / 4;
    int bid = blockIdx.x;
    const int l = bid / (tiles_x * tiles_y);
    bid -= l * tiles_x * tiles_y;
    const int ty = bid / tiles_x, tx = bid - ty * tiles_x;
    const int oy0 = ty * 4, ox0 = tx * 4;
    const int n0 = blockIdx.y * 16, NB16 = a.N >> 4, nb = n0 >> 4;
    const float* W16 = a.Wp + (size_t)a.N * a.C * 9;
    const float* in_l = a.in + (size_t)l * a.H * a.W * a.C + ks * 32;

    // all global loads first: this wave's 32-channel slice of the 6x6 halo, then its 18 weight fragments
    f32x4 hv[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int idx = lane + 64 * i;
        hv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < HALO * 8) {
            const int pix = idx >> 3, c4 = idx & 7;
            const int iy = oy0 - 1 + pix / 6, ix = ox0 - 1 + pix % 6;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) hv[i] = *(const f32x4*)(in_l + ((size_t)iy * a.W + ix) * a.C + c4 * 4);
        }
    }
    f32x4 bv[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g)
            bv[t][g] = *(const f32x4*)(W16 + ((size_t)(((ks * 9 + t) * 2 + g) * NB16 + nb) * 64 + lane) * 4);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int idx = lane + 64 * i;
        if (idx < HALO * 8) *(f32x4*)&As[ks][(idx >> 3) * PK + (idx & 7) * 4] = hv[i];
    }
    __builtin_amdgcn_wave_barrier();        // wave-private patch: LDS executes a wave's accesses in order

    const int pi = lane & 15;
    const float* ab = &As[ks][((pi >> 2) * 6 + (pi & 3)) * PK + (lane >> 4) * 4];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int toff = ((t / 3) * 6 + (t % 3)) * PK;
        const f32x4 a0 = *(const f32x4*)(ab + toff), a1 = *(const f32x4*)(ab + toff + 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc0 = mfma16(a0[u], bv[t][0][u], acc0);
            acc1 = mfma16(a1[u], bv[t][1][u], acc1);
        }
    }
    const f32x4 acc = acc0 + acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) part[ks][((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
    __syncthreads();
    if (ks == 0) {
        const int epx = lane >> 2, ec4 = lane & 3;
        f32x4 o = *(const f32x4*)&part[0][epx * 16 + ec4 * 4];
#pragma unroll
        for (int w = 1; w < NW; ++w) o += *(const f32x4*)&part[w][epx * 16 + ec4 * 4];
        const int col = n0 + ec4 * 4;
        o += *(const f32x4*)(a.bias + col);
        if (a.relu) {
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = fmaxf(o[t], 0.f);
        }
        const int oy = oy0 + (epx >> 2), ox = ox0 + (epx & 3);
        if (oy < a.OH && ox < a.OW) *(f32x4*)(a.out + (((size_t)l * a.OH + oy) * a.OW + ox) * a.N + col) = o;
    }
}

int launch_conv3x3_small(const ConvArgs& a, hipStream_t s) {
    const int tiles = ((a.OW + 3) / 4) * ((a.OH + 3) / 4) * a.L;
    dim3 grid(tiles, a.N / 16);
    if (a.C == 128) hipLaunchKernelGGL((conv3x3_small_kernel<4>), grid, dim3(256), 0, s, a);
    else if (a.C == 64) hipLaunchKernelGGL((conv3x3_small_kernel<2>), grid, dim3(128), 0, s, a);
    else if (a.C == 32) hipLaunchKernelGGL((conv3x3_small_kernel<1>), grid, dim3(64), 0, s, a);
    else { suo_set_error("conv3x3_small: C=%d unsupported", a.C); return SUO_ERR_ARG; }
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo

// Persistent 1x1 convolution (GEMM over NHWC pixels) for the large feature maps.
//
// Same operation, operands and epilogue as gemm1x1_kernel (csrc/conv.hip; Residual.forward conv1 / conv3 (+conv4, +skip),
// /root/reference/lib/models/layers/Residual.py:20-35; the lin_ / ll_ / tmpOut_ heads of hg.py:106-117).  Different
// schedule.  The channel counts of this network are tiny for a GEMM (K = 64..320, i.e. 2..10 chunks of 32): a chunk is
// only 64 MFMAs per wave (~1 us), shorter than an HBM round trip, and a tile's prologue + epilogue are as long as its
// main loop, so the one-tile-per-workgroup kernel measures  t = flops / 141 TFLOP/s + bytes / 4.2 TB/s  -- a SUM, no
// overlap (the 3x3 kernel has 9x longer chunks and does not suffer).  Here
//   * a workgroup is persistent and walks a list of tiles; (tile, chunk) pairs form one flat sequence of STEPS,
//   * activations are fetched TWO steps ahead (two static register sets; weights one step ahead, and issued BEFORE
//     the activation loads because vmcnt retires in order: a fast L2 weight load must not queue behind an HBM load
//     it does not depend on),
//   * so the first chunks of tile i+1 are already in flight while tile i finishes, and tile i's stores drain
//     under tile i+1's MFMAs.  The epilogue uses its own wave-private LDS patches (the chunk buffers already hold the
//     next tile).
#include "suo_internal.h"

namespace suo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row_s(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int TM, int TN, int WGM, int WGN, bool HAS_R>
__global__ __launch_bounds__(WGM* WGN * 64, 2) void gemm_persist_kernel(const GemmArgs a, int ntiles) {
    constexpr int BK = 32, PK = BK + 4;
    constexpr int BM = TM * 32 * WGM, BN = TN * 32 * WGN, NT = WGM * WGN * 64;
    constexpr int NLD = BM * 8 / NT;
    static_assert(BM * 8 % NT == 0, "staging must divide evenly");
    __shared__ __attribute__((aligned(16))) float As[2][BM * PK];
    __shared__ __attribute__((aligned(16))) float Tp[WGM * WGN][32 * PK];

    const int tid = threadIdx.x, lane = tid & 63;
#ifndef SUO_GEMM_SCALAR_WAVE
#define SUO_GEMM_SCALAR_WAVE 1
#endif
    const int w = SUO_GEMM_SCALAR_WAVE ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);     // wave index as an SGPR
    const int wm = w / WGN, wn = w % WGN;
    const int ntn = a.N / BN;
    const int nch1 = a.K1 >> 5, nch = nch1 + (a.K2 >> 5);
    const int NB = a.N >> 5;
    const int c4 = tid & 7, r0 = tid >> 3;
    const bool has_pro = a.pro_scale != nullptr;

    // tile list of this workgroup.  XCD-aware: workgroup b runs on XCD b % 8; each XCD owns a contiguous range of tiles
    // and its workgroups sweep it side by side, so the sibling N-tiles of a pixel tile are in flight together in one L2.
    const int G = gridDim.x;
    int t_first, t_stride, n_mine;
    if ((G & 7) == 0 && (ntiles & 7) == 0) {
        const int per_xcd = ntiles >> 3, gx = G >> 3, j = blockIdx.x >> 3;
        t_first = (blockIdx.x & 7) * per_xcd + j;
        t_stride = gx;
        n_mine = j < per_xcd ? (per_xcd - j + gx - 1) / gx : 0;
    } else {
        t_first = blockIdx.x;
        t_stride = G;
        n_mine = blockIdx.x < ntiles ? (ntiles - blockIdx.x + G - 1) / G : 0;
    }
    const int nsteps = n_mine * nch;
    if (nsteps == 0) return;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ar0[NLD], ar1[NLD], sc0, sh0, sc1, sh1;
    f32x4 b0[4][TN], b1[4][TN];

    // step -> (tile origin, chunk); fetch cursor and compute cursor advance independently (no divisions in the loop)
    // EVERY load of the steady state is unconditional (rows clamped instead of predicated, the cursor parks on the
    // last step instead of stopping): a load under a branch makes hipcc fall back to s_waitcnt vmcnt(0) at the next
    // use of ANY loaded value -- which would wait for the prefetches just issued and expose a full memory round
    // trip every step.
    int f_kc = 0, f_m0 = (t_first / ntn) * BM, f_tile = t_first, f_left = nsteps - 1;      // fetch cursor (activations)
    auto fetch_advance = [&]() {
        if (f_left > 0) {
            --f_left;
            if (++f_kc == nch) { f_kc = 0; f_tile += t_stride; f_m0 = (f_tile / ntn) * BM; }
        }
    };
    const float* const ps_base = has_pro ? a.pro_scale : a.bias;               // any readable floats when there is no prologue
    const float* const ph_base = has_pro ? a.pro_shift : a.bias;
    auto gload = [&](f32x4(&ar)[NLD], f32x4& sc, f32x4& sh) {
        const float* A;
        int lda, kk;
        if (f_kc < nch1) { A = a.A1; lda = a.lda1; kk = f_kc * BK; }
        else { A = a.A2; lda = a.lda2; kk = (f_kc - nch1) * BK; }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = f_m0 + r0 + i * (NT / 8);                         // M % BM == 0 (checked by the launcher)
            ar[i] = *(const f32x4*)(A + (size_t)row * lda + kk + c4 * 4);
        }
        const int pk = has_pro && f_kc < nch1 ? kk + c4 * 4 : 0;
        sc = *(const f32x4*)(ps_base + pk);
        sh = *(const f32x4*)(ph_base + pk);
        fetch_advance();
    };
    auto sstore = [&](const f32x4(&ar)[NLD], const f32x4& sc, const f32x4& sh, bool pro, int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            f32x4 v = ar[i];
            if (pro) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = fmaxf(fmaf(v[t], sc[t], sh[t]), 0.f);
            }
            *(f32x4*)&As[buf][(r0 + i * (NT / 8)) * PK + c4 * 4] = v;
        }
    };
    auto bload = [&](int kc, int n0, f32x4(&b)[4][TN]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nb = (n0 >> 5) + wn * TN + j;
                b[s][j] = *(const f32x4*)(a.Wp + ((size_t)((kc * 4 + s) * NB + nb) * 64 + lane) * 4);
            }
    };

    // compute cursor
    int kc = 0, tile = t_first, m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    // weights of the step after the current one: same tile unless the current chunk is the last
    auto next_b = [&](int& nkc, int& nn0) {
        nkc = kc + 1; nn0 = n0;
        if (nkc == nch) { nkc = 0; nn0 = (min(tile + t_stride, ntiles - 1) % ntn) * BN; }
    };

    bload(0, n0, b0);
    gload(ar0, sc0, sh0);
    gload(ar1, sc1, sh1);
    sstore(ar0, sc0, sh0, has_pro && 0 < nch1, 0);
    __syncthreads();

    auto epilogue = [&]() {
        float* T = &Tp[w][0];
        // every residual / bias value of the tile is requested BEFORE the first accumulator is transposed: one HBM round
        // trip per tile instead of one per 32 x 32 accumulator (K = 128 layers: the epilogue used to outlast the main loop)
        f32x4 rvall[TM][TN][4], bvall[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bvall[j] = *(const f32x4*)(a.bias + n0 + (wn * TN + j) * 32 + (lane & 7) * 4);
        if (HAS_R) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int row = m0 + (wm * TM + i) * 32 + (lane >> 3) + 8 * k;
                        rvall[i][j][k] = *(const f32x4*)(a.R + (size_t)row * a.ldr + n0 + (wn * TN + j) * 32 + (lane & 7) * 4);
                    }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { T[acc_row_s(r, lane) * PK + (lane & 31)] = acc[i][j][r]; acc[i][j][r] = 0.f; }
                __builtin_amdgcn_wave_barrier();
                const int col = n0 + (wn * TN + j) * 32 + (lane & 7) * 4;
                const f32x4 bv = bvall[j];
                f32x4 v[4], rv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    rv[k] = HAS_R ? rvall[i][j][k] : f32x4{0.f, 0.f, 0.f, 0.f};
                    v[k] = *(const f32x4*)&T[((lane >> 3) + 8 * k) * PK + (lane & 7) * 4];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = m0 + (wm * TM + i) * 32 + (lane >> 3) + 8 * k;
                    f32x4 o = v[k] + bv + rv[k];
                    if (a.relu) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) o[t] = fmaxf(o[t], 0.f);
                    }
                    *(f32x4*)(a.out + (size_t)row * a.ldo + col) = o;     // unpredicated: stores count in vmcnt too
                }
                __builtin_amdgcn_wave_barrier();
            }
    };

    // One step: weights for step s+1 and activations for step s+2 are requested, then the 16*TM*TN MFMAs of step s run
    // from LDS buffer s&1; afterwards step s+1's activations (requested during step s-1) go to the other buffer.
    //   b / bn        weight sets of step s / s+1
    //   a_free        register set that step s+2 is fetched into (it held step s, stored at the end of step s-1)
    //   a_next        register set holding step s+1
    auto step = [&](int s, const f32x4(&b)[4][TN], f32x4(&bn)[4][TN], f32x4(&a_free)[NLD], f32x4& sc_free, f32x4& sh_free,
                    const f32x4(&a_next)[NLD], const f32x4& sc_next, const f32x4& sh_next) {
        const int buf = s & 1;
        int nkc, nn0;
        next_b(nkc, nn0);
        bload(nkc, nn0, bn);                                   // (past the last step: a harmless re-read of chunk 0)
        gload(a_free, sc_free, sh_free);
        __builtin_amdgcn_sched_barrier(0);
        const float* as = &As[buf][((wm * TM * 32) + (lane & 31)) * PK + (lane >> 5) * 4];
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            f32x4 af[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(as + i * 32 * PK + ss * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], b[ss][j][t], acc[i][j], 0, 0, 0);
        }
        sstore(a_next, sc_next, sh_next, has_pro && nkc < nch1, buf ^ 1);
        __syncthreads();
        if (kc + 1 == nch) {                                   // workgroup-uniform: the tile is complete
            epilogue();
            kc = 0; tile += t_stride; m0 = (tile / ntn) * BM; n0 = (tile % ntn) * BN;
        } else {
            ++kc;
        }
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(s, b0, b1, ar0, sc0, sh0, ar1, sc1, sh1);
        if (s + 1 < nsteps) step(s + 1, b1, b0, ar1, sc1, sh1, ar0, sc0, sh0);
    }
}

template <int TM, int TN, int WGM, int WGN>
static int launch_persist_cfg(const GemmArgs& a, int max_wgs, hipStream_t s) {
    constexpr int BM = TM * 32 * WGM, BN = TN * 32 * WGN;
    if (a.N % BN || a.M % BM || a.n_valid != a.N) {
        suo_set_error("gemm_persist: M=%d N=%d n_valid=%d must be whole %dx%d tiles", a.M, a.N, a.n_valid, BM, BN);
        return SUO_ERR_ARG;
    }
    const int ntiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    const int g = ntiles < max_wgs ? ntiles : max_wgs;
    if (a.R) hipLaunchKernelGGL((gemm_persist_kernel<TM, TN, WGM, WGN, true>), dim3(g), dim3(WGM * WGN * 64), 0, s, a, ntiles);
    else hipLaunchKernelGGL((gemm_persist_kernel<TM, TN, WGM, WGN, false>), dim3(g), dim3(WGM * WGN * 64), 0, s, a, ntiles);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

// cfg: 1 = 128x128 tiles, 2 = 128x64 tiles, 3 = 64x64 tiles.  The grid is the number of RESIDENT workgroups (2 per CU
// for the 128-row tiles, 4 for the 64x64 one, on the 256-CU part), not the number of tiles.
int launch_gemm_persist(const GemmArgs& a, int cfg, hipStream_t s) {
    static const int max_wgs = getenv("SUO_GEMM_WGS") ? atoi(getenv("SUO_GEMM_WGS")) : 512;       // tuning aid
    if (cfg == 3) return launch_persist_cfg<1, 1, 2, 2>(a, 2 * max_wgs, s);
    if (cfg == 2) return launch_persist_cfg<2, 1, 2, 2>(a, max_wgs, s);
    return launch_persist_cfg<2, 2, 2, 2>(a, max_wgs, s);
}

}  // namespace suo

// Persistent 1x1 convolution (GEMM over NHWC pixels) for the large feature maps.
//
// Same operation, operands and epilogue as gemm1x1_kernel (csrc/conv.hip; Residual.forward conv1 / conv3 (+conv4, +skip),
// /root/reference/lib/models/layers/Residual.py:20-35; the lin_ / ll_ / tmpOut_ heads of hg.py:106-117).  Different
// schedule.  The channel counts of this network are tiny for a GEMM (K = 64..320, i.e. 2..10 chunks of 32): a chunk is
// only 64 MFMAs per wave (~1 us), shorter than an HBM round trip, and a tile's prologue + epilogue are as long as its
// main loop, so the one-tile-per-workgroup kernel measures  t = flops / 141 TFLOP/s + bytes / 4.2 TB/s  -- a SUM, no
// overlap (the 3x3 kernel has 9x longer chunks and does not suffer).  Here
//   * a workgroup is persistent and walks a list of tiles; (tile, chunk) pairs form one flat sequence of STEPS,
//   * activations are fetched one step ahead (one register set; -DSUO_GEMM_A_DEPTH=2: two steps, two sets -- measured
//     no faster), weights ride a static ring of four k-group slots requested three groups ahead (vmcnt retires in order,
//     so a weight load issued after an activation load also waits for it: the ring bounds how long),
//   * so the first chunk of tile i+1 is already in flight while tile i finishes, and tile i's stores drain
//     under tile i+1's MFMAs.  The epilogue uses its own wave-private LDS patches (the chunk buffers already hold the
//     next tile),
//   * every stream goes through a buffer descriptor (csrc/buffer_ops.h): one fixed 32-bit VGPR offset per stream, tile
//     and chunk in the descriptor base / scalar offset -- no 64-bit address registers, which together with the 16-row
//     epilogue patch (50 KB of LDS in all) lets THREE workgroups share a CU.
#include "buffer_ops.h"
#include "suo_internal.h"
#include "tune.h"

namespace suo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row_s(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

constexpr int GP_MAX_PRO_K = 512;       // channels of a fused BN-ReLU prologue (scale / shift kept in LDS)

#ifndef SUO_GEMM_A_DEPTH
#define SUO_GEMM_A_DEPTH 1          // steps the activations are fetched ahead (1: one register set, 2: two)
#endif
#ifndef SUO_GEMM_STORE_AT
#define SUO_GEMM_STORE_AT 1         // (A_DEPTH 2) k-group of the step before whose MFMAs the next activations go to LDS
#endif
#ifndef SUO_GEMM_STORE_MIX
#define SUO_GEMM_STORE_MIX 0
#endif
#ifndef SUO_GEMM_EPI_PREFETCH
#define SUO_GEMM_EPI_PREFETCH 0     // 1: request bias / first residual block during the tile's last K-step (measured: no gain
#endif                              //    with 3 workgroups per CU, and the 128x128 residual variant starts to spill)
#ifndef SUO_GEMM_WAVES_PER_EU
#define SUO_GEMM_WAVES_PER_EU 3
#endif

// POOL: the workgroup's 128 rows are not 128 consecutive pixels but a 2 x 64 patch of the image (rows y, y+1; wave row wm owns 32
// columns, its two 32-row MFMA blocks are the two image rows), so the four pixels of a 2x2 pooling window are register i / i+1 of the
// same lane after the transposition through the LDS patch plus one DPP rotate: the epilogue writes the pooled tensor beside (or
// instead of) the full-resolution one, pooling AFTER bias / residual / ReLU exactly as the separate kernel would.
template <int TM, int TN, int WGM, int WGN, bool HAS_R, bool POOL = false>
__global__ __launch_bounds__(WGM* WGN * 64) __attribute__((amdgpu_waves_per_eu(SUO_GEMM_WAVES_PER_EU))) void gemm_persist_kernel(const GemmArgs a, int ntiles) {
    static_assert(!POOL || (TM == 2 && WGM == 2), "POOL: 128-row tiles = 2 image rows x 64 columns");
    constexpr int BK = 32, PK = BK + 4;
    constexpr int BM = TM * 32 * WGM, BN = TN * 32 * WGN, NT = WGM * WGN * 64;
    constexpr int NLD = BM * 8 / NT;
    static_assert(BM * 8 % NT == 0, "staging must divide evenly");
    __shared__ __attribute__((aligned(16))) float As[2][BM * PK];
    __shared__ __attribute__((aligned(16))) float Tp[WGM * WGN][16 * PK];      // 128-row tiles: 50 KB in all = 3 workgroups per CU
    __shared__ __attribute__((aligned(16))) float Ps[2][GP_MAX_PRO_K];         // prologue scale | shift

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
    // POOL: pixel index of a tile's first row, and of the i-th group of 32 staged rows / the wave's second block relative to it
    const int pW = POOL ? a.pool_W : 0, pxb = POOL ? a.pool_W >> 6 : 1;
    auto tile_m0 = [&](int tm) { return POOL ? (tm / pxb) * 2 * pW + (tm % pxb) * 64 : tm * BM; };
    const int span = POOL ? pW + 64 : BM;                      // pixels a tile's descriptor has to cover

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
        n_mine = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + G - 1) / G : 0;
    }
    const int nsteps = n_mine * nch;
    if (nsteps == 0) return;

    if (has_pro)
        for (int k = tid; k < a.K1; k += NT) { Ps[0][k] = a.pro_scale[k]; Ps[1][k] = a.pro_shift[k]; }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ar0[NLD];
#if SUO_GEMM_A_DEPTH == 2
    f32x4 ar1[NLD];
#endif
    f32x4 bring[4][TN];

    // fixed per-lane byte offsets of the four streams
    const int av1 = (r0 * a.lda1 + c4 * 4) * 4, av2 = (r0 * a.lda2 + c4 * 4) * 4;           // activations (operand 1 / 2)
    const int wv = (wn * TN * 64 + lane) * 16;                                                // packed weights
    const int wrow = POOL ? wm * 32 : wm * TM * 32;                                           // the wave's first pixel within the tile
    const int blk = POOL ? pW : 32;                                                           // pixels from its block i to block i + 1
    const int ov = ((wrow + (lane >> 3)) * a.ldo + wn * TN * 32 + (lane & 7) * 4) * 4;        // output tile
    const int rv = ((wrow + (lane >> 3)) * a.ldr + wn * TN * 32 + (lane & 7) * 4) * 4;        // residual tile
    const int bv = (wn * TN * 32 + (lane & 7) * 4) * 4;                                       // bias
    const __amdgpu_buffer_rsrc_t w_srd = make_srd(a.Wp, (size_t)a.N * (a.K1 + a.K2) * sizeof(float));
    const __amdgpu_buffer_rsrc_t bias_srd = make_srd(a.bias, (size_t)a.N * sizeof(float));

    // step -> (tile origin, chunk); fetch cursor and compute cursor advance independently (no divisions in the loop)
    // EVERY load of the steady state is unconditional (the cursor parks on the last step instead of stopping): a load
    // under a branch makes hipcc fall back to s_waitcnt vmcnt(0) at the next use of ANY loaded value -- which would wait
    // for the prefetches just issued and expose a full memory round trip every step.
    int f_kc = 0, f_m0 = tile_m0(t_first / ntn), f_tile = t_first, f_left = nsteps - 1;      // fetch cursor (activations)
    auto fetch_advance = [&]() {
        if (f_left > 0) {
            --f_left;
            if (++f_kc == nch) { f_kc = 0; f_tile += t_stride; f_m0 = tile_m0(f_tile / ntn); }
        }
    };
    auto gload = [&](f32x4(&ar)[NLD]) {
        const bool first = f_kc < nch1;
        const int lda = first ? a.lda1 : a.lda2;
        const float* A = (first ? a.A1 + f_kc * BK : a.A2 + (f_kc - nch1) * BK) + (size_t)f_m0 * lda;    // M % BM == 0 (launcher)
        const __amdgpu_buffer_rsrc_t srd = make_srd(A, ((size_t)(span - 1) * lda + BK) * sizeof(float));
        const int v = first ? av1 : av2;
#pragma unroll
        for (int i = 0; i < NLD; ++i) ar[i] = buf_load(srd, v, (POOL ? (i & 1) * pW + (i >> 1) * 32 : i * (NT / 8)) * lda * 4);      // LDS rows 32 i ..: wave row i / 2, block i % 2
        fetch_advance();
    };
    auto sstore = [&](const f32x4(&ar)[NLD], int pro_k, int buf) {      // pro_k < 0: no prologue for this chunk
        f32x4 sc, sh;
        if (pro_k >= 0) { sc = *(const f32x4*)&Ps[0][pro_k + c4 * 4]; sh = *(const f32x4*)&Ps[1][pro_k + c4 * 4]; }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            f32x4 v = ar[i];
            if (pro_k >= 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = fmaxf(fmaf(v[t], sc[t], sh[t]), 0.f);
            }
            *(f32x4*)&As[buf][(r0 + i * (NT / 8)) * PK + c4 * 4] = v;
        }
    };
    auto bgroup = [&](int kc, int n0, int ss, f32x4(&b)[TN]) {
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = buf_load(w_srd, wv + j * 1024, ((kc * 4 + ss) * NB + (n0 >> 5)) * 1024);
    };

    // compute cursor
    int kc = 0, tile = t_first, m0 = tile_m0(tile / ntn), n0 = (tile % ntn) * BN;
    // weights of the step after the current one: same tile unless the current chunk is the last
    auto next_b = [&](int& nkc, int& nn0) {
        nkc = kc + 1; nn0 = n0;
        if (nkc == nch) { nkc = 0; nn0 = (min(tile + t_stride, ntiles - 1) % ntn) * BN; }
    };

#pragma unroll
    for (int ss = 0; ss < 3; ++ss) bgroup(0, n0, ss, bring[ss]);
    gload(ar0);
#if SUO_GEMM_A_DEPTH == 2
    gload(ar1);
#endif
    __syncthreads();                                          // Ps visible
    sstore(ar0, has_pro && 0 < nch1 ? 0 : -1, 0);
    __syncthreads();

    // Epilogue operands: the bias and the residual values of the tile's first 32-row block are requested before anything
    // else of the epilogue (or, with SUO_GEMM_EPI_PREFETCH, during the tile's last K-step); the second block's residuals
    // reuse the same registers and are requested as soon as the first block has been consumed.
    f32x4 rvall[TN][4], bvall[TN];
    auto rload = [&](int i) {
        const __amdgpu_buffer_rsrc_t r_srd = make_srd(HAS_R ? a.R + (size_t)m0 * a.ldr + n0 : a.bias, HAS_R ? ((size_t)(span - 1) * a.ldr + BN) * sizeof(float) : 16);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) rvall[j][k] = buf_load(r_srd, rv, ((i * blk + 8 * k) * a.ldr + j * 32) * 4);
    };
    auto epi_prefetch = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j) bvall[j] = buf_load(bias_srd, bv, (n0 + j * 32) * 4);
        if (HAS_R) rload(0);
    };
    // (Tried, not kept: storing / loading straight in the accumulator layout -- one dword op of a wave covers two full
    // 128-byte row segments, no LDS patch, no wave barriers -- is 2-10 % SLOWER than this transposed 16-byte form: four
    // times the VMEM instructions cost more than the LDS round trips they replace.)
    auto epilogue = [&]() {
        float* T = &Tp[w][0];
        const bool full = !POOL || a.out != nullptr;          // POOL may be asked for the pooled tensor only
        const __amdgpu_buffer_rsrc_t o_srd = make_srd(full ? a.out + (size_t)m0 * a.ldo + n0 : a.bias, full ? ((size_t)(span - 1) * a.ldo + BN) * sizeof(float) : 0);
        // pooled tile: 16 pixels of image row y / 2 per wave row; pixel index m0 / 4 + ... because m0 = (2 W) * (row pair) + 64 * (column block)
        const int pm0 = POOL ? (m0 / (2 * pW)) * (pW >> 1) + ((m0 % (2 * pW)) >> 1) : 0;
        const __amdgpu_buffer_rsrc_t p_srd = make_srd(POOL ? a.pool_out + (size_t)pm0 * a.ldo + n0 : a.bias, POOL ? ((size_t)31 * a.ldo + BN) * sizeof(float) : 0);
        // lane (row q = lane >> 3 of a 16-row patch, k-th half) holds pixel q + 8 k; after max with the lane 8 further (DPP row_ror:8
        // = lane ^ 8 within a row of 16) both lanes of a pair hold the window's maximum: even q stores half 0, odd q half 1
        const int pq = lane >> 3;
        const int pv = ((wm * 16 + (pq >> 1) + 4 * (pq & 1)) * a.ldo + wn * TN * 32 + (lane & 7) * 4) * 4;
        f32x4 keep[POOL ? TN : 1][4];                          // block 0's finished values (image row y) until block 1 (row y + 1) arrives
#if !SUO_GEMM_EPI_PREFETCH
        epi_prefetch();
#endif
        // each 32 x 32 accumulator goes through the 16-row patch in two halves: registers 0-7 hold rows 0-15, 8-15 rows 16-31
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (HAS_R && i > 0) rload(i);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int r = 8 * h; r < 8 * h + 8; ++r) { T[(acc_row_s(r, lane) - 16 * h) * PK + (lane & 31)] = acc[i][j][r]; acc[i][j][r] = 0.f; }
                    __builtin_amdgcn_wave_barrier();
                    f32x4 v[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) v[k] = *(const f32x4*)&T[((lane >> 3) + 8 * k) * PK + (lane & 7) * 4];
                    f32x4 pm[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        f32x4 o = v[k] + bvall[j] + (HAS_R ? rvall[j][2 * h + k] : f32x4{0.f, 0.f, 0.f, 0.f});
                        if (a.relu) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) o[t] = fmaxf(o[t], 0.f);
                        }
                        if (full) buf_store(o, o_srd, ov, ((i * blk + 8 * (2 * h + k)) * a.ldo + j * 32) * 4);     // unpredicated: stores count in vmcnt too
                        if constexpr (POOL) {
                            if (i == 0) keep[j][2 * h + k] = o;
                            else {
#pragma unroll
                                for (int t = 0; t < 4; ++t) {
                                    const float m = fmaxf(o[t], keep[j][2 * h + k][t]);
                                    pm[k][t] = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x128, 0xf, 0xf, false)));
                                }
                            }
                        }
                    }
                    if constexpr (POOL) {
                        if (i == 1) {                           // 8 pooled pixels x 32 channels per pass: 8 h + {0..3} from half 0 (even q), + 4 from half 1 (odd q)
                            const f32x4 o = (pq & 1) ? pm[1] : pm[0];
                            buf_store(o, p_srd, pv, (8 * h * a.ldo + j * 32) * 4);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    };

    // One step = the four k-groups of one 32-channel chunk: before each group's 4*TM*TN MFMAs (from LDS buffer s&1) the
    // weights three groups ahead are requested, before the first also the next activations; afterwards the activations
    // of step s+1 go to the other LDS buffer.
    //   a_free        register set the new activations are fetched into
    //   a_next        register set holding step s+1 (the same set with SUO_GEMM_A_DEPTH 1, the other one with 2)
    auto step = [&](int s, f32x4(&a_free)[NLD], const f32x4(&a_next)[NLD]) {
        const int buf = s & 1;
        int nkc, nn0;
        next_b(nkc, nn0);
        const float* as = &As[buf][((wm * TM * 32) + (lane & 31)) * PK + (lane >> 5) * 4];
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            // weights ride a static ring of 4 k-group slots (slot = group of the chunk), requested 3 groups ahead into
            // the slot the previous group just released; past the last step: a harmless re-read of chunk 0
            if (ss == 0) bgroup(kc, n0, 3, bring[3]);      //               4 no weight loads, 8 no residual loads
            else bgroup(nkc, nn0, ss - 1, bring[ss - 1]);
            if (ss == 0) gload(a_free);
#if SUO_GEMM_EPI_PREFETCH
            if (ss == 1 && kc + 1 == nch) epi_prefetch();      // workgroup-uniform: last K-step of the tile
#endif
#if SUO_GEMM_A_DEPTH == 2
            // two register sets: step s+1's activations were requested a whole step ago, so they can go to the other LDS
            // buffer (free since the barrier that ended step s-1) in the MIDDLE of this step's MFMAs instead of between
            // the last MFMA and the barrier
            if (ss == SUO_GEMM_STORE_AT && !SUO_GEMM_STORE_MIX) sstore(a_next, has_pro && nkc < nch1 ? nkc * BK : -1, buf ^ 1);
#endif
            __builtin_amdgcn_sched_barrier(0);
            f32x4 af[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(as + i * 32 * PK + ss * 8);
#if SUO_GEMM_A_DEPTH == 2 && SUO_GEMM_STORE_MIX
            // same scheduling region as the group's MFMAs: hipcc may slot the prologue VALU and ds_write between them
            if (ss == SUO_GEMM_STORE_AT) sstore(a_next, has_pro && nkc < nch1 ? nkc * BK : -1, buf ^ 1);
#endif
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bring[ss][j][t], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#if SUO_GEMM_A_DEPTH != 2
        sstore(a_next, has_pro && nkc < nch1 ? nkc * BK : -1, buf ^ 1);
#endif
        __syncthreads();
        if (kc + 1 == nch) {                                // workgroup-uniform: the tile is complete
            epilogue();
            kc = 0; tile += t_stride; m0 = tile_m0(tile / ntn); n0 = (tile % ntn) * BN;
        } else {
            ++kc;
        }
    };
    for (int s = 0; s < nsteps; s += 2) {
#if SUO_GEMM_A_DEPTH == 2
        step(s, ar0, ar1);
        if (s + 1 < nsteps) step(s + 1, ar1, ar0);
#else
        step(s, ar0, ar0);
        if (s + 1 < nsteps) step(s + 1, ar0, ar0);
#endif
    }
}

template <int TM, int TN, int WGM, int WGN>
static int launch_persist_cfg(const GemmArgs& a, int max_wgs, hipStream_t s) {
    constexpr int BM = TM * 32 * WGM, BN = TN * 32 * WGN;
    if (a.N % BN || a.M % BM || a.n_valid != a.N) {
        suo_set_error("gemm_persist: M=%d N=%d n_valid=%d must be whole %dx%d tiles", a.M, a.N, a.n_valid, BM, BN);
        return SUO_ERR_ARG;
    }
    if (a.pro_scale && a.K1 > GP_MAX_PRO_K) { suo_set_error("gemm_persist: prologue over K1=%d > %d channels", a.K1, GP_MAX_PRO_K); return SUO_ERR_ARG; }
    const int ntiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    const int g = ntiles < max_wgs ? ntiles : max_wgs;
    if constexpr (TM == 2 && TN == 2 && WGM == 2 && WGN == 2) {
        if (a.pool_out) {
            if (a.R) hipLaunchKernelGGL((gemm_persist_kernel<TM, TN, WGM, WGN, true, true>), dim3(g), dim3(WGM * WGN * 64), 0, s, a, ntiles);
            else hipLaunchKernelGGL((gemm_persist_kernel<TM, TN, WGM, WGN, false, true>), dim3(g), dim3(WGM * WGN * 64), 0, s, a, ntiles);
            SUO_HIP_CHECK(hipGetLastError());
            return SUO_OK;
        }
    }
    if (a.pool_out) { suo_set_error("gemm_persist: the fused max-pool needs the 128x128 configuration"); return SUO_ERR_ARG; }
    if (a.R) hipLaunchKernelGGL((gemm_persist_kernel<TM, TN, WGM, WGN, true>), dim3(g), dim3(WGM * WGN * 64), 0, s, a, ntiles);
    else hipLaunchKernelGGL((gemm_persist_kernel<TM, TN, WGM, WGN, false>), dim3(g), dim3(WGM * WGN * 64), 0, s, a, ntiles);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

// cfg: 1 = 128x128 tiles, 2 = 128x64 tiles, 3 = 64x64 tiles.  The grid is the number of RESIDENT workgroups (3 per CU
// for the 128-row tiles, 6 for the 64x64 one, on the 256-CU part), not the number of tiles.
bool gemm1x1_can_pool(const GemmArgs& a) {
    return a.pool_W > 0 && a.pool_H > 0 && (a.pool_W & 63) == 0 && (a.pool_H & 1) == 0 && a.M % (a.pool_H * a.pool_W) == 0 && (a.N & 127) == 0 &&
           a.n_valid == a.N && a.nchw_hw == 0 && !(a.K1 & 31) && !(a.K2 & 31) && (!a.pro_scale || a.K1 <= GP_MAX_PRO_K);
}

int launch_gemm_persist(const GemmArgs& a, int cfg, hipStream_t s) {
    static const int max_wgs = (int)SUO_TUNE("SUO_GEMM_WGS", 768);       // tuning aid
    if (cfg == 3) return launch_persist_cfg<1, 1, 2, 2>(a, 2 * max_wgs, s);
    if (cfg == 2) return launch_persist_cfg<2, 1, 2, 2>(a, max_wgs, s);
    return launch_persist_cfg<2, 2, 2, 2>(a, max_wgs, s);
}

}  // namespace suo

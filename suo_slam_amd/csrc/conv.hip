// fp32 MFMA convolution kernels for the stacked-hourglass keypoint CNN (gfx950 / CDNA4).
//
// Reference ops replaced (all stock ATen/cuDNN in the reference):
//   Conv2d 1x1  + BatchNorm2d(eval) + ReLU + residual add   /root/reference/lib/models/layers/Residual.py:20-35
//   Conv2d 3x3 p1 + BN + ReLU                               /root/reference/lib/models/layers/Residual.py:12-14,27-29
//   Conv2d 7x7 s2 p3 + BN + ReLU (stem)                     /root/reference/lib/models/hg.py:67-69,96-98
//
// Design (MI355X-first, not a cuDNN translation):
//  * activations are NHWC so the GEMM-K dimension (input channels) is contiguous;
//  * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD): rows = pixels, cols = output channels;
//  * A (activations) is staged once per K-chunk through LDS (halo tile for KxK: each of the 9/49
//    taps re-reads the same LDS tile at a shifted address), with the pre-activation BN+ReLU of
//    Residual.forward applied while staging (prologue), so the raw tensor is never re-written;
//  * B (weights) is pre-packed at load time so each lane fetches its next 4 MFMA operands with one
//    coalesced 16-byte load straight from L2 -- no LDS traffic for weights;
//  * the K loop is software-pipelined: next chunk's global loads are issued before the current
//    chunk's MFMAs and written to the other LDS buffer afterwards (one barrier per chunk);
//  * epilogue fuses bias (BN folded on the host), residual add, ReLU, and for the final head the
//    NCHW transpose (operands swapped so the wave's 32 columns are 32 consecutive pixels).
#include <type_traits>

#include "buffer_ops.h"
#include "suo_internal.h"
#include "tune.h"

#ifndef SUO_CONV_SCALAR_WAVE
#define SUO_CONV_SCALAR_WAVE 1
#endif
#ifndef SUO_CONV_AF_PIPE
#define SUO_CONV_AF_PIPE 0          // 1: A fragments one k-group ahead through a register ring (+8 VGPRs; measured slower:
#endif                              //    with 3 workgroups per CU the other waves already hide the LDS latency)
#ifndef SUO_CONV_WAVES_PER_EU
#define SUO_CONV_WAVES_PER_EU 2     // amdgpu_waves_per_eu(2) is a FLOOR: it caps the allocation at 256 registers per lane.  What is
#endif                              // resident is decided by what the build then uses: the plain 3x3 kernels take 144 registers and
                                    // 51.8 KB of LDS = THREE workgroups per CU (measured 140.9 TFLOP/s; forced to two: 139.2, to one:
                                    // 130.2 -- SUO_CONV_DYN_LDS experiment, round 2); the fused conv2 -> conv3 kernel takes 210 = two.
#ifndef SUO_CONV_BRING3
#define SUO_CONV_BRING3 4           // weight-ring slots of the 3x3 kernels (must divide 9 * CK / 8)
#endif
#ifndef SUO_GEMM_SCALAR_WAVE
#define SUO_GEMM_SCALAR_WAVE 1
#endif

namespace suo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// -DSUO_CONV_PROFILE (tools/micro/conv_prof.hip only): per-workgroup timestamps of the KxK kernel's phases
#ifdef SUO_CONV_PROFILE
__device__ long long g_conv_prof[16384 * 8];
__device__ long long g_conv_prof_clk[16384 * 8];     // the same points on the shader clock (s_memtime)
#define CPROF(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) { g_conv_prof[blockIdx.x * 8 + (i)] = wall_clock64(); \
                                                                    g_conv_prof_clk[blockIdx.x * 8 + (i)] = clock64(); } } while (0)
#else
#define CPROF(i) do { } while (0)
#endif

size_t packed_weight_floats(int n_pad, int k_pad) { return (size_t)n_pad * (size_t)k_pad; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// row of accumulator register r for this lane inside a 32x32 tile (col = lane & 31)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// =================================================================================================
// 1x1 convolution: out[M,N] = epi( pro(A1)[M,K1] * W1 + A2[M,K2] * W2 + bias (+ R) )
// =================================================================================================
template <int TM, int TN, int WGM, int WGN, bool NCHW>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm1x1_kernel(const GemmArgs a) {
    constexpr int BK = 32, PK = BK + 4;
    constexpr int BM = TM * 32 * WGM, BN = TN * 32 * WGN, NT = WGM * WGN * 64;
    constexpr int NLD = BM * 8 / NT;
    static_assert(BM * 8 % NT == 0, "staging must divide evenly");
    __shared__ __attribute__((aligned(16))) float As[2][BM * PK];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = SUO_GEMM_SCALAR_WAVE ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);     // wave index as an SGPR
    const int wm = w / WGN, wn = w % WGN;
    // 1-D grid, XCD-aware order (workgroup b runs on XCD b % 8): each XCD gets a contiguous run of tiles with the
    // N-tiles of one pixel tile adjacent, so the A tile they share is fetched from HBM once and re-read from that L2
    const int ntn = a.N / BN;
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    const int nch1 = a.K1 >> 5, nch = nch1 + (a.K2 >> 5);
    const int NB = a.N >> 5;
    const int c4 = tid & 7, r0 = tid >> 3;
    const bool has_pro = a.pro_scale != nullptr;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 areg[NLD];
    f32x4 bcur[4][TN], bnxt[4][TN];
    f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};

    auto gload = [&](int kc) {
        const float* A;
        int lda, kk;
        if (kc < nch1) { A = a.A1; lda = a.lda1; kk = kc * BK; }
        else { A = a.A2; lda = a.lda2; kk = (kc - nch1) * BK; }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = m0 + r0 + i * (NT / 8);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < a.M) v = *(const f32x4*)(A + (size_t)row * lda + kk + c4 * 4);
            areg[i] = v;
        }
        if (has_pro && kc < nch1) {
            psc = *(const f32x4*)(a.pro_scale + kk + c4 * 4);
            psh = *(const f32x4*)(a.pro_shift + kk + c4 * 4);
        }
    };
    auto sstore = [&](int kc, int buf) {
        const bool pro = has_pro && kc < nch1;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            f32x4 v = areg[i];
            if (pro) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = fmaxf(fmaf(v[t], psc[t], psh[t]), 0.f);
            }
            *(f32x4*)&As[buf][(r0 + i * (NT / 8)) * PK + c4 * 4] = v;
        }
    };
    auto bload = [&](int kc, f32x4(&b)[4][TN]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nb = (n0 >> 5) + wn * TN + j;
                b[s][j] = *(const f32x4*)(a.Wp + ((size_t)((kc * 4 + s) * NB + nb) * 64 + lane) * 4);
            }
    };

    gload(0);
    bload(0, bcur);
    sstore(0, 0);
    __syncthreads();

    // One K-chunk: prefetch chunk kc+1 (A into registers, B into the OTHER weight buffer), then 16*TM*TN MFMAs from
    // LDS buffer kc&1 and weight buffer `b`.  The two weight buffers swap roles every chunk (static ring: no
    // register copies, so hipcc keeps the loads a full chunk ahead of their first use).
    auto chunk = [&](int kc, const f32x4(&b)[4][TN], f32x4(&bn)[4][TN]) {
        const int buf = kc & 1;
        const bool more = kc + 1 < nch;
        if (more) gload(kc + 1);
        bload(more ? kc + 1 : kc, bn);
        __builtin_amdgcn_sched_barrier(0);
        const float* as = &As[buf][((wm * TM * 32) + (lane & 31)) * PK + (lane >> 5) * 4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 af[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(as + i * 32 * PK + s * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (NCHW) acc[i][j] = mfma32(b[s][j][t], af[i][t], acc[i][j]);
                        else acc[i][j] = mfma32(af[i][t], b[s][j][t], acc[i][j]);
                    }
        }
        if (more) sstore(kc + 1, buf ^ 1);
        __syncthreads();
    };
    for (int kc = 0; kc < nch; kc += 2) {
        chunk(kc, bcur, bnxt);
        if (kc + 1 < nch) chunk(kc + 1, bnxt, bcur);
    }

    // ---- epilogue --------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!NCHW) {
                // transpose the 32x32 accumulator tile through a wave-private LDS patch (the A buffers are free after
                // the last barrier) so every lane owns 4 consecutive channels of one pixel: 16-byte residual loads and
                // stores, 8 full 128-byte lines per instruction, all residual loads in flight before the first store
                float* T = &As[0][0] + w * (32 * PK);
#pragma unroll
                for (int r = 0; r < 16; ++r) T[acc_row(r, lane) * PK + (lane & 31)] = acc[i][j][r];
                __builtin_amdgcn_wave_barrier();
                const int col = n0 + (wn * TN + j) * 32 + (lane & 7) * 4;
                const f32x4 bv = *(const f32x4*)(a.bias + col);
                f32x4 v[4], rv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = m0 + (wm * TM + i) * 32 + (lane >> 3) + 8 * k;
                    rv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (a.R && row < a.M && col < a.n_valid) rv[k] = *(const f32x4*)(a.R + (size_t)row * a.ldr + col);
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
                    if (row < a.M && col < a.n_valid) *(f32x4*)(a.out + (size_t)row * a.ldo + col) = o;
                }
                __builtin_amdgcn_wave_barrier();
            } else {
                const int m = m0 + (wm * TM + i) * 32 + (lane & 31);
                const int crop = m / a.nchw_hw, pix = m - crop * a.nchw_hw;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + (wn * TN + j) * 32 + acc_row(r, lane);
                    if (m < a.M && n < a.n_valid) {
                        float v = acc[i][j][r] + a.bias[n];
                        if (a.relu) v = fmaxf(v, 0.f);
                        a.out[((size_t)crop * a.n_valid + n) * a.nchw_hw + pix] = v;
                    }
                }
            }
        }
}

template <int TM, int TN, int WGM, int WGN, bool NCHW>
static int launch_gemm_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = TM * 32 * WGM, BN = TN * 32 * WGN;
    if (a.N % BN) { suo_set_error("gemm1x1: N=%d not a multiple of %d", a.N, BN); return SUO_ERR_ARG; }
    dim3 grid(((a.M + BM - 1) / BM) * (a.N / BN));
    hipLaunchKernelGGL((gemm1x1_kernel<TM, TN, WGM, WGN, NCHW>), grid, dim3(WGM * WGN * 64), 0, s, a);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

int launch_gemm1x1(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || (a.K1 & 31) || (a.K2 & 31) || (a.N & 63) || a.K1 <= 0) {
        suo_set_error("gemm1x1: bad shape M=%d K1=%d K2=%d N=%d", a.M, a.K1, a.K2, a.N);
        return SUO_ERR_ARG;
    }
    if (a.pool_out) {                                    // fused 2x2 max-pool: the persistent 128x128 kernel only (csrc/gemm_persist.hip: POOL)
        if (!gemm1x1_can_pool(a)) { suo_set_error("gemm1x1: shape M=%d N=%d H=%d W=%d cannot take the fused max-pool", a.M, a.N, a.pool_H, a.pool_W); return SUO_ERR_ARG; }
        return launch_gemm_persist(a, 1, s);
    }
    if (!a.out) { suo_set_error("gemm1x1: no output"); return SUO_ERR_ARG; }
    if (a.nchw_hw > 0) return launch_gemm_cfg<1, 1, 2, 2, true>(a, s);
    // small feature maps are latency problems: 16x16 tiles + split-K (csrc/conv_small.hip)
    if (a.M <= 4096 && ((a.K1 | a.K2) & 15) == 0 && (a.K1 + a.K2) <= 640) return launch_gemm_small(a, s);
    // tile choice: keep >= ~2 workgroups per CU where the problem allows it
    const long tiles128 = (long)((a.M + 127) / 128) * (a.N / 128 > 0 ? a.N / 128 : 1);
    const long tiles12864 = (long)((a.M + 127) / 128) * (a.N / 64);
    static const int force_cfg = (int)SUO_TUNE("SUO_GEMM_CFG", 0);           // tuning aid only
    int cfg = ((a.N % 128) == 0 && tiles128 >= 512) ? 1 : (tiles12864 >= 384 ? 2 : 3);      // 128x128 | 128x64 | 64x64
    if (force_cfg >= 2 || (force_cfg == 1 && (a.N % 128) == 0)) cfg = force_cfg;
    // whole tiles (every shape of this network): the persistent, branch-free kernel of csrc/gemm_persist.hip
    static const int persist = (int)SUO_TUNE("SUO_GEMM_PERSIST", 1);     // 0: A/B against the one-tile kernel
    if (persist && a.n_valid == a.N && a.M % (cfg == 3 ? 64 : 128) == 0) return launch_gemm_persist(a, cfg, s);
    if (cfg == 1) return launch_gemm_cfg<2, 2, 2, 2, false>(a, s);
    if (cfg == 2) return launch_gemm_cfg<2, 1, 2, 2, false>(a, s);
    return launch_gemm_cfg<1, 1, 2, 2, false>(a, s);
}

// =================================================================================================
// KxK convolution as implicit GEMM over an LDS-resident halo tile
// =================================================================================================
// FUSE (3x3, 128-pixel x 128-channel tiles only): the workgroup holds the COMPLETE conv2 tile of a Residual block in its
// accumulators, so conv3 (1x1, 128 -> 256) + bias + skip runs right here (Residual.py:27-35): relu(acc + bias2) goes
// accumulator -> LDS (32 channels at a time, double-buffered) -> MFMA A operand, the 128-channel `mid` tensor never
// exists in HBM.  Output channels in two passes of 128 (a second 64-register accumulator set; 2 workgroups per CU).
// Same summation order as the separate launches (K chunks of 32 in order, bias after the sum, then the skip): the fused
// block is bit-identical to conv3x3 followed by gemm_persist.
template <int KS, int ST, int CK, int TH, int TW, int TM, int TN, int WGM, int WGN, bool FUSE = false>
__global__ __launch_bounds__(WGM* WGN * 64) __attribute__((amdgpu_waves_per_eu(SUO_CONV_WAVES_PER_EU))) void convk_kernel(const ConvArgs a) {
    constexpr int PAD = KS / 2;
    constexpr int BM = TM * 32 * WGM, BN = TN * 32 * WGN, NT = WGM * WGN * 64;
    static_assert(BM == TH * TW, "pixel tile mismatch");
    // CK == 4 (the image-only stem: 3 real channels + 1 pad in the staging buffer): the GEMM-K axis is the DENSE sequence
    // kk = tap * 3 + channel (147 terms), two consecutive kk per MFMA k-step (lanes 0-31 the even one, lanes 32-63 the odd
    // one).  No k-step multiplies padding (76 instead of 196 MFMAs per accumulator), and since v_mfma_f32_32x32x2_f32 adds
    // its two products in k order, the nonzero products are summed in exactly the order of the 8- / 48-channel kernels
    // (tap-major, channel-minor, zeros in between): bit-identical results.
    constexpr bool PAIR = CK == 4;
    constexpr int KDENSE = KS * KS * 3;
    constexpr int PK = PAIR ? 5 : CK + 4, S = PAIR ? 1 : CK / 8, C4 = CK / 4;      // (pitch 5: stride-2 pixel reads spread over the banks)
    constexpr int IH = (TH - 1) * ST + KS, IW = (TW - 1) * ST + KS, NPIX = IH * IW;
    constexpr int NF4 = NPIX * C4, NLD = (NF4 + NT - 1) / NT;
    constexpr int ASZ = NPIX * PK > WGM * WGN * 32 * 36 / 2 ? NPIX * PK : WGM * WGN * 32 * 36 / 2;      // (also hosts the epilogue patches)
    __shared__ __attribute__((aligned(16))) float As[2][ASZ];

    const int tid = threadIdx.x, lane = tid & 63;
    // Wave index as an SGPR: the weight addresses and tile offsets derived from it become scalar, which frees 50-70
    // VGPRs (120 + 16 instead of 192).  Not for the 2x2-tile configuration: there the scalarised addresses let hipcc
    // hoist more loads and the allocation jumps from 136 + 64 to 248 + 64 registers, i.e. one wave per SIMD.
    const int w = SUO_CONV_SCALAR_WAVE && (TM * TN < 4 || SUO_CONV_SCALAR_WAVE > 1) ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int wm = w / WGN, wn = w % WGN;
    const int tiles_x = (a.OW + TW - 1) / TW, tiles_y = (a.OH + TH - 1) / TH;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed placement; used for speed only), so give each
    // XCD a contiguous run of tiles -- neighbouring tiles share halo rows / columns and then hit the same L2
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int l = bid / (tiles_x * tiles_y);
    bid -= l * tiles_x * tiles_y;
    const int ty = bid / tiles_x, tx = bid - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * ST - PAD, ix0 = ox0 * ST - PAD;
    const int n0 = blockIdx.y * BN;
    const int NB = a.N >> 5;
    const int nch = a.C / CK;
    const float* in_l = a.in + (size_t)l * a.H * a.W * a.C;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // k-groups: G per channel chunk (one group = 8 channels of one tap = 4 MFMA k-steps per accumulator tile).
    // Weights travel through a STATIC ring of R group slots (group gg of the whole K loop lives in slot gg % R and is
    // requested R-1 groups before its MFMAs; G % R == 0 keeps the slot of every unrolled group a compile-time constant
    // across chunks -- no register copies for hipcc to coalesce away, no second copy of the loop body).
    constexpr int G = PAIR ? ((KDENSE + 1) / 2 + 3) / 4 : KS * KS * S;
    constexpr int R = KS == 3 ? SUO_CONV_BRING3 : (PAIR ? 4 : 7);
    static_assert((PAIR || G % R == 0) && R >= 2, "weight ring must divide the groups of a chunk (single-chunk PAIR mode excepted)");
    // activations: the halo tile of the next chunk is fetched in two halves (registers for half a tile only)
    constexpr int NLH = (NLD + 1) / 2;
    f32x4 areg[NLH];
    f32x4 bring[R][TN];

    const __amdgpu_buffer_rsrc_t in_srd = make_srd(in_l, (size_t)a.H * a.W * a.C * sizeof(float));
    const __amdgpu_buffer_rsrc_t w_srd = make_srd(a.Wp, (size_t)nch * G * a.N * 8 * sizeof(float));      // G k-groups of N x 8 weights per chunk
    const __amdgpu_buffer_rsrc_t out_srd = make_srd(a.out + (size_t)l * a.OH * a.OW * a.N, (size_t)a.OH * a.OW * a.N * sizeof(float));
    // halo staging: byte offset of this thread's i-th 16-byte piece inside the crop (channel chunk 0), out of range
    // for the zero padding and for the unused tail of the last round
    int avoff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * NT;
        const int pix = idx / C4, cc = idx - pix * C4;
        const int py = pix / IW, px = pix - py * IW;
        const int iy = iy0 + py, ix = ix0 + px;
        const bool ok = idx < NF4 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        avoff[i] = ok ? ((iy * a.W + ix) * a.C + cc * 4) * 4 : BUF_OOB;
    }
    auto gload = [&](int c, int h) {
#pragma unroll
        for (int i = 0; i < NLH; ++i)
            if (h * NLH + i < NLD) areg[i] = buf_load(in_srd, avoff[h * NLH + i], c * CK * 4);
    };
    auto sstore = [&](int buf, int h) {
#pragma unroll
        for (int i = 0; i < NLH; ++i) {
            const int idx = tid + (h * NLH + i) * NT;
            if (h * NLH + i < NLD && idx < NF4) {
                const int pix = idx / C4, cc = idx - pix * C4;
                if (PAIR) {                                          // 3 real channels at pitch 5 (no 16-byte alignment)
                    float* d = &As[buf][pix * PK];
                    d[0] = areg[i][0]; d[1] = areg[i][1]; d[2] = areg[i][2];
                } else {
                    *(f32x4*)&As[buf][pix * PK + cc * 4] = areg[i];
                }
            }
        }
    };
    // weights: Wp[gg][nb][lane][4] with gg = chunk * G + group; the lane / N-tile part is one VGPR, gg an SGPR
    const int wvoff = (((n0 >> 5) + wn * TN) * 64 + lane) * 16;
    const int gtot = nch * G;
    auto bload = [&](int gg, f32x4(&b)[TN]) {
        const int gc = gg < gtot ? gg : gtot - 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = buf_load(w_srd, wvoff + j * 1024, gc * NB * 1024);
    };

    int abase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = (wm * TM + i) * 32 + (lane & 31);
        const int py = p / TW, px = p - py * TW;
        abase[i] = ((py * ST) * IW + px * ST) * PK + (PAIR ? 0 : (lane >> 5) * 4);
    }

    CPROF(0);
    gload(0, 0);
#pragma unroll
    for (int r = 0; r < R - 1; ++r) bload(r, bring[r]);
    sstore(0, 0);
    gload(0, 1);
    sstore(0, 1);
    __syncthreads();
    CPROF(1);

    // A fragments of group g (tap t = g / S, channels s*8 .. s*8+7 of the chunk): one ds_read_b128 per M-tile
    auto aread = [&](const float* as, int g, f32x4(&af)[TM]) {
        if (PAIR) {                                                  // k-step t of group g: kk = 8 g + 2 t + (lane >> 5)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ka = 8 * g + 2 * t < KDENSE ? 8 * g + 2 * t : KDENSE - 1;          // (past the end: zero weights)
                const int kb = 8 * g + 2 * t + 1 < KDENSE ? 8 * g + 2 * t + 1 : KDENSE - 1;
                const int offa = (((ka / 3) / KS) * IW + ((ka / 3) % KS)) * PK + ka % 3;
                const int offb = (((kb / 3) / KS) * IW + ((kb / 3) % KS)) * PK + kb % 3;
                const int off = (lane >> 5) ? offb : offa;
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i][t] = as[abase[i] + off];
            }
            return;
        }
        const int t = g / S, s = g - t * S;
        const int toff = ((t / KS) * IW + (t % KS)) * PK;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(as + abase[i] + toff + s * 8);
    };
    auto group = [&](const f32x4(&af)[TM], const f32x4(&bs)[TN]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(af[i][t], bs[j][t], acc[i][j]);
    };
    // (Tried and measured, not kept: sched_group_barrier hints instead of the explicit fragment ring -- hipcc then
    // pipelines the LDS reads but the 2x2 configuration grows past 256 registers.  DESIGN.md section 4.)

    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        const float* as = &As[buf][0];
        const bool more = c + 1 < nch;
        // the G groups of this chunk, fully unrolled; every load is requested well ahead of its use and pinned there
        f32x4 afr[2][TM];
        aread(as, 0, afr[0]);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g == 0 && more) gload(c + 1, 0);
            if (g == G / 2 && more) { sstore(buf ^ 1, 0); gload(c + 1, 1); }
            bload(c * G + g + R - 1, bring[(g + R - 1) % R]);
#if SUO_CONV_AF_PIPE
            if (g + 1 < G) aread(as, g + 1, afr[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            group(afr[g & 1], bring[g % R]);
#else
            __builtin_amdgcn_sched_barrier(0);
            aread(as, g, afr[0]);
            group(afr[0], bring[g % R]);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) sstore(buf ^ 1, 1);
        __syncthreads();
        if (c < 4) CPROF(2 + c);
    }

    if constexpr (FUSE) {
        static_assert(KS == 3 && TM == 2 && TN == 2 && WGM == 2 && WGN == 2 && TW == 16 && !PAIR, "fused conv3: 128 x 128 tiles");
        constexpr int MP = 36;                                   // pitch of a staged 32-channel slice of the mid tile
        constexpr int MSZ = BM * MP;
        static_assert(2 * MSZ <= 2 * ASZ && 2 * ASZ >= WGM * WGN * 32 * 36, "mid slices / epilogue patches must fit the A buffers");
        float* M2 = &As[0][0];
        const int NB2 = a.N2 >> 5;
        const __amdgpu_buffer_rsrc_t w3_srd = make_srd(a.W3p, (size_t)a.N * a.N2 * sizeof(float));
        const size_t crop2 = (size_t)a.OH * a.OW * a.N2;
        const __amdgpu_buffer_rsrc_t r_srd = make_srd(a.R + (size_t)l * crop2, crop2 * sizeof(float));
        const __amdgpu_buffer_rsrc_t o2_srd = make_srd(a.out2 + (size_t)l * crop2, crop2 * sizeof(float));
        float b2v[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) b2v[j] = a.bias[(wn * TN + j) * 32 + (lane & 31)];
        // slice kc = mid channels [32 kc, 32 kc + 32) = accumulator column tile (wn, j) = (kc >> 1, kc & 1): its two owner
        // waves write relu(acc + bias2) transposed (lane & 31 = channel: 32 consecutive floats per pixel row)
        auto stage = [&](int kc, int buf) {
            if (wn == (kc >> 1)) {
                float* d = M2 + buf * MSZ;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = (kc & 1) ? acc[i][1][r] + b2v[1] : acc[i][0][r] + b2v[0];
                        d[((wm * TM + i) * 32 + acc_row(r, lane)) * MP + (lane & 31)] = fmaxf(v, 0.f);
                    }
            }
        };
        // conv3 weights: Wp[kg][nb][lane][4] (kg = k-group of 8 mid channels, nb = 32-channel tile of the 256 outputs),
        // through the same static 4-slot ring as the 3x3 weights: 32 groups (2 passes x 16), requested 3 groups ahead
        const int w3voff = lane * 16;
        auto b3load = [&](int q, f32x4(&b)[TN]) {               // q = pass * 16 + kg
            const int qc = q < 32 ? q : 31;
            const int kg = qc & 15, nb0 = (qc >> 4) * 4 + wn * TN;
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = buf_load(w3_srd, w3voff, (kg * NB2 + nb0 + j) * 1024);
        };
        __syncthreads();                                         // every wave is done with the last activation chunk
#pragma unroll
        for (int r = 0; r < R - 1; ++r) b3load(r, bring[r]);
        stage(0, 0);
        __syncthreads();
#pragma unroll 1
        for (int p = 0; p < 2; ++p) {
            f32x16 acc2[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const int buf = kc & 1;
                if (kc + 1 < 4) stage(kc + 1, buf ^ 1);
                const float* ms = M2 + buf * MSZ + ((wm * TM) * 32 + (lane & 31)) * MP + (lane >> 5) * 4;
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    const int q = kc * 4 + sg;                   // static slot: 16 groups per pass, R divides 16
                    b3load(p * 16 + q + R - 1, bring[(q + R - 1) % R]);
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 af[TM];
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(ms + i * 32 * MP + sg * 8);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) acc2[i][j] = mfma32(af[i][t], bring[q % R][j][t], acc2[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
            }
            // epilogue of this pass: + bias3 + skip, 16-byte stores (wave-private patches over the now idle slices)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float* T = M2 + w * (32 * 36);
                    const int col = p * 128 + (wn * TN + j) * 32 + (lane & 7) * 4;
                    const f32x4 bv = *(const f32x4*)(a.bias3 + col);
                    int off[4];
                    f32x4 rv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int pp = (wm * TM + i) * 32 + (lane >> 3) + 8 * k;
                        const int oy = oy0 + pp / TW, ox = ox0 + pp % TW;
                        off[k] = (oy < a.OH && ox < a.OW) ? ((oy * a.OW + ox) * a.N2 + col) * 4 : BUF_OOB;
                        rv[k] = buf_load(r_srd, off[k], 0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) T[acc_row(r, lane) * 36 + (lane & 31)] = acc2[i][j][r];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4 o = (*(const f32x4*)&T[((lane >> 3) + 8 * k) * 36 + (lane & 7) * 4] + bv) + rv[k];
                        buf_store(o, o2_srd, off[k]);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            if (p == 0) {
                __syncthreads();                                 // patches done before slice 0 is staged again
                stage(0, 0);
                __syncthreads();
            }
        }
        return;
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            // LDS transpose of the accumulator tile (wave-private patch of the free A buffers) -> 16-byte stores
            static_assert(2 * ASZ >= WGM * WGN * 32 * 36, "epilogue patch must fit the A buffers");
            float* T = &As[0][0] + w * (32 * 36);
#pragma unroll
            for (int r = 0; r < 16; ++r) T[acc_row(r, lane) * 36 + (lane & 31)] = acc[i][j][r];
            __builtin_amdgcn_wave_barrier();
            const int col = n0 + (wn * TN + j) * 32 + (lane & 7) * 4;
            const f32x4 bv = *(const f32x4*)(a.bias + col);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pr = (lane >> 3) + 8 * k;
                const int p = (wm * TM + i) * 32 + pr;
                const int py = p / TW, px = p - py * TW;
                const int oy = oy0 + py, ox = ox0 + px;
                f32x4 o = *(const f32x4*)&T[pr * 36 + (lane & 7) * 4] + bv;
                if (a.relu) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = fmaxf(o[t], 0.f);
                }
                buf_store(o, out_srd, (oy < a.OH && ox < a.OW) ? ((oy * a.OW + ox) * a.N + col) * 4 : BUF_OOB);
            }
            __builtin_amdgcn_wave_barrier();
        }
    CPROF(6);
#ifdef SUO_CONV_PROFILE
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        unsigned hw_id, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(xcc));
        g_conv_prof[blockIdx.x * 8 + 7] = ((long long)xcc << 32) | hw_id;
    }
#endif
}

template <int KS, int ST, int CK, int TH, int TW, int TM, int TN, int WGM, int WGN, bool FUSE = false>
static int launch_conv_cfg(const ConvArgs& a, hipStream_t s) {
    constexpr int BN = TN * 32 * WGN;
    if (a.N % BN || a.C % CK) { suo_set_error("conv%dx%d: N=%d C=%d unsupported", KS, KS, a.N, a.C); return SUO_ERR_ARG; }
    const int tiles = ((a.OW + TW - 1) / TW) * ((a.OH + TH - 1) / TH) * a.L;
    dim3 grid(tiles, a.N / BN);
    static const int dyn_lds = (int)SUO_TUNE("SUO_CONV_DYN_LDS", 0);      // occupancy experiments only
    hipLaunchKernelGGL((convk_kernel<KS, ST, CK, TH, TW, TM, TN, WGM, WGN, FUSE>), grid, dim3(WGM * WGN * 64), dyn_lds, s, a);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

int launch_conv3x3(const ConvArgs& a, hipStream_t s) {
    if (a.OH != a.H || a.OW != a.W || (a.C & 31) || (a.N & 63)) {
        suo_set_error("conv3x3: bad shape H=%d W=%d C=%d N=%d", a.H, a.W, a.C, a.N);
        return SUO_ERR_ARG;
    }
    const long px = (long)a.L * a.OH * a.OW;
    if (px <= 4096 && (a.C == 128 || a.C == 64 || a.C == 32)) return launch_conv3x3_small(a, s);
    const long t128 = ((px + 127) / 128) * (a.N / 64);
    static const int force = (int)SUO_TUNE("SUO_CONV3_CFG", 0);     // tuning aid only
    if (force == 1) return launch_conv_cfg<3, 1, 32, 8, 8, 1, 1, 2, 2>(a, s);
    if (force == 2 && (a.N % 128) == 0) return launch_conv_cfg<3, 1, 32, 8, 16, 2, 2, 2, 2>(a, s);
    if (force == 3) return launch_conv_cfg<3, 1, 32, 8, 16, 1, 1, 4, 2>(a, s);
    if (force == 4) return launch_conv_cfg<3, 1, 32, 8, 16, 2, 1, 2, 2>(a, s);
    // 128-pixel x 128-channel workgroup tiles when all output channels fit one tile: every activation tile is then
    // fetched once (PMC: HBM fetch 2.8x -> 1.4x of the input, the 1.4x being the 3x3 halo)
    // -- but only with >= 8 such tiles per CU: three workgroups share a CU, and a handful of long tiles per CU quantises
    // badly (tools/bench_conv_shapes.py, 128 crops: 32x32 maps 117 -> 138 TFLOP/s with 128 x 64 tiles, 16x16 130 -> 134)
    const bool big = a.OH >= 8 && a.OW >= 16;
    if (big && (a.N % 128) == 0 && ((px + 127) / 128) * (a.N / 128) >= 2048) return launch_conv_cfg<3, 1, 32, 8, 16, 2, 2, 2, 2>(a, s);
    if (big && (a.N % 128) == 0 && t128 >= 1024) return launch_conv_cfg<3, 1, 32, 8, 16, 1, 1, 4, 2>(a, s);
    if (big && t128 >= 2048) return launch_conv_cfg<3, 1, 32, 8, 16, 2, 1, 2, 2>(a, s);
    const long t64 = ((px + 63) / 64) * (a.N / 64);
    if (a.OH >= 8 && a.OW >= 8 && t64 >= 256) return launch_conv_cfg<3, 1, 32, 8, 8, 1, 1, 2, 2>(a, s);
    return launch_conv_cfg<3, 1, 32, 4, 8, 1, 1, 1, 2>(a, s);
}

// conv2 (3x3, 128 -> 128) + conv3 (1x1, 128 -> 256) + skip of a Residual block in one launch (FUSE above)
bool conv3x3_fusable(const ConvArgs& a) {
    const long px = (long)a.L * a.OH * a.OW;
    static const int fuse = (int)SUO_TUNE("SUO_CONV_FUSE", 1);              // 0: A/B against the separate launches
    static const long min_tiles = (long)SUO_TUNE("SUO_CONV_FUSE_TILES", 1024);
    return fuse && a.N == 128 && a.C == 128 && a.OH == a.H && a.OW == a.W && a.OH >= 8 && a.OW >= 16 && (px + 127) / 128 >= min_tiles;
}
int launch_conv3x3_fused(const ConvArgs& a, hipStream_t s) {
    if (!conv3x3_fusable(a) || a.N2 != 256 || !a.W3p || !a.bias3 || !a.R || !a.out2) {
        suo_set_error("conv3x3_fused: unsupported shape C=%d N=%d N2=%d", a.C, a.N, a.N2);
        return SUO_ERR_ARG;
    }
    return launch_conv_cfg<3, 1, 32, 8, 16, 2, 2, 2, 2, true>(a, s);
}

int launch_conv7x7s2(const ConvArgs& a, hipStream_t s) {
    if (a.OH * 2 != a.H || a.OW * 2 != a.W || ((a.C & 7) && a.C != 4) || (a.N & 63)) {
        suo_set_error("conv7x7s2: bad shape H=%d W=%d C=%d N=%d", a.H, a.W, a.C, a.N);
        return SUO_ERR_ARG;
    }
    // image-only stem: 3 channels.  As one 8-channel chunk the kernel is MFMA-bound on PADDED work (5 of 8 channels are
    // zeros: 777 us at 128 crops; 8 x 16-pixel tiles with 4 / 8 waves are slower, 823 / 810 us); with 4-channel staging
    // (what the network uses, suo_internal.h: IMG_C) the K axis is dense -- see PAIR in convk_kernel.
    if (a.C == 4) return launch_conv_cfg<7, 2, 4, 8, 8, 1, 1, 2, 2>(a, s);
    if (a.C == 8) return launch_conv_cfg<7, 2, 8, 8, 8, 1, 1, 2, 2>(a, s);
    if (a.C & 15) { suo_set_error("conv7x7s2: C=%d must be 8 or a multiple of 16", a.C); return SUO_ERR_ARG; }
    return launch_conv_cfg<7, 2, 16, 8, 8, 1, 1, 2, 2>(a, s);
}

}  // namespace suo

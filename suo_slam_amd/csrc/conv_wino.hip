// 3x3 convolution by Winograd F(2x2, 3x3) on fp32 MFMA (gfx950): 16 element-wise products per 2x2 output tile and channel
// pair instead of 36 -- 2.25x fewer MFMA MACs than the direct form of csrc/conv.hip, for the 128 -> 128 channel convolutions
// of the Residual blocks (/root/reference/lib/models/layers/Residual.py:12-14,27-29: 60 % of a network call).
//
//   Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A          (Lavin & Gray; correlation form, as torch.nn.Conv2d)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
//
// Mapping (one workgroup = 8 x 16 output pixels = 32 Winograd tiles x 128 output channels, 4 waves):
//   * U = G g G^T is computed on the host in fp64, rounded once, packed per (channel chunk, component, k-group, 32-channel
//     tile) in MFMA B-operand order; it streams from L2 through a static register ring like the direct kernel's weights;
//   * per 16-channel chunk the 10 x 18 halo tile is staged in LDS (double-buffered, register prefetch), then ALL threads
//     transform it: thread (tile, channel quad, half) reads 3 x 4 halo pixels and writes 8 of the 16 components of
//     V = B^T d B (additions only) into LDS in A-operand order;
//   * MFMA phase: the 32 tiles are the 32 rows of v_mfma_f32_32x32x2_f32, wave w owns output channels [32 w, 32 w + 32):
//     per component 8 MFMAs (K = 16) into a scratch accumulator, which is then added with its A^T (.) A sign (0, +1, -1)
//     to the four output-position accumulators -- the output transform costs VALU additions, never MFMAs;
//   * two barriers per chunk (V ready / V free); two workgroups per CU (70 KB of LDS each) run the phases against each other.
// Numerics: fp32 throughout; the transforms add at most 4 terms on either side, so the result differs from the direct
// kernel by summation order and by the rounding of U (observed <= 2e-6 of the output range).
#include "buffer_ops.h"
#include "suo_internal.h"
#include "tune.h"

namespace suo {

typedef float w_f32x4 __attribute__((ext_vector_type(4)));
typedef float w_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ w_f32x16 w_mfma32(float a, float b, w_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int w_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

constexpr int W_CK = 16, W_PKH = 20, W_PKV = 20, W_TH = 8, W_TW = 16, W_IH = 10, W_IW = 18, W_NPIX = W_IH * W_IW;
constexpr int W_RING = 8;

// host: U[n][c][comp] = (G g G^T)[xi][nu], comp = 4 xi + nu, in fp64; packed as
//   Up[chunk][comp][s][nb][lane][t] = +-U[nb*32 + (lane&31)][chunk*16 + s*8 + (lane>>5)*4 + t][comp]      (N, C multiples of 32 / 16;
//   the components of row xi = 3 are stored NEGATED: A^T[.][3] = (0, -1), so their products accumulate straight into Z[1][nu])
void pack_wino_weight(const float* W, int N, int C, int Np, int Cp, const float* out_scale, float* out) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nch = Cp / W_CK, NB = Np / 32;
    for (size_t i = 0; i < (size_t)Np * Cp * 16; ++i) out[i] = 0.f;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float* g = W + ((size_t)n * C + c) * 9;
            const double sc = out_scale ? (double)out_scale[n] : 1.0;
            double Gg[4][3], U[4][4];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) Gg[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) U[i][j] = (Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2]) * sc;
            const int chunk = c / W_CK, cc = c % W_CK, s = cc / 8, half = (cc % 8) / 4, t = cc % 4;
            const int nb = n / 32, lane = half * 32 + (n % 32);
            for (int comp = 0; comp < 16; ++comp)
                out[((((size_t)(chunk * 16 + comp) * 2 + s) * NB + nb) * 64 + lane) * 4 + t] = (float)(comp >= 12 ? -U[comp >> 2][comp & 3] : U[comp >> 2][comp & 3]);      // xi = 3 enters the output transform with -1 only: stored negated
        }
    (void)nch;
}

// NT = output channels / 32: 4 (128 channels: wave w owns n-tile w and all 16 components) or 2 (64 channels: wave (wc, wn) owns
// n-tile wn and the 8 components [8 wc, 8 wc + 8); the two partial sums are exchanged through LDS before the epilogue).
template <bool FUSE, int NT = 4, bool UP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wino3x3_kernel(const ConvArgs a) {
    // one LDS array: halo double buffer | V; the fused tail re-uses ALL of it for the complete conv2 tile (128 pixels x 128 channels)
    constexpr int HSZ = W_NPIX * W_PKH, VSZ = 16 * 32 * W_PKV;
    __shared__ __attribute__((aligned(16))) float S[2 * HSZ + VSZ];
    float (*Hin)[HSZ] = reinterpret_cast<float (*)[HSZ]>(&S[0]);
    float* V = &S[2 * HSZ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (a.OW + W_TW - 1) / W_TW, tiles_y = (a.OH + W_TH - 1) / W_TH;
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);          // XCD-aware tile order (csrc/conv.hip)
    const int l = bid / (tiles_x * tiles_y);
    bid -= l * tiles_x * tiles_y;
    const int ty0 = bid / tiles_x, tx0 = bid - ty0 * tiles_x;
    const int oy0 = ty0 * W_TH, ox0 = tx0 * W_TW;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int nch = a.C / W_CK, NB = a.N >> 5;
    const float* in_l = a.in + (size_t)l * a.H * a.W * a.C;
    const __amdgpu_buffer_rsrc_t in_srd = make_srd(in_l, (size_t)a.H * a.W * a.C * sizeof(float));
    const __amdgpu_buffer_rsrc_t w_srd = make_srd(a.Wp, (size_t)a.N * a.C * 16 * sizeof(float));
    const __amdgpu_buffer_rsrc_t out_srd = make_srd(a.out + (size_t)l * a.OH * a.OW * a.N, (size_t)a.OH * a.OW * a.N * sizeof(float));

    // ---- halo staging: 180 pixels x 4 float4 per chunk = 720 pieces over 256 threads ------------------------------------
    constexpr int NF4 = W_NPIX * 4, NLD = (NF4 + 255) / 256;
    int avoff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * 256;
        const int pix = idx >> 2, cc = idx & 3;
        const int py = pix / W_IW, px = pix - py * W_IW;
        const int iy = iy0 + py, ix = ix0 + px;
        const bool ok = idx < NF4 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        avoff[i] = ok ? ((iy * a.W + ix) * a.C + cc * 4) * 4 : BUF_OOB;
    }
    w_f32x4 areg[NLD];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) areg[i] = buf_load(in_srd, avoff[i], c * W_CK * 4);
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            if (idx < NF4) *(w_f32x4*)&Hin[buf][(idx >> 2) * W_PKH + (idx & 3) * 4] = areg[i];
        }
    };
    // ---- weights: Up[gg][nb][lane][4], gg = (chunk * 16 + comp) * 2 + s; wave w owns n-tile w ---------------------------
    static_assert(NT == 4 || (NT == 2 && !FUSE), "64-channel form: plain convolution only");
    const int wn = NT == 4 ? w : (w & 1), wc = NT == 4 ? 0 : (w >> 1);      // n-tile, component half
    constexpr int NPAIR = NT == 4 ? 8 : 4, KSEQ = NPAIR * 4;                 // component pairs / weight groups per chunk and wave
    const int wvoff = (wn * 64 + lane) * 16;
    const int gtot = nch * 32;
    w_f32x4 bring[W_RING];
    auto bload = [&](int gg, w_f32x4& b) {
        const int gc = gg < gtot ? gg : gtot - 1;
        b = buf_load(w_srd, wvoff, gc * NB * 1024);
    };
    // ---- transform: thread = (tile t, channel quad q, half h); h is wave-uniform (waves 0,1 / 2,3) -----------------------
    const int tt = tid & 31, tq = (tid >> 5) & 3, th = tid >> 7;
    const int t_ty = tt >> 3, t_tx = tt & 7;
    const int hbase = ((2 * t_ty + th) * W_IW + 2 * t_tx) * W_PKH + tq * 4;       // input rows th, th+1, th+2 of the 4x4 patch
    const int vbase = tt * W_PKV + tq * 4;
    auto transform = [&](int buf) {
        const float* hs = &Hin[buf][hbase];
        w_f32x4 L0[4], L1[4], L2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            L0[c] = *(const w_f32x4*)(hs + c * W_PKH);
            L1[c] = *(const w_f32x4*)(hs + (W_IW + c) * W_PKH);
            L2[c] = *(const w_f32x4*)(hs + (2 * W_IW + c) * W_PKH);
        }
        // rows of B^T d:  half 0 (input rows 0,1,2): xi0 = r0 - r2, xi1 = r1 + r2;  half 1 (rows 1,2,3): xi3 = r1 - r3, xi2 = r2 - r1
        w_f32x4 eA[4], eB[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            eA[c] = L0[c] - L2[c];
            eB[c] = th ? (L1[c] - L0[c]) : (L1[c] + L2[c]);
        }
        const int xiA = th ? 3 : 0, xiB = th ? 2 : 1;
        float* va = &V[(xiA * 4) * 32 * W_PKV + vbase];
        float* vb = &V[(xiB * 4) * 32 * W_PKV + vbase];
        // columns: nu0 = c0 - c2, nu1 = c1 + c2, nu2 = c2 - c1, nu3 = c1 - c3
        *(w_f32x4*)(va + 0 * 32 * W_PKV) = eA[0] - eA[2];
        *(w_f32x4*)(va + 1 * 32 * W_PKV) = eA[1] + eA[2];
        *(w_f32x4*)(va + 2 * 32 * W_PKV) = eA[2] - eA[1];
        *(w_f32x4*)(va + 3 * 32 * W_PKV) = eA[1] - eA[3];
        *(w_f32x4*)(vb + 0 * 32 * W_PKV) = eB[0] - eB[2];
        *(w_f32x4*)(vb + 1 * 32 * W_PKV) = eB[1] + eB[2];
        *(w_f32x4*)(vb + 2 * 32 * W_PKV) = eB[2] - eB[1];
        *(w_f32x4*)(vb + 3 * 32 * W_PKV) = eB[1] - eB[3];
    };

    w_f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    // Output transform Y = A^T M A in two steps.  128-channel form: the accumulators are Z[i][nu] = sum_xi A^T[i][xi] M(xi,nu), 8 of
    // them: the components of rows xi = 0 and xi = 3 touch ONE Z each (A^T columns (1,0) and (0,-1), the sign is in the packed
    // weights), so their MFMAs accumulate into it directly; only rows xi = 1, 2 go through a scratch accumulator and VALU
    // additions (Z[0] += M1 + M2, Z[1] += M1 - M2): 16 vector additions per chunk instead of 36.  On gfx950 VALU instructions do
    // NOT overlap with MFMAs of the same SIMD (tools/micro/mfma_valu.hip: every v_add_f32 between MFMAs adds its 2.6-4.7 cycles),
    // so the fold is paid in MFMA time.  Y = Z A once per tile after the channel loop.  64-channel form: a wave owns 8 components
    // (rows xi = 2 wc, 2 wc + 1) and one accumulator per component -- no additions in the channel loop at all; the rows meet in the
    // exchange after it.
    constexpr int NZ = 8;
    w_f32x16 out[4], Z[NZ];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[p][r] = 0.f;
#pragma unroll
    for (int p = 0; p < NZ; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) Z[p][r] = 0.f;
    // consumption order of the components: 128-channel form pairs 0-3 = (xi 0, xi 3) of nu = pair (direct), pairs 4-7 = (xi 1, xi 2)
    // of nu = pair - 4 (scratch); 64-channel form pair p = components 2 p, 2 p + 1 of the wave's half
    auto comp_of = [&](int pair, int which) -> int {
        if (NT == 4) return pair < 4 ? (which ? 12 + pair : pair) : (which ? 8 + (pair - 4) : 4 + (pair - 4));
        return wc * 8 + 2 * pair + which;
    };

    gload(0);
#pragma unroll
    for (int r = 0; r < W_RING - 2; ++r) bload(comp_of(r >> 2, r & 1) * 2 + ((r >> 1) & 1), bring[r]);      // consumption order, see below
    sstore(0);
    __syncthreads();

    const float* vs = &V[(lane & 31) * W_PKV + (lane >> 5) * 4];
#ifdef SUO_WINO_PROF
    long long pt[5] = {0, 0, 0, 0, 0}, p0 = clock64();
#define WPROF(i) do { const long long _t = clock64(); pt[i] += _t - p0; p0 = _t; } while (0)
#else
#define WPROF(i) do { } while (0)
#endif
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < nch;
        if (more) gload(c + 1);
        transform(buf);
        WPROF(0);
        __syncthreads();
        WPROF(1);
        if (more) sstore(buf ^ 1);
        WPROF(2);
        // Components in pairs: two independent scratch accumulators keep the MFMA pipe fed back to back (one chain alone
        // waits for its own result every time); the first MFMA of a component takes the persistent all-zero accumulator as C.
        // Weight groups are consumed in the order k = 4 pair + 2 s + which <-> (comp = 2 pair + which, s); the ring slot of a
        // group is its consumption index mod W_RING, its address comes from (comp, s).
        auto seq_gg = [&](int k) -> int {                         // consumption index (may run into the next chunk) -> packed group
            const int cc = c + k / KSEQ, kk = k % KSEQ;
            const int comp = comp_of(kk >> 2, kk & 1), sg = (kk >> 1) & 1;
            return cc * 32 + comp * 2 + sg;
        };
#pragma unroll
        for (int pair = 0; pair < NPAIR; ++pair) {
            w_f32x16 ta, tb;
            const bool direct = NT == 2 || pair < 4;              // accumulate into their own Z, no scratch, no additions
            const int za = NT == 4 ? pair : 2 * pair, zb = NT == 4 ? 4 + pair : 2 * pair + 1;       // (64-channel form: Z[local component])
            const int ca = comp_of(pair, 0), cb = comp_of(pair, 1);
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
                const int k = pair * 4 + sg * 2;
                bload(seq_gg(k + W_RING - 2), bring[(k + W_RING - 2) % W_RING]);      // the two slots the previous step consumed
                bload(seq_gg(k + W_RING - 1), bring[(k + W_RING - 1) % W_RING]);
                const w_f32x4 afa = *(const w_f32x4*)(vs + ca * 32 * W_PKV + sg * 8);
                const w_f32x4 afb = *(const w_f32x4*)(vs + cb * 32 * W_PKV + sg * 8);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (direct) {
                        Z[za] = w_mfma32(afa[t], bring[k % W_RING][t], Z[za]);
                        Z[zb] = w_mfma32(afb[t], bring[(k + 1) % W_RING][t], Z[zb]);
                    } else {
                        ta = w_mfma32(afa[t], bring[k % W_RING][t], (sg == 0 && t == 0) ? zero16 : ta);
                        tb = w_mfma32(afb[t], bring[(k + 1) % W_RING][t], (sg == 0 && t == 0) ? zero16 : tb);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!direct) { Z[pair - 4] += ta; Z[pair - 4] += tb; Z[pair] += ta; Z[pair] -= tb; }      // (pair - 4 = nu: Z[0][nu], Z[1][nu] = Z[4 + nu])
        }
        WPROF(3);
        __syncthreads();
        WPROF(4);
    }
    // second step of the output transform: R[h][j] = sum_nu A^T[j][nu] Z[4 h + nu].  128-channel form: h = i, R = Y.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        out[2 * h] = Z[4 * h] + Z[4 * h + 1] + Z[4 * h + 2];
        out[2 * h + 1] = Z[4 * h + 1] - Z[4 * h + 2] - Z[4 * h + 3];
    }
#ifdef SUO_WINO_PROF
    if (blockIdx.x == 1000 && (tid & 63) == 0) printf("wave %d cycles: transform %lld  barrier1 %lld  sstore %lld  mfma+fold %lld  barrier2 %lld\n", w, pt[0], pt[1], pt[2], pt[3], pt[4]);
#endif

    if constexpr (FUSE) {
        // ---- tail of the Residual block in the same launch (layers/Residual.py:27-35): conv3 (1x1, 128 -> 256) + bias + skip on
        // relu(conv2 + bias2), as in csrc/conv.hip (FUSE): the conv2 tile is complete in `out` (wave w: channels [32 w, 32 w + 32)
        // of the 128 pixels, as 4 output positions x 32 Winograd tiles); slice kc of 32 channels is staged by wave kc through LDS
        // (pixel-major, double-buffered) as the A operand, output channels in two passes of 128 with the waves as a 2 x 2 grid.
        // Single pass: every wave stages ITS 32 channels of relu(conv2 + bias2) for all 128 pixels at once (pitch 132: the A-operand
        // reads of 16 consecutive pixels hit 16 different bank quads), one barrier, then 16 k-groups of MFMAs into 2 x 4 accumulators
        // (wave grid 2 x 2: 64 pixels x 128 output channels each) with no barrier in between -- `out` is dead from the barrier on, which
        // is what makes room for the 128 accumulator registers (and for the up-sampled addend of the UP variant without spilling).
        constexpr int MP = 132;
        static_assert(128 * MP <= 2 * HSZ + VSZ, "the conv2 tile must fit the workgroup's LDS");
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
        auto b3load = [&](int q, w_f32x4(&b)[4]) {                // q = k-group of 8 mid channels; the wave's four 32-channel tiles
            const int kg = q < 16 ? q : 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = buf_load(w3_srd, w3voff, (kg * NB2 + wn * 4 + j) * 1024);
        };
        constexpr int R3 = 3;
        w_f32x4 b3[R3][4];
#pragma unroll
        for (int r = 0; r < R3 - 1; ++r) b3load(r, b3[r]);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = w_acc_row(r, lane);                               // Winograd tile -> pixel of the 8 x 16 tile
                const int pix = (2 * (t >> 3) + (p >> 1)) * W_TW + 2 * (t & 7) + (p & 1);
                M2[pix * MP + w * 32 + (lane & 31)] = fmaxf(out[p][r] + b2v, 0.f);
            }
        __syncthreads();
        w_f32x16 acc2[2][4];
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
            w_f32x4 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const w_f32x4*)(ms + i * 32 * MP + q * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc2[i][j] = w_mfma32(af[i][t], b3[q % R3][j][t], acc2[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                          // every wave has read the tile: LDS becomes the transposition patches
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* T = M2 + w * (32 * 36);
                const int col = (wn * 4 + j) * 32 + (lane & 7) * 4;
                const w_f32x4 bv = *(const w_f32x4*)(a.bias3 + col);
                int off[4];
                w_f32x4 rv[4], uv[UP ? 4 : 1];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pp = (wm * 2 + i) * 32 + (lane >> 3) + 8 * k;
                    const int oy = oy0 + pp / W_TW, ox = ox0 + pp % W_TW;
                    const bool in = oy < a.OH && ox < a.OW;
                    off[k] = in ? ((oy * a.OW + ox) * a.N2 + col) * 4 : BUF_OOB;
                    rv[k] = buf_load(r_srd, off[k], 0);
                    // the Hourglass's "up1 + up2(low3)" (hg.py:56-58) folded into this block's output: + low[oy / 2][ox / 2]
                    // (added LAST, as the separate up-sample kernel would: the fused block stays bit-identical to the two launches)
                    if constexpr (UP) uv[k] = buf_load(up_srd, in ? (((oy >> 1) * (a.OW >> 1) + (ox >> 1)) * a.N2 + col) * 4 : BUF_OOB, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) T[w_acc_row(r, lane) * 36 + (lane & 31)] = acc2[i][j][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    w_f32x4 o = (*(const w_f32x4*)&T[((lane >> 3) + 8 * k) * 36 + (lane & 7) * 4] + bv) + rv[k];
                    if constexpr (UP) o += uv[k];
                    buf_store(o, o2_srd, off[k]);
                }
                __builtin_amdgcn_wave_barrier();
            }
        return;
    }

    if constexpr (NT == 2) {
        // the two component halves of an n-tile meet.  Wave wc = 0 holds rows xi = 0, 1 (out[0..1] = R of row 0, out[2..3] = R of row 1),
        // wave wc = 1 rows xi = 2, 3 (row 3 negated in the weights).  Y[0][j] = R0 + R1 + R2, Y[1][j] = R1 - R2 + R3': wave wc keeps
        // output row i = wc and hands its MIDDLE row (1 or 2) to the partner through LDS ([position][register][lane]: conflict-free).
        float* X = &V[0];
        __syncthreads();                                          // V is free
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[((w * 2 + q) * 16 + r) * 64 + lane] = wc == 0 ? out[2 + q][r] : out[q][r];
        __syncthreads();
        const int pw = (1 - wc) * 2 + wn;
        w_f32x16 mine[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[q][r] = (wc == 0 ? out[q][r] + out[2 + q][r] : out[2 + q][r] - out[q][r]) + X[((pw * 2 + q) * 16 + r) * 64 + lane];
        __syncthreads();                                          // the exchange area becomes the transposition patches
        float* T = &V[0] + w * (32 * 36);
        const int col = wn * 32 + (lane & 7) * 4;
        const w_f32x4 bv = *(const w_f32x4*)(a.bias + col);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = 2 * wc + q, pi = p >> 1, pj = p & 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) T[w_acc_row(r, lane) * 36 + (lane & 31)] = mine[q][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = (lane >> 3) + 8 * k;
                const int oy = oy0 + 2 * (t >> 3) + pi, ox = ox0 + 2 * (t & 7) + pj;
                w_f32x4 o = *(const w_f32x4*)&T[t * 36 + (lane & 7) * 4] + bv;
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

    // ---- epilogue: per output position (i,j) transpose the 32 tiles x 32 channels through a wave-private patch -> 16-byte stores
    float* T = &V[0] + w * (32 * 36);
    const int col = w * 32 + (lane & 7) * 4;
    const w_f32x4 bv = *(const w_f32x4*)(a.bias + col);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int pi = p >> 1, pj = p & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) T[w_acc_row(r, lane) * 36 + (lane & 31)] = out[p][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = (lane >> 3) + 8 * k;
            const int oy = oy0 + 2 * (t >> 3) + pi, ox = ox0 + 2 * (t & 7) + pj;
            w_f32x4 o = *(const w_f32x4*)&T[t * 36 + (lane & 7) * 4] + bv;
            if (a.relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = fmaxf(o[q], 0.f);
            }
            buf_store(o, out_srd, (oy < a.OH && ox < a.OW) ? ((oy * a.OW + ox) * a.N + col) * 4 : BUF_OOB);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Measured at 128 -> 128 channels (tools/bench_wino.py): 1.46-1.75x the direct kernel from 256 tiles of 8 x 16 pixels up
// (64x64 maps of 8 crops, 16x16 maps of 128 crops), 0.5x at 64 tiles -- there the direct kernel's smaller tiles fill more CUs.
bool conv3x3_wino_pays(const ConvArgs& a, long min_tiles_arg) {
    static const int on = (int)SUO_TUNE("SUO_CONV_WINO", 1);                  // 0: A/B against the direct kernels
    static const long min_tiles_env = (long)SUO_TUNE("SUO_CONV_WINO_TILES", -1);
    const long min_tiles = min_tiles_env >= 0 ? min_tiles_env : (min_tiles_arg >= 0 ? min_tiles_arg : 256);
    const long tiles = (long)((a.OW + W_TW - 1) / W_TW) * ((a.OH + W_TH - 1) / W_TH) * a.L;
    return on && ((a.N == 128 && a.C == 128) || (a.N == 64 && a.C == 64)) && a.OH == a.H && a.OW == a.W && a.OH >= 8 && a.OW >= 16 && tiles >= min_tiles;
}

int launch_conv3x3_wino(const ConvArgs& a, hipStream_t s) {
    if (a.OH != a.H || a.OW != a.W || (a.N != 128 && a.N != 64) || (a.C % W_CK) || a.C <= 0) {
        suo_set_error("conv3x3_wino: unsupported shape H=%d W=%d C=%d N=%d", a.H, a.W, a.C, a.N);
        return SUO_ERR_ARG;
    }
    const int tiles = ((a.OW + W_TW - 1) / W_TW) * ((a.OH + W_TH - 1) / W_TH) * a.L;
    if (a.N == 64) hipLaunchKernelGGL((wino3x3_kernel<false, 2>), dim3(tiles), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wino3x3_kernel<false, 4>), dim3(tiles), dim3(256), 0, s, a);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo

namespace suo {
// conv2 (3x3, 128 -> 128, Winograd) + ReLU -> conv3 (1x1, 128 -> 256) + bias + skip in one launch
int launch_conv3x3_wino_fused(const ConvArgs& a, hipStream_t s) {
    if (a.OH != a.H || a.OW != a.W || a.N != 128 || a.C != 128 || a.N2 != 256 || !a.W3p || !a.bias3 || !a.R || !a.out2) {
        suo_set_error("conv3x3_wino_fused: unsupported shape C=%d N=%d N2=%d", a.C, a.N, a.N2);
        return SUO_ERR_ARG;
    }
    const int tiles = ((a.OW + W_TW - 1) / W_TW) * ((a.OH + W_TH - 1) / W_TH) * a.L;
    if (a.up) hipLaunchKernelGGL((wino3x3_kernel<true, 4, true>), dim3(tiles), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wino3x3_kernel<true, 4, false>), dim3(tiles), dim3(256), 0, s, a);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}
}  // namespace suo

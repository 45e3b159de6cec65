// EXPERIMENT: the bf16x3 Winograd convolution of csrc/conv_wino_x3.hip as ONE PERSISTENT workgroup per CU (one wave per SIMD, 512 registers).
//   * all 16 component accumulators stay in registers: no fold in the channel loop (-256 VALU per chunk and wave);
//   * V is double-buffered and the input transform of chunk q + 1 runs in the gaps of the MFMA stream of chunk q in the same wave (on gfx950 two
//     waves that share a SIMD do not hide each other's VALU work under MFMAs; one wave's own stream does: MI355X_MICROARCH.md);
//   * the chunk stream runs on across tile boundaries (tile j + 1's first halo is loaded and transformed during tile j's last chunks), so the only
//     exposed per-tile work is the output transform + stores -- the non-persistent form of this design lost 31 k of 75 k cycles per workgroup
//     to prologue / epilogue / launch gaps.
// Weights and packing as csrc/conv_wino_x3.hip (pack_wino_weight_bf16x3).
#include "buffer_ops.h"
#include "suo_internal.h"

namespace suo {

typedef float p_f32x4 __attribute__((ext_vector_type(4)));
typedef float p_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned p_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned p_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 p_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int p_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ unsigned p_pack_hi(float a, float b) { return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u); }
__device__ __forceinline__ float p_hi(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

#ifndef SUO_P_WR
#define SUO_P_WR 2
#endif
constexpr int P_WR = SUO_P_WR;
constexpr int P_CK = 16, P_PKH = 20, P_TH = 8, P_TW = 16, P_IH = 10, P_IW = 18, P_NPIX = P_IH * P_IW;

template <bool FUSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino3x3_x3p_kernel(const ConvArgs a, int ntiles) {
    constexpr int HSZ = P_NPIX * P_PKH;                       // floats per halo buffer
    constexpr int VROW = 16;                                  // bf16 per (tile, chunk) row
    constexpr int VPL = 16 * 32 * VROW;                       // bf16 per plane
    constexpr int VFLOATS = 3 * VPL / 2;
    __shared__ __attribute__((aligned(16))) float S[2 * HSZ + 2 * VFLOATS + 4 * 32 * 36];      // halo x 2 | V x 2 | epilogue patches
    float (*Hin)[HSZ] = reinterpret_cast<float (*)[HSZ]>(&S[0]);
    uint16_t* V = reinterpret_cast<uint16_t*>(&S[2 * HSZ]);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* T = &S[2 * HSZ + 2 * VFLOATS] + w * (32 * 36);
    const int tiles_x = (a.OW + P_TW - 1) / P_TW, tiles_y = (a.OH + P_TH - 1) / P_TH;
    const int per_crop = tiles_x * tiles_y;
    const int G = gridDim.x;
    const int nmine = (ntiles - (int)blockIdx.x + G - 1) / G; // tiles of this workgroup: blockIdx.x, blockIdx.x + G, ...
    if (nmine <= 0) return;
    // tile number -> (crop, tile row, tile column); XCD-aware: workgroup b lives on XCD b % 8 and walks a contiguous eighth of the tile list
    auto tile_coords = [&](int j, int& l, int& oy0, int& ox0) {
        int t = j * G + (int)blockIdx.x;
        if (t >= ntiles) t = ntiles - 1;                      // (prefetches past the end: harmless)
        if ((ntiles & 7) == 0 && (G & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
        l = t / per_crop;
        const int r = t - l * per_crop;
        const int ty0 = r / tiles_x, tx0 = r - ty0 * tiles_x;
        oy0 = ty0 * P_TH; ox0 = tx0 * P_TW;
    };
    const int nch = a.C / P_CK;                               // 8
    const size_t crop_in = (size_t)a.H * a.W * a.C;
    const __amdgpu_buffer_rsrc_t in_srd = make_srd(a.in, (size_t)a.L * crop_in * sizeof(float));          // (whole tensor: < 2 GB checked by the launcher)
    const __amdgpu_buffer_rsrc_t w_srd = make_srd(a.Wp, (size_t)a.N * a.C * 16 * 3 * sizeof(uint16_t));
    const __amdgpu_buffer_rsrc_t out_srd = make_srd(a.out, (size_t)a.L * a.OH * a.OW * a.N * sizeof(float));

    // ---- halo staging: 180 pixels x 4 float4 per chunk over 256 threads; the offsets follow the tile the stream is loading ------------------
    constexpr int NF4 = P_NPIX * 4, NLD = (NF4 + 255) / 256;
    int avoff[NLD];
    auto set_tile_offsets = [&](int j) {
        int l, oy0, ox0;
        tile_coords(j, l, oy0, ox0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            const int pix = idx >> 2, cc = idx & 3;
            const int py = pix / P_IW, px = pix - py * P_IW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool ok = idx < NF4 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            avoff[i] = ok ? (((l * a.H + iy) * a.W + ix) * a.C + cc * 4) * 4 : BUF_OOB;
        }
    };
    p_f32x4 areg[NLD];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) areg[i] = buf_load(in_srd, avoff[i], c * P_CK * 4);
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            if (idx < NF4) *(p_f32x4*)&Hin[buf][(idx >> 2) * P_PKH + (idx & 3) * 4] = areg[i];
        }
    };
    // every 128-byte line of a tile's input touched once: from then on its halo loads come from L2 / MALL like the weights (vmcnt retires in
    // order: a load that goes to HBM holds up every weight load issued after it)
    auto touch_tile = [&](int j) -> unsigned {
        int l, oy0, ox0;
        tile_coords(j, l, oy0, ox0);
        unsigned t = 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            const int pix = idx >> 2, ln = idx & 3;
            const int py = pix / P_IW, px = pix - py * P_IW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool ok = idx < NF4 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            t |= __builtin_amdgcn_raw_buffer_load_b32(in_srd, ok ? (((l * a.H + iy) * a.W + ix) * a.C + ln * 32) * 4 : BUF_OOB, 0, 0);
        }
        return t;
    };
    // ---- weights: Up3[(chunk * 16 + comp)][nb][plane][lane][8 bf16] (128 output channels: 4 n-tiles) ------------------------------------
    const int wvoff = lane * 16;
    const int wsbase = w * 3 * 1024;
    auto bload = [&](int gc, p_u32x4 (&b)[3]) {               // gc = chunk * 16 + comp
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(p_u32x4, buf_load(w_srd, wvoff + p * 1024, gc * (4 * 3 * 1024) + wsbase));
    };
    // ---- transform: thread = (tile tt, channel quad tq, half th) -----------------------------------------------------------------------------
    const int tt = tid & 31, tq = (tid >> 5) & 3, th = tid >> 7;
    const int t_ty = tt >> 3, t_tx = tt & 7;
    const int hbase = ((2 * t_ty + th) * P_IW + 2 * t_tx) * P_PKH + tq * 4;
    const int vbase = tt * VROW + ((((tq >> 1) ^ ((tt >> 3) & 1)) * 8) + (tq & 1) * 4);
    const int hother = th ? 0 : 2 * P_IW * P_PKH;
    const float hsign = th ? -1.f : 1.f;
    const int xiA = th ? 3 : 0, xiB = th ? 2 : 1;
    auto vstore = [&](int comp, p_f32x4 v, int vb) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            *(p_u32x2*)&V[vb * 3 * VPL + p * VPL + comp * 32 * VROW + vbase] = p_u32x2{p_pack_hi(v[0], v[1]), p_pack_hi(v[2], v[3])};
            if (p < 2) v = v - p_f32x4{p_hi(v[0]), p_hi(v[1]), p_hi(v[2]), p_hi(v[3])};      // exact residual
        }
    };
    p_f32x4 eA[4], eB[4], hl[16];
    auto xf_read = [&](int hb) {
        const float* hs = &Hin[hb][hbase];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            hl[q] = *(const p_f32x4*)(hs + q * P_PKH);
            hl[4 + q] = *(const p_f32x4*)(hs + (2 * P_IW + q) * P_PKH);
            hl[8 + q] = *(const p_f32x4*)(hs + (P_IW + q) * P_PKH);
            hl[12 + q] = *(const p_f32x4*)(hs + hother + q * P_PKH);
        }
    };
    auto xf_rows = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            eA[q] = hl[q] - hl[4 + q];
            eB[q] = p_f32x4{__builtin_fmaf(hl[12 + q][0], hsign, hl[8 + q][0]), __builtin_fmaf(hl[12 + q][1], hsign, hl[8 + q][1]),
                            __builtin_fmaf(hl[12 + q][2], hsign, hl[8 + q][2]), __builtin_fmaf(hl[12 + q][3], hsign, hl[8 + q][3])};
        }
    };
    auto xf_comp = [&](int j, int vb) {                        // j = 0..3: row xiA, nu = j; 4..7: row xiB, nu = j - 4
        const p_f32x4 (&e)[4] = j < 4 ? eA : eB;
        const int nu = j & 3, comp = (j < 4 ? xiA : xiB) * 4 + nu;
        vstore(comp, nu == 0 ? e[0] - e[2] : nu == 1 ? e[1] + e[2] : nu == 2 ? e[2] - e[1] : e[1] - e[3], vb);
    };
    const int afoff = (lane & 31) * VROW + (((lane >> 5) ^ ((lane >> 3) & 1)) * 8);
    p_bf16x8 af[2][2][3];
    auto aread = [&](int vb, int pair, p_bf16x8 (&f)[2][3]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) f[i][p] = *(const p_bf16x8*)&V[vb * 3 * VPL + p * VPL + (2 * pair + i) * 32 * VROW + afoff];
    };
    constexpr int WR = P_WR;                                  // weight ring, in pairs of components
    p_u32x4 bring[WR][2][3];
    p_f32x16 M[16];
    constexpr int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};

    // ---- prologue: tile 0's chunk 0 staged and transformed, chunk 1 staged, the first weights on their way ---------------------------------
    set_tile_offsets(0);
    unsigned touch = touch_tile(0) | touch_tile(1);
    gload(0);
#pragma unroll
    for (int r = 0; r < WR - 1; ++r) { bload(2 * r, bring[r][0]); bload(2 * r + 1, bring[r][1]); }
    sstore(0);
    asm volatile("" :: "v"(touch));
    gload(1);
    __syncthreads();
    xf_read(0);
    xf_rows();
#pragma unroll
    for (int j = 0; j < 8; ++j) xf_comp(j, 0);
    sstore(1);
    __syncthreads();

#ifdef SUO_WX3P_PROF
    long long pt[4] = {0, 0, 0, 0}, ps[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, p0 = clock64();
    const long long pstart = p0;
#define PPROF(i) do { const long long _t = clock64(); pt[i] += _t - p0; p0 = _t; } while (0)
#else
#define PPROF(i) do { } while (0)
#endif
    for (int j = 0; j < nmine; ++j) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) M[i][r] = 0.f;
      for (int c = 0; c < 8; ++c) {
        const int vb = c & 1, nb_ = vb ^ 1;                   // V buffer of this chunk / of the next one = halo buffer of the next one (8 chunks per tile: even)
        if (c == 6) set_tile_offsets(j + 1);                  // the stream's loads move on to the next tile (chunk c + 2 is its chunk 0)
        aread(vb, 0, af[0]);
        __builtin_amdgcn_sched_barrier(0);
#ifdef SUO_WX3P_PROF
        { const long long _t = clock64(); ps[8] += _t - p0; }
#endif
#pragma unroll
        for (int pair = 0; pair < 8; ++pair) {
            {   // weights WR - 1 pairs ahead (this chunk's, or the next one's: the stream wraps at the tile's last chunk)
                const int np = (pair + WR - 1) & 7, nc = (c + ((pair + WR - 1) >> 3)) & 7;
                bload(nc * 16 + 2 * np, bring[(pair + WR - 1) % WR][0]);
                bload(nc * 16 + 2 * np + 1, bring[(pair + WR - 1) % WR][1]);
            }
            if (pair < 7) aread(vb, pair + 1, af[(pair + 1) & 1]);
            if (pair == 0) {
                gload((c + 2) & 7);
                xf_read(nb_);
            }
            const p_bf16x8 (&fa)[3] = af[pair & 1][0];
            const p_bf16x8 (&fb)[3] = af[pair & 1][1];
            const p_u32x4 (&wa)[3] = bring[pair % WR][0];
            const p_u32x4 (&wb)[3] = bring[pair % WR][1];
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                M[2 * pair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TI[t]], __builtin_bit_cast(p_bf16x8, wa[TJ[t]]), M[2 * pair], 0, 0, 0);
                M[2 * pair + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[TI[t]], __builtin_bit_cast(p_bf16x8, wb[TJ[t]]), M[2 * pair + 1], 0, 0, 0);
            }
            // transform slices of the next chunk: step 1 the row combinations, steps 2-5 one component each, steps 6, 7 two
            if (pair == 1) xf_rows();
            else if (pair >= 2 && pair < 6) xf_comp(pair - 2, nb_);
            else if (pair == 6) { xf_comp(4, nb_); xf_comp(5, nb_); }
            else if (pair == 7) { xf_comp(6, nb_); xf_comp(7, nb_); sstore(vb); }      // + the halo of chunk q + 2 -> the buffer chunk q's transform read
            // issue order: an MFMA, then at most one or two memory instructions and a few VALU
            if (pair == 0) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 9) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (i < 11) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            } else if (pair == 1) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
            } else if (pair < 6) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    if (i == 8) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    else if (pair == 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    if (i == 4 || i == 7 || i == 10) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef SUO_WX3P_PROF
            { const long long _t = clock64(); ps[pair] += _t - p0; }
#endif
        }
        PPROF(c == 0 ? 0 : 1);
        __syncthreads();
        PPROF(2);
      }
        {
            // ---- the tile is complete: Y = A^T M A (row xi = 3 was packed negated), then per output position a transposition through the wave's
            // patch -> bias (+ ReLU) -> 16-byte stores.  The next tile's input is touched first: its latency passes under this epilogue.
            const unsigned tch = touch_tile(j + 2);
            int l, oy0, ox0;
            tile_coords(j, l, oy0, ox0);
            const int col = w * 32 + (lane & 7) * 4;
            const p_f32x4 bv = *(const p_f32x4*)(a.bias + col);
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                // out[2 i + j] = sum_nu A^T[j][nu] Z[i][nu],  Z[0][nu] = M[nu] + M[4 + nu] + M[8 + nu],  Z[1][nu] = M[4 + nu] - M[8 + nu] + M[12 + nu]
                p_f32x16 oo[2];
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) {              // one Z at a time (registers): out[.][0] = Z0 + Z1 + Z2, out[.][1] = Z1 - Z2 - Z3
                    const p_f32x16 z = pi == 0 ? M[nu] + M[4 + nu] + M[8 + nu] : M[4 + nu] - M[8 + nu] + M[12 + nu];
                    if (nu == 0) oo[0] = z;
                    else if (nu < 3) oo[0] += z;
                    if (nu == 1) oo[1] = z;
                    else if (nu > 1) oo[1] -= z;
                }
#pragma unroll
                for (int pj = 0; pj < 2; ++pj) {
                    const p_f32x16 o = oo[pj];
#pragma unroll
                    for (int r = 0; r < 16; ++r) T[p_acc_row(r, lane) * 36 + (lane & 31)] = o[r];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int t = (lane >> 3) + 8 * k;
                        const int oy = oy0 + 2 * (t >> 3) + pi, ox = ox0 + 2 * (t & 7) + pj;
                        p_f32x4 v = *(const p_f32x4*)&T[t * 36 + (lane & 7) * 4] + bv;
                        if (a.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        buf_store(v, out_srd, (oy < a.OH && ox < a.OW) ? (((l * a.OH + oy) * a.OW + ox) * a.N + col) * 4 : BUF_OOB);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            asm volatile("" :: "v"(tch));
            PPROF(3);
        }
    }
#ifdef SUO_WX3P_PROF
    if (blockIdx.x == 100 && (tid & 63) == 0)
        printf("wave %d: %d tiles, cycles: chunk 0 of a tile %lld  chunks 1-7 %lld  barriers %lld  epilogues %lld  total %lld | since chunk start, cumulative, at top %lld steps %lld %lld %lld %lld %lld %lld %lld %lld\n", w, nmine, pt[0], pt[1], pt[2], pt[3], p0 - pstart, ps[8], ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], ps[6], ps[7]);
#endif
}

int launch_conv3x3_wino_x3p(const ConvArgs& a, hipStream_t s) {
    if (a.OH != a.H || a.OW != a.W || a.N != 128 || a.C != 128 || (size_t)a.L * a.H * a.W * 128 * sizeof(float) >= ((size_t)1 << 31)) {
        suo_set_error("conv3x3_wino_x3p: unsupported shape L=%d H=%d W=%d C=%d N=%d", a.L, a.H, a.W, a.C, a.N);
        return SUO_ERR_ARG;
    }
    const int tiles = ((a.OW + P_TW - 1) / P_TW) * ((a.OH + P_TH - 1) / P_TH) * a.L;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    hipLaunchKernelGGL((wino3x3_x3p_kernel<false>), dim3(tiles < cus ? tiles : cus), dim3(256), 0, s, a, tiles);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo

// A CHAIN of 256 -> 256 Residual blocks on the small maps of a one-frame call (8x8 and 4x4 at 8 crops: the innermost two levels of an Hourglass,
// hg.py:37-58 -- twelve blocks per stack) in ONE cooperative launch.
//
// Why: at these sizes a layer is 0.1-0.6 MFLOP-microseconds of matrix-pipe work behind 0.13-0.59 MB of weights.  Per-layer launches cost their latency
// (4.6-6.3 us each, three per block + the pools: 72 launches and ~0.4 ms per frame, 21 % of the network call for 0.4 % of its FLOPs); one workgroup per
// tile doing a whole block (csrc/res_small_x3.hip) streams ALL of a block's weights through one CU (16.6 us at 8x8).  Here the workgroups of ONE XCD
// (<= 32, one per CU; they share an L2) split every layer by (32-pixel tile x 32..128-channel slice), so each CU streams 1/4 .. 1/32 of a layer's weights,
// and the layers are separated by the light grid barrier of csrc/lm_grid.hip (L2 atomics; no L2 write-back: one XCD, verified from XCC_ID at kernel start,
// general agent-scope fences otherwise).  Activations travel through L2 as fp32 ([M,128] scratch for the two inner tensors, the blocks' own [M,256] outputs).
//
// Arithmetic: the two-term fp16 form of csrc/f16x2.h (activations times 16, weight rows times 2^t_n, three v_mfma_f32_32x32x16_f16 per product block, fp32
// accumulate, per-channel rescale in the epilogue, range guard) on the SAME packed weights as the one-launch block kernel (ResBlockArgs; pack_gemm_weight_f16x2
// for the 1x1s, pack_res_conv3x3_f16x2: k-step = tap * 8 + channel / 16).  A task = 32 pixels x NTW n-tiles; its K range is split over the 8 waves of the
// workgroup (each wave a contiguous run of k-steps, A and B fragments straight from L2 into registers), partial accumulators meet in LDS and are summed in
// wave order (deterministic).  Layer l + 1 reads what layer l wrote only behind a grid barrier.
#include <stdlib.h>

#include <algorithm>

#include "buffer_ops.h"
#include "f16x2.h"
#include "suo_internal.h"

namespace suo {

constexpr int RC_THREADS = 512, RC_WAVES = RC_THREADS / 64;
constexpr int RC_PITCH = 36;                                  // floats per row of a wave's 32 x 32 partial in LDS

typedef float rc_f32x2 __attribute__((ext_vector_type(2)));
typedef float rc_f32x4 __attribute__((ext_vector_type(4)));
typedef float rc_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned rc_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 rc_f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int rc_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

struct RcGrid { unsigned* bar; int G; bool same_xcd; int mode; };

// csrc/lm_grid.hip: grid_sync -- bar[0] arrivals, bar[1] generation
__device__ __forceinline__ void rc_sync(const RcGrid& g) {
    if (g.same_xcd) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned gen = __hip_atomic_fetch_add(&g.bar[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(&g.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)g.G - 1) {
                __hip_atomic_exchange(&g.bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&g.bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_fetch_add(&g.bar[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (g.mode & 2) asm volatile("buffer_inv sc1" ::: "memory");      // (experiment; ~15 us on this multi-XCD part: readers go to L2 by themselves instead)
        if (g.mode & 4) {                                                  // (experiment: ONE wave of the workgroup invalidates the CU's L1 for all)
            if (threadIdx.x < 64) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        return;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned gen = __hip_atomic_load(&g.bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&g.bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)g.G - 1) {
            __hip_atomic_store(&g.bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&g.bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&g.bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    __threadfence();
}

__device__ __forceinline__ rc_f32x2 buf_load2_l2(__amdgpu_buffer_rsrc_t r, int voff) {      // 8 bytes at device scope (see buf_load_l2)
    return __builtin_bit_cast(rc_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 16));
}
__device__ __forceinline__ rc_f32x4 rc_max4(rc_f32x4 a, rc_f32x4 b) {
    return rc_f32x4{fmaxf(a[0], b[0]), fmaxf(a[1], b[1]), fmaxf(a[2], b[2]), fmaxf(a[3], b[3])};
}

// One layer of one block.  PH 0: conv1 (BN + ReLU prologue on x [, x = 2x2 max-pool of the source], 256 -> 128, ReLU) -> mid_out;  1: conv 3x3 on mid_in
// (128 -> 128, zero padding, ReLU) -> mid_out;  2: conv3 (128 -> 256) + skip [+ up-sampled addend] -> a.out.
// Everything another workgroup wrote earlier in this launch (x of a later block, mid_in, up) is read at device scope (buf_load_l2: from L2, not from this CU's L1 --
// the grid barrier does not invalidate L1, see rc_sync); weights, biases and BatchNorm terms are read-only for the launch and take the ordinary path.
template <int PH, int NTW>
__device__ __forceinline__ void rc_phase(const ResBlockArgs& a, const float* __restrict__ mid_in, float* __restrict__ mid_out, int wg, int G, float* red, float& gmax) {
    constexpr int KS = PH == 0 ? 16 : (PH == 1 ? 72 : 8), KSW = KS / RC_WAVES, NB = PH == 2 ? 8 : 4, NG = NB / NTW;
    constexpr int DMAX = NTW == 4 ? 1 : (NTW == 2 ? 2 : 3), D = KSW < DMAX ? KSW : DMAX;      // k-steps requested ahead (128 registers per lane: two workgroups per CU)
    const int tid = threadIdx.x;
    int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);      // the wave's index is uniform: k-step arithmetic on the scalar unit
    // (opaque to the optimiser: everything below derives from these two, and hoisted out of the block loop as loop invariants of all nine phase bodies at once it spilt 400 registers)
    asm volatile("" : "+v"(lane), "+s"(w));
    w = __builtin_amdgcn_readfirstlane(w);
    const int H = a.H, W = a.W, HW = H * W, M = a.L * HW, MT = (M + 31) >> 5;
    const uint16_t* __restrict__ Wp = reinterpret_cast<const uint16_t*>(PH == 0 ? a.W1 : (PH == 1 ? a.W2 : a.W3));
    const float* __restrict__ osc = PH == 0 ? a.osc1 : (PH == 1 ? a.osc2 : a.osc3);
    const float* __restrict__ bias = PH == 0 ? a.b1 : (PH == 1 ? a.b2 : a.b3);
    const int row = lane & 31, kh = lane >> 5;
    const size_t x_bytes = (size_t)M * (a.pool_in ? 4 : 1) * 256 * sizeof(float);
    const __amdgpu_buffer_rsrc_t in_srd = PH == 0 ? make_srd(a.x, x_bytes) : make_srd(mid_in, (size_t)M * 128 * sizeof(float));
    const __amdgpu_buffer_rsrc_t x_srd = make_srd(a.x, x_bytes);           // (the skip path of PH 2)
    const __amdgpu_buffer_rsrc_t up_srd = make_srd(a.up ? a.up : a.x, a.up ? (size_t)(M / 4) * 256 * sizeof(float) : 0);
    for (int task = wg; task < MT * NG; task += G) {
        const int mt = __builtin_amdgcn_readfirstlane(task / NG), ng = __builtin_amdgcn_readfirstlane(task - mt * NG);
        const int p = mt * 32 + row;
        const bool valid = p < M;
        const int crop = p / HW, rem = p - crop * HW, y = rem / W, x = rem - y * W;
        // this lane's byte offset into the layer's input for k-step 0 (its pixel, its half of the 16 channels); BUF_OOB reads zeros
        int voff, voff_row1 = 0;
        if constexpr (PH == 0) {
            const int pix = a.pool_in ? (crop * 2 * H + 2 * y) * (2 * W) + 2 * x : p;
            voff = valid ? (pix * 256 + 8 * kh) * 4 : BUF_OOB;
            voff_row1 = valid ? voff + 2 * W * 256 * 4 : BUF_OOB;            // (pool_in: the source row below)
        } else {
            voff = (p * 128 + 8 * kh) * 4;                                  // (validity per tap / per row below)
        }
        rc_f32x16 acc[NTW];
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        rc_f32x4 av[D][2];
        rc_u32x4 bw[D][NTW][2];
        auto request = [&](int i, int slot) {                 // k-step i of this wave's run
            const int ks = __builtin_amdgcn_readfirstlane(w * KSW + i);
            const uint16_t* wb = Wp + (size_t)(ks * NB + ng * NTW) * 1024;      // (uniform) 2 planes x 64 lanes x 8 per n-tile
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    bw[slot][n][pl] = *reinterpret_cast<const rc_u32x4*>(wb + (n * 2 + pl) * 512 + lane * 8);
            if constexpr (PH == 0) {
                const int so = ks * 64;
                if (a.pool_in) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const rc_f32x4 t0 = buf_load_l2(in_srd, voff + 16 * q, so), t1 = buf_load_l2(in_srd, voff + 1024 + 16 * q, so);
                        const rc_f32x4 t2 = buf_load_l2(in_srd, voff_row1 + 16 * q, so), t3 = buf_load_l2(in_srd, voff_row1 + 1024 + 16 * q, so);
                        av[slot][q] = rc_max4(rc_max4(t0, t1), rc_max4(t2, t3));
                    }
                } else {
                    av[slot][0] = buf_load_l2(in_srd, voff, so);
                    av[slot][1] = buf_load_l2(in_srd, voff + 16, so);
                }
            } else if constexpr (PH == 1) {
                const int tap = ks >> 3, cc = ks & 7, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const bool ok = valid && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
                const int vo = ok ? voff + ((dy * W + dx) * 128 + cc * 16) * 4 : BUF_OOB;
                av[slot][0] = buf_load_l2(in_srd, vo, 0);
                av[slot][1] = buf_load_l2(in_srd, vo + 16, 0);
            } else {
                const int vo = valid ? voff : BUF_OOB;
                av[slot][0] = buf_load_l2(in_srd, vo, ks * 64);
                av[slot][1] = buf_load_l2(in_srd, vo + 16, ks * 64);
            }
        };
#pragma unroll
        for (int i = 0; i < D; ++i) request(i, i);
#pragma unroll
        for (int i = 0; i < KSW; ++i) {
            const int slot = i % D;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = av[slot][j >> 2][j & 3];
            if constexpr (PH == 0) {
                const int c0 = __builtin_amdgcn_readfirstlane(w * KSW + i) * 16 + 8 * kh;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const rc_f32x4 sc = *reinterpret_cast<const rc_f32x4*>(a.pro_scale + c0 + 4 * q), sh = *reinterpret_cast<const rc_f32x4*>(a.pro_shift + c0 + 4 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[4 * q + j] = valid ? fmaxf(fmaf(v[4 * q + j], S2_XSCALE * sc[j], S2_XSCALE * sh[j]), 0.f) : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= S2_XSCALE;
            }
            rc_u32x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gmax = s2_track(gmax, v[2 * j], v[2 * j + 1]);
                hi[j] = s2_pack_rn(v[2 * j], v[2 * j + 1]);
                lo[j] = s2_lo_pack(v[2 * j], v[2 * j + 1], hi[j]);
            }
            const rc_f16x8 ah = __builtin_bit_cast(rc_f16x8, hi), al = __builtin_bit_cast(rc_f16x8, lo);
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const rc_f16x8 bh = __builtin_bit_cast(rc_f16x8, bw[slot][n][0]), bl = __builtin_bit_cast(rc_f16x8, bw[slot][n][1]);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[n], 0, 0, 0);
            }
            if (i + D < KSW) request(i + D, slot);
            __builtin_amdgcn_sched_barrier(0);                // (keeps the compiler from hoisting every k-step's loads to the top of the unrolled loop -- and spilling)
        }
        // the eight partial tiles meet in LDS, red[n][wave & 3][32][RC_PITCH]: waves 4-7 hand theirs to waves 0-3 (same lane, same element), whose sums the epilogue adds
        __syncthreads();                                      // (the previous task's epilogue has read red)
        if (w >= RC_WAVES / 2) {
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((n * (RC_WAVES / 2) + (w - RC_WAVES / 2)) * 32 + rc_acc_row(r, lane)) * RC_PITCH + (lane & 31)] = acc[n][r];
        }
        __syncthreads();
        if (w < RC_WAVES / 2) {
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* q = red + ((n * (RC_WAVES / 2) + w) * 32 + rc_acc_row(r, lane)) * RC_PITCH + (lane & 31);
                    *q = acc[n][r] + *q;
                }
        }
        __syncthreads();
        // epilogue: thread -> (pixel er, channel pair ec) of every n-tile; partials summed in wave order
        const int er = tid >> 4, ec = (tid & 15) * 2, ep = mt * 32 + er;
        if (ep < M) {
            const int ecrop = ep / HW, erem = ep - ecrop * HW, ey = erem / W, ex = erem - ey * W;
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int q = 0; q < RC_WAVES / 2; ++q) {
                    const rc_f32x2 t = *reinterpret_cast<const rc_f32x2*>(red + ((n * (RC_WAVES / 2) + q) * 32 + er) * RC_PITCH + ec);
                    s0 += t[0]; s1 += t[1];
                }
                const int ch = (ng * NTW + n) * 32 + ec;
                const rc_f32x2 oc = *reinterpret_cast<const rc_f32x2*>(osc + ch), bb = *reinterpret_cast<const rc_f32x2*>(bias + ch);
                float o0 = fmaf(s0, oc[0], bb[0]), o1 = fmaf(s1, oc[1], bb[1]);
                if constexpr (PH < 2) {
                    *reinterpret_cast<rc_f32x2*>(mid_out + (size_t)ep * 128 + ch) = rc_f32x2{fmaxf(o0, 0.f), fmaxf(o1, 0.f)};
                } else {
                    rc_f32x2 sk;
                    if (a.pool_in) {                          // the skip path takes the pooled input as well
                        const int o00 = (((ecrop * 2 * H + 2 * ey) * (2 * W) + 2 * ex) * 256 + ch) * 4, o10 = o00 + 2 * W * 256 * 4;
                        const rc_f32x2 t0 = buf_load2_l2(x_srd, o00), t1 = buf_load2_l2(x_srd, o00 + 1024), t2 = buf_load2_l2(x_srd, o10), t3 = buf_load2_l2(x_srd, o10 + 1024);
                        sk = rc_f32x2{fmaxf(fmaxf(t0[0], t1[0]), fmaxf(t2[0], t3[0])), fmaxf(fmaxf(t0[1], t1[1]), fmaxf(t2[1], t3[1]))};
                    } else {
                        sk = buf_load2_l2(x_srd, (ep * 256 + ch) * 4);
                    }
                    o0 += sk[0]; o1 += sk[1];
                    if (a.up) {                               // + nearest-neighbour 2x up-sampling of the low branch (hg.py:56-58)
                        const rc_f32x2 u = buf_load2_l2(up_srd, (((ecrop * (H / 2) + ey / 2) * (W / 2) + ex / 2) * 256 + ch) * 4);
                        o0 += u[0]; o1 += u[1];
                    }
                    *reinterpret_cast<rc_f32x2*>(a.out + (size_t)ep * 256 + ch) = rc_f32x2{o0, o1};
                }
            }
        }
    }
}

template <int PH>
__device__ __forceinline__ void rc_layer(const ResBlockArgs& a, const float* mid_in, float* mid_out, int wg, int G, float* red, float& gmax) {
    // n-tiles per task: as few as keep the task count at or below the workgroup count (each CU then streams NTW / NB of the layer's weights once)
    constexpr int NB = PH == 2 ? 8 : 4;
    const int MT = (a.L * a.H * a.W + 31) >> 5;
    if (MT * NB <= G) rc_phase<PH, 1>(a, mid_in, mid_out, wg, G, red, gmax);
    else if (MT * NB / 2 <= G || PH < 2) rc_phase<PH, 2>(a, mid_in, mid_out, wg, G, red, gmax);      // (four n-tiles per task only for conv3: one k-step per wave, 64 accumulator registers)
    else if constexpr (PH == 2) rc_phase<PH, 4>(a, mid_in, mid_out, wg, G, red, gmax);
}

// Two of these workgroups fit a CU (<= 128 registers, 72 KB LDS): two chains can be resident on one XCD at the same time whatever else runs there -- see
// launch_res_chain for why that matters.
__global__ __launch_bounds__(RC_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void res_chain_kernel(const ResChainArgs c) {
    if ((int)(blockIdx.x & 7) != c.xcd) return;               // workgroup b runs on XCD b % 8 (observed; checked below): keep one XCD
    __shared__ __attribute__((aligned(16))) float red[4 * (RC_WAVES / 2) * 32 * RC_PITCH];
    __shared__ unsigned sh_mask;
    const int wg = blockIdx.x >> 3, G = c.G;
    RcGrid g;
    g.bar = c.bar; g.G = G; g.same_xcd = true; g.mode = c.bar_mode;
    // Which XCD is every cooperating workgroup on (HW_REG_XCC_ID = 20, bits [3:0])?  Each ORs its bit into bar[2] (agent-scope atomic), one barrier on atomics only
    // (no data behind it), then all read the same mask: one bit -> one L2 -> the light barrier from here on.
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;
        __hip_atomic_fetch_or(&c.bar[2], 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    rc_sync(g);
    if (threadIdx.x == 0) sh_mask = __hip_atomic_fetch_add(&c.bar[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    g.same_xcd = __builtin_popcount(sh_mask) == 1;
    if (!g.same_xcd && threadIdx.x == 0 && wg == 0) __hip_atomic_fetch_add(&c.bar[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (diagnostic: launches that fell back)
    rc_sync(g);                                               // everyone has read the mask ...
    if (threadIdx.x == 0 && wg == 0) __hip_atomic_exchange(&c.bar[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                     // ... clear it for the next launch
    float gmax = 0.f;
    for (int b = 0; b < c.n; ++b) {
        const ResBlockArgs& a = c.b[b];
        rc_layer<0>(a, nullptr, c.mid1, wg, G, red, gmax);
        rc_sync(g);
        rc_layer<1>(a, c.mid1, c.mid2, wg, G, red, gmax);
        rc_sync(g);
        rc_layer<2>(a, c.mid2, nullptr, wg, G, red, gmax);
        if (b + 1 < c.n) rc_sync(g);
    }
    for (int i = 0; i < c.extra_barriers; ++i) rc_sync(g);    // (tools/bench_res_chain.py: the barrier's own cost)
    s2_raise(c.range_flag, gmax);
}

size_t res_chain_scratch_floats(int max_pixels) { return (size_t)2 * max_pixels * 128 + 16; }

bool res_chain_takes(const ResBlockArgs& a) {
    return a.x && a.out && a.W1 && a.W2 && a.W3 && a.b1 && a.b2 && a.b3 && a.osc1 && a.osc2 && a.osc3 && a.pro_scale && a.pro_shift && a.L > 0 && a.H > 0 && a.W > 0 &&
           (!a.up || (a.H % 2 == 0 && a.W % 2 == 0));
}

// scratch: res_chain_scratch_floats(max over the blocks of L * H * W) floats of device memory whose LAST 16 floats (the barrier words) were zero before the first launch
// that used it (they return to zero / stay consistent from launch to launch); launches sharing a scratch must be stream-ordered.
// xcd (0..7): the XCD whose CUs run the chain.  CO-RESIDENCY: a grid barrier needs all G workgroups on the chip at once, and nothing in a plain launch promises that --
// other kernels drain by themselves, but two CHAINS each holding part of an XCD's slots while waiting for the rest would wait for ever.  Two workgroups of this kernel
// fit a CU, so an XCD (32 CUs) holds two whole chains: as long as at most two chains target the same XCD at a time both become fully resident.  The caller spreads its
// concurrent users over `xcd` accordingly (csrc/net.hip: network k uses XCD k % 8 and only the first 16 networks of a process take this path).
int launch_res_chain(const ResBlockArgs* blocks, int n, float* scratch, size_t scratch_floats, unsigned* range_flag, int xcd, int extra_barriers, int min_pixels, hipStream_t s) {
    if (n < 0 || n > RC_MAX_BLOCKS || !scratch || !range_flag || xcd < 0 || xcd > 7) { suo_set_error("res_chain: bad arguments (n = %d)", n); return SUO_ERR_ARG; }
    ResChainArgs c = {};
    int maxpix = std::max(1, min_pixels);                     // (min_pixels: the probe of tools/bench_res_chain.py sizes an empty chain's grid with it)
    for (int i = 0; i < n; ++i) {
        if (!res_chain_takes(blocks[i])) { suo_set_error("res_chain: block %d: unsupported arguments", i); return SUO_ERR_ARG; }
        c.b[i] = blocks[i];
        maxpix = std::max(maxpix, blocks[i].L * blocks[i].H * blocks[i].W);
    }
    if (res_chain_scratch_floats(maxpix) > scratch_floats) { suo_set_error("res_chain: scratch of %zu floats, %zu needed", scratch_floats, res_chain_scratch_floats(maxpix)); return SUO_ERR_ARG; }
    c.n = n;
    c.mid1 = scratch; c.mid2 = scratch + (size_t)maxpix * 128;
    c.bar = reinterpret_cast<unsigned*>(scratch + scratch_floats - 16);
    c.range_flag = range_flag;
    c.extra_barriers = extra_barriers;
    c.xcd = xcd;
    static const int bar_mode = getenv("SUO_RC_BAR_MODE") ? atoi(getenv("SUO_RC_BAR_MODE")) : 0;
    c.bar_mode = bar_mode;
    // workgroups: enough for the widest layer (conv3: 8 n-tiles per 32 pixels), at most one XCD's CUs
    const int MT = (maxpix + 31) / 32;
    c.G = std::max(1, std::min(RC_MAX_WGS, MT * 8));
    hipLaunchKernelGGL(res_chain_kernel, dim3(8 * c.G), dim3(RC_THREADS), 0, s, c);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo

// Do the VALU instructions of OTHER waves overlap with the MFMAs of a wave on the same SIMD?  (tools/micro/mfma_valu.hip asks it of a single
// wave: no.)  Workgroup = 4 (1 + NVW) waves: waves 0-3 (one per SIMD) run a chain of v_mfma_f32_32x32x16_bf16 (two accumulators), the other
// NVW waves per SIMD a stream of independent v_fma_f32.  MODE 1: MFMA waves only, 2: VALU waves only, 3: both.  Overlap <=> the waves' own
// shader-clock counts together ~ those alone; serialised <=> the sum.
// Measured (MI355X, 2.0-2.3 GHz, per-wave s_memtime of workgroup 0): the MFMA wave takes 1.025 M clocks with the VALU waves beside it and 1.04 M
// alone -- its pipe stays full whatever the VALU load (1 or 3 waves, 16-64 v_fma_f32 per 16 MFMAs).  The heaviest VALU stream (3 waves x 64)
// takes 0.94 M clocks alone and 1.08 M beside the MFMA wave (+15 %, not +110 %): across waves the two pipes overlap.  Lighter VALU streams end
// WITH the MFMA wave (1.025 M) although alone they need 0.29-0.5 M: beside a wave that always has an MFMA ready they are issued at a
// throttled rate -- VALU work that is on the critical path (a barrier behind it) is slowed by a co-resident MFMA burst even though the
// pipes overlap.  Within ONE wave they do not overlap at all (mfma_valu.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_xwave.hip -o /tmp/xwave && /tmp/xwave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, int nv, float* clk, int prio) {
    const int w = threadIdx.x >> 6;
    float s = 0;
    const unsigned long long c0 = __builtin_readcyclecounter();               // s_memtime: shader clocks
    const unsigned long long t0 = wall_clock64();                             // s_memrealtime: 100 MHz
    if (w < 4) {
        if (MODE & 1) {
            f32x16 acc[2];
            for (int i = 0; i < 2; ++i)
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(blockIdx.x * 1e-3f + i); }
            if (prio) __builtin_amdgcn_s_setprio(3);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 1], 0, 0, 0);
            }
            for (int i = 0; i < 2; ++i)
                for (int r = 0; r < 16; ++r) s += acc[i][r];
        }
    } else if (MODE & 2) {
        float v[16];
        const float inc = blockIdx.x * 1e-3f + 1.f, m = 1.0001f;
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
        for (int it = 0; it < iters; ++it)
            for (int q = 0; q < nv; ++q) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(inc));
            }
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (w == 0 || w == 4)) {
        const unsigned long long c1 = __builtin_readcyclecounter();
        const unsigned long long t1 = wall_clock64();
        clk[w ? 2 : 0] = (float)(c1 - c0);
        clk[w ? 3 : 1] = (float)(t1 - t0);
    }
}

template <int MODE>
float run(float* out, int iters, int nv, int nvw, float* clk, int prio = 0) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * (1 + nvw)), 0, 0, out, iters, nv, clk, prio);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * (1 + nvw)), 0, 0, out, iters, nv, clk, prio);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 2000;
    float* clk;
    hipHostMalloc((void**)&clk, 16, 0);
    // per iteration: 16 MFMAs x 32 cycles = 512 cycles of matrix pipe; nv x 16 v_fma_f32 x 4 cycles = 64 nv cycles of VALU
    // Per-wave shader clocks (s_memtime) of workgroup 0: the grid's wall clock also depends on how the dispatcher packs the 256 workgroups onto CUs
    for (int nvw : {1, 3})
        for (int nv : {1, 2, 4}) {
            run<1>(out, iters, nv, nvw, clk);
            hipDeviceSynchronize();
            const float ca = clk[0], ta = clk[1];
            run<2>(out, iters, nv, nvw, clk);
            hipDeviceSynchronize();
            const float cb = clk[2], tb = clk[3];
            run<3>(out, iters, nv, nvw, clk);
            hipDeviceSynchronize();
            printf("%d VALU wave(s) per SIMD, %3d v_fma_f32 each per 16 MFMAs (512 matrix-pipe cycles).  Shader clocks of a wave: MFMA wave alone %8.0f (%.2f GHz), VALU wave alone %8.0f (%.2f GHz); "
                   "together: MFMA wave %8.0f, VALU wave %8.0f (%.2f GHz)\n", nvw, 16 * nv, ca, ca / ta * 0.1, cb, cb / tb * 0.1, clk[0], clk[2], clk[0] / clk[1] * 0.1);
        }
    return 0;
}

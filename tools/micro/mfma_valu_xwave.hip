// Do the VALU instructions of OTHER waves overlap with the MFMAs of a wave on the same SIMD?  (tools/micro/mfma_valu.hip asks it of a single
// wave: no.)  Workgroup = 4 (1 + NVW) waves: waves 0-3 (one per SIMD) run a chain of v_mfma_f32_32x32x16_bf16 (two accumulators), the other
// NVW waves per SIMD a stream of independent v_fma_f32 (one wave alone issues a VALU instruction every ~9 cycles; three saturate the SIMD's
// 4-cycle issue).  MODE 1: MFMA waves only, 2: VALU waves only, 3: both.  Overlap <=> t(3) ~ max(t(1), t(2)); serialised <=> t(1) + t(2).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_xwave.hip -o /tmp/xwave && /tmp/xwave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, int nv) {
    const int w = threadIdx.x >> 6;
    float s = 0;
    if (w < 4) {
        if (MODE & 1) {
            f32x16 acc[2];
            for (int i = 0; i < 2; ++i)
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(blockIdx.x * 1e-3f + i); }
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
}

template <int MODE>
float run(float* out, int iters, int nv, int nvw) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * (1 + nvw)), 0, 0, out, iters, nv);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * (1 + nvw)), 0, 0, out, iters, nv);
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
    // per iteration: 16 MFMAs x 32 cycles = 512 cycles of matrix pipe; nv x 16 v_fma_f32 x 4 cycles = 64 nv cycles of VALU
    for (int nvw : {1, 3})
        for (int nv : {1, 2, 4}) {
            const float a = run<1>(out, iters, nv, nvw), b = run<2>(out, iters, nv, nvw), c = run<3>(out, iters, nv, nvw);
            printf("%d VALU wave(s) per SIMD, %3d v_fma_f32 each per 16 MFMAs (512 matrix-pipe cycles): MFMA waves alone %7.1f us, VALU waves alone %7.1f us, both %7.1f us  (sum %7.1f, max %7.1f)\n",
                   nvw, 16 * nv, a, b, c, a + b, a > b ? a : b);
        }
    return 0;
}

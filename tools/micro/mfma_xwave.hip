// Does VALU / LDS work of ONE wave overlap with the MFMAs of ANOTHER wave on the same SIMD?  (tools/micro/mfma_valu.hip answers it for one wave: no.)
// A workgroup of eight waves, one workgroup per CU: waves 0-3 (one per SIMD) issue v_mfma_f32_32x32x16_f16 on four independent accumulators, waves 4-7 (the second
// wave of each SIMD) issue independent v_fma_f32 chains, or ds_write_b128 + v_fma (a transform-and-stage stand-in).  Times of: MFMA waves alone, the other waves alone, both.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_xwave.hip -o tools/micro/mfma_xwave.bin && tools/micro/mfma_xwave.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float vf4 __attribute__((ext_vector_type(4)));

// mode bit 0: MFMA waves work, bit 1: the other waves work; KIND 0: VALU only, 1: VALU + LDS writes
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    __shared__ float4 lds[6144];                       // 96 KB: one workgroup per CU
    const int w = threadIdx.x >> 6;
    float s = 0;
    if (w < 4) {
        if (!(mode & 1)) return;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 1e-3f + e); b[e] = (_Float16)(blockIdx.x * 1e-3f + e); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 3], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        if (!(mode & 2)) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
        const float m = 1.0001f, c = blockIdx.x * 1e-6f;
        vf4 st = {1.f, 2.f, 3.f, 4.f};
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)&lds[0] + (threadIdx.x - 256) * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("ds_write_b128 %1, %0" : : "v"(st), "v"(base + (u & 7) * 4096) : "memory");
            }
            if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        for (int i = 0; i < 8; ++i) s += v[i];
        if (KIND == 1) s += lds[threadIdx.x & 255].x;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND>
static float run(float* out, int iters, int mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 4000;
    for (int kind = 0; kind < 2; ++kind) {
        float t[4];
        for (int mode = 1; mode <= 3; ++mode) t[mode] = kind ? run<1>(out, iters, mode) : run<0>(out, iters, mode);
        printf("%s: MFMA waves alone %.1f us | the other wave of each SIMD alone %.1f us | both %.1f us  (sum %.1f, max %.1f)\n",
               kind ? "128 v_fma_f32 + 16 ds_write_b128 per 16 MFMAs" : "128 v_fma_f32 per 16 MFMAs", t[1], t[2], t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
    }
    return 0;
}

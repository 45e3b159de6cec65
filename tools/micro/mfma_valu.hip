// Do VALU instructions overlap with MFMAs on a SIMD?  Each iteration: 16 x v_mfma_f32_32x32x2_f32 (two independent accumulator
// chains) interleaved with NV independent VALU operations per MFMA (v_add_f32 / v_pk_add_f32 on registers the MFMAs do not touch).
// If the pipes overlap, time stays at 64 cycles per MFMA until NV * 4 cycles exceed it; if they serialise, it grows from NV = 1.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, bool PK>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    float v[16];
    f32x2 p[8];
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    for (int i = 0; i < 8; ++i) p[i] = f32x2{a + i, b + i};
    const float inc = b + 1.f;
    const f32x2 pinc = {inc, inc};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 1], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(u * NV + q) & 7]) : "v"(pinc));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(u * NV + q) & 15]) : "v"(inc));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool PK>
void run(int wgs_per_cu, int iters) {
    float* out;
    const int grid = 256 * wgs_per_cu;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, PK><<<grid, 256>>>(out, 10);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<NV, PK><<<grid, 256>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double mfmas = (double)iters * 16 * wgs_per_cu;                 // per SIMD
    printf("%s NV=%2d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  -> %.1f ns per MFMA per SIMD (64 cycles at 2.4 GHz = 26.7 ns)\n", PK ? "v_pk_add_f32" : "v_add_f32   ", NV,
           wgs_per_cu, best, (double)grid * 4 * iters * 16 * 4096.0 / best / 1e9, best * 1e6 / mfmas);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 3; ++w) {
        run<0, false>(w, 20000);
        run<1, false>(w, 20000);
        run<2, false>(w, 20000);
        run<4, false>(w, 20000);
        run<6, false>(w, 20000);
        run<8, false>(w, 20000);
        run<12, false>(w, 20000);
        run<16, false>(w, 20000);
        run<4, true>(w, 20000);
        run<8, true>(w, 20000);
    }
    return 0;
}

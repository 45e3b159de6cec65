// Sustained fp32 MFMA rate of the part (v_mfma_f32_32x32x2_f32 from registers only: no LDS, no memory).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int wgs_per_cu, int iters) {
    float* out;
    const int grid = 256 * wgs_per_cu;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, 256>>>(out, 10);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<NACC><<<grid, 256>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 16 * NACC * 4096.0;
        printf("acc=%d wgs/cu=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, iters, ms, flops / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run<4>(1, 2000);      // ~0.5 ms bursts
    run<4>(2, 2000);
    run<4>(1, 40000);     // ~10 ms sustained
    run<4>(2, 40000);
    run<2>(2, 40000);
    run<4>(4, 20000);
    return 0;
}

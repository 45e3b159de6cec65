// Sustained fp32 MFMA rate with REALISTIC operand data (random floats, different per lane and per instruction) vs the
// constant-operand loop of mfma_peak.hip: is the part power/clock-limited once the multiplier arrays actually toggle?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak_data.hip -o /tmp/mfma_peak_data && /tmp/mfma_peak_data
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: constants, 1: random operands held in 16 registers each, cycled
__global__ __launch_bounds__(256) void k(const float* src, float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = MODE ? src[(threadIdx.x * 16 + i) & 65535] : 1.0f;
        b[i] = MODE ? src[(threadIdx.x * 16 + i + 7777) & 65535] : 1.0f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 15], b[(u * 3 + i) & 15], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const float* src, int wgs_per_cu, int iters) {
    float* out;
    const int grid = 256 * wgs_per_cu;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(src, out, 10);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<grid, 256>>>(src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode=%d wgs/cu=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", MODE, wgs_per_cu, iters, ms, (double)grid * 4 * iters * 64 * 4096.0 / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float* src;
    hipMalloc(&src, 65536 * 4);
    hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
    run<0>(src, 2, 20000);
    run<1>(src, 2, 20000);
    run<1>(src, 1, 20000);
    run<0>(src, 2, 20000);
    return 0;
}

// What would the fp32-accurate product block cost if its three smallest cross terms ran as fp8 MFMAs?  Registers only (no LDS, no memory), random
// operand bits so that the multipliers toggle; sustained launches (the part runs at its power cap under this load: wall clock is what counts).
//   A: 6 x v_mfma_f32_32x32x16_bf16 per 16-wide k-group (today's bf16x3 form)           = 24 per 64 k
//   B: 3 x v_mfma_f32_32x32x16_bf16 per k-group + 3 x v_mfma_f32_32x32x64_f8f6f4 (fp8) per 64 k   = 12 + 3 per 64 k
//   C: 3 x bf16 per k-group only (lower bound: the small terms for free)
// hipcc --offload-arch=gfx950 -O3 -w tools/micro/mfma_mix.hip -o tools/micro/mfma_mix.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    bf16x8 a[3], b[3];
    i32x8 fa[3], fb[3];
    for (int p = 0; p < 3; ++p) {
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u; a[p][e] = (__bf16)((float)(h >> 8) * 1e-7f - 0.8f);
            h = h * 1664525u + 1013904223u; b[p][e] = (__bf16)((float)(h >> 8) * 1e-7f - 0.8f);
            h = h * 1664525u + 1013904223u; fa[p][e] = (int)(h & 0x3f3f3f3f) | 0x20202020;       // fp8 e4m3 bytes of moderate magnitude
            h = h * 1664525u + 1013904223u; fb[p][e] = (int)(h & 0x3f3f3f3f) | 0x20202020;
        }
    }
    for (int it = 0; it < iters; ++it) {                       // one iteration = 64 k for 4 accumulator tiles (2 x 2 of a wave)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int t = 0; t < (MODE == 0 ? 6 : 3); ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % 3], b[(t + i) % 3], acc[i], 0, 0, 0);
        }
        if (MODE == 1) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[t], fb[(t + i) % 3], acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<512, 256>>>(out, iters / 10, 1);
    hipEventRecord(e0);
    k<MODE><<<512, 256>>>(out, iters, 7);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    const int iters = 20000;                                   // tens of ms per launch: sustained, power-limited clocks
    for (int rep = 0; rep < 2; ++rep) {
        const float a = run<0>(out, iters), b = run<1>(out, iters), c = run<2>(out, iters);
        const double blocks = 512.0 * 4 * iters * 4 * 4;       // (wave, tile, 16-wide k-group) product blocks
        printf("6 bf16 per block: %7.2f ms (%.0f TFLOP/s executed)   3 bf16 + fp8 K=64 for the small terms: %7.2f ms (%.2fx)   3 bf16 only: %7.2f ms (%.2fx)\n",
               a, blocks * 6 * 32768.0 / a / 1e9, b, a / b, c, a / c);
    }
    return 0;
}

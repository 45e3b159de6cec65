// Can the fp32-accurate product run as THREE fp16 MFMAs (two-term split: x = hi + lo, hi = rn16(x), lo = rn16(x - hi); hi*hi + hi*lo + lo*hi)
// instead of the six bf16 MFMAs of the three-term bf16 split?  Three questions, one binary:
//   A  does v_mfma_f32_32x32x16_f16 honour SUBNORMAL fp16 inputs (lo of a small x is one) or flush them?
//   B  error per output element, in units of 2^-24 * sum |x||w| against an fp64 sum, of
//        0: the fp32 pipe (v_mfma_f32_32x32x2_f32)         1: bf16 x 3, six terms (today's kernels)
//        2: fp16 x 2, three terms, ONE accumulator, operands pre-scaled by 2^sx / 2^sw          3: fp16 x 2 with lo scaled by 2^11 into a SECOND accumulator
//      on post-ReLU-like, one-signed, tiny and large data
//   C  sustained rate of 3 fp16 MFMAs per product block against 6 bf16
// hipcc --offload-arch=gfx950 -O3 -w tools/micro/f16_split.hip -o tools/micro/f16_split.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__global__ void denorm_probe(float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = __builtin_bit_cast(_Float16, (unsigned short)(blockIdx.x == 0 ? 0x0010 : 0x0001)); b[e] = (_Float16)1.0f; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[blockIdx.x] = acc[0];
    // and the conversion: does float -> half keep subnormal results?
    if (lane == 0 && blockIdx.x == 0) {
        volatile float tiny = 3.0e-6f;      // fp16 subnormal range (< 6.1e-5)
        const f16x2 h = __builtin_convertvector(f32x2{tiny, tiny * 0.01f}, f16x2);
        out[2] = (float)h[0];
        out[3] = (float)h[1];
    }
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// one wave = one 32 x 32 output tile; x[tile][32][K], w[tile][32][K]
template <int MODE>
__global__ __launch_bounds__(64) void acc_kernel(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ out, int K, float sx, float sw) {
    const int lane = threadIdx.x, t = blockIdx.x;
    const float* xr = X + ((size_t)t * 32 + (lane & 31)) * K;
    const float* wr = Wt + ((size_t)t * 32 + (lane & 31)) * K;
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[k + (lane >> 5)], wr[k + (lane >> 5)], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            const int kb = k0 + 8 * (lane >> 5);
            if (MODE == 1) {
                bf16x8 a[3], b[3];
                for (int e = 0; e < 8; ++e) {
                    float x = xr[kb + e], w = wr[kb + e];
                    for (int p = 0; p < 3; ++p) {
                        a[p][e] = (__bf16)x; x -= (float)a[p][e];
                        b[p][e] = (__bf16)w; w -= (float)b[p][e];
                    }
                }
                const int TI[6] = {0, 1, 2, 0, 1, 0}, TJ[6] = {2, 1, 0, 1, 0, 0};
                for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TI[q]], b[TJ[q]], acc, 0, 0, 0);
            } else {
                f16x8 a[2], b[2];
                const float lsc = MODE == 3 ? 2048.f : 1.f;
                for (int e = 0; e < 8; ++e) {
                    float x = xr[kb + e] * sx, w = wr[kb + e] * sw;
                    a[0][e] = (_Float16)x; a[1][e] = (_Float16)((x - (float)a[0][e]) * lsc);
                    b[0][e] = (_Float16)w; b[1][e] = (_Float16)((w - (float)b[0][e]) * lsc);
                }
                if (MODE == 2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
                } else {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
                }
            }
        }
    }
    const float inv = 1.f / (sx * sw);
    for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        if (MODE == 3) v += acc2[r] * (1.f / 2048.f);
        if (MODE >= 2) v *= inv;
        out[((size_t)t * 32 + acc_row(r, lane)) * 32 + (lane & 31)] = v;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    bf16x8 a[3], b[3];
    f16x8 fa[2], fb[2];
    for (int p = 0; p < 3; ++p)
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u; a[p][e] = (__bf16)((float)(h >> 8) * 1e-7f - 0.8f);
            h = h * 1664525u + 1013904223u; b[p][e] = (__bf16)((float)(h >> 8) * 1e-7f - 0.8f);
            h = h * 1664525u + 1013904223u; fa[p & 1][e] = (_Float16)((float)(h >> 8) * 1e-7f - 0.8f);
            h = h * 1664525u + 1013904223u; fb[p & 1][e] = (_Float16)((float)(h >> 8) * 1e-7f - 0.8f);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (MODE == 0) {
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % 3], b[(t + i) % 3], acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[t & 1], fb[(t + i) & 1], acc[i], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
float run_rate(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<512, 256>>>(out, iters / 10, 1);
    hipEventRecord(e0);
    rate_kernel<MODE><<<512, 256>>>(out, iters, 7);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

static double gauss(unsigned long long& s) {
    double u = 0;
    for (int i = 0; i < 12; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; u += (double)(s >> 11) / 9007199254740992.0; }
    return u - 6.0;
}

int main() {
    float* dout;
    hipMalloc(&dout, 512 * 256 * 4);
    denorm_probe<<<2, 64>>>(dout);
    float pr[4];
    hipMemcpy(pr, dout, 16, hipMemcpyDeviceToHost);
    printf("A  16 x (2^-20 * 1) through v_mfma_f32_32x32x16_f16 = %.9g (expected %.9g); 16 x (2^-24 * 1) = %.9g (expected %.9g): subnormal fp16 inputs are %s\n",
           pr[0], 16 * ldexp(1.0, -20), pr[1], 16 * ldexp(1.0, -24), pr[0] > 0 ? "HONOURED" : "FLUSHED");
    printf("A  float -> half of 3.0e-6 = %.9g, of 3.0e-8 = %.9g (subnormal results of the conversion %s)\n", pr[2], pr[3], pr[2] > 0 ? "kept" : "flushed");

    const int T = 256;
    const char* dist_name[5] = {"post-ReLU x (half zeros), signed w ~ N/sqrt(K)", "one-signed x, w", "tiny: x ~ 1e-3 |N|, w ~ 1e-2 N/sqrt(K)", "large: x ~ 1e3 |N|", "wide: x ~ |N| * 2^U(-12,4)"};
    for (int K : {128, 256, 1152}) {
        for (int dist = 0; dist < 5; ++dist) {
            std::vector<float> X((size_t)T * 32 * K), W((size_t)T * 32 * K);
            unsigned long long s = 1234567 + dist * 77 + K;
            for (size_t i = 0; i < X.size(); ++i) {
                double x = gauss(s), w = gauss(s) / sqrt((double)K);
                if (dist == 0) x = x > 0 ? x : 0;
                if (dist == 1) { x = fabs(x); w = fabs(w); }
                if (dist == 2) { x = fabs(x) * 1e-3; w *= 1e-2; }
                if (dist == 3) x = fabs(x) * 1e3;
                if (dist == 4) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = fabs(x) * ldexp(1.0, (int)((s >> 33) % 17) - 12); }
                X[i] = (float)x; W[i] = (float)w;
            }
            float *dX, *dW, *dO;
            hipMalloc(&dX, X.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dO, (size_t)T * 1024 * 4);
            hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
            std::vector<double> ref((size_t)T * 1024), mag((size_t)T * 1024);
            for (int t = 0; t < T; ++t)
                for (int i = 0; i < 32; ++i)
                    for (int j = 0; j < 32; ++j) {
                        double a = 0, m = 0;
                        const float* xr = &X[((size_t)t * 32 + i) * K];
                        const float* wr = &W[((size_t)t * 32 + j) * K];
                        for (int k = 0; k < K; ++k) { a += (double)xr[k] * wr[k]; m += fabs((double)xr[k] * wr[k]); }
                        ref[((size_t)t * 32 + i) * 32 + j] = a; mag[((size_t)t * 32 + i) * 32 + j] = m;
                    }
            printf("B  K = %4d, %s\n", K, dist_name[dist]);
            struct V { int mode; float sx, sw; const char* name; };
            const V vs[] = {{0, 1, 1, "fp32 pipe"}, {1, 1, 1, "bf16 x 3 (6 MFMA)"}, {2, 1, 1, "fp16 x 2, one acc, unscaled"}, {2, 256, 4096, "fp16 x 2, one acc, x*2^8 w*2^12"},
                            {2, 16, 256, "fp16 x 2, one acc, x*2^4 w*2^8"}, {3, 1, 1, "fp16 x 2, lo*2^11, two acc"}, {3, 1, 256, "fp16 x 2, lo*2^11, two acc, w*2^8"}};
            for (const V& v : vs) {
                switch (v.mode) {
                    case 0: acc_kernel<0><<<T, 64>>>(dX, dW, dO, K, v.sx, v.sw); break;
                    case 1: acc_kernel<1><<<T, 64>>>(dX, dW, dO, K, v.sx, v.sw); break;
                    case 2: acc_kernel<2><<<T, 64>>>(dX, dW, dO, K, v.sx, v.sw); break;
                    default: acc_kernel<3><<<T, 64>>>(dX, dW, dO, K, v.sx, v.sw); break;
                }
                std::vector<float> got((size_t)T * 1024);
                hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost);
                double sum = 0, sq = 0, mx = 0;
                int bad = 0;
                for (size_t i = 0; i < got.size(); ++i) {
                    if (!isfinite(got[i])) { ++bad; continue; }
                    const double e = ((double)got[i] - ref[i]) / (ldexp(1.0, -24) * mag[i]);
                    sum += e; sq += e * e; if (fabs(e) > mx) mx = fabs(e);
                }
                const double n = (double)got.size() - bad;
                printf("     %-36s mean %+8.3f  std %8.3f  max %9.3f  non-finite %d   (gate 2 sqrt(K) = %.1f)\n", v.name, sum / n, sqrt(sq / n - (sum / n) * (sum / n)), mx, bad, 2 * sqrt((double)K));
            }
            hipFree(dX); hipFree(dW); hipFree(dO);
        }
    }
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        const float a = run_rate<0>(dout, iters), b = run_rate<1>(dout, iters);
        const double blocks = 512.0 * 4 * iters * 4 * 4;
        printf("C  6 bf16 per block: %7.2f ms (%.0f TFLOP/s executed)   3 fp16 per block: %7.2f ms (%.0f TFLOP/s executed, %.2fx)\n", a, blocks * 6 * 32768.0 / a / 1e9, b,
               blocks * 3 * 32768.0 / b / 1e9, a / b);
    }
    return 0;
}

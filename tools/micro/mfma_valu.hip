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


// LDS instructions between the MFMAs: MODE 0 ds_read_b128, 1 ds_write_b128, 2 ds_write_b32 (conflict-free addresses).  The read
// destinations are 8 variables that stay live to the end of the kernel (an asynchronous return into a reallocated register would
// corrupt it); s_waitcnt only after the loop body.
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int NL, int MODE>
__global__ __launch_bounds__(256) void kl(float* out, int iters) {
    __shared__ float4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = float4{1.f * i, 0.f, 0.f, 0.f};
    __syncthreads();
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    vf4 v0 = {a, b, a, b}, v1 = v0, v2 = v0, v3 = v0, v4 = v0, v5 = v0, v6 = v0, v7 = v0;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)&lds[0] + threadIdx.x * 16;      // 256 threads x 16 B = 4 KB rows
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 1], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                const int slot = (u * NL + q) & 7;
                const unsigned la = base + (slot & 7) * 4096;
#define SUO_LDS_OP(V)                                                                                    \
                if (MODE == 0) asm volatile("ds_read_b128 %0, %1" : "+v"(V) : "v"(la) : "memory");      \
                else if (MODE == 1) asm volatile("ds_write_b128 %1, %0" : : "v"(V), "v"(la) : "memory"); \
                else asm volatile("ds_write_b32 %1, %0" : : "v"(V[0]), "v"(la) : "memory");
                if (slot == 0) { SUO_LDS_OP(v0) } else if (slot == 1) { SUO_LDS_OP(v1) } else if (slot == 2) { SUO_LDS_OP(v2) } else if (slot == 3) { SUO_LDS_OP(v3) }
                else if (slot == 4) { SUO_LDS_OP(v4) } else if (slot == 5) { SUO_LDS_OP(v5) } else if (slot == 6) { SUO_LDS_OP(v6) } else { SUO_LDS_OP(v7) }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    s += v0[0] + v1[0] + v2[0] + v3[0] + v4[0] + v5[0] + v6[0] + v7[0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NL, int MODE>
void runl(int wgs_per_cu, int iters) {
    float* out;
    const int grid = 256 * wgs_per_cu;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kl<NL, MODE><<<grid, 256>>>(out, 10);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kl<NL, MODE><<<grid, 256>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double mfmas = (double)iters * 16 * wgs_per_cu;
    printf("%s N=%d waves/SIMD=%d: %.3f ms -> %.1f ns per MFMA per SIMD\n", MODE == 0 ? "ds_read_b128 " : MODE == 1 ? "ds_write_b128" : "ds_write_b32 ", NL, wgs_per_cu, best,
           best * 1e6 / mfmas);
    hipFree(out);
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
    for (int w = 1; w <= 2; ++w) {
        runl<1, 0>(w, 20000); runl<2, 0>(w, 20000); runl<4, 0>(w, 20000);
        runl<1, 1>(w, 20000); runl<2, 1>(w, 20000); runl<4, 1>(w, 20000);
        runl<1, 2>(w, 20000); runl<4, 2>(w, 20000);
    }
    return 0;
}

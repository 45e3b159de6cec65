// What does an LDS-fed MFMA loop sustain?  Register-only MFMA reaches 155 TFLOP/s (mfma_peak.hip); the conv main loop
// with all global traffic removed only 129.  Variants of the conv's inner structure, in isolation:
//   0  MFMA only, operands in registers
//   1  + 2 ds_read_b128 per 16 MFMA (A fragments of 2 M-tiles, read just before use as hipcc schedules them)
//   2  as 1 but the fragments for group g+1 are read before the MFMAs of group g (explicit double buffer)
//   3  as 2 + one workgroup barrier per 576 MFMA (one conv chunk)
//   4  as 3 + 8 global (L2-resident) 16-byte loads per 64 MFMA feeding the B operands (static two-buffer ring)
//   5  as 4 + the A staging of a real chunk: 6 streaming 16-byte global loads per thread, written to the other LDS buffer
//   6  as 5 + a tile epilogue every 4 chunks: 64 accumulator registers -> LDS patch -> 16 16-byte global stores per thread
//   7  as 6 but the A loads stream from a 512 MB buffer (HBM, not L2)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, const float* __restrict__ wsrc, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[2 * 180 * 36];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 180 * 36; i += 256) As[i] = src[i];
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 b0[4][2], b1[4][2];
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < 2; ++j) { b0[s][j] = *(const f32x4*)(wsrc + ((s * 2 + j) * 64 + lane) * 4); b1[s][j] = b0[s][j]; }
    const float* as0 = &As[((w >> 1) * 64 + (lane & 31)) * 36 % (148 * 36) + (lane >> 5) * 4];
    const float* as = as0;
    f32x4 afA[2], afB[2];
    afA[0] = *(const f32x4*)(as); afA[1] = *(const f32x4*)(as + 32 * 36);
    afB[0] = afA[0]; afB[1] = afA[1];
    const float* wl = wsrc + lane * 4;
    auto group = [&](const f32x4(&af)[2], const f32x4(&b)[2]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], b[j][t], acc[i][j], 0, 0, 0);
    };
    const f32x4* abig = (const f32x4*)src;          // 1M floats = 256K float4, streamed
    f32x4* obig = (f32x4*)out;
    f32x4 areg[6];
    for (int it = 0; it < iters; ++it) {            // one iteration = one "chunk" = 9 taps x 4 groups x 16 MFMA
        if (MODE >= 5) {
#pragma unroll
            for (int i = 0; i < 6; ++i) areg[i] = abig[MODE >= 7 ? ((size_t)blockIdx.x * iters + it) * 1536 % 33554432 + i * 256 + tid + 262144 : ((size_t)(blockIdx.x * 997 + it * 1536 + i * 256 + tid)) & 262143];
        }
        if (MODE >= 5) as = as0 + (it & 1) * 180 * 36;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = ((tap / 3) * 18 + tap % 3) * 36;
            if (MODE >= 4) {
                // next tap's weights into the other ring buffer (L2 hits), requested before this tap's MFMAs
                const int q = (it * 9 + tap + 1) & 63;
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4 v = *(const f32x4*)(wl + (size_t)((q * 4 + s) * 2 + j) * 256);
                        if (tap & 1) b0[s][j] = v; else b1[s][j] = v;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4(&b)[2] = (MODE >= 4 && (tap & 1)) ? b1[s] : b0[s];
                if (MODE == 0) {
                    group(afA, b);
                } else if (MODE == 1) {
                    f32x4 af[2];
                    af[0] = *(const f32x4*)(as + toff + s * 8);
                    af[1] = *(const f32x4*)(as + 32 * 36 + toff + s * 8);
                    group(af, b);
                } else {
                    const int noff = s < 3 ? toff + (s + 1) * 8 : (((tap + 1) % 9 / 3) * 18 + (tap + 1) % 9 % 3) * 36;
                    if ((s & 1) == 0) {
                        afB[0] = *(const f32x4*)(as + noff); afB[1] = *(const f32x4*)(as + 32 * 36 + noff);
                        __builtin_amdgcn_sched_barrier(0);
                        group(afA, b);
                    } else {
                        afA[0] = *(const f32x4*)(as + noff); afA[1] = *(const f32x4*)(as + 32 * 36 + noff);
                        __builtin_amdgcn_sched_barrier(0);
                        group(afB, b);
                    }
                }
            }
        }
        if (MODE >= 5) {
#pragma unroll
            for (int i = 0; i < 6; ++i) { const int idx = tid + i * 256; if (idx < 1440) *(f32x4*)&As[((it + 1) & 1) * 180 * 36 + (idx >> 3) * 36 + (idx & 7) * 4] = areg[i]; }
        }
        if (MODE >= 3) __syncthreads();
        if (MODE >= 6 && (it & 3) == 3) {   // (mode 7 included)
            float* T = &As[w * 32 * 36];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[i][j][r]; acc[i][j][r] = 0.f; }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        f32x4 o = *(const f32x4*)&T[((lane >> 3) + 8 * kk) * 36 + (lane & 7) * 4];
                        obig[65536 + (((size_t)blockIdx.x * 4099 + it * 64 + (i * 2 + j) * 16 + kk * 4) * 64 + lane) % 4000000] = o;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            __syncthreads();
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const float* src, const float* wsrc, int wgs_per_cu, int iters, int grid_override = 0) {
    float* out;
    const int grid = grid_override ? grid_override : 256 * wgs_per_cu;
    hipMalloc(&out, (size_t)80 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(src, wsrc, out, 2);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<grid, 256>>>(src, wsrc, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("mode=%d grid=%d iters=%d: best %.3f ms  %.1f TFLOP/s\n", MODE, grid, iters, best, (double)grid * 4 * iters * 576 * 4096.0 / best / 1e9);
    hipFree(out);
}

int main() {
    const int n = 1 << 20;
    const size_t nbig = ((size_t)512 << 20) / 4 + (1 << 20);
    float* h = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *src, *wsrc;
    hipMalloc(&src, nbig * 4 + (64 << 20)); hipMalloc(&wsrc, n * 4);
    hipMemset(src, 0, nbig * 4);
    hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    hipMemcpy(wsrc, h, n * 4, hipMemcpyHostToDevice);
    // workgroup turnover: the same work as 4096 one-tile workgroups (4 chunks each) or as 512 persistent ones (32 chunks)
    run<7>(src, wsrc, 0, 4, 4096);
    run<7>(src, wsrc, 0, 32, 512);
    run<7>(src, wsrc, 0, 128, 512);
    run<7>(src, wsrc, 0, 4, 16384);
    run<6>(src, wsrc, 0, 4, 4096);
    run<6>(src, wsrc, 0, 32, 512);
    run<6>(src, wsrc, 0, 4, 16384);
    run<6>(src, wsrc, 0, 128, 512);
    run<3>(src, wsrc, 0, 4, 4096);
    run<3>(src, wsrc, 0, 32, 512);
    run<4>(src, wsrc, 0, 32, 512);
    run<4>(src, wsrc, 0, 128, 512);
    run<5>(src, wsrc, 0, 32, 512);
    run<5>(src, wsrc, 0, 128, 512);
    run<6>(src, wsrc, 0, 8, 512);
    run<6>(src, wsrc, 0, 16, 512);
    run<6>(src, wsrc, 0, 64, 512);
    run<6>(src, wsrc, 0, 512, 512);
    const int it = 600;
    for (int wg = 1; wg <= 2; ++wg) {
        run<0>(src, wsrc, wg, it);
        run<1>(src, wsrc, wg, it);
        run<2>(src, wsrc, wg, it);
        run<3>(src, wsrc, wg, it);
        run<4>(src, wsrc, wg, it);
        run<5>(src, wsrc, wg, it);
        run<6>(src, wsrc, wg, it);
    }
    return 0;
}

// The 2-way fp16 operand split behind the "f16x2" kernels (csrc/gemm_bf16x3.hip and csrc/conv_wino_x3.hip with NP = 2): the network's default
// matrix-pipe form since round 5.  THREE v_mfma_f32_32x32x16_f16 per product block where the bf16 split (csrc/bf16x3.h) issues six.
//
//     x = hi + lo + e,   hi = rn16(x),  lo = rn16(x - hi)  (x - hi exact in fp32),  |e| <= 2^-22 |x|  (2^-11 of a residual of <= 2^-11 |x|)
//     x w  ~  hi_x lo_w + lo_x hi_w + hi_x hi_w        (fp16 x fp16 is exact in fp32; fp32 accumulate; dropped: lo_x lo_w <= 2^-22 |x w|)
// The representation error e is RANDOM in sign and 2^-22 only in the worst case (rms ~ 0.25 * 2^-24 |x|): over a K-term dot product it adds
// ~ 0.3 / sqrt(K) units of 2^-24 * sum |x||w| -- nothing beside the accumulation's own rounding (2-4 units at K = 128 ... 1152, either pipe).  Measured on
// the matrix pipe itself, per output element in those units (tools/micro/f16_split.hip, profiles/r05_f16_split_micro.txt; K = 256):
//     post-ReLU data        fp32 pipe std 0.47 max 5.5 | bf16 x 3 std 0.53 max 4.9 | this split std 0.43 max 4.7
//     one-signed data       fp32 pipe std 4.12 max 19.6 | bf16 x 3 std 3.34 max 15.9 | this split std 2.46 max 11.3   (3 K / 16 roundings against 6 K / 16 against K)
// What fp16 does NOT have is bf16's range, so both operands are brought into it by EXACT powers of two:
//   * weights: row n (one output channel) times 2^t_n with max_k |w_nk| 2^t_n in [2^12, 2^13)  (host, when a network is built)
//   * activations: times 2^S2_XSHIFT (folded into the BN prologue's scale / shift where there is one -- fmaf(x, 16 a, 16 b) = 16 fmaf(x, a, b) exactly)
//   * the epilogue multiplies the accumulator by 2^-(t_n + S2_XSHIFT) (one fused multiply-add with the bias instead of an add)
// Small values: v_mfma_f32_32x32x16_f16 HONOURS subnormal fp16 inputs and v_cvt_pk_f16_f32 produces them (probed: tools/micro/f16_split.hip, part A), so a
// residual below 2^-14 keeps an absolute precision of 2^-25: an activation loses relative accuracy only below ~2^-7 / 16, where its product no longer
// matters to a sum of O(1) terms (measured: activations of 1e-3 -> std 1.8 units; the bf16 split 0.4; gate 2 sqrt(K) = 32).
// Large values -- the RANGE GUARD: a scaled activation of 65504 or more would round to +-inf.  Every kernel tracks max |x'| of what it splits and raises
// *range_flag (system-scope store: the flag may live in mapped host memory) when it reaches S2_LIMIT; the launch's results are then INVALID and whoever
// owns the flag re-issues the work on the bf16x3 kernels (csrc/net.hip: Net::range_exceeded; the C entries return the flag to the caller).
// An infinite input raises the flag too; a nan passes through to the output as it does on either other pipe (fmaxf drops it from the running max).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <hip/hip_runtime.h>

namespace suo {

constexpr int S2_XSHIFT = 4;                      // activations enter the split times 16: range guard at |x| >= 4094
constexpr float S2_XSCALE = 16.f;
constexpr float S2_LIMIT = 65504.f;               // largest finite fp16

typedef _Float16 s2_f16x2 __attribute__((ext_vector_type(2)));
typedef float s2_f32x2 __attribute__((ext_vector_type(2)));

// fp16(a) in the low half, fp16(b) in the high half, round-to-nearest-even (one v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned s2_pack_rn(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(s2_f32x2{a, b}, s2_f16x2)); }
// the two halves of such a pair as fp32 values (v_cvt_f32_f16, the upper half by SDWA)
__device__ __forceinline__ float s2_lo(unsigned p) { return (float)__builtin_bit_cast(s2_f16x2, p)[0]; }
__device__ __forceinline__ float s2_hi(unsigned p) { return (float)__builtin_bit_cast(s2_f16x2, p)[1]; }
// the residual pair of (a, b) against their rounded pair h: fp16(a - hi_a) | fp16(b - hi_b) << 16.  v_fma_mixlo / mixhi_f16 read the fp16 half as an fp32 operand,
// form fma(hi, -1, x) in fp32 (exact here) and round the result to fp16 in ONE instruction each -- 2 per pair where cvt_f32_f16 x 2 + v_pk_add_f32 + v_cvt_pk_f16_f32
// take 4.  Same value bit for bit (tests/test_gpu_f16x2.py: test_device_split_equals_numpy_float16_rounding); -DSUO_S2_MIX=0 builds the four-instruction form.
#ifndef SUO_S2_MIX
#define SUO_S2_MIX 1
#endif
__device__ __forceinline__ unsigned s2_lo_pack(float a, float b, unsigned h) {
#if SUO_S2_MIX
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(b));
    return r;
#else
    return s2_pack_rn(a - s2_lo(h), b - s2_hi(h));
#endif
}
// running max of magnitudes for the range guard: m <- max(m, |a|, |b|) (one v_max3_f32 with source modifiers)
__device__ __forceinline__ float s2_track(float m, float a, float b) { return fmaxf(fmaxf(m, fabsf(a)), fabsf(b)); }
// raise the flag when the lane saw a value at or beyond the fp16 range
__device__ __forceinline__ void s2_raise(unsigned* flag, float m) {
    if (!(m < S2_LIMIT) && flag) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// host: fp32 -> fp16 bits, round-to-nearest-even, subnormals kept, overflow -> inf (the device's v_cvt_pk_f16_f32)
static inline uint16_t s2_rn_host(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0u));
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);              // >= 65520: rounds to inf
    if (u < 0x38800000u) {                                                 // below 2^-14: a multiple of 2^-24 (nearbyintf: ties to even in the default mode)
        float f;
        memcpy(&f, &u, 4);
        return (uint16_t)(sign | (uint16_t)nearbyintf(f * 16777216.0f));  // (1024 = the smallest normal: the carry lands where it belongs)
    }
    const uint32_t r = u + 0xfffu + ((u >> 13) & 1u);
    return (uint16_t)(sign | (uint16_t)((r - 0x38000000u) >> 13));
}
static inline float s2_to_float_host(uint16_t h) {
    const int e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(1024 + m), e - 25);
    return (h & 0x8000u) ? -v : v;
}
// the two terms of x (already scaled into range by the caller)
static inline void s2_split_host(float x, uint16_t out[2]) {
    out[0] = s2_rn_host(x);
    out[1] = s2_rn_host(x - s2_to_float_host(out[0]));                     // the subtraction is exact
}
// exponent t with max 2^t in [2^12, 2^13) (0 for an all-zero row): the per-output-channel weight scale
static inline int s2_row_shift(float max_abs) {
    if (!(max_abs > 0.f) || !isfinite(max_abs)) return 0;
    int e;
    (void)frexpf(max_abs, &e);                                             // max_abs = f 2^e, f in [0.5, 1): max_abs in [2^(e-1), 2^e)
    int t = 13 - e;
    if (t > 100) t = 100;                                                  // (keeps 2^-(t + S2_XSHIFT) a normal fp32 number)
    if (t < -100) t = -100;
    return t;
}

}  // namespace suo

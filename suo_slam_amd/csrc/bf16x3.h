// The 3-way bf16 operand split behind every "bf16x3" kernel (csrc/gemm_bf16x3.hip, csrc/conv_wino_x3.hip, csrc/res_small.hip).
//
// An fp32 number is EXACTLY the sum of three bf16 numbers when each term is the round-to-nearest-even bf16 of what is left:
//     x0 = rn(x),  x1 = rn(x - x0),  x2 = x - x0 - x1
// x - x0 is a multiple of ulp(x) no larger than half a bf16 ulp of x (<= 2^15 units: exact in fp32), rounding it to 8 bits leaves a
// multiple of the same unit no larger than 2^6 -- which IS a bf16 number.  So every subtraction and the last conversion are exact.
// The kernels accumulate the six cross terms x_i w_j, i + j <= 2, in fp32 (bf16 x bf16 is exact in fp32); the dropped ones are
//     x1 w2 + x2 w1 + x2 w2,   |x1| <= 2^-8 |x|_binade,  |x2|, |w2| <= 2^-17  ->  <= 2^-25 |x w| each, of EITHER sign.
// Rounds 1-3 split by truncation (x0 = the leading 16 bits): also exact, but every residual then has the sign of x, the dropped terms
// the sign of x w and four times the size -- on all-positive activations against one-signed weights a systematic pull toward zero.
// Measured per output element in units of 2^-24 * sum |x||w| (profiles/r04_bias_ab_split.txt, tools/bias_ab.sh; the fp32-pipe kernels:
// mean 0.00, standard deviation 2.9-4.1): truncating split mean -0.78 ... -0.81 (fused Residual tail -1.50), this split -0.11 ... -0.18
// (-0.30).  What is left is not the split's: v_mfma_f32_32x32x16_bf16 aligns its 16 products and the accumulator and drops the
// shifted-out bits (toward -inf, ~12 guard bits) instead of rounding, 6 K / 16 times per output; the same -0.02 shows on sign-mixed
// data.  It is 1/20 of the rounding noise either pipe has, and the standard deviation and the worst element of the bf16x3 kernels are
// BELOW the fp32 pipe's in every case measured (fewer roundings: 6 K / 16 against K).
// v_cvt_pk_bf16_f32 (gfx950) converts two values per instruction with round-to-nearest-even; per pair of values the split is
// 3 conversions + 4 expansions + 2 packed subtractions, no more instructions than the truncating form took.
#pragma once
#include <stdint.h>
#include <string.h>

#include <hip/hip_runtime.h>

namespace suo {

typedef __bf16 s3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float s3_f32x2 __attribute__((ext_vector_type(2)));

// bf16(a) in the low half, bf16(b) in the high half, round-to-nearest-even (one v_cvt_pk_bf16_f32)
// (-DSUO_S3_TRUNC builds the truncating split of rounds 1-3 again, device and host: tools/bias_ab.sh measures the two side by side)
__device__ __forceinline__ unsigned s3_pack_rn(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(s3_f32x2{a, b}, s3_bf16x2));
}
// the two halves of such a pair as fp32 values
__device__ __forceinline__ float s3_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float s3_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// host: the same split (weights are split once, when a network is built)
static inline uint16_t s3_rn_host(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);      // inf / nan: keep the leading bits
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline void s3_split_host(float x, uint16_t out[3]) {
    for (int p = 0; p < 3; ++p) {
        out[p] = s3_rn_host(x);
        const uint32_t u = (uint32_t)out[p] << 16;
        float t;
        memcpy(&t, &u, 4);
        x -= t;                                                             // exact
    }
}

}  // namespace suo

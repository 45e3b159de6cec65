// Buffer-descriptor (SRD) memory operations shared by the MFMA kernels (csrc/conv.hip, csrc/gemm_persist.hip).
//
// A stream is addressed as  descriptor base (4 SGPRs) + per-lane byte offset (ONE 32-bit VGPR) + scalar byte offset
// (an SGPR: the loop-variant part).  Compared with flat 64-bit addresses this frees the VGPR pairs and the 64-bit VALU
// address arithmetic of every load / store, and the hardware range check gives zero padding (loads) and discarded
// out-of-tile elements (stores) without a branch -- every steady-state load and store is unconditional, so hipcc's
// `s_waitcnt vmcnt(N)` stays exact instead of falling back to vmcnt(0).
// The descriptor inputs go through readfirstlane so hipcc can prove them wave-uniform; otherwise it wraps each buffer
// op in a waterfall loop.
#pragma once
#include <hip/hip_runtime.h>

namespace suo {

typedef float bo_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int bo_u32x4 __attribute__((ext_vector_type(4)));

constexpr int BUF_OOB = (int)0x80000000;      // a per-lane offset beyond any num_records used here

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* p, size_t bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((unsigned)bytes), 0x00020000);
}
__device__ __forceinline__ bo_f32x4 buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(bo_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store(bo_f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff = 0) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bo_u32x4, v), r, voff, soff, 0);
}

}  // namespace suo

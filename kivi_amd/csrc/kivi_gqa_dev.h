// Shared device pieces of the matrix-pipe kernels over the KT / VT cache layouts (kivi_mfma_layout.h): storage
// addressing, packed-half helpers, wave reductions.
#pragma once
#include "kivi_common.h"
#include "kivi_mfma_layout.h"

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Segments of a score row for the softmax statistics of the decode step: one per 512-token super-block of packed K
// plus KIVI_GQA_RES_SEGS pieces of the fp16 residual (each produced by its own small block of the qK^T launch).
#define KIVI_GQA_RES_SEGS 4

struct MfStore {              // one cache side (K or V) in the super-block layout
    uint32_t* base;
    int64_t sb_b, sb_h, sb_s; // word strides: batch row, kv head, super-block
};

__device__ __forceinline__ uint32_t* mf_sb(const MfStore& s, int b, int hk, int64_t sb) {
    return s.base + b * s.sb_b + hk * s.sb_h + sb * s.sb_s;
}

__device__ __forceinline__ h8 as_h8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_bit_cast(h8, (u32x4){a, b, c, d});
}
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, a) * __builtin_bit_cast(h2, b));
}
// a * b - c, one rounding (v_pk_fma_f16): with c = fp16(a * b) the exact remainder of the product
__device__ __forceinline__ uint32_t pk_fms(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b),
                                                                  -__builtin_bit_cast(h2, c)));
}
// A rows carry 2^(4 + 2 (i >> 1)) for the register i = 0..3 of an MFMA operand (kivi_mfma_layout.h)
__device__ __forceinline__ constexpr int aexp(int i) { return KIVI_MF_SHIFT + 2 * (i >> 1); }
// ... for BITS-wide codes: 4-bit codes all sit on mantissa bits 9:6, the four registers carry 2^6
template <int BITS>
__device__ __forceinline__ constexpr int aexp_b(int i) { return BITS == 2 ? aexp(i) : 6; }

// Wave-wide max / sum with DPP inside the 16-lane rows and four v_readlane across them: no LDS round trips (the
// __shfl_xor form is 6 dependent ds_bpermute, ~600 cycles per reduction; the softmax statistics need 2 R of them per
// super-block).  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_max(float v) {
    v = __builtin_fmaxf(v, dpp_f<0xB1>(v));     // quad_perm [1,0,3,2]
    v = __builtin_fmaxf(v, dpp_f<0x4E>(v));     // quad_perm [2,3,0,1]
    v = __builtin_fmaxf(v, dpp_f<0x141>(v));    // row_half_mirror
    v = __builtin_fmaxf(v, dpp_f<0x140>(v));    // row_mirror: every lane of a row holds the row's max
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return __builtin_fmaxf(__builtin_fmaxf(a, b), __builtin_fmaxf(c, d));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (a + b) + (c + d);
}

}  // namespace

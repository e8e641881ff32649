// Shared device pieces of the matrix-pipe kernels over the KT / VT cache layouts (kivi_mfma_layout.h): storage
// addressing, packed-half helpers, the A-operand hi / lo split, the MFMA pair over one masked code word, wave reductions.
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
// A rows carry 2^(4 + 2 (i >> 1)) for the register i = 0..3 of an MFMA operand (kivi_mfma_layout.h): afac = that factor in
// both halves, zfac = its inverse (brings a zero-point operand to the scale of the A rows)
__device__ __forceinline__ constexpr uint32_t afac(int i) { return i < 2 ? 0x4C004C00u : 0x54005400u; }
__device__ __forceinline__ constexpr uint32_t zfac(int i) { return i < 2 ? 0x2C002C00u : 0x24002400u; }
__device__ __forceinline__ constexpr int aexp(int i) { return KIVI_MF_SHIFT + 2 * (i >> 1); }

// hi / lo rows of the A operand without a branch: `xf` is x in the lanes of a "lo" row and 0 in the lanes of a "hi" row,
// so  fma(x, s, -fp16(xf * s))  is the rounded product in hi rows and its exact remainder in lo rows (two packed ops).
// Without the split (HILO = false) `xf` is x in hi rows and 0 in lo rows and the element is one packed multiply.
template <bool HILO>
__device__ __forceinline__ uint32_t a_elem(uint32_t x, uint32_t xf, uint32_t s) {
    if constexpr (HILO) return pk_fms(x, s, pk_mul(xf, s));
    else return pk_mul(xf, s);
}

// one 32-channel (K) / 32-token (V) chunk: two MFMAs, the masked code words are the B operands
__device__ __forceinline__ void mfma_pair(const uint32_t* A, uint32_t w, f4& acc0, f4& acc1) {
    // four views of the word put every 2-bit field on bits 9:8 or 7:6 of its half (kivi_mfma_layout.h)
    const uint32_t x1 = w << 4, x2 = w >> 4, x3 = __builtin_amdgcn_perm(w, w, 0x02030001u);
    const h8 a = as_h8(A[0], A[1], A[2], A[3]);
    const h8 b0 = as_h8(w & 0x03000300u, x1 & 0x03000300u, w & 0x00C000C0u, x1 & 0x00C000C0u);
    const h8 b1 = as_h8(x2 & 0x03000300u, x3 & 0x03000300u, x2 & 0x00C000C0u, x3 & 0x00C000C0u);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, acc1, 0, 0, 0);
}

// rows hi + lo of two accumulator registers (tile 0 in x, tile 1 in y) in one swap + add:
//   R = 4: lanes 0-15 <- tile 0 rows j + (4 + j), lanes 16-31 <- tile 1 (v_permlane16_swap: odd rows of x <-> even rows of y)
//   R = 8: lanes 0-31 <- tile 0 rows (0..7) + (8..15), lanes 32-63 <- tile 1 (v_permlane32_swap)
// (inline asm: on ROCm 7.2 the __builtin_amdgcn_permlane16_swap / 32_swap builtins return the FIRST result in both slots
// -- hipcc emits v_add v, v, v after the swap; checked with hipcc -S)
template <int R>
__device__ __forceinline__ float fold_rows(float x, float y) {
    if constexpr (R == 4) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}

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

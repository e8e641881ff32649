// Shared device helpers for the KIVI gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kivi_hip.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));

#define KIVI_WAVE 64

// Tuning / diagnostic environment knobs exist only in -DKIVI_TUNING builds (tools/build_variant.sh tuning -DKIVI_TUNING):
// the product library reads no environment variable and carries no losing or result-changing instantiation.
#include <stdlib.h>
#ifdef KIVI_TUNING
#define KIVI_TUNE_ENV(name) getenv(name)
#else
#define KIVI_TUNE_ENV(name) ((const char*)nullptr)
#endif

// ---- argument / launch error plumbing (host side) -------------------------
void kivi_set_error(const char* fmt, ...);
int* kivi_device_error_word(hipStream_t stream);   // device pointer to the process's host-visible error word (or null): kivi_abi.hip
int kivi_take_device_error(int* unit);     // returns and clears it

#define KIVI_REQUIRE(cond, code, ...)      \
    do {                                   \
        if (!(cond)) {                     \
            kivi_set_error(__VA_ARGS__);   \
            return (code);                 \
        }                                  \
    } while (0)

static inline int kivi_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        kivi_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// ---- launch with optional per-dispatch timing events (kivi_set_launch_events, bench instrumentation):
// hipExtLaunchKernelGGL stamps the events with the dispatch's own begin / end, like a profiler does.
struct KiviLaunchEvents {
    hipEvent_t start, stop;
};
KiviLaunchEvents kivi_take_launch_events();
void kivi_note_timed_kernel(const char* name);
unsigned long long* kivi_debug_stamps();         // kivi_debug_set_stamps buffer or null   // remembers which kernel the last event pair bracketed

#define KIVI_LAUNCH_LDS(kernel, grid, block, lds, stream, ...)                                              \
    do {                                                                                                   \
        KiviLaunchEvents ev__ = kivi_take_launch_events();                                                 \
        if (ev__.start || ev__.stop) {                                                                     \
            kivi_note_timed_kernel(#kernel);                                                               \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ev__.start, ev__.stop, 0, __VA_ARGS__); \
        } else                                                                                               \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                             \
    } while (0)
#define KIVI_LAUNCH(kernel, grid, block, stream, ...) KIVI_LAUNCH_LDS(kernel, grid, block, 0, stream, __VA_ARGS__)

// max / sum over the NW * 64 threads of a block through NW floats of LDS (same tree in every kernel that uses it with
// the same NW, so the stand-alone softmax and the one fused into the sV kernel round identically)
template <int NW = 4>
__device__ __forceinline__ float kivi_block_reduce(float v, bool is_max, float* lds) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float o = __shfl_xor(v, m);
        v = is_max ? __builtin_fmaxf(v, o) : v + o;
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();   // previous use of lds is over
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    const float a = lds[0], b = lds[1], c = lds[2], d = lds[3];
    const float lo = is_max ? __builtin_fmaxf(__builtin_fmaxf(a, b), __builtin_fmaxf(c, d)) : (a + b) + (c + d);
    if constexpr (NW == 8) {
        const float e = lds[4], f = lds[5], g = lds[6], h = lds[7];
        const float hi = is_max ? __builtin_fmaxf(__builtin_fmaxf(e, f), __builtin_fmaxf(g, h)) : (e + f) + (g + h);
        return is_max ? __builtin_fmaxf(lo, hi) : lo + hi;
    }
    return lo;
}

// Block barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, i.e. s_waitcnt vmcnt(0) first: a wave
// that has requested global data ahead (the first blocks of its packed-V ring, the fp16 window rows) would stall at the barrier until
// all of it has landed -- measured 6.7 us from "residual scores done" to "past the barrier" in mf_row4_kernel (profiles/r05_row4_phases.log).
// Here only this wave's LDS (and scalar) operations are waited for; global loads stay in flight, global stores are not ordered.
__device__ __forceinline__ void kivi_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exp() of the softmax kernels: one v_exp_f32 on x * log2(e).  The argument product rounds at 2^-24 relative, i.e. the
// result carries a relative error <= ~|x| * 1e-7 (x <= 0 here, |x| < 100 where the result matters) -- three orders
// below the fp16 rounding of the probabilities, and a tenth of the instructions of the libm expf.  Every softmax in
// the library (stand-alone, row launches, sV prologue) uses this helper and multiplies by one correctly rounded
// reciprocal of the row sum, so the same row gives the same probabilities on every path.
__device__ __forceinline__ float kivi_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// One attention score -> the fp16 value the reference feeds its softmax (llama_kivi.py:339, :364-372):
// fp16(s * inv_scale) [then fp16(+ mask) clamped at the fp16 minimum].
__device__ __forceinline__ uint16_t kivi_scaled_score(uint16_t s, float inv_scale, bool has_mask, uint16_t m) {
    uint16_t h = __builtin_bit_cast(uint16_t, (_Float16)((float)__builtin_bit_cast(_Float16, s) * inv_scale));
    if (has_mask) {
        h = __builtin_bit_cast(uint16_t, (_Float16)((float)__builtin_bit_cast(_Float16, h) + (float)__builtin_bit_cast(_Float16, m)));
        if ((float)__builtin_bit_cast(_Float16, h) < -65504.0f) h = 0xFBFFu;
    }
    return h;
}

// ---- device helpers -------------------------------------------------------
template <typename T, bool NT>
__device__ __forceinline__ T ld_stream(const T* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// Buffer (SRSRC) loads: wave-uniform 128-bit descriptor + scalar row offset +
// 32-bit per-lane offset; out-of-range lanes read 0 (hardware bounds check).
// The descriptor inputs must be provably wave-uniform (blockIdx / readfirstlane).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    // readfirstlane makes the uniformity provable: otherwise hipcc wraps every
    // buffer op in a waterfall loop (cdna_hip_programming.md T20).
    const uint64_t p = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    bytes = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, bytes, 0x00020000);
}
template <typename T, bool NT>
__device__ __forceinline__ T buf_load(rsrc_t r, uint32_t voff, uint32_t soff) {
    constexpr int AUX = NT ? 2 : 0;  // aux bit 1 = nt (streamed once)
    if constexpr (sizeof(T) == 2) return (T)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, AUX);
    else if constexpr (sizeof(T) == 4) return (T)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX);
    else if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
    else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}

__device__ __forceinline__ float h2f_bits(uint16_t h) { return (float)__builtin_bit_cast(f16, h); }
__device__ __forceinline__ uint16_t f2h_bits(float f) { return __builtin_bit_cast(uint16_t, (f16)f); }

// acc + f32(half in the low / high 16 bits of m) * b, one instruction.  The
// half operand is an fp16 SUBNORMAL holding a masked code (value
// code * 4^k * 2^-24); v_fma_mix_f32 converts it exactly and accumulates in fp32.
__device__ __forceinline__ float fma_mix_lo(uint32_t m, float b, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(m), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(uint32_t m, float b, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(m), "v"(b), "v"(c));
    return d;
}

// f32(half `qhi` of the SGPR pair-word `q`) * f32(half `vhi` of v) [+ c]: exact fp16 x fp16 product
// (22 significand bits) in one VOP3P instruction, no separate conversions.  `qhi` / `vhi` must fold to
// constants at the call site (they select the asm string).
__device__ __forceinline__ float mul_hh_s(uint32_t q, bool qhi, uint32_t v, bool vhi) {
    float d;
    if (!qhi && !vhi) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,1,0]" : "=v"(d) : "s"(q), "v"(v));
    else if (qhi && !vhi) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "s"(q), "v"(v));
    else if (!qhi && vhi) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "s"(q), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "s"(q), "v"(v));
    return d;
}
__device__ __forceinline__ float fma_hh_s(uint32_t q, bool qhi, uint32_t v, bool vhi, float c) {
    float d;
    if (!qhi && !vhi) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(d) : "s"(q), "v"(v), "v"(c));
    else if (qhi && !vhi) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "s"(q), "v"(v), "v"(c));
    else if (!qhi && vhi) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "s"(q), "v"(v), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "s"(q), "v"(v), "v"(c));
    return d;
}
// same with a per-lane (VGPR) half operand
__device__ __forceinline__ float mul_hh_vv(uint32_t a_lo, uint32_t v, bool hi) {
    float d;
    if (hi) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a_lo), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a_lo), "v"(v));
    return d;
}
__device__ __forceinline__ float fma_hh_vv(uint32_t a_lo, uint32_t v, float c, bool hi) {
    float d;
    if (hi) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a_lo), "v"(v), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a_lo), "v"(v), "v"(c));
    return d;
}

// Unpack strategies for the fused GEMV inner loop (see DESIGN.md "VALU budget").
enum : int {
    KIVI_UNPACK_BFE = 0,    // v_bfe_u32 + v_cvt_f32_u32 + v_fma_f32            (3 op / code)
    KIVI_UNPACK_UBYTE = 1,  // byte-plane mask + v_cvt_f32_ubyteN + v_fma_f32    (2.25 op / code)
    KIVI_UNPACK_MIX = 2,    // half-plane mask + v_fma_mix_f32 on fp16 subnormals (1.56 op / code)
    KIVI_UNPACK_NONE = 3,   // DIAGNOSTIC ONLY (wrong results): one xor per word, shows the memory-side ceiling
    KIVI_UNPACK_DEN32 = 4,  // in-place mask read as an fp32 SUBNORMAL + v_fmac_f32 (two 2-cycle ops / code)
};

// Accumulate one packed word `w` (FPI = 32/BITS codes) into acc[FPI]:
//   acc[p] += code_p * POSTINV[p] * qs          (POSTINV = power of two, undone by post_scale)
// qs must carry qs_factor<MODE>().
template <int BITS, int MODE>
__device__ __forceinline__ void accum_word(uint32_t w, float qs, float* acc) {
    constexpr int FPI = 32 / BITS;
    if constexpr (MODE == KIVI_UNPACK_NONE) {
        acc[0] = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, acc[0]) ^ w) & 0x3FFFFFFFu) + qs;
    } else if constexpr (MODE == KIVI_UNPACK_BFE) {
#pragma unroll
        for (int p = 0; p < FPI; p++) {
            float c = (float)((w >> (BITS * p)) & ((1u << BITS) - 1u));
            acc[p] = __builtin_fmaf(c, qs, acc[p]);
        }
    } else if constexpr (MODE == KIVI_UNPACK_UBYTE) {
        constexpr int PER_BYTE = 8 / BITS;                 // codes per byte: 4 (2-bit) or 2 (4-bit)
        constexpr uint32_t M0 = (BITS == 2) ? 0x03030303u : 0x0F0F0F0Fu;
#pragma unroll
        for (int k = 0; k < PER_BYTE; k++) {
            uint32_t mb = w & (M0 << (BITS * k));
#pragma unroll
            for (int b = 0; b < 4; b++) {
                float c = (float)((mb >> (8 * b)) & 0xFFu);  // v_cvt_f32_ubyte{b}: code * 2^(BITS*k)
                acc[b * PER_BYTE + k] = __builtin_fmaf(c, qs, acc[b * PER_BYTE + k]);
            }
        }
    } else if constexpr (MODE == KIVI_UNPACK_DEN32) {
        // fp32 mantissa = bits 0..22: 11 two-bit (5 four-bit) fields readable in place as subnormals
        constexpr int INPLACE = (BITS == 2) ? 11 : 5;
        constexpr int SHIFT = (BITS == 2) ? 10 : 12;
        constexpr uint32_t M0 = (1u << BITS) - 1u;
#pragma unroll
        for (int p = 0; p < INPLACE; p++) {
            const float c = __builtin_bit_cast(float, w & (M0 << (BITS * p)));
            acc[p] = __builtin_fmaf(c, qs, acc[p]);
        }
        const uint32_t ws = w >> SHIFT;
#pragma unroll
        for (int p = INPLACE; p < FPI; p++) {
            const float c = __builtin_bit_cast(float, ws & (M0 << (BITS * p - SHIFT)));
            acc[p] = __builtin_fmaf(c, qs, acc[p]);
        }
    } else {
        // fp16 mantissa = bits 0..9 of each half: 5 two-bit fields (or 2 four-bit fields) are
        // readable in place; the remaining fields of each half are brought down by ONE shift.
        constexpr int HALF = FPI / 2;                      // codes per 16-bit half: 8 or 4
        constexpr int INPLACE = (BITS == 2) ? 5 : 2;       // fields that sit inside the mantissa
        constexpr int SHIFT = (BITS == 2) ? 6 : 8;         // brings field INPLACE.. down to >= bit 4 / 0
        constexpr int K0 = (BITS == 2) ? 2 : 0;            // first field index after the shift
        constexpr uint32_t M0 = (BITS == 2) ? 0x00030003u : 0x000F000Fu;
#pragma unroll
        for (int k = 0; k < INPLACE; k++) {
            uint32_t m = w & (M0 << (BITS * k));
            acc[k] = fma_mix_lo(m, qs, acc[k]);
            acc[k + HALF] = fma_mix_hi(m, qs, acc[k + HALF]);
        }
        uint32_t ws = w >> SHIFT;
#pragma unroll
        for (int k = INPLACE; k < HALF; k++) {
            uint32_t m = ws & (M0 << (BITS * (k - INPLACE + K0)));
            acc[k] = fma_mix_lo(m, qs, acc[k]);
            acc[k + HALF] = fma_mix_hi(m, qs, acc[k + HALF]);
        }
    }
}

// Factor that turns the accumulated value of code position p back into
// sum(code * qs_true):  acc[p] * post_scale(p).
template <int BITS, int MODE>
__device__ __forceinline__ constexpr float post_scale(int p) {
    constexpr int FPI = 32 / BITS;
    if constexpr (MODE == KIVI_UNPACK_BFE || MODE == KIVI_UNPACK_NONE) {
        return 1.0f;
    } else if constexpr (MODE == KIVI_UNPACK_UBYTE) {
        constexpr int PER_BYTE = 8 / BITS;
        int k = p % PER_BYTE;
        return 1.0f / (float)(1u << (BITS * k));
    } else if constexpr (MODE == KIVI_UNPACK_DEN32) {
        // value read = code * 2^(bit position) * 2^-149, qs carries 2^100 (qs_factor): undo 2^-49 * 2^pos
        constexpr int INPLACE = (BITS == 2) ? 11 : 5;
        constexpr int SHIFT = (BITS == 2) ? 10 : 12;
        const int pos = (p < INPLACE) ? BITS * p : BITS * p - SHIFT;
        return 562949953421312.0f / (float)(1u << pos);  // 2^49 / 2^pos
    } else {
        // value read = code * 2^(bit position in the half) * 2^-24 (fp16 subnormal): undo both
        constexpr int HALF = FPI / 2;
        constexpr int INPLACE = (BITS == 2) ? 5 : 2;
        constexpr int K0 = (BITS == 2) ? 2 : 0;
        int k = p % HALF;
        int field = (k < INPLACE) ? k : (k - INPLACE + K0);
        return 16777216.0f / (float)(1u << (BITS * field));
    }
}

// qs pre-factor.  MIX needs none (products of an fp16 subnormal and qs stay normal in fp32; the 2^24 is
// undone in post_scale).  DEN32 reads fp32 subnormals (x 2^-149): qs carries 2^100 so products stay normal.
template <int MODE>
__device__ __forceinline__ constexpr float qs_factor() {
    return MODE == KIVI_UNPACK_DEN32 ? 1.2676506002282294e30f : 1.0f;
}

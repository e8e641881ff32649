// Group-wise asymmetric quantiser shared by the pack kernels and the fused decode kernels.
// Reference arithmetic, op for op (quant/new_pack.py:236-241); see kivi_pack.hip for the derivation.
#pragma once
#include "kivi_common.h"

// Order-preserving fp16 -> u16 key, with -0 < +0 (what the oracle's h_lt does).
__device__ __forceinline__ uint32_t h_key(uint32_t h) { return (h & 0x8000u) ? (h ^ 0xFFFFu) : (h | 0x8000u); }
__device__ __forceinline__ uint32_t h_unkey(uint32_t k) { return (k & 0x8000u) ? (k ^ 0x8000u) : (k ^ 0xFFFFu); }

struct GroupQ {
    float fmn, fs, fmaxq;
    uint16_t mn, scale;
};

__device__ __forceinline__ GroupQ make_group(uint32_t kmin, uint32_t kmax, int maxq) {
    GroupQ g;
    g.mn = (uint16_t)h_unkey(kmin);
    const uint16_t mx = (uint16_t)h_unkey(kmax);
    g.fmn = h2f_bits(g.mn);
    const uint16_t range = f2h_bits(h2f_bits(mx) - g.fmn);        // new_pack.py:238 (mx - mn)
    g.scale = f2h_bits(h2f_bits(range) / (float)maxq);            //                 / max_int
    g.fs = h2f_bits(g.scale);
    g.fmaxq = (float)maxq;
    return g;
}

__device__ __forceinline__ uint32_t quant_one(uint16_t x, const GroupQ& g) {
    const uint16_t d = f2h_bits(h2f_bits(x) - g.fmn);             // new_pack.py:239
    const uint16_t q = f2h_bits(h2f_bits(d) / g.fs);              // :240, correctly rounded division
    float fq = h2f_bits(q);
    fq = __builtin_fmaxf(fq, 0.0f);                               // NaN -> 0 (fmax drops the NaN)
    fq = __builtin_fminf(fq, g.fmaxq);                            // :241 clamp_
    return (uint32_t)__builtin_rintf(fq);                         //      round_ (half to even), to(int32)
}


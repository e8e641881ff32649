// Group-wise asymmetric quantiser shared by the pack kernels and the fused decode kernels.
// Reference arithmetic, op for op (quant/new_pack.py:236-241); see kivi_pack.hip for the derivation.
#pragma once
#include "kivi_common.h"

// Order-preserving fp16 -> u16 key, with -0 < +0 (what the oracle's h_lt does).
__device__ __forceinline__ uint32_t h_key(uint32_t h) { return (h & 0x8000u) ? (h ^ 0xFFFFu) : (h | 0x8000u); }
__device__ __forceinline__ uint32_t h_unkey(uint32_t k) { return (k & 0x8000u) ? (k ^ 0x8000u) : (k ^ 0xFFFFu); }

struct GroupQ {
    float fmn, fs, fmaxq;
    float rfs;     // fp32(1 / scale), IEEE: the 4- / 8-bit quantiser multiplies by it instead of dividing per element
    float th[3];   // 2-bit decision thresholds tau_k * scale (exact fp32 products)
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
    g.rfs = 1.0f / g.fs;                                          // inf for scale 0, 0 for scale inf (used by 4 / 8 bits only)
    g.fmaxq = (float)maxq;
    // 2-bit fast path (quant_one<2>): code = #{k : d > tau_k * scale} with
    //   tau_0 = 0.5 + 2^-12, tau_1 = 1.5 - 2^-11 (>=), tau_2 = 2.5 + 2^-10
    // = the fp16 rounding boundaries of d/scale around k + 0.5 combined with round-half-even of the quotient.
    // tau (12 bits) x scale (11 bits) is exact in fp32, so the comparison is exact and the result is identical to
    // rint(clamp(fp16(d / scale))): checked against the division for every scale and every d within 3 ulps of a
    // boundary (and 4e7 random pairs) -- tests/test_oracle_golden.py::test_threshold_quantiser_equals_division.
    // Degenerate scales, as the reference's CUDA path resolves them (float -> int of NaN = 0):
    //   scale inf / NaN (range overflow, NaN input): x / inf = 0, inf / inf = NaN -> every code 0: NaN thresholds make
    //     every comparison false;
    //   scale 0: a constant group (d = 0: 0 / 0 = NaN -> 0) OR a group whose range is ONE fp16-subnormal ulp (2^-24 / 3
    //     rounds to 0): d > 0 there gives d / 0 = inf -> clamp -> 3.  Thresholds below the smallest positive fp16 do
    //     both: d = 0 stays under them, any d > 0 passes all three.
    const float fsx = (g.fs > 0.0f && g.fs < __builtin_inff()) ? g.fs : (g.fs == 0.0f ? 0x1p-30f : __builtin_nanf(""));
    g.th[0] = 0.500244140625f * fsx;
    g.th[1] = 1.49951171875f * fsx;
    g.th[2] = 2.5009765625f * fsx;
    return g;
}

template <int BITS>
__device__ __forceinline__ uint32_t quant_one(uint16_t x, const GroupQ& g) {
    const uint16_t d = f2h_bits(h2f_bits(x) - g.fmn);             // new_pack.py:239
    if constexpr (BITS == 2) {
        const float fd = h2f_bits(d);
        return (uint32_t)(fd > g.th[0]) + (uint32_t)(fd >= g.th[1]) + (uint32_t)(fd > g.th[2]);
    }
    // :240 through the reciprocal of the group: fp16(d * fp32(1 / s)) is not always fp16(d / s) (1 495 of the 10^9 fp16 pairs
    // differ by an ulp) but rint(clamp(.)) of the two never differs, for maxq 3 / 15 / 255 -- exhaustive CPU check,
    // tests/test_oracle_golden.py::test_reciprocal_quantiser_codes_are_exact; d / 0 = d * inf, 0 / 0 = 0 * inf = NaN,
    // x / inf = x * 0, inf / inf = inf * 0 = NaN: the degenerate scales resolve as with the division
    const uint16_t q = f2h_bits(h2f_bits(d) * g.rfs);
    float fq = h2f_bits(q);
    fq = __builtin_fmaxf(fq, 0.0f);                               // NaN -> 0 (fmax drops the NaN)
    fq = __builtin_fminf(fq, g.fmaxq);                            // :241 clamp_
    return (uint32_t)__builtin_rintf(fq);                         //      round_ (half to even), to(int32)
}

// ---- packed 16-bit forms (round 2): two fp16 values per register through the same arithmetic (kivi_pack.hip has the derivation)
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short us16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hf2 __attribute__((ext_vector_type(2)));

// logical shift right by 15 of both halves (count in a register for both halves: see the key computation below)
__device__ __forceinline__ uint32_t pk_lshr15(uint32_t v) {
    uint32_t r;
    asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "s"(0x000F000Fu), "v"(v));
    return r;
}
// One group of the 2-bit packed-math kernel: mn / scale as make_group does (kivi_quant.h), the three decision thresholds as
// fp16 bit patterns.  Two shortcuts, both checked exhaustively on the CPU (tests/test_oracle_golden.py):
//   * scale = fp16(range / 3) = fp16(range * fp32(1/3)): range / 3 is never within 2^-13 (relative) of an fp16 rounding
//     boundary, the product is within 2^-23 of the quotient;
//   * tau_k * scale (a 12-bit odd factor times an 11-bit mantissa) is never an fp16 value, so "d > th" and "d >= th"
//     are both  bits(d) > bits(RTZ(th))  for d >= +0: v_cvt_pkrtz_f16_f32 makes two thresholds per instruction.
// min / max of three packed fp16 pairs, IEEE 754-2019 minimum / maximum: a NaN in any operand gives NaN (torch.min / max
// propagate it: new_pack.py:236-237), -0 < +0.  gfx950 instructions; two new elements per instruction where the
// order-preserving integer key costs five per element pair.
__device__ __forceinline__ uint32_t pk_min3_f16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_max3_f16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

struct Group2 {
    uint32_t t02, t11;   // (T0 | T2 << 16), (T1 | T1 << 16)
    uint16_t mn, scale;
    bool live;           // false: scale inf / NaN -> every code 0
};
__device__ __forceinline__ Group2 make_group2_bits(uint16_t mn, uint16_t mx);
__device__ __forceinline__ Group2 make_group2(uint32_t kmin, uint32_t kmax) {
    return make_group2_bits((uint16_t)h_unkey(kmin), (uint16_t)h_unkey(kmax));
}
// the same from the fp16 bit patterns of the group's minimum and maximum
__device__ __forceinline__ Group2 make_group2_bits(uint16_t mn, uint16_t mx) {
    Group2 g;
    g.mn = mn;
    const uint16_t range = f2h_bits(h2f_bits(mx) - h2f_bits(g.mn));          // new_pack.py:238 (mx - mn)
    g.scale = f2h_bits(h2f_bits(range) * 0.3333333432674408f);               //                 / max_int
    const float fs0 = h2f_bits(g.scale);
    g.live = fs0 >= 0.0f && fs0 < __builtin_inff();                          // inf / NaN: every code 0
    const float fs = fs0 == 0.0f ? 0x1p-30f : fs0;                           // scale 0: d > 0 -> d / 0 = inf -> 3 (kivi_quant.h)
    g.t02 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(0.500244140625f * fs, 2.5009765625f * fs));
    g.t11 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(1.49951171875f * fs, 1.49951171875f * fs));
    return g;
}


// 2-bit codes of both halves from the thresholds T0 <= T1 <= T2 of their groups (dt = T2 - T0): [bits(d) > T1] picks the
// second threshold to compare with (T2 or T0) -- the same three comparisons as [d > T0] + [d > T1] + [d > T2], evaluated as
// a tree: 6 packed instructions per pair instead of 8.  "bits(d) > T" is the sign of T - bits(d) (both < 0x8000).
__device__ __forceinline__ uint32_t pk_code2(us16x2 db, us16x2 t0, us16x2 t1, us16x2 dt) {
    const uint32_t b1 = pk_lshr15(__builtin_bit_cast(uint32_t, t1 - db));
    const us16x2 ts = __builtin_bit_cast(us16x2, b1) * dt + t0;                   // v_pk_mad_u16: b1 ? T2 : T0
    const uint32_t b0 = pk_lshr15(__builtin_bit_cast(uint32_t, ts - db));
    return (b1 << 1) + b0;                                                        // v_lshl_add_u32 (no carry between the halves)
}

// min / max of the two channels (or rows) a lane holds in the halves of x[0..N): (mn lo | mn hi << 16), (mx lo | mx hi << 16)
template <int N>
__device__ __forceinline__ void pk16_pair_minmax(const uint32_t* x, uint32_t& mnb, uint32_t& mxb) {
    static_assert(N >= 2 && N % 2 == 0, "an even number of elements");
    mnb = pk_min3_f16(x[0], x[1], x[1]);
    mxb = pk_max3_f16(x[0], x[1], x[1]);
#pragma unroll
    for (int t = 2; t < N; t += 2) {
        mnb = pk_min3_f16(mnb, x[t], x[t + 1]);
        mxb = pk_max3_f16(mxb, x[t], x[t + 1]);
    }
}

// The 2-bit codes of the TWO channels (or rows) a lane holds in the halves of x[0..N): min / max, groups and threshold
// compares on both halves at once.  cq[t] = code of the low half | code of the high half << 16; scale2 / mn2 likewise.
template <int N>
__device__ __forceinline__ void pk16_pair_quant2(const uint32_t* x, uint32_t* cq, uint32_t& scale2, uint32_t& mn2) {
    uint32_t mnb, mxb;
    pk16_pair_minmax<N>(x, mnb, mxb);
    const Group2 g0 = make_group2_bits((uint16_t)(mnb & 0xFFFFu), (uint16_t)(mxb & 0xFFFFu)),
                 g1 = make_group2_bits((uint16_t)(mnb >> 16), (uint16_t)(mxb >> 16));
    scale2 = (uint32_t)g0.scale | ((uint32_t)g1.scale << 16);
    mn2 = (uint32_t)g0.mn | ((uint32_t)g1.mn << 16);
    const uint32_t live = (g0.live ? 0x00000003u : 0u) | (g1.live ? 0x00030000u : 0u);   // scale inf / NaN: code 0
    const us16x2 t0 = __builtin_bit_cast(us16x2, (g0.t02 & 0xFFFFu) | (g1.t02 << 16));
    const us16x2 t2 = __builtin_bit_cast(us16x2, (g0.t02 >> 16) | (g1.t02 & 0xFFFF0000u));
    const us16x2 t1 = __builtin_bit_cast(us16x2, (g0.t11 & 0xFFFFu) | (g1.t11 & 0xFFFF0000u));
    const hf2 mnv = __builtin_bit_cast(hf2, mn2);
    const us16x2 dt = t2 - t0;
#pragma unroll
    for (int t = 0; t < N; t++) {
        const uint32_t xt = x[t];
        const us16x2 db = __builtin_bit_cast(us16x2, __builtin_bit_cast(hf2, xt) - mnv);
        cq[t] = pk_code2(db, t0, t1, dt) & live;
    }
}

// The 4- / 8-bit codes of the two channels (or rows) a lane holds in the halves of x[0..N): the reciprocal quantiser of
// quant_pack_lastdimN_kernel / quant_pack_k_tmajor_tiled (kivi_pack.hip: scale = fp16(range * fp32(1 / maxq)), code =
// rint(clamp(fp16(d * fp32(1 / scale)))) through the 1024 magic add -- equal to the reference's divisions for every fp16 pair, checked
// exhaustively on the CPU) on both halves at once.  cq[t] = code of the low half | code of the high half << 16.
template <int N, int BITS>
__device__ __forceinline__ void pk16_pair_quantN(const uint32_t* x, uint32_t* cq, uint32_t& scale2, uint32_t& mn2) {
    static_assert(BITS == 4 || BITS == 8, "2 bits: pk16_pair_quant2");
    constexpr int MAXQ = (1 << BITS) - 1;
    uint32_t mnb, mxb;
    pk16_pair_minmax<N>(x, mnb, mxb);
    const uint16_t mn0 = (uint16_t)(mnb & 0xFFFFu), mn1 = (uint16_t)(mnb >> 16);
    const uint16_t r0 = f2h_bits(h2f_bits((uint16_t)(mxb & 0xFFFFu)) - h2f_bits(mn0));       // new_pack.py:238 (mx - mn)
    const uint16_t r1 = f2h_bits(h2f_bits((uint16_t)(mxb >> 16)) - h2f_bits(mn1));
    const uint16_t sc0 = f2h_bits(h2f_bits(r0) * (1.0f / (float)MAXQ)), sc1 = f2h_bits(h2f_bits(r1) * (1.0f / (float)MAXQ));
    scale2 = (uint32_t)sc0 | ((uint32_t)sc1 << 16);
    mn2 = (uint32_t)mn0 | ((uint32_t)mn1 << 16);
    const float rc0 = 1.0f / h2f_bits(sc0), rc1 = 1.0f / h2f_bits(sc1);                      // IEEE; inf for scale 0, 0 for scale inf
    const hf2 mnv = __builtin_bit_cast(hf2, mn2);
    const hf2 zero2 = {(_Float16)0.0f, (_Float16)0.0f}, maxq2 = {(_Float16)(float)MAXQ, (_Float16)(float)MAXQ};
    const hf2 magic = {(_Float16)1024.0f, (_Float16)1024.0f};
#pragma unroll
    for (int t = 0; t < N; t++) {
        const uint32_t xt = x[t];
        const hf2 d = __builtin_bit_cast(hf2, xt) - mnv;                                      // :239
        hf2 q;
        q[0] = (_Float16)((float)d[0] * rc0);                                                 // :240 through the reciprocal
        q[1] = (_Float16)((float)d[1] * rc1);
        const hf2 cl = __builtin_elementwise_min(__builtin_elementwise_max(q, zero2), maxq2);  // :241 clamp_ (NaN -> 0)
        cq[t] = __builtin_bit_cast(uint32_t, cl + magic) & (BITS == 4 ? 0x000F000Fu : 0x00FF00FFu);   // round_ + to(int32)
    }
}


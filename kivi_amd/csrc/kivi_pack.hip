// Fused group-wise quantise + pack, unpack + dequantise (gfx950).
//
// Replaces triton_quantize_and_pack_along_last_dim (quant/new_pack.py:217-252:
// _minmax_along_last_dim + five torch elementwise kernels + _pack_along_last_dim
// and an int32 temporary 2x the input) with ONE pass: 2 B read, bits/8 + 4/g B
// written per element.  Arithmetic is the reference's, op for op, so the
// outputs are bit-identical:
//   scale = fp16(fp16(mx - mn) / (2^bits - 1))
//   code  = rint_half_even(clamp(fp16(fp16(x - mn) / scale), 0, 2^bits - 1))
// every op evaluated in fp32 on fp16 operands and rounded to fp16 (innocuous
// double rounding), with a correctly rounded fp32 division
// (-fhip-fp32-correctly-rounded-divide-sqrt; never x * rcp(scale)).
// A constant group has scale 0 -> 0/0 = NaN -> code 0 (the reference's CUDA
// float->int conversion; its CPU run yields INT_MIN instead);
// a group whose range is one fp16-subnormal ulp also has scale 0 and its
// non-minimum elements quantise to d/0 = inf -> max code (kivi_quant.h).
#include <stdlib.h>

#include "kivi_common.h"
#include "kivi_quant.h"

namespace {

// One lane = 8 consecutive elements (16 B); LPG = g/8 lanes share a group.  A thread takes NU chunks 256 apart (every
// wave-instruction still reads 1 KiB contiguous) and requests all of them before it touches the first: with one load
// per thread the kernel ran at half of what the memory system gives (4.0 TB/s algorithmic; `NU` loads in flight per lane).
template <int BITS, int NU>
__global__ __launch_bounds__(256) void quant_pack_lastdim_kernel(const uint16_t* __restrict__ x,
                                                                 uint32_t* __restrict__ code,
                                                                 uint16_t* __restrict__ scale,
                                                                 uint16_t* __restrict__ mn, int64_t nchunk,
                                                                 int lpg) {
    const int64_t c0 = (int64_t)blockIdx.x * (256 * NU) + threadIdx.x;
    u16x8 vv[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int64_t c = c0 + 256 * u;
        vv[u] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (c < nchunk) vv[u] = __builtin_nontemporal_load((const u16x8*)(x + c * 8));
    }
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int64_t c = c0 + 256 * u;
        const bool valid = c < nchunk;  // whole groups fall out together (nchunk % lpg == 0, lpg | 64)
        const u16x8 v = vv[u];
        uint32_t kmin = 0xFFFFu, kmax = 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t k = h_key(v[i]);
            kmin = k < kmin ? k : kmin;
            kmax = k > kmax ? k : kmax;
        }
        for (int m = 1; m < lpg; m <<= 1) {
            const uint32_t omin = __shfl_xor(kmin, m), omax = __shfl_xor(kmax, m);
            kmin = omin < kmin ? omin : kmin;
            kmax = omax > kmax ? omax : kmax;
        }
        const GroupQ g = make_group(kmin, kmax, (1 << BITS) - 1);
        uint32_t q[8];
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = quant_one<BITS>(v[i], g);
        if constexpr (BITS == 2) {
            uint32_t part = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) part |= q[i] << (2 * i);
            const uint32_t other = __shfl_xor(part, 1);
            if (valid && !(threadIdx.x & 1)) code[c >> 1] = part | (other << 16);
        } else if constexpr (BITS == 4) {
            uint32_t w = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) w |= q[i] << (4 * i);
            if (valid) code[c] = w;
        } else {
            u32x2 w;
            w[0] = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
            w[1] = q[4] | (q[5] << 8) | (q[6] << 16) | (q[7] << 24);
            if (valid) *(u32x2*)(code + c * 2) = w;
        }
        if (valid && (c % lpg) == 0) {
            scale[c / lpg] = g.scale;
            mn[c / lpg] = g.mn;
        }
    }
}

// 2-bit variant on packed 16-bit math (round 2).  The kernel above spends ~190 VALU instructions per 16-byte chunk and is
// bound by them (332 us per GiB in), so this one keeps the same arithmetic but does it two halves at a time:
//   * order-preserving keys, min / max: v_pk_ashrrev_i16 + xor, v_pk_min_u16 / v_pk_max_u16; the lanes of a group meet
//     through DPP (no LDS round trips) while the group has <= 16 lanes;
//   * d = fp16(x - mn): one v_pk_add_f16 per pair (IEEE, fp16 subnormals on) = the reference's fp32 subtract + round
//     (24 >= 2 * 11 + 2 bits: the double rounding is innocuous);
//   * the three decisions d > th0, d >= th1, d > th2 (kivi_quant.h) compare a NON-NEGATIVE fp16 with fp32 thresholds:
//     they are equal to integer comparisons of the bit patterns with the thresholds rounded toward zero to fp16 once per
//     group (make_group2), i.e. sign(T - bits(d)) by v_pk_sub_i16 + v_pk_lshrrev_b16; groups with scale inf or NaN
//     get code 0 by a mask.
// Bit-exact against the same fixtures as the kernel above (tests/test_pack_gpu.py).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// Front end shared by the packed-math kernels: min / max of the 8 halves of a chunk, then the lpg lanes of a group (aligned,
// power of two) meet through DPP inside a 16-lane row and shuffles beyond.  All in the fp16 domain with the NaN-propagating
// three-operand minimum / maximum of gfx950 (kivi_quant.h): the maximum travels NEGATED in the high half next to the minimum
// in the low half, so one packed minimum per step reduces both (-max = min of the negated values; negation is exact and keeps
// -0 < +0 in order).  Returns the fp16 bit patterns.
__device__ __forceinline__ void pk16_group_minmax(const u32x4& v, int lpg, uint32_t& gmn, uint32_t& gmx) {
    const uint32_t v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    const uint32_t a = pk_min3_f16(pk_min3_f16(v0, v1, v2), v3, v3), b = pk_max3_f16(pk_max3_f16(v0, v1, v2), v3, v3);
    // (min lo | -max lo << 16) and (min hi | -max hi << 16)
    const uint32_t c = __builtin_amdgcn_perm(b, a, 0x05040100u) ^ 0x80000000u, d = __builtin_amdgcn_perm(b, a, 0x07060302u) ^ 0x80000000u;
    uint32_t r = pk_min3_f16(c, d, d);
    if (lpg > 1) { const uint32_t o = dpp_u<0xB1>(r); r = pk_min3_f16(r, o, o); }
    if (lpg > 2) { const uint32_t o = dpp_u<0x4E>(r); r = pk_min3_f16(r, o, o); }
    if (lpg > 4) { const uint32_t o = dpp_u<0x141>(r); r = pk_min3_f16(r, o, o); }
    if (lpg > 8) { const uint32_t o = dpp_u<0x140>(r); r = pk_min3_f16(r, o, o); }
    for (int m = 16; m < lpg; m <<= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)r, m);
        r = pk_min3_f16(r, o, o);
    }
    gmn = r & 0xFFFFu;
    gmx = (r >> 16) ^ 0x8000u;
}

// LPG = lanes per group when known at compile time (4 / 8 / 16 for group 32 / 64 / 128), 0 = the runtime value;
// FULL = every block has all of its 256 * NU chunks (no bounds checks).
template <int NU, int LPG, bool FULL>
__global__ __launch_bounds__(256) void quant_pack_lastdim2_kernel(const uint16_t* __restrict__ x, uint32_t* __restrict__ code,
                                                                  uint16_t* __restrict__ scale, uint16_t* __restrict__ mn,
                                                                  int64_t nchunk, int lpg_rt) {
    const int64_t c0 = (int64_t)blockIdx.x * (256 * NU) + threadIdx.x;
    const int lpg = LPG ? LPG : lpg_rt;
    const int lg = LPG ? __builtin_ctz((unsigned)(LPG ? LPG : 1)) : __builtin_ctz((unsigned)lpg_rt);
    u32x4 vv[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int64_t c = c0 + 256 * u;
        vv[u] = u32x4{0, 0, 0, 0};
        if (FULL || c < nchunk) vv[u] = __builtin_nontemporal_load((const u32x4*)(x + c * 8));
    }
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int64_t c = c0 + 256 * u;
        const bool valid = FULL || c < nchunk;
        const u32x4 v = vv[u];
        uint32_t gmn_b, gmx_b;
        pk16_group_minmax(v, lpg, gmn_b, gmx_b);
        const Group2 g = make_group2_bits((uint16_t)gmn_b, (uint16_t)gmx_b);
        const us16x2 t02 = __builtin_bit_cast(us16x2, g.t02);
        const us16x2 t0 = {t02[0], t02[0]}, t1 = __builtin_bit_cast(us16x2, g.t11), t2 = {t02[1], t02[1]};
        const _Float16 hmn = __builtin_bit_cast(_Float16, g.mn);
        const hf2 mnv = {hmn, hmn};
        uint32_t cq[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t xk = v[k];   // by value: __builtin_bit_cast of a vector ELEMENT reads element 0 (hipcc 7.2)
            const hf2 d = __builtin_bit_cast(hf2, xk) - mnv;                              // new_pack.py:239
            const us16x2 db = __builtin_bit_cast(us16x2, d);
            // 1 where bits(d) > T: the sign of T - bits(d) (both < 0x8000), per half
            cq[k] = pk_code2(db, t0, t1, t2 - t0);
        }
        // lo halves: codes 0, 2, 4, 6; hi halves: codes 1, 3, 5, 7
        const uint32_t t = cq[0] | (cq[1] << 4) | (cq[2] << 8) | (cq[3] << 12);
        // dead groups (scale inf / NaN): code 0 whatever d is -- d itself may then be a NaN with the sign bit set
        // (-inf - -inf), which the sign trick above would count
        const uint32_t part = (t | (t >> 14)) & (g.live ? 0xFFFFu : 0u);
        const uint32_t other = dpp_u<0xB1>(part);                                          // lane ^ 1
        if (valid && !(threadIdx.x & 1)) code[c >> 1] = part | (other << 16);
        if (valid && (c & (lpg - 1)) == 0) {                        // lpg is a power of two
            const int64_t gi = c >> lg;
            scale[gi] = g.scale;
            mn[gi] = g.mn;
        }
    }
}

// 4- and 8-bit on packed 16-bit math (round 2): the front end of the kernel above (keys, min / max, DPP group reduce), then
//   q    = fp16(d * r),  r = fp32(1 / scale) once per chunk (IEEE division), instead of one IEEE division per element;
//   code = low bits of  fp16(min(max(q, 0), maxq) + 1024)   -- the fp16 add rounds to nearest even at ulp 1 = rint().
// q itself is NOT always the reference's fp16(d / scale) (1 495 of the 10^9 (d, scale) pairs differ by an ulp with the
// product rounded twice, 17 069 with v_fma_mixlo_f16's single rounding) but the CODE is, for every fp16 d >= 0, every
// positive fp16 scale and maxq in {3, 15, 255}: checked exhaustively on the CPU for both roundings
// (tests/test_oracle_golden.py::test_reciprocal_quantiser_codes_are_exact).  Degenerate scales need no special case:
// scale 0 -> r = inf: d > 0 -> inf -> maxq, d = 0 -> NaN -> 0; scale inf -> r = 0: 0, inf * 0 = NaN -> 0 (v_pk_max_f16
// returns the non-NaN operand) -- the reference's CUDA results (kivi_quant.h).
template <int BITS, int NU, int LPG, bool FULL>
__global__ __launch_bounds__(256) void quant_pack_lastdimN_kernel(const uint16_t* __restrict__ x, uint32_t* __restrict__ code,
                                                                  uint16_t* __restrict__ scale, uint16_t* __restrict__ mn,
                                                                  int64_t nchunk, int lpg_rt) {
    static_assert(BITS == 4 || BITS == 8, "2 bits: quant_pack_lastdim2_kernel");
    constexpr int MAXQ = (1 << BITS) - 1;
    const int64_t c0 = (int64_t)blockIdx.x * (256 * NU) + threadIdx.x;
    const int lpg = LPG ? LPG : lpg_rt;
    const int lg = LPG ? __builtin_ctz((unsigned)(LPG ? LPG : 1)) : __builtin_ctz((unsigned)lpg_rt);
    u32x4 vv[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int64_t c = c0 + 256 * u;
        vv[u] = u32x4{0, 0, 0, 0};
        if (FULL || c < nchunk) vv[u] = __builtin_nontemporal_load((const u32x4*)(x + c * 8));
    }
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int64_t c = c0 + 256 * u;
        const bool valid = FULL || c < nchunk;
        const u32x4 v = vv[u];
        uint32_t gmn_b, gmx_b;
        pk16_group_minmax(v, lpg, gmn_b, gmx_b);
        const uint16_t gmn = (uint16_t)gmn_b, gmx = (uint16_t)gmx_b;
        const uint16_t range = f2h_bits(h2f_bits(gmx) - h2f_bits(gmn));                       // new_pack.py:238 (mx - mn)
        const uint16_t gscale = f2h_bits(h2f_bits(range) * (1.0f / (float)MAXQ));             //   / max_int (equal to the division for every fp16 range)
        const float r = 1.0f / h2f_bits(gscale);                                              // IEEE; inf for scale 0, 0 for scale inf
        const _Float16 hmn = __builtin_bit_cast(_Float16, gmn);
        const hf2 mnv = {hmn, hmn};
        const hf2 zero2 = {(_Float16)0.0f, (_Float16)0.0f}, maxq2 = {(_Float16)(float)MAXQ, (_Float16)(float)MAXQ};
        const hf2 magic = {(_Float16)1024.0f, (_Float16)1024.0f};
        uint32_t cb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t xk = v[k];   // by value (see quant_pack_lastdim2_kernel)
            const hf2 d = __builtin_bit_cast(hf2, xk) - mnv;                                  // new_pack.py:239
            hf2 q;
            q[0] = (_Float16)((float)d[0] * r);                                               // :240 through the reciprocal
            q[1] = (_Float16)((float)d[1] * r);
            const hf2 cl = __builtin_elementwise_min(__builtin_elementwise_max(q, zero2), maxq2);   // :241 clamp_ (NaN -> 0)
            cb[k] = __builtin_bit_cast(uint32_t, cl + magic) & (BITS == 4 ? 0x000F000Fu : 0x00FF00FFu);   // round_ + to(int32)
        }
        if constexpr (BITS == 4) {
            // element 2k + h -> nibble 2k + h: byte k = lo | hi << 4
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) w |= ((cb[k] | (cb[k] >> 12)) & 0xFFu) << (8 * k);
            if (valid) code[c] = w;
        } else {
            u32x2 w;
            w[0] = ((cb[0] | (cb[0] >> 8)) & 0xFFFFu) | (((cb[1] | (cb[1] >> 8)) & 0xFFFFu) << 16);
            w[1] = ((cb[2] | (cb[2] >> 8)) & 0xFFFFu) | (((cb[3] | (cb[3] >> 8)) & 0xFFFFu) << 16);
            if (valid) *(u32x2*)(code + c * 2) = w;
        }
        if (valid && (c & (lpg - 1)) == 0) {
            const int64_t gi = c >> lg;
            scale[gi] = gscale;
            mn[gi] = gmn;
        }
    }
}

// Any group size (multiple of fpi): one thread per group, scalar loops.
template <int BITS>
__global__ __launch_bounds__(256) void quant_pack_lastdim_generic(const uint16_t* x, uint32_t* code, uint16_t* scale,
                                                                  uint16_t* mn, int64_t ngroups, int g) {
    constexpr int FPI = 32 / BITS;
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    const uint16_t* xp = x + gi * g;
    uint32_t kmin = 0xFFFFu, kmax = 0u;
    for (int i = 0; i < g; i++) {
        const uint32_t k = h_key(xp[i]);
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
    }
    const GroupQ gq = make_group(kmin, kmax, (1 << BITS) - 1);
    for (int w = 0; w < g / FPI; w++) {
        uint32_t word = 0;
        for (int i = 0; i < FPI; i++) word |= quant_one<BITS>(xp[w * FPI + i], gq) << (BITS * i);
        code[gi * (g / FPI) + w] = word;
    }
    scale[gi] = gq.scale;
    mn[gi] = gq.mn;
}

// Per-channel K straight from the un-transposed tensor: a lane owns two adjacent
// channels (one 4-byte load per token, 256 B per wave-instruction) and the G tokens
// of one group; codes of its channels are packed along t in registers.
template <int BITS, int G>
__global__ __launch_bounds__(64) void quant_pack_k_tmajor_kernel(const uint16_t* __restrict__ k, int64_t k_sb,
                                                                 int64_t k_sh, int64_t k_st,
                                                                 uint32_t* __restrict__ code, int64_t code_sb,
                                                                 int64_t code_sh, int64_t code_sr, int64_t code_off,
                                                                 uint16_t* __restrict__ scale,
                                                                 uint16_t* __restrict__ mn, int64_t sm_sb,
                                                                 int64_t sm_sh, int64_t sm_sr, int64_t sm_off, int nh,
                                                                 int D, int64_t ngroups) {
    constexpr int FPI = 32 / BITS;
    constexpr int NW = G / FPI;
    const int dpair = blockIdx.y * 64 + threadIdx.x;  // channels 2*dpair, 2*dpair+1
    const int64_t gi = blockIdx.x % ngroups;
    const int bh = (int)(blockIdx.x / ngroups);
    const int b = bh / nh, h = bh - b * nh;
    if (2 * dpair >= D) return;
    const uint16_t* kp = k + b * k_sb + h * k_sh + gi * G * k_st + 2 * dpair;
    uint32_t v[G];
#pragma unroll
    for (int t = 0; t < G; t++) v[t] = __builtin_nontemporal_load((const uint32_t*)(kp + t * k_st));
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
        uint32_t kmin = 0xFFFFu, kmax = 0u;
#pragma unroll
        for (int t = 0; t < G; t++) {
            const uint32_t kk = h_key((v[t] >> (16 * ch)) & 0xFFFFu);
            kmin = kk < kmin ? kk : kmin;
            kmax = kk > kmax ? kk : kmax;
        }
        const GroupQ gq = make_group(kmin, kmax, (1 << BITS) - 1);
        const int d = 2 * dpair + ch;
        uint32_t* cp = code + b * code_sb + h * code_sh + (int64_t)d * code_sr + code_off + gi * NW;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            uint32_t word = 0;
#pragma unroll
            for (int i = 0; i < FPI; i++)
                word |= quant_one<BITS>((uint16_t)((v[w * FPI + i] >> (16 * ch)) & 0xFFFFu), gq) << (BITS * i);
            cp[w] = word;
        }
        const int64_t so = b * sm_sb + h * sm_sh + (int64_t)d * sm_sr + sm_off + gi;
        scale[so] = gq.scale;
        mn[so] = gq.mn;
    }
}

// Same arithmetic, tiled for the store side: a 256-thread block takes NG = 16 consecutive groups (512 tokens at
// g = 32) of one (b, h); each wave quantises 4 of them exactly like the kernel above, but the packed words and the
// scale / mn go through an LDS tile so that every channel row is written as one contiguous 128-byte (codes) and
// 32-byte (scale, mn) segment instead of 64 scattered 8-byte stores.  D <= 128 per block column (blockIdx.y).
template <int BITS, int G, bool PK16 = true>
__global__ __launch_bounds__(256) void quant_pack_k_tmajor_tiled(const uint16_t* __restrict__ k, int64_t k_sb, int64_t k_sh,
                                                                 int64_t k_st, uint32_t* __restrict__ code,
                                                                 int64_t code_sb, int64_t code_sh, int64_t code_sr,
                                                                 int64_t code_off, uint16_t* __restrict__ scale,
                                                                 uint16_t* __restrict__ mn, int64_t sm_sb, int64_t sm_sh,
                                                                 int64_t sm_sr, int64_t sm_off, int nh, int D,
                                                                 int64_t ngroups, int64_t nchunks) {
    constexpr int FPI = 32 / BITS;
    constexpr int NW = G / FPI;          // words per group per channel
    constexpr int NG = 16;               // groups per block
    constexpr int CP = NG * NW + 1;      // LDS row pitch (words), +1 against bank conflicts
    __shared__ uint32_t codeL[128 * CP];
    __shared__ uint16_t scaleL[128 * NG], mnL[128 * NG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t chunk = blockIdx.x % nchunks;
    const int bh = (int)(blockIdx.x / nchunks);
    const int b = bh / nh, h = bh - b * nh;
    const int d0 = blockIdx.y * 128;     // first channel of this block column
    const int dl = 2 * lane;             // this lane's channel pair inside the column
    const int64_t g_first = chunk * NG;
    const int ng_here = (int)((ngroups - g_first) < NG ? (ngroups - g_first) : NG);
    if (d0 + dl < D) {
        for (int gl = wave; gl < ng_here; gl += 4) {
            const uint16_t* kp = k + b * k_sb + h * k_sh + (g_first + gl) * G * k_st + d0 + dl;
            uint32_t v[G];
#pragma unroll
            for (int t = 0; t < G; t++) v[t] = __builtin_nontemporal_load((const uint32_t*)(kp + t * k_st));
            if constexpr (PK16) {
                // packed 16-bit math (round 2): the lane's two channels are the two halves of every register, so min / max,
                // d = x - mn, the threshold compares (2 bits) or the reciprocal quantiser (4 / 8 bits) and the word assembly
                // run on both channels at once and nothing crosses lanes; the arithmetic of quant_pack_lastdim2_kernel /
                // quant_pack_lastdimN_kernel (bit-exact, see there)
                constexpr int TPR = 16 / BITS;           // tokens per 16-bit half of an accumulation register
                uint32_t W[G / TPR];
#pragma unroll
                for (int i = 0; i < G / TPR; i++) W[i] = 0u;
                uint32_t scale2, mn2;
                if constexpr (BITS == 2) {
                    uint32_t cq[G];
                    pk16_pair_quant2<G>(v, cq, scale2, mn2);
#pragma unroll
                    for (int t = 0; t < G; t++) W[t / TPR] |= cq[t] << (BITS * (t % TPR));
                } else {
                    constexpr int MAXQ = (1 << BITS) - 1;
                    uint32_t mnb, mxb;
                    pk16_pair_minmax<G>(v, mnb, mxb);
                    const uint16_t mn0 = (uint16_t)(mnb & 0xFFFFu), mn1 = (uint16_t)(mnb >> 16);
                    const uint16_t r0 = f2h_bits(h2f_bits((uint16_t)(mxb & 0xFFFFu)) - h2f_bits(mn0));
                    const uint16_t r1 = f2h_bits(h2f_bits((uint16_t)(mxb >> 16)) - h2f_bits(mn1));
                    const uint16_t sc0 = f2h_bits(h2f_bits(r0) * (1.0f / (float)MAXQ)), sc1 = f2h_bits(h2f_bits(r1) * (1.0f / (float)MAXQ));
                    scale2 = (uint32_t)sc0 | ((uint32_t)sc1 << 16);
                    mn2 = (uint32_t)mn0 | ((uint32_t)mn1 << 16);
                    const float rc0 = 1.0f / h2f_bits(sc0), rc1 = 1.0f / h2f_bits(sc1);
                    const hf2 mnv = __builtin_bit_cast(hf2, mn2);
                    const hf2 zero2 = {(_Float16)0.0f, (_Float16)0.0f}, maxq2 = {(_Float16)(float)MAXQ, (_Float16)(float)MAXQ};
                    const hf2 magic = {(_Float16)1024.0f, (_Float16)1024.0f};
#pragma unroll
                    for (int t = 0; t < G; t++) {
                        const uint32_t xt = v[t];
                        const hf2 d = __builtin_bit_cast(hf2, xt) - mnv;
                        hf2 q;
                        q[0] = (_Float16)((float)d[0] * rc0);
                        q[1] = (_Float16)((float)d[1] * rc1);
                        const hf2 cl = __builtin_elementwise_min(__builtin_elementwise_max(q, zero2), maxq2);
                        const uint32_t cb = __builtin_bit_cast(uint32_t, cl + magic) & (BITS == 4 ? 0x000F000Fu : 0x00FF00FFu);
                        W[t / TPR] |= cb << (BITS * (t % TPR));
                    }
                }
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    const uint32_t a = W[2 * w], bq = W[2 * w + 1];
                    codeL[dl * CP + gl * NW + w] = (a & 0xFFFFu) | (bq << 16);
                    codeL[(dl + 1) * CP + gl * NW + w] = (a >> 16) | (bq & 0xFFFF0000u);
                }
                scaleL[dl * NG + gl] = (uint16_t)scale2; scaleL[(dl + 1) * NG + gl] = (uint16_t)(scale2 >> 16);
                mnL[dl * NG + gl] = (uint16_t)mn2; mnL[(dl + 1) * NG + gl] = (uint16_t)(mn2 >> 16);
            } else {
#pragma unroll
            for (int ch = 0; ch < 2; ch++) {
                uint32_t kmin = 0xFFFFu, kmax = 0u;
#pragma unroll
                for (int t = 0; t < G; t++) {
                    const uint32_t kk = h_key((v[t] >> (16 * ch)) & 0xFFFFu);
                    kmin = kk < kmin ? kk : kmin;
                    kmax = kk > kmax ? kk : kmax;
                }
                const GroupQ gq = make_group(kmin, kmax, (1 << BITS) - 1);
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    uint32_t word = 0;
#pragma unroll
                    for (int i = 0; i < FPI; i++)
                        word |= quant_one<BITS>((uint16_t)((v[w * FPI + i] >> (16 * ch)) & 0xFFFFu), gq) << (BITS * i);
                    codeL[(dl + ch) * CP + gl * NW + w] = word;
                }
                scaleL[(dl + ch) * NG + gl] = gq.scale;
                mnL[(dl + ch) * NG + gl] = gq.mn;
            }
            }
        }
    }
    __syncthreads();
    // cooperative row stores: two threads per channel row, each a contiguous half
    const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
    if (d0 + row < D) {
        const int nwords = ng_here * NW;
        uint32_t* cp = code + b * code_sb + h * code_sh + (int64_t)(d0 + row) * code_sr + code_off + g_first * NW;
        const int w0 = half * (NG * NW / 2);
        for (int w = w0; w < w0 + NG * NW / 2 && w < nwords; w++) cp[w] = codeL[row * CP + w];
        const int64_t so = b * sm_sb + h * sm_sh + (int64_t)(d0 + row) * sm_sr + sm_off + g_first;
        const int s0 = half * (NG / 2);
        for (int gq_ = s0; gq_ < s0 + NG / 2 && gq_ < ng_here; gq_++) {
            scale[so + gq_] = scaleL[row * NG + gq_];
            mn[so + gq_] = mnL[row * NG + gq_];
        }
    }
}

// Any group size / odd D: one thread per (channel, group), strided scalar loads.
template <int BITS>
__global__ __launch_bounds__(64) void quant_pack_k_tmajor_generic(const uint16_t* k, int64_t k_sb, int64_t k_sh,
                                                                  int64_t k_st, uint32_t* code, int64_t code_sb,
                                                                  int64_t code_sh, int64_t code_sr, int64_t code_off,
                                                                  uint16_t* scale, uint16_t* mn, int64_t sm_sb,
                                                                  int64_t sm_sh, int64_t sm_sr, int64_t sm_off, int nh,
                                                                  int D, int64_t ngroups, int g) {
    constexpr int FPI = 32 / BITS;
    const int d = blockIdx.y * 64 + threadIdx.x;
    const int64_t gi = blockIdx.x % ngroups;
    const int bh = (int)(blockIdx.x / ngroups);
    const int b = bh / nh, h = bh - b * nh;
    if (d >= D) return;
    const uint16_t* kp = k + b * k_sb + h * k_sh + gi * g * k_st + d;
    uint32_t kmin = 0xFFFFu, kmax = 0u;
    for (int t = 0; t < g; t++) {
        const uint32_t kk = h_key(kp[t * k_st]);
        kmin = kk < kmin ? kk : kmin;
        kmax = kk > kmax ? kk : kmax;
    }
    const GroupQ gq = make_group(kmin, kmax, (1 << BITS) - 1);
    uint32_t* cp = code + b * code_sb + h * code_sh + (int64_t)d * code_sr + code_off + gi * (g / FPI);
    for (int w = 0; w < g / FPI; w++) {
        uint32_t word = 0;
        for (int i = 0; i < FPI; i++) word |= quant_one<BITS>(kp[(int64_t)(w * FPI + i) * k_st], gq) << (BITS * i);
        cp[w] = word;
    }
    const int64_t so = b * sm_sb + h * sm_sh + (int64_t)d * sm_sr + sm_off + gi;
    scale[so] = gq.scale;
    mn[so] = gq.mn;
}

// out = fp16(fp16(fp16(q) * scale) + mn): one thread per packed word.
template <int BITS>
__global__ __launch_bounds__(256) void unpack_dequant_lastdim_kernel(const uint32_t* __restrict__ code,
                                                                     const uint16_t* __restrict__ scale,
                                                                     const uint16_t* __restrict__ mn,
                                                                     uint16_t* __restrict__ out, int64_t nwords,
                                                                     int64_t Tw, int64_t ng, int g) {
    constexpr int FPI = 32 / BITS;
    const int64_t wi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= nwords) return;
    const int64_t row = wi / Tw, w = wi - row * Tw;
    const int64_t gi = row * ng + (w * FPI) / g;
    const float fs = h2f_bits(scale[gi]), fm = h2f_bits(mn[gi]);
    const uint32_t word = code[wi];
    uint16_t o[FPI];
#pragma unroll
    for (int i = 0; i < FPI; i++) {
        const float q = (float)((word >> (BITS * i)) & ((1u << BITS) - 1u));  // exact in fp16
        const uint16_t p = f2h_bits(q * fs);                                 // new_pack.py:82  data * scale
        o[i] = f2h_bits(h2f_bits(p) + fm);                                   //                 + mn
    }
    uint16_t* op = out + wi * FPI;
#pragma unroll
    for (int i = 0; i < FPI; i += 4) {
        typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
        u16x4 t = {o[i], o[i + 1], o[i + 2], o[i + 3]};
        *(u16x4*)(op + i) = t;
    }
}

template <int BITS>
__global__ __launch_bounds__(256) void unpack_codes_lastdim_kernel(const uint32_t* __restrict__ code,
                                                                   int16_t* __restrict__ out, int64_t nwords) {
    constexpr int FPI = 32 / BITS;
    const int64_t wi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= nwords) return;
    const uint32_t word = code[wi];
#pragma unroll
    for (int i = 0; i < FPI; i++) out[wi * FPI + i] = (int16_t)((word >> (BITS * i)) & ((1u << BITS) - 1u));
}

// pack_tensor along the last dim (quant/new_pack.py:86-107, Triton twin :132-154):
// word = OR_i data[j*fpi + i] << (bits*i) with int32 wrap-around, no masking (as the reference).
template <int BITS>
__global__ __launch_bounds__(256) void pack_codes_lastdim_kernel(const int32_t* __restrict__ data,
                                                                 uint32_t* __restrict__ code, int64_t nwords) {
    constexpr int FPI = 32 / BITS;
    const int64_t wi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= nwords) return;
    const int32_t* dp = data + wi * FPI;
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < FPI; i += 4) {
        const u32x4 v = *(const u32x4*)(dp + i);
#pragma unroll
        for (int e = 0; e < 4; e++) word |= v[e] << (BITS * (i + e));
    }
    code[wi] = word;
}

bool bits_ok_pack(int bits) { return bits == 2 || bits == 4 || bits == 8; }

}  // namespace

extern "C" int kivi_quant_pack_lastdim(const void* x, void* code, void* scale, void* mn, int64_t rows, int64_t T,
                                       int group_size, int bits, kivi_stream_t stream) {
    KIVI_REQUIRE(bits_ok_pack(bits), KIVI_EINVAL, "kivi_quant_pack_lastdim: bits must be 2, 4 or 8 (new_pack.py:90), got %d",
                 bits);
    const int fpi = 32 / bits;
    KIVI_REQUIRE(group_size > 0 && T % group_size == 0, KIVI_EINVAL,
                 "kivi_quant_pack_lastdim: T=%lld not a multiple of group_size=%d (new_pack.py:222)", (long long)T,
                 group_size);
    KIVI_REQUIRE(T % fpi == 0 && group_size % fpi == 0, KIVI_EINVAL,
                 "kivi_quant_pack_lastdim: T and group_size must be multiples of %d codes per word", fpi);
    KIVI_REQUIRE(rows >= 0, KIVI_EINVAL, "kivi_quant_pack_lastdim: negative rows");
    const int64_t n = rows * T;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int lpg = group_size / 8;
    const bool fast = (group_size % 8 == 0) && lpg <= 64 && (lpg & (lpg - 1)) == 0 && ((uintptr_t)x % 16 == 0) &&
                      ((uintptr_t)code % 8 == 0);
    if (fast) {
        const int64_t nchunk = n / 8;
        static const char* fu = KIVI_TUNE_ENV("KIVI_PACK_UNROLL");   // tuning aid: chunks per thread (1, 2, 4 or 8)
        const int nu = fu ? atoi(fu) : (nchunk >= (int64_t)1 << 20 ? 8 : 1);
#define KIVI_QP(BB)                                                                                                    \
    do {                                                                                                               \
        if (nu == 8) hipLaunchKernelGGL((quant_pack_lastdim_kernel<BB, 8>), dim3((unsigned)((nchunk + 2047) / 2048)), dim3(256), 0, s, \
                                        (const uint16_t*)x, (uint32_t*)code, (uint16_t*)scale, (uint16_t*)mn, nchunk, lpg);          \
        else if (nu == 4) hipLaunchKernelGGL((quant_pack_lastdim_kernel<BB, 4>), dim3((unsigned)((nchunk + 1023) / 1024)), dim3(256), 0, s, \
                                             (const uint16_t*)x, (uint32_t*)code, (uint16_t*)scale, (uint16_t*)mn, nchunk, lpg);     \
        else if (nu == 2) hipLaunchKernelGGL((quant_pack_lastdim_kernel<BB, 2>), dim3((unsigned)((nchunk + 511) / 512)), dim3(256), 0, s, \
                                             (const uint16_t*)x, (uint32_t*)code, (uint16_t*)scale, (uint16_t*)mn, nchunk, lpg);     \
        else hipLaunchKernelGGL((quant_pack_lastdim_kernel<BB, 1>), dim3((unsigned)((nchunk + 255) / 256)), dim3(256), 0, s,         \
                                (const uint16_t*)x, (uint32_t*)code, (uint16_t*)scale, (uint16_t*)mn, nchunk, lpg);                  \
    } while (0)
        static const char* nopk = KIVI_TUNE_ENV("KIVI_PACK_NO_PK16");   // tuning aid: the scalar-math kernels
        if (bits == 2 && !nopk) {
#define KIVI_QP2(NUU, LL, FF)                                                                                          \
    hipLaunchKernelGGL((quant_pack_lastdim2_kernel<NUU, LL, FF>), dim3((unsigned)((nchunk + 256 * NUU - 1) / (256 * NUU))), dim3(256), 0, \
                       s, (const uint16_t*)x, (uint32_t*)code, (uint16_t*)scale, (uint16_t*)mn, nchunk, lpg)
            const int nu2 = nu >= 4 ? 4 : 1;
            const bool full = nchunk % (256 * nu2) == 0;
            if (nu2 == 4 && full && lpg == 4) KIVI_QP2(4, 4, true);
            else if (nu2 == 4 && full && lpg == 8) KIVI_QP2(4, 8, true);
            else if (nu2 == 4 && full && lpg == 16) KIVI_QP2(4, 16, true);
            else if (nu2 == 4) KIVI_QP2(4, 0, false);
            else if (lpg == 4) KIVI_QP2(1, 4, false);
            else KIVI_QP2(1, 0, false);
#undef KIVI_QP2
        } else if (bits == 2) KIVI_QP(2);
        else if (!nopk) {
#define KIVI_QPN(BB, NUU, LL, FF)                                                                                      \
    hipLaunchKernelGGL((quant_pack_lastdimN_kernel<BB, NUU, LL, FF>), dim3((unsigned)((nchunk + 256 * NUU - 1) / (256 * NUU))), dim3(256), \
                       0, s, (const uint16_t*)x, (uint32_t*)code, (uint16_t*)scale, (uint16_t*)mn, nchunk, lpg)
            const int nu2 = nu >= 4 ? 4 : 1;
            const bool full = nchunk % (256 * nu2) == 0;
            if (bits == 4) {
                if (nu2 == 4 && full && lpg == 4) KIVI_QPN(4, 4, 4, true);
                else if (nu2 == 4 && full && lpg == 8) KIVI_QPN(4, 4, 8, true);
                else if (nu2 == 4 && full && lpg == 16) KIVI_QPN(4, 4, 16, true);
                else if (nu2 == 4) KIVI_QPN(4, 4, 0, false);
                else KIVI_QPN(4, 1, 0, false);
            } else {
                if (nu2 == 4 && full && lpg == 4) KIVI_QPN(8, 4, 4, true);
                else if (nu2 == 4) KIVI_QPN(8, 4, 0, false);
                else KIVI_QPN(8, 1, 0, false);
            }
#undef KIVI_QPN
        } else if (bits == 4) KIVI_QP(4);
        else KIVI_QP(8);
#undef KIVI_QP
        return kivi_launch_status("quant_pack_lastdim");
    }
    const int64_t ngroups = n / group_size;
    dim3 grid((unsigned)((ngroups + 255) / 256));
    if (bits == 2)
        hipLaunchKernelGGL(quant_pack_lastdim_generic<2>, grid, dim3(256), 0, s, (const uint16_t*)x, (uint32_t*)code,
                           (uint16_t*)scale, (uint16_t*)mn, ngroups, group_size);
    else if (bits == 4)
        hipLaunchKernelGGL(quant_pack_lastdim_generic<4>, grid, dim3(256), 0, s, (const uint16_t*)x, (uint32_t*)code,
                           (uint16_t*)scale, (uint16_t*)mn, ngroups, group_size);
    else
        hipLaunchKernelGGL(quant_pack_lastdim_generic<8>, grid, dim3(256), 0, s, (const uint16_t*)x, (uint32_t*)code,
                           (uint16_t*)scale, (uint16_t*)mn, ngroups, group_size);
    return kivi_launch_status("quant_pack_lastdim_generic");
}

#define KIVI_K_ARGS                                                                                                   \
    (const uint16_t*)k, k_sb, k_sh, k_st, (uint32_t*)code, code_sb, code_sh, code_sr, code_off, (uint16_t*)scale,     \
        (uint16_t*)mn, sm_sb, sm_sh, sm_sr, sm_off, nh, D, ngroups

extern "C" int kivi_quant_pack_k_tmajor(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_st, void* code,
                                        int64_t code_sb, int64_t code_sh, int64_t code_sr, int64_t code_off,
                                        void* scale, void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr,
                                        int64_t sm_off, int B, int nh, int64_t T, int D, int group_size, int bits,
                                        kivi_stream_t stream) {
    KIVI_REQUIRE(bits_ok_pack(bits), KIVI_EINVAL, "kivi_quant_pack_k_tmajor: bits must be 2, 4 or 8, got %d", bits);
    const int fpi = 32 / bits;
    KIVI_REQUIRE(group_size > 0 && group_size % fpi == 0 && T % group_size == 0, KIVI_EINVAL,
                 "kivi_quant_pack_k_tmajor: T=%lld must be a multiple of group_size=%d (new_pack.py:13), group_size of %d",
                 (long long)T, group_size, fpi);
    KIVI_REQUIRE(B > 0 && nh > 0 && D > 0 && T >= 0, KIVI_EINVAL, "kivi_quant_pack_k_tmajor: bad shape");
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int64_t ngroups = T / group_size;
    const int64_t nblk = ngroups * B * nh;
    KIVI_REQUIRE(nblk < ((int64_t)1 << 31), KIVI_EINVAL, "kivi_quant_pack_k_tmajor: grid too large");
    const bool fast = (D % 2 == 0) && (k_sb % 2 == 0) && (k_sh % 2 == 0) && (k_st % 2 == 0) && ((uintptr_t)k % 4 == 0) &&
                      (group_size == 32 || group_size == 64 || group_size == 128) && bits != 8;
    if (fast && ngroups >= 8) {   // store-coalescing tile kernel (prefill-sized calls)
        const int64_t nchunks = (ngroups + 15) / 16;
        dim3 grid((unsigned)(nchunks * B * nh), (unsigned)((D + 127) / 128));
        static const char* nopk_k = KIVI_TUNE_ENV("KIVI_PACK_NO_PK16");   // tuning aid: the scalar-math instantiation
#define KIVI_KT_CASE(BITS, G)                                                                                   \
    if (bits == BITS && group_size == G) {                                                                      \
        if (nopk_k) hipLaunchKernelGGL((quant_pack_k_tmajor_tiled<BITS, G, false>), grid, dim3(256), 0, s, KIVI_K_ARGS, nchunks); \
        else hipLaunchKernelGGL((quant_pack_k_tmajor_tiled<BITS, G, true>), grid, dim3(256), 0, s, KIVI_K_ARGS, nchunks);         \
        return kivi_launch_status("quant_pack_k_tmajor_tiled");                                                 \
    }
        KIVI_KT_CASE(2, 32) KIVI_KT_CASE(2, 64) KIVI_KT_CASE(4, 32) KIVI_KT_CASE(4, 64)
#undef KIVI_KT_CASE
    }
    if (fast) {
        dim3 grid((unsigned)nblk, (unsigned)((D / 2 + 63) / 64));
#define KIVI_K_CASE(BITS, G)                                                                            \
    if (bits == BITS && group_size == G) {                                                              \
        hipLaunchKernelGGL((quant_pack_k_tmajor_kernel<BITS, G>), grid, dim3(64), 0, s, KIVI_K_ARGS);   \
        return kivi_launch_status("quant_pack_k_tmajor");                                               \
    }
        KIVI_K_CASE(2, 32) KIVI_K_CASE(2, 64) KIVI_K_CASE(2, 128) KIVI_K_CASE(4, 32) KIVI_K_CASE(4, 64) KIVI_K_CASE(4, 128)
#undef KIVI_K_CASE
    }
    dim3 grid((unsigned)nblk, (unsigned)((D + 63) / 64));
    if (bits == 2)
        hipLaunchKernelGGL(quant_pack_k_tmajor_generic<2>, grid, dim3(64), 0, s, KIVI_K_ARGS, group_size);
    else if (bits == 4)
        hipLaunchKernelGGL(quant_pack_k_tmajor_generic<4>, grid, dim3(64), 0, s, KIVI_K_ARGS, group_size);
    else
        hipLaunchKernelGGL(quant_pack_k_tmajor_generic<8>, grid, dim3(64), 0, s, KIVI_K_ARGS, group_size);
    return kivi_launch_status("quant_pack_k_tmajor_generic");
}

extern "C" int kivi_unpack_dequant_lastdim(const void* code, const void* scale, const void* mn, void* out, int64_t rows,
                                           int64_t T, int group_size, int bits, kivi_stream_t stream) {
    KIVI_REQUIRE(bits_ok_pack(bits), KIVI_EINVAL, "kivi_unpack_dequant_lastdim: bits must be 2, 4 or 8 (new_pack.py:75)");
    const int fpi = 32 / bits;
    KIVI_REQUIRE(group_size > 0 && group_size % fpi == 0 && T % group_size == 0, KIVI_EINVAL,
                 "kivi_unpack_dequant_lastdim: T=%lld must be a multiple of group_size=%d", (long long)T, group_size);
    KIVI_REQUIRE((uintptr_t)out % 8 == 0, KIVI_EALIGN, "kivi_unpack_dequant_lastdim: out must be 8-byte aligned");
    const int64_t nwords = rows * (T / fpi);
    if (nwords == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((nwords + 255) / 256));
    const int64_t Tw = T / fpi, ng = T / group_size;
    if (bits == 2)
        hipLaunchKernelGGL(unpack_dequant_lastdim_kernel<2>, grid, dim3(256), 0, s, (const uint32_t*)code,
                           (const uint16_t*)scale, (const uint16_t*)mn, (uint16_t*)out, nwords, Tw, ng, group_size);
    else if (bits == 4)
        hipLaunchKernelGGL(unpack_dequant_lastdim_kernel<4>, grid, dim3(256), 0, s, (const uint32_t*)code,
                           (const uint16_t*)scale, (const uint16_t*)mn, (uint16_t*)out, nwords, Tw, ng, group_size);
    else
        hipLaunchKernelGGL(unpack_dequant_lastdim_kernel<8>, grid, dim3(256), 0, s, (const uint32_t*)code,
                           (const uint16_t*)scale, (const uint16_t*)mn, (uint16_t*)out, nwords, Tw, ng, group_size);
    return kivi_launch_status("unpack_dequant_lastdim");
}

extern "C" int kivi_unpack_codes_lastdim(const void* code, void* out_i16, int64_t rows, int64_t T, int bits,
                                         kivi_stream_t stream) {
    KIVI_REQUIRE(bits_ok_pack(bits), KIVI_EINVAL, "kivi_unpack_codes_lastdim: bits must be 2, 4 or 8 (new_pack.py:113)");
    const int fpi = 32 / bits;
    KIVI_REQUIRE(T % fpi == 0, KIVI_EINVAL, "kivi_unpack_codes_lastdim: T must be a multiple of %d", fpi);
    const int64_t nwords = rows * (T / fpi);
    if (nwords == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((nwords + 255) / 256));
    if (bits == 2)
        hipLaunchKernelGGL(unpack_codes_lastdim_kernel<2>, grid, dim3(256), 0, s, (const uint32_t*)code, (int16_t*)out_i16,
                           nwords);
    else if (bits == 4)
        hipLaunchKernelGGL(unpack_codes_lastdim_kernel<4>, grid, dim3(256), 0, s, (const uint32_t*)code, (int16_t*)out_i16,
                           nwords);
    else
        hipLaunchKernelGGL(unpack_codes_lastdim_kernel<8>, grid, dim3(256), 0, s, (const uint32_t*)code, (int16_t*)out_i16,
                           nwords);
    return kivi_launch_status("unpack_codes_lastdim");
}

extern "C" int kivi_pack_codes_lastdim(const void* data_i32, void* code, int64_t rows, int64_t T, int bits,
                                       kivi_stream_t stream) {
    KIVI_REQUIRE(bits_ok_pack(bits), KIVI_EINVAL, "kivi_pack_codes_lastdim: only 2, 4, 8 bits are supported (new_pack.py:90)");
    const int fpi = 32 / bits;
    KIVI_REQUIRE(T % fpi == 0, KIVI_EINVAL,
                 "kivi_pack_codes_lastdim: dimension length must be divisible by %d features per int (new_pack.py:91)", fpi);
    KIVI_REQUIRE((uintptr_t)data_i32 % 16 == 0, KIVI_EALIGN, "kivi_pack_codes_lastdim: data must be 16-byte aligned");
    const int64_t nwords = rows * (T / fpi);
    if (nwords == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((nwords + 255) / 256));
    if (bits == 2)
        hipLaunchKernelGGL(pack_codes_lastdim_kernel<2>, grid, dim3(256), 0, s, (const int32_t*)data_i32, (uint32_t*)code,
                           nwords);
    else if (bits == 4)
        hipLaunchKernelGGL(pack_codes_lastdim_kernel<4>, grid, dim3(256), 0, s, (const int32_t*)data_i32, (uint32_t*)code,
                           nwords);
    else
        hipLaunchKernelGGL(pack_codes_lastdim_kernel<8>, grid, dim3(256), 0, s, (const int32_t*)data_i32, (uint32_t*)code,
                           nwords);
    return kivi_launch_status("pack_codes_lastdim");
}

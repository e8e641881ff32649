// Device bodies of the matrix-pipe kernels over the KT / VT cache layouts (kivi_mfma_layout.h): the packed qK^T
// and sV products of quant/csrc/gemv_cuda.cu:348-427 (call sites models/llama_kivi.py:324 / :382) for R = nh / nh_kv
// in {1, 4, 8} query heads per kv head (R = 8: round 4, rows = 2 groups x 8 heads / two row sets of (channel group, head)).
//
// What changed against the round-2 bodies (kivi_gqa.hip), per 32-token block of 128 channels (64 codes per lane):
//   * qK^T: an MFMA ROW is a quantisation GROUP (R = 1: the 16 groups of a super-block; R = 4: 4 groups x 4 heads), not
//     a (hi / lo, head) pair.  The A operand q * scale[., group] is then built ONCE per 16 / R groups by the one lane
//     that owns the row -- 2 packed ops per group instead of 32 -- and every group multiplies its own code words with it;
//     row g of the result is the group's scores, the other rows (codes of group g against the scales of other groups)
//     are never read.  hi and lo (the exact fp16 split of the 22-bit product) are two chained MFMAs.  The zero-point term
//     sum_d q * mn[d, g] comes out of 4 MFMAs per super-block in exactly the lanes / registers of the useful rows.
//   * sV: an MFMA row is a CHANNEL GROUP (R = 1: (channel group, hi | lo); R = 4: (channel group, head) with hi and lo
//     chained), the code sums are CENTRED: one block in RING also accumulates A x (-1.5 RING), so the running sums
//     through the C operand stay at the size of the output instead of growing to sum p * scale * code ~ 100x larger
//     (the matrix pipe aligns its 32 products to the largest addend and drops what falls ~2^-26 below it: with the
//     uncentred sums that loss, accumulated over 128 blocks, was 2e-3 of the output once code sum and zero-point sum
//     cancelled, which is why round 2 started every MFMA from zero and added 32 registers on the VALU per block).
//     What was subtracted and the zero-point sum p * mn are v_dot2_f32_f16 on the operands the lane holds anyway.
// VALU instructions per block: qK^T 106 -> ~50 (R = 1) / ~70 (R = 4), sV 115 -> ~62; matrix instructions 9-18 -> 16-24
// on a pipe that was under 10 % busy.
#pragma once
#include "kivi_gqa_dev.h"

namespace {

// fp16 constants in both halves
constexpr uint32_t MF_ONE2 = 0x3C003C00u;      // 1.0
constexpr uint32_t MF_M1 = 0x03000300u, MF_M2 = 0x00C000C0u;
// -1.5 in the units of a masked code (kivi_mfma_layout.h): registers 0, 1 hold code * 2^-16, registers 2, 3 code * 2^-18
constexpr uint32_t MF_C15A = 0x81808180u, MF_C15B = 0x80608060u;

__device__ __forceinline__ float dot2_f16(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// the two B operands (token tile 0 / 1 for K, channel tile 0 / 1 for V) of one code word: 3 views + 8 masks
struct MfB { h8 b0, b1; };
__device__ __forceinline__ MfB mf_views(uint32_t w) {
#if defined(KIVI_TUNING) && defined(KIVI_PROBE_NO_UNPACK)
    // bound probe (tools/build_variant.sh nounpack -DKIVI_TUNING -DKIVI_PROBE_NO_UNPACK; WRONG results): ONE instruction per code word
    // instead of 11 -- what a launch would cost if the unpack were free (profiles/r05_unpack_bound.log)
    const uint32_t y = w & 0x03FF03FFu;
    MfB p;
    p.b0 = as_h8(y, y, y, y);
    p.b1 = as_h8(y, y, y, y);
    return p;
#endif
    const uint32_t x1 = w << 4, x2 = w >> 4, x3 = __builtin_amdgcn_perm(w, w, 0x02030001u);
    MfB r;
    r.b0 = as_h8(w & MF_M1, x1 & MF_M1, w & MF_M2, x1 & MF_M2);
    r.b1 = as_h8(x2 & MF_M1, x3 & MF_M1, x2 & MF_M2, x3 & MF_M2);
    return r;
}

// The code words of one 32-token block a lane holds: 2 bits: one 16-byte load (every word feeds both tiles); 4 bits: two (tile 0,
// tile 1: kivi_mfma_layout.h) -- a word is then the B operand of ONE matrix instruction after 4 shifts + 4 masks.
template <int BITS> struct MfW { u32x4 w[BITS / 2]; };
__device__ __forceinline__ h8 mf_views4(uint32_t w) {
    constexpr uint32_t M = 0x03C003C0u;
    return as_h8((w << 6) & M, (w << 2) & M, (w >> 2) & M, (w >> 6) & M);
}
template <int BITS>
__device__ __forceinline__ MfB mf_views_c(const MfW<BITS>& x, int c) {
    if constexpr (BITS == 2) {
        return mf_views(x.w[0][c]);
    } else {
        MfB r;
        r.b0 = mf_views4(x.w[0][c]);
        r.b1 = mf_views4(x.w[1][c]);
        return r;
    }
}
// voff: per-lane byte offset of the lane's 16 bytes of tile 0 (or a dead offset), soff: the block's scalar byte offset
template <int BITS>
__device__ __forceinline__ void mf_load_block(MfW<BITS>& x, rsrc_t r, uint32_t voff, uint32_t soff) {
    x.w[0] = buf_load<u32x4, true>(r, voff, soff);
    if constexpr (BITS == 4) x.w[1] = buf_load<u32x4, true>(r, voff + 1024u, soff);
}

// ------------------------------------------------------------------------------------------------ q operand
// Lane (m, kb) of a wave, m = lane & 15: the query of head r = m % R, channels 32 c + 8 kb + 2 i (+ 1) as fp16 pairs,
// normalised to max |q| in [1, 2) (exponent sq) and placed by the unit's range shift `rsh` (mf_range_shift of the store's range
// word, kivi_mfma_layout.h; exponent sa = sq + rsh): 2^10 lower when the store holds a scale >= 256, so that A = q'' * scale *
// 2^(4 | 6) is a finite fp16 for EVERY finite scale, 2^8 higher when all its scales are < 2^-8, so that A keeps a normal hi part
// down to subnormal scales -- and pre-multiplied by 2^aexp(i).  The same registers are the A operand's q factor (row m) and, because a
// lane's A row and B column have the same index, the B operand of the zero-point product (column m -> head m % R): that
// product takes them back to [1, 2) first (mf_zfac), so the zero-point sums do not depend on the placement.
template <int R>
struct MfQ {
    uint32_t qq[4][4];
    int sq, sa;        // exponent of the zero-point operand, exponent of the A operand
};

// packed factor that takes a q'' register (times 2^aexp(i), placed at sa = sq + rsh) to q * 2^sq: 2^(-aexp(i) - rsh), a normal
// fp16 for rsh in {-10, 0, 8} (2^6 ... 2^-14)
template <int BITS = 2>
__device__ __forceinline__ uint32_t mf_zfac(int i, int rsh) {
    const uint32_t h = (uint32_t)(15 - aexp_b<BITS>(i) - rsh) << 10;
    return h | (h << 16);
}

template <int R, int BITS = 2>
__device__ __forceinline__ void mf_load_q(const uint16_t* q_h0, int64_t q_sh, MfQ<R>& Q, int rsh) {
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kb = lane >> 4;
    const uint16_t* qrow = q_h0 + (int64_t)(m % R) * q_sh + 8 * kb;
    u16x8 qv[4];
#pragma unroll
    for (int c = 0; c < 4; c++) qv[c] = *(const u16x8*)(qrow + 32 * c);
    uint32_t amax = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t v = qv[c][e] & 0x7FFFu;
            amax = v > amax ? v : amax;
        }
    amax = max(amax, (uint32_t)__shfl_xor((int)amax, 16));
    amax = max(amax, (uint32_t)__shfl_xor((int)amax, 32));
    const int ex = (int)(amax >> 10);                              // biased exponent of the row maximum (0: zero / subnormal)
    Q.sq = amax >= 0x7C00u ? 0 : 15 - (ex ? ex : 1);               // inf / nan rows: no scaling (they poison the row anyway)
    Q.sa = Q.sq + rsh;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float f0 = __builtin_ldexpf(h2f_bits(qv[c][2 * i]), Q.sa + aexp_b<BITS>(i));
            const float f1 = __builtin_ldexpf(h2f_bits(qv[c][2 * i + 1]), Q.sa + aexp_b<BITS>(i));
            Q.qq[c][i] = (uint32_t)f2h_bits(f0) | ((uint32_t)f2h_bits(f1) << 16);
        }
}

// ------------------------------------------------------------------------------------------------ qK^T
// ---- R = 1: rows = (8 groups) x (hi | lo).  Row m -> group (m & 7) of the current HALF super-block, m < 8: fp16(q'' scale),
// m >= 8: the exact remainder -- one MFMA per (channel chunk, token tile) gives hi and lo sums in two rows, 8 MFMAs per group.
// A wave walks a SEQUENCE of super-blocks (sb_first + i * sb_stride, i < n_sb; the last one may be partial) half by half:
//   * everything the first half needs is requested before the first wait (q, scale, zero points, RING code blocks): one
//     memory round trip, not three (the round-2 / first round-3 form paid q -> scale -> codes one after the other, per wave
//     and super-block: ~10 us of a 46 us launch, profiles/r03_mfk_ablation.log);
//   * the scale / zero points of half h + 1 are requested right after the A operands of half h have been built from the
//     registers they land in; the code ring runs across halves and super-blocks.
constexpr uint32_t MF_DEAD_OFF = 0xFFFE0000u;    // + any in-super-block offset (< 25 KiB) stays below 2^32; the host keeps every descriptor's extent <= this value (MF_DESC_LIMIT, kivi_gqa.hip), so such a request is out of range whatever the store's size

template <int V> struct mf_ic { static constexpr int value = V; };

struct MfKSeq {
    uint32_t sb_bytes;              // byte stride between consecutive super-blocks of the unit
    int sb_first, sb_stride, n_sb;  // this wave's super-blocks
    int ng_total;                   // 32-token groups in the whole sequence (counted from the first group of sb_first)
    int g_first = 0;                // mf_k_seqR only (round 6): the sequence STARTS at this group of sb_first -- a multiple of the ring --, so
                                    // that a row can be dealt to the waves of a block in contiguous runs of groups instead of whole super-blocks
};

// Scores go to sink(super-block index, token inside it, fp32 score).  RING = code blocks in flight (2 or 4).
// done(super-block index, its number of groups) is called when the last score of a super-block has been handed to sink.
// q_lds: 64 words of this wave's LDS (the normalised q operand is parked there: kept in registers, it and the loop-invariant
// B operand of the zero-point product hipcc derives from it hold 32 registers across the whole loop).
// rsh: the unit's range shift (wave-uniform; mf_load_q).
// BITS = 4 (round 6: multi-head KIVI-4, e.g. LongChat-7B / Llama-2-7B with 4-bit K / V): the same walk over the 4-bit super-blocks
// (kivi_mfma_layout.h, "KT4"): a ring slot is two 16-byte loads (token tile 0 / 1), the four registers of a B operand carry ONE
// exponent (aexp_b<4> = 6), a view is 2 instructions per register instead of 11 per 8.
template <int RING, int BITS = 2, typename Sink, typename Done>
__device__ __forceinline__ void mf_k_seq1(rsrc_t rk, const MfKSeq& W, const uint16_t* q_row, uint32_t* q_lds, int rsh, Sink&& sink, Done&& done) {
    static_assert(RING == 2 || RING == 4 || RING == 8, "ring of 2, 4 or 8 code blocks");
    typedef MfL<BITS> LY;
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kb = lane >> 4;
    const int gp = m & 7;
    const uint32_t lomask = (m & 8) ? 0xFFFFFFFFu : 0u;
    if (W.ng_total <= 0) return;
    // ---- requests: q, scale / zero points of half 0, the ring
    u16x8 qv[4];
#pragma unroll
    for (int c = 0; c < 4; c++) qv[c] = *(const u16x8*)(q_row + 8 * kb + 32 * c);
    const uint32_t row_off = (uint32_t)(kb * 128 + gp * 16);       // kt_sm_word4(gp, kb, 0) * 4: per chunk c 512 dense bytes per load
    auto sb_off = [&](int sbi) { return (uint32_t)(W.sb_first + sbi * W.sb_stride) * W.sb_bytes; };
    u32x4 sv[4], mv[4];
    // Requests past the end of the sequence keep the loop branch-free but must not cost traffic: their per-lane offset is
    // pushed past the descriptor's range (MF_DEAD_OFF: a buffer load out of range returns zeros without touching memory; the
    // scalar offset stays a valid one).  Repeating the last block instead cost 5 % of the launch's HBM reads.
    auto request_half = [&](int hq, bool live) {
        const uint32_t so = sb_off(hq >> 1) + (uint32_t)(hq & 1) * 2048u;      // half a super-block: 8 groups x 256 bytes
        const uint32_t ro = live ? row_off : MF_DEAD_OFF;
#pragma unroll
        for (int c = 0; c < 4; c++) sv[c] = buf_load<u32x4, true>(rk, LY::SCALE_WORD0 * 4 + ro + c * 512, so);
#pragma unroll
        for (int c = 0; c < 4; c++) mv[c] = buf_load<u32x4, true>(rk, LY::MN_WORD0 * 4 + ro + c * 512, so);
    };
    request_half(0, true);
    const int g_last = W.ng_total - 1;
    MfW<BITS> wr[RING];
    auto request_group = [&](int slot, int gi) {
        const bool live = gi <= g_last;
        const int gc = live ? gi : g_last;                          // (a valid scalar offset either way)
        mf_load_block<BITS>(wr[slot], rk, live ? (uint32_t)(lane * 16) : MF_DEAD_OFF, sb_off(gc >> 4) + (uint32_t)(gc & 15) * (uint32_t)(LY::BLOCK_WORDS * 4));
    };
#pragma unroll
    for (int i = 0; i < RING; i++) {
        request_group(i, i);
        __builtin_amdgcn_sched_barrier(0);      // same request order as inside the loop: see mf_v_run
    }
    // ---- q operand (kivi_mf_dev.h, MfQ): normalised to max |q| in [1, 2), times 2^aexp(i)
    uint32_t amax = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t v = qv[c][e] & 0x7FFFu;
            amax = v > amax ? v : amax;
        }
    amax = max(amax, (uint32_t)__shfl_xor((int)amax, 16));
    amax = max(amax, (uint32_t)__shfl_xor((int)amax, 32));
    const int ex = (int)(amax >> 10);
    const int sq = amax >= 0x7C00u ? 0 : 15 - (ex ? ex : 1);
    const int sa = sq + rsh;                                        // placement of the A operand (mf_load_q)
    const uint32_t zf01 = mf_zfac<BITS>(0, rsh), zf23 = mf_zfac<BITS>(2, rsh);
    {
        uint32_t qq0[4][4];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float f0 = __builtin_ldexpf(h2f_bits(qv[c][2 * i]), sa + aexp_b<BITS>(i));
                const float f1 = __builtin_ldexpf(h2f_bits(qv[c][2 * i + 1]), sa + aexp_b<BITS>(i));
                qq0[c][i] = (uint32_t)f2h_bits(f0) | ((uint32_t)f2h_bits(f1) << 16);
            }
        if (m == 0) {
#pragma unroll
            for (int c = 0; c < 4; c++) *(u32x4*)(q_lds + kb * 16 + c * 4) = u32x4{qq0[c][0], qq0[c][1], qq0[c][2], qq0[c][3]};
        }
        __builtin_amdgcn_wave_barrier();
    }
    const float cmul = __builtin_ldexpf(1.0f, KIVI_MF_PROD_SHIFT - sa);
    const float zmul = __builtin_ldexpf(0.5f, -sq);                 // 0.5: the hi and the lo lane of a group each add the zero-point term
    const int n_half = (W.ng_total + 7) >> 3;
    for (int hq = 0; hq < n_half; hq++) {
        // ---- A operands of this half from the registers the requests landed in; zero-point sums of its 8 groups
        uint32_t A[4][4];
        u32x4 qq[4];
#pragma unroll
        for (int c = 0; c < 4; c++) qq[c] = *(const u32x4*)(q_lds + kb * 16 + c * 4);
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t hi = pk_mul(qq[c][i], sv[c][i]);
                A[c][i] = pk_fms(qq[c][i], sv[c][i], hi & lomask);
            }
        f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const h8 bq = as_h8(pk_mul(qq[c][0], zf01), pk_mul(qq[c][1], zf01), pk_mul(qq[c][2], zf23), pk_mul(qq[c][3], zf23));
            z = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(mv[c][0], mv[c][1], mv[c][2], mv[c][3]), bq, z, 0, 0, 0);
        }
        // the zero-point sums are pinned HERE: hipcc otherwise sinks their MFMAs below the group loop (where z is first
        // read) and keeps their 32 operand registers alive across it
        float zs[4] = {z[0] * zmul, z[1] * zmul, z[2] * zmul, z[3] * zmul};
        asm volatile("" : "+v"(zs[0]), "+v"(zs[1]), "+v"(zs[2]), "+v"(zs[3]));
        request_half(hq + 1 < n_half ? hq + 1 : hq, hq + 1 < n_half);   // lands during this half's groups (after the last one: nothing)
        __builtin_amdgcn_sched_barrier(0);
        float o0[4] = {0.f, 0.f, 0.f, 0.f}, o1[4] = {0.f, 0.f, 0.f, 0.f};
        // one quad of groups: rolled for rings of <= 4 (the slot of group j is a constant either way), unrolled for the ring of 8 --
        // the few-rows instantiations, where a wave has the registers to keep a whole half super-block in flight
        auto quad = [&](int u, auto ubase) {
            constexpr int UB = decltype(ubase)::value;              // ring slot of the quad's first group
            const bool mine = (kb & 1) == u;                        // output rows 4 kb' + j: group 4 (kb' & 1) + j of the half
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int gi = hq * 8 + u * 4 + j;
                const MfW<BITS>& w = wr[(UB + j) % RING];
                f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const MfB b = mf_views_c<BITS>(w, c);
                    const h8 av = as_h8(A[c][0], A[c][1], A[c][2], A[c][3]);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b.b0, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b.b1, a1, 0, 0, 0);
                }
                o0[j] = mine ? a0[j] : o0[j];
                o1[j] = mine ? a1[j] : o1[j];
                request_group((UB + j) % RING, gi + RING);          // after the last use: the load lands in the slot directly
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (RING == 8) {
            quad(0, mf_ic<0>{});
            quad(1, mf_ic<4>{});
        } else {
#pragma unroll 1
            for (int u = 0; u < 2; u++) quad(u, mf_ic<0>{});
        }
        // lane (n, kb'), register j: the hi (kb' < 2) or lo part of group 4 (kb' & 1) + j of the half, tokens n (o0) / 16 + n (o1)
        const int sbi = hq >> 1;
        const int gbase = (hq & 1) * 8 + 4 * (kb & 1);
        const int g_lim = W.ng_total - sbi * 16;                    // groups of this super-block that exist
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v0 = __builtin_fmaf(o0[j], cmul, zs[j]), v1 = __builtin_fmaf(o1[j], cmul, zs[j]);
            v0 += __shfl_xor(v0, 32);
            v1 += __shfl_xor(v1, 32);
            const int g = gbase + j;
            if (g < g_lim) {
                if (kb < 2) sink(W.sb_first + sbi * W.sb_stride, g * 32 + m, v0);
                else sink(W.sb_first + sbi * W.sb_stride, g * 32 + 16 + m, v1);
            }
        }
        if ((hq & 1) || hq + 1 == n_half) done(W.sb_first + sbi * W.sb_stride, g_lim < 16 ? g_lim : 16);
    }
}

// Zero-point sums of a whole super-block for R = 4 / 8: rows = heads (row m -> head m % R), columns = groups.  zz[j] at lane
// (n = group, kb') = sum_d q[head (4 kb' + j) % R, d] * mn[d, group n] in score units (R = 4: register j = head j in every lane;
// R = 8: heads 4 (kb' & 1) + j); `zmul`: 2^-sq of those heads.  `mv`: this lane's 4 x 16 bytes of the zero points of group n
// (B layout = the row layout).
// `qsrc(c)` returns the lane's q'' registers of channel chunk c (from registers, or from LDS when they are parked there).
template <int R, int BITS = 2, typename QSrc>
__device__ __forceinline__ void mf_k_zero(QSrc&& qsrc, const u32x4* mv, const float* zmul, float* zz, int rsh) {
    f4 z = {0.f, 0.f, 0.f, 0.f};
    const uint32_t zf01 = mf_zfac<BITS>(0, rsh), zf23 = mf_zfac<BITS>(2, rsh);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        // A = q * 2^sq: q'' without the 2^aexp and the placement
        const u32x4 qc = qsrc(c);
        const h8 aq = as_h8(pk_mul(qc[0], zf01), pk_mul(qc[1], zf01), pk_mul(qc[2], zf23), pk_mul(qc[3], zf23));
        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq, as_h8(mv[c][0], mv[c][1], mv[c][2], mv[c][3]), z, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) zz[j] = z[j] * zmul[j];
    // pinned here: hipcc otherwise sinks the four MFMAs (and keeps their 32 operand registers alive) down to the first use
    asm volatile("" : "+v"(zz[0]), "+v"(zz[1]), "+v"(zz[2]), "+v"(zz[3]));
}

// R = 4 / 8: rows = (16 / R groups) x (R heads) -- row m -> group m / R of the round, head m % R; one row set per round of
// 16 / R groups, hi and lo operands chained (16 MFMAs per group).  The walker of mf_row4_kernel and of the two-launch qK^T
// (mf_k_kernel<4 | 8>).  With the round-3 placement of the scale (kivi_mfma_layout.h) the 16 bytes (kb, chunk c) of consecutive
// groups are contiguous, so the A-operand load of a round touches fully used 64-byte lines (four per instruction for R = 4);
// the scale of round rq + 1 is requested right after the operands of round rq have been built from the registers it lands in,
// the zero points of the next super-block while the current one is multiplied, the code ring runs across rounds and super-
// blocks (cf. mf_k_seq1).
// Scores go to sink(super-block index, token tt inside it, register r, fp32 scores of tokens tt and tt + 16): the head is
// (4 kb) % R + r, i.e. r itself for R = 4 and 4 (kb & 1) + r for R = 8 (the two scores of a call share everything but the token
// tile, so a sink can convert / scale / compare them as a packed pair); done(super-block index, its number of groups) is called
// when the last score of a super-block has been handed to sink.  RING code blocks in flight, a multiple of the 16 / R groups of a round.
// (Round 4 also built the walker with hi and lo of q'' * scale in ROWS -- 2 groups x (hi | lo) x 4 heads, 8 instead of 16 matrix
// instructions per group, the sums meeting through v_permlane16_swap -- : SQ_VALU_MFMA_BUSY_CYCLES fell from 0.74 to 0.40 of the
// wave cycles and the launch did not get faster (BASELINE config 4: 108.0 us against 107.2 on the same box); not kept,
// profiles/r04_row4_levers.log.)
// OFF: the sequence starts at W.g_first (otherwise that field is ignored and the walk starts at group 0 of sb_first)
template <int R, int RING, int BITS = 2, bool OFF = false, typename Sink, typename Done>
__device__ __forceinline__ void mf_k_seqR(rsrc_t rk, const MfKSeq& W, const uint16_t* q_h0, int64_t q_sh, int rsh, Sink&& sink, Done&& done) {
    static_assert(R == 4 || R == 8, "4 or 8 query heads per kv head");
    typedef MfL<BITS> LY;
    constexpr int RR = R;                                           // rows per group
    constexpr int GPR = 16 / RR;                                     // groups per round
    static_assert((RING >= GPR ? RING % GPR == 0 : GPR % RING == 0) && RING <= 8, "whole rounds per ring, or whole rings per round");
    constexpr int RPT = RING > GPR ? RING / GPR : 1;               // rounds per trip of the loop below
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kb = lane >> 4;
    const int gl = (4 * kb) / RR;                                   // the group of the round whose scores this lane's registers hold
    const int hb = (4 * kb) % R;                                    // ... for the heads hb .. hb + 3
    const int g_first = OFF ? W.g_first : 0;
    if (W.ng_total <= g_first) return;
    const int rq0 = g_first / GPR;                                // first round of the sequence (0 unless it starts inside sb_first)
    auto sb_off = [&](int sbi) { return (uint32_t)(W.sb_first + sbi * W.sb_stride) * W.sb_bytes; };
    const int g_last = W.ng_total - 1;
    const int n_round = (W.ng_total + GPR - 1) / GPR;
    // ---- requests: scale of round 0, zero points of super-block 0, the ring
    u32x4 sv[4], zv[4];
    // (requests past the end: out-of-range per-lane offsets, no traffic -- see mf_k_seq1)
    auto request_round = [&](int rq, bool live) {
        const int g0 = GPR * rq;
        const uint32_t so = sb_off(g0 >> 4);
        const int g = (g0 & 15) + m / RR;
        const uint32_t dead = live ? 0u : MF_DEAD_OFF;
#pragma unroll
        for (int c = 0; c < 4; c++) sv[c] = buf_load<u32x4, true>(rk, (uint32_t)(LY::SCALE_WORD0 * 4 + kt_sm_word4(g, kb, c) * 4) + dead, so);
    };
    auto request_z = [&](int sbi, bool live) {
        const uint32_t dead = live ? 0u : MF_DEAD_OFF;
#pragma unroll
        for (int c = 0; c < 4; c++) zv[c] = buf_load<u32x4, true>(rk, (uint32_t)(LY::MN_WORD0 * 4 + kt_sm_word4(m, kb, c) * 4) + dead, sb_off(sbi));
    };
    request_round(rq0, true);
    request_z(rq0 / RR, true);
    MfW<BITS> wr[RING];
    auto request_group = [&](int slot, int gi) {
        const bool live = gi <= g_last;
        const int gc = live ? gi : g_last;
        mf_load_block<BITS>(wr[slot], rk, live ? (uint32_t)(lane * 16) : MF_DEAD_OFF, sb_off(gc >> 4) + (uint32_t)(gc & 15) * (uint32_t)(LY::BLOCK_WORDS * 4));
    };
#pragma unroll
    for (int i = 0; i < RING; i++) {
        request_group(i, g_first + i);
        __builtin_amdgcn_sched_barrier(0);
    }
    MfQ<R> Q;
    mf_load_q<R, BITS>(q_h0, q_sh, Q, rsh);
    float zmul[4], cmul[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int sqj = __shfl(Q.sq, hb + j);                       // lane h (kb = 0, row h) holds head h's exponent
        zmul[j] = __builtin_ldexpf(1.0f, -sqj);
        cmul[j] = __builtin_ldexpf(1.0f, KIVI_MF_PROD_SHIFT - rsh - sqj);
    }
    auto qsrc = [&](int c) -> u32x4 { return u32x4{Q.qq[c][0], Q.qq[c][1], Q.qq[c][2], Q.qq[c][3]}; };
    float zz[4] = {0.f, 0.f, 0.f, 0.f};
    // one round = GPR groups on ring slots S0 .. S0 + GPR - 1; a ring of several rounds walks S0 = 0, GPR, ... inside one loop
    // trip (so that every slot index is a constant)
    auto do_round = [&](int rq, auto slot0) {
        constexpr int S0 = decltype(slot0)::value;
        const int sbi = rq / RR;                                    // 16 / GPR = RR rounds per super-block
        const int rs = rq - sbi * RR;
        if (rs == 0 || rq == rq0) {                                 // a new super-block (or the sequence's first round): its zero-point sums, then the next one's zero points
            mf_k_zero<R, BITS>(qsrc, zv, zmul, zz, rsh);
            request_z(sbi + 1 < W.n_sb ? sbi + 1 : sbi, sbi + 1 < W.n_sb);
        }
        uint32_t Ah[4][4], Al[4][4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const u32x4 qc = qsrc(c);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                Ah[c][i] = pk_mul(qc[i], sv[c][i]);
                Al[c][i] = pk_fms(qc[i], sv[c][i], Ah[c][i]);
            }
        }
        request_round(rq + 1 < n_round ? rq + 1 : rq, rq + 1 < n_round);
        __builtin_amdgcn_sched_barrier(0);
        // zero points of (group GPR rs + gl, heads (4 kb) % R + j) from the lane of that group in this 16-lane row
        float zs[4];
        const int src = ((lane & 48) + GPR * rs + gl) * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) zs[j] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, zz[j])));
        float o0[4] = {0.f, 0.f, 0.f, 0.f}, o1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < GPR; j++) {
            f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const MfB b = mf_views_c<BITS>(wr[(S0 + j) % RING], c);
                const h8 ah = as_h8(Ah[c][0], Ah[c][1], Ah[c][2], Ah[c][3]);
                const h8 al = as_h8(Al[c][0], Al[c][1], Al[c][2], Al[c][3]);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b.b0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b.b1, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b.b0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b.b1, a1, 0, 0, 0);
            }
            const bool mine = gl == j;                              // rows 4 kb .. 4 kb + 3 = group GPR rq + gl, heads hb + 0 .. 3
#pragma unroll
            for (int r = 0; r < 4; r++) {
                o0[r] = mine ? a0[r] : o0[r];
                o1[r] = mine ? a1[r] : o1[r];
            }
            request_group((S0 + j) % RING, GPR * rq + j + RING);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (GPR * rq + gl < W.ng_total) {
            const int sb = W.sb_first + sbi * W.sb_stride;
            const int g = GPR * rs + gl;
#pragma unroll
            for (int r = 0; r < 4; r++) sink(sb, g * 32 + m, r, __builtin_fmaf(o0[r], cmul[r], zs[r]), __builtin_fmaf(o1[r], cmul[r], zs[r]));
        }
        if (rs == RR - 1 || rq == n_round - 1) {                   // the super-block is complete
            const int left = W.ng_total - 16 * sbi;
            done(W.sb_first + sbi * W.sb_stride, left < 16 ? left : 16);
        }
    };
    static_assert(RPT == 1 || RPT == 2 || RPT == 4, "1, 2 or 4 rounds per trip");
    for (int rq = rq0; rq < n_round; rq += RPT) {
        do_round(rq, mf_ic<0>{});
        if constexpr (RPT >= 2) {
            if (rq + 1 < n_round) do_round(rq + 1, mf_ic<GPR>{});
        }
        if constexpr (RPT == 4) {
            if (rq + 2 < n_round) do_round(rq + 2, mf_ic<2 * GPR>{});
            if (rq + 3 < n_round) do_round(rq + 3, mf_ic<3 * GPR>{});
        }
    }
}

// ------------------------------------------------------------------------------------------------ sV
// Accumulators of a wave over its token blocks: acc[c][tile] = rows x channels 32 c + 16 tile + n, chained through the
// C operand over all blocks.
// R = 1: row 4 cg + j, j even = hi, j odd = lo part of p'' * scale[t, cg]; the lanes of rows j >= 2 load the ZERO POINTS
//        instead of the scale (their rows are never read) and so accumulate sum p'' * mn while rows j < 2 accumulate
//        sum p'' * scale -- one 16-byte load per lane brings both.
// R = 4: row 4 cg + r = (channel group cg, head r); hi and lo are two operands.
// p'' = the fp16 probability times 2^(Sp + aexp(i)) (exact), i = ((t & 7) >> 1): written that way into LDS by the softmax.
// CENTRING: the codes are all >= 0 and the zero points < 0, so sum p s code grows to ~100x the output and the matrix pipe
// (which aligns its 32 products to the largest addend, C included, and drops what falls ~2^-26 below) loses 2e-3 of the
// output over a 4k row.  So the FIRST block of every ring round also accumulates A x (-1.5 RING) -- the expected code sum of
// the round -- and the lanes keep the exact sum of the A they centred with (cs4 / cs6, v_dot2_f32_f16); the running sums
// then stay within ~RING blocks' worth of the output.  (A centring MFMA for every tile of every block -- the first round-3
// form -- doubled the matrix instructions for nothing.)
// HL (R = 4 only): hi and lo parts of p'' * scale in ROWS -- row m -> channel group m >> 3 of the row set, hi (m & 4 == 0) | lo,
// head m & 3; two row sets as for R = 8, ONE MFMA per (channel chunk, tile): 8 (+ the centring ones) per block instead of 16.
// The hi and lo sums of an output meet in the caller's final reduction (mf_v_finish writes them to separate slots).
template <int R, bool HL = false>
struct MfVAcc {
    static_assert(!HL || R == 4, "hi / lo rows: 2 channel groups x (hi | lo) x 4 heads");
    static constexpr int NS = (R == 8 || HL) ? 2 : 1;      // row sets (channel groups {0, 1} and {2, 3})
    f4 acc[4][2];
    float z4[NS], z6[NS];      // R = 1: sum p'' * (scale | mn) of every block; R = 4 / 8: sum p'' * mn        (registers with 2^4 / 2^6)
    float c4[NS], c6[NS];      // sums over the centring blocks: R = 1: p'' * (scale | mn); R = 4 / 8: the hi operand
};

template <int R, bool HL>
__device__ __forceinline__ void mf_v_init(MfVAcc<R, HL>& A) {
#pragma unroll
    for (int c = 0; c < 4; c++) A.acc[c][0] = A.acc[c][1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < MfVAcc<R, HL>::NS; s++) A.z4[s] = A.z6[s] = A.c4[s] = A.c6[s] = 0.f;
}

// -1.5 * RING in the units of a masked code: registers 0, 1 hold code * 2^-16, registers 2, 3 code * 2^-18 (fp16 bits, both halves)
template <int RING, int BITS = 2> struct MfCentre;
template <> struct MfCentre<2, 2> { static constexpr uint32_t a = 0x83008300u, b = 0x80C080C0u; static constexpr float f = 3.0f; };
template <> struct MfCentre<3, 2> { static constexpr uint32_t a = 0x84808480u, b = 0x81208120u; static constexpr float f = 4.5f; };
template <> struct MfCentre<4, 2> { static constexpr uint32_t a = 0x86008600u, b = 0x81808180u; static constexpr float f = 6.0f; };
template <> struct MfCentre<8, 2> { static constexpr uint32_t a = 0x8A008A00u, b = 0x83008300u; static constexpr float f = 12.0f; };   // (-12 x 2^-16 is a normal fp16)
// 4-bit codes: -7.5 * RING x 2^-18 in every register (ring 2: a subnormal, from 3 on a normal fp16)
template <> struct MfCentre<2, 4> { static constexpr uint32_t a = 0x83C083C0u, b = 0x83C083C0u; static constexpr float f = 15.0f; };
template <> struct MfCentre<3, 4> { static constexpr uint32_t a = 0x85A085A0u, b = 0x85A085A0u; static constexpr float f = 22.5f; };
template <> struct MfCentre<4, 4> { static constexpr uint32_t a = 0x87808780u, b = 0x87808780u; static constexpr float f = 30.0f; };
template <> struct MfCentre<8, 4> { static constexpr uint32_t a = 0x8B808B80u, b = 0x8B808B80u; static constexpr float f = 60.0f; };

// one 32-token block.  w: code words; ps: the lane's 8 scaled probabilities (tokens 8 kb + e of its row's head);
// R = 1: sm[0] = scale (rows j < 2) or zero points (rows j >= 2), lomask = all ones in lo rows;
// R = 4: row (channel group m >> 2, head m & 3): sm[0] = scale, mn[0] = zero points of that channel group;
// R = 8: row (channel group 2 s + (m >> 3), head m & 7) for the row sets s = 0, 1: sm[s], mn[s]; channel chunk c multiplies with
//        row set c >> 1 (its rows for channel group c are the useful ones).
// CENTRE: this block also accumulates A x (-1.5 RING).
template <int R, int RING, bool CENTRE, bool HL, int BITS = 2>
__device__ __forceinline__ void mf_v_block(MfVAcc<R, HL>& A, const MfW<BITS>& w, const u32x4& ps, const u32x4* sm, const u32x4* mn,
                                           uint32_t lomask) {
    typedef MfCentre<RING, BITS> CE;
    const h8 bc = as_h8(CE::a, CE::a, CE::b, CE::b);
    if constexpr (HL) {
        // lomask: all ones in the lanes of a lo row.  hi rows: fp16(p'' s); lo rows: the exact remainder (cf. mf_k_seqR)
        uint32_t psm[4], a[2][4];
#pragma unroll
        for (int i = 0; i < 4; i++) psm[i] = ps[i] & lomask;
#pragma unroll
        for (int s = 0; s < 2; s++) {
#pragma unroll
            for (int i = 0; i < 4; i++) a[s][i] = pk_fms(ps[i], sm[s][i], pk_mul(psm[i], sm[s][i]));
            A.z4[s] = dot2_f16(ps[0], mn[s][0], A.z4[s]);            // (hi and lo lanes both: halved in mf_v_finish)
            A.z4[s] = dot2_f16(ps[1], mn[s][1], A.z4[s]);
            A.z6[s] = dot2_f16(ps[2], mn[s][2], A.z6[s]);
            A.z6[s] = dot2_f16(ps[3], mn[s][3], A.z6[s]);
            if constexpr (CENTRE) {                                // what this lane's own row (hi or lo) is centred with
                A.c4[s] = dot2_f16(a[s][0], MF_ONE2, A.c4[s]);
                A.c4[s] = dot2_f16(a[s][1], MF_ONE2, A.c4[s]);
                A.c6[s] = dot2_f16(a[s][2], MF_ONE2, A.c6[s]);
                A.c6[s] = dot2_f16(a[s][3], MF_ONE2, A.c6[s]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const h8 av = as_h8(a[c >> 1][0], a[c >> 1][1], a[c >> 1][2], a[c >> 1][3]);
            const MfB b = mf_views_c<BITS>(w, c);
            f4 x0 = A.acc[c][0], x1 = A.acc[c][1];
            if constexpr (CENTRE) {
                x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bc, x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bc, x1, 0, 0, 0);
            }
            A.acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b.b0, x0, 0, 0, 0);
            A.acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b.b1, x1, 0, 0, 0);
        }
    } else if constexpr (R == 1) {
        uint32_t a[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t hi = pk_mul(ps[i], sm[0][i]);
            a[i] = pk_fms(ps[i], sm[0][i], hi & lomask);          // hi rows: fp16(p'' s); lo rows: the exact remainder
        }
        A.z4[0] = dot2_f16(ps[0], sm[0][0], A.z4[0]);
        A.z4[0] = dot2_f16(ps[1], sm[0][1], A.z4[0]);
        A.z6[0] = dot2_f16(ps[2], sm[0][2], A.z6[0]);
        A.z6[0] = dot2_f16(ps[3], sm[0][3], A.z6[0]);
        if constexpr (CENTRE) {
            // what THIS row (hi or lo) is centred with: the operand as the matrix pipe gets it.  (Until round 6 every lane summed the
            // exact product p'' s here and mf_v_finish took the hi rows' sum for hi + lo -- the same number while lo is the exact
            // remainder, but an operand below 2^-14 has no representable remainder, and the difference came back times the
            // ring's centring weight: 2.5 % of the packed part in tests/test_mfma_gpu.py::test_big_value_units_... with rings of 8.)
            A.c4[0] = dot2_f16(a[0], MF_ONE2, A.c4[0]);
            A.c4[0] = dot2_f16(a[1], MF_ONE2, A.c4[0]);
            A.c6[0] = dot2_f16(a[2], MF_ONE2, A.c6[0]);
            A.c6[0] = dot2_f16(a[3], MF_ONE2, A.c6[0]);
        }
        const h8 av = as_h8(a[0], a[1], a[2], a[3]);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const MfB b = mf_views_c<BITS>(w, c);
            f4 x0 = A.acc[c][0], x1 = A.acc[c][1];
            if constexpr (CENTRE) {
                x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bc, x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bc, x1, 0, 0, 0);
            }
            A.acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b.b0, x0, 0, 0, 0);
            A.acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b.b1, x1, 0, 0, 0);
        }
    } else {
        constexpr int NS = MfVAcc<R, HL>::NS;
        uint32_t hi[NS][4], lo[NS][4];
#pragma unroll
        for (int s = 0; s < NS; s++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                hi[s][i] = pk_mul(ps[i], sm[s][i]);
                lo[s][i] = pk_fms(ps[i], sm[s][i], hi[s][i]);
            }
            A.z4[s] = dot2_f16(ps[0], mn[s][0], A.z4[s]);
            A.z4[s] = dot2_f16(ps[1], mn[s][1], A.z4[s]);
            A.z6[s] = dot2_f16(ps[2], mn[s][2], A.z6[s]);
            A.z6[s] = dot2_f16(ps[3], mn[s][3], A.z6[s]);
            if constexpr (CENTRE) {
                A.c4[s] = dot2_f16(hi[s][0], MF_ONE2, A.c4[s]);
                A.c4[s] = dot2_f16(hi[s][1], MF_ONE2, A.c4[s]);
                A.c6[s] = dot2_f16(hi[s][2], MF_ONE2, A.c6[s]);
                A.c6[s] = dot2_f16(hi[s][3], MF_ONE2, A.c6[s]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int s = (NS == 2) ? (c >> 1) : 0;               // (a constant after unrolling)
            const h8 ah = as_h8(hi[s][0], hi[s][1], hi[s][2], hi[s][3]), al = as_h8(lo[s][0], lo[s][1], lo[s][2], lo[s][3]);
            const MfB b = mf_views_c<BITS>(w, c);
            f4 x0 = A.acc[c][0], x1 = A.acc[c][1];
            if constexpr (CENTRE) {
                x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bc, x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bc, x1, 0, 0, 0);
            }
            x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b.b0, x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b.b1, x1, 0, 0, 0);
            A.acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b.b0, x0, 0, 0, 0);
            A.acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b.b1, x1, 0, 0, 0);
        }
    }
}

// The stream of token blocks [b_lo, b_hi) of one unit's V store.  `rv`: buffer over the unit's V store (all its super-blocks),
// sb_bytes = byte stride between consecutive super-blocks.  prime() requests the first RING blocks (callers do it as early
// as they can: the fused row kernel before its softmax), run() consumes: ps_lds = the R rows of scaled probabilities
// (halves), row pitch `pitch` halves, indexed by token - tok0.
template <int R, int RING, bool HL = false, int BITS = 2>
struct MfVStream {
    static constexpr int NS = MfVAcc<R, HL>::NS;
    typedef MfL<BITS> LY;
    MfW<BITS> wr[RING];
    u32x4 sr[RING][NS], mr[R == 1 ? 1 : RING][NS];
    uint32_t sm_off, mn_off, sb_bytes;
    int b_last;
    int sb_first, sb_stride;       // block q of the stream lives in super-block sb_first + (q >> 4) * sb_stride, block q & 15

    __device__ __forceinline__ void request(rsrc_t rv, int slot, int bl) {
        const int lane = threadIdx.x & 63;
        const bool live = bl <= b_last;                            // past the stream: out-of-range per-lane offsets (zeros, no traffic)
        const int bc = live ? bl : b_last;
        const uint32_t dead = live ? 0u : MF_DEAD_OFF;
        const uint32_t so = (uint32_t)(sb_first + (bc >> 4) * sb_stride) * sb_bytes;
        mf_load_block<BITS>(wr[slot], rv, (uint32_t)(lane * 16) + dead, so + (uint32_t)(bc & 15) * (uint32_t)(LY::BLOCK_WORDS * 4));
#pragma unroll
        for (int s = 0; s < NS; s++) {                             // (row set s: two channel groups = 32 bytes further)
            sr[slot][s] = buf_load<u32x4, true>(rv, sm_off + 32u * s + dead, so + (uint32_t)(bc & 15) * 256u);
            if constexpr (R != 1) mr[slot][s] = buf_load<u32x4, true>(rv, mn_off + 32u * s + dead, so + (uint32_t)(bc & 15) * 256u);
        }
    }

    // blocks [b_lo, b_hi) in stream numbering; (first, stride) = (0, 1): stream numbering = the unit's block numbering
    __device__ __forceinline__ void prime(rsrc_t rv, uint32_t sb_bytes_, int b_lo, int b_hi, int first = 0, int stride = 1) {
        const int lane = threadIdx.x & 63;
        const int m = lane & 15, kb = lane >> 4;
        const int cg = (R == 8 || HL) ? (m >> 3) : (m >> 2), j = m & 3;   // channel group of the lane's row (row set 0)
        sm_off = (uint32_t)(((R == 1 && j >= 2) ? LY::MN_WORD0 : LY::SCALE_WORD0) * 4 + kb * 64 + cg * 16);
        mn_off = (uint32_t)(LY::MN_WORD0 * 4 + kb * 64 + cg * 16);
        sb_bytes = sb_bytes_;
        sb_first = first;
        sb_stride = stride;
        b_last = b_hi > b_lo ? b_hi - 1 : b_lo;
        if (b_hi <= b_lo) return;
#pragma unroll
        for (int i = 0; i < RING; i++) {
            request(rv, i, b_lo + i);
            // the slots are requested in the order the loop re-requests them: hipcc merges the wait counters of the loop's
            // two entries, and a different order here makes every wait inside the loop a vmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // Consumes blocks [b_lo, b_hi) (b_lo on a ring-round boundary relative to the primed block; b_hi <= the primed end: the
    // ring keeps requesting up to b_last, so a caller may run() the stream piece by piece -- one super-block at a time with
    // the probabilities of the next one made in between -- without ever draining it).  ps_lds: the R rows of scaled
    // probabilities, row pitch `pitch` halves, indexed by (stream block * 32 + token in block) - tok0.
    // ksh (wave-uniform; mf_ksh: > 0 only for a peaked row of a unit whose V store holds a scale >= 256; the largest over the R rows
    // of the block): the SCALES of a ring round are taken 2^ksh lower (<= 2^3) before its blocks are multiplied (in place; exact for
    // every scale >= 2^-11; R = 1: not in the lanes whose registers hold zero points) and mf_v_finish brings the products back (`up`).
    // One branch per ring round that ordinary data never takes.
    __device__ __forceinline__ void run(MfVAcc<R, HL>& A, rsrc_t rv, int b_lo, int b_hi, const uint16_t* ps_lds, int pitch, int tok0, int ksh = 0) {
        const int lane = threadIdx.x & 63;
        const int m = lane & 15, kb = lane >> 4;
        const int j = m & 3;
        const uint32_t lomask = (HL ? (m & 4) != 0 : (R == 1 && (j & 1))) ? 0xFFFFFFFFu : 0u;
        // 2^-ksh in both halves (fp16 exponent field 15 - ksh); R = 1: the lanes of rows j >= 2 hold zero points
        ksh = __builtin_amdgcn_readfirstlane(ksh);
        const uint32_t bigf = (R == 1 && j >= 2) ? 0x3C003C00u : (uint32_t)(((15 - ksh) << 10) * 0x00010001u);
        const bool big = ksh > 0;
        const uint16_t* prow = ps_lds + (R == 1 ? 0 : (m % R) * pitch) + 8 * kb - tok0;      // the head of the lane's row
        if (b_hi <= b_lo) return;
        for (int b0 = b_lo; b0 < b_hi; b0 += RING) {
            if (big) {                                             // (at the top of the round: the blocks below stay as they are for everyone else)
#pragma unroll
                for (int s = 0; s < RING; s++) {
#pragma unroll
                    for (int q = 0; q < NS; q++) {
#pragma unroll
                        for (int i = 0; i < 4; i++) sr[s][q][i] = pk_mul(sr[s][q][i], bigf);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < RING; s++) {
                const int bl = b0 + s;
                // blocks past the range repeat the last block with zero probabilities
                u32x4 ps = *(const u32x4*)(prow + (bl < b_hi ? bl : b_hi - 1) * 32);
                if (bl >= b_hi) ps = u32x4{0, 0, 0, 0};
                if (s == 0) mf_v_block<R, RING, true, HL, BITS>(A, wr[s], ps, sr[s], mr[R == 1 ? 0 : s], lomask);
                else mf_v_block<R, RING, false, HL, BITS>(A, wr[s], ps, sr[s], mr[R == 1 ? 0 : s], lomask);
                request(rv, s, bl + RING);
                // nothing moves across this point: without it hipcc gathers all RING re-requests at the end of the round, i.e.
                // a block's data is asked for one block before its use
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
};

// Per-wave result: O[r][d] (before the 2^-Sp of the head) into `dst` (fp32, [R][128]; HL: [2][R][128], the hi and the lo
// part of every output, to be added by the caller) -- 2^12 * (hi + lo sums) + zero-point term + 1.5 * sum p'' s.
// `zl`: 128 floats of scratch LDS of this wave.
// `up`: 2^ksh of MfVStream::run (the scales went in that much lower: the products and the centring sums come
// back here, the zero-point term never left), 1 otherwise.
template <int R, int RING, bool HL, int BITS = 2>
__device__ __forceinline__ void mf_v_finish(const MfVAcc<R, HL>& A, float* zl, float* dst, float up = 1.0f) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, kb = lane >> 4;
    const float CF = MfCentre<RING, BITS>::f * up;                // what the centring blocks subtracted per unit of A
    const float PS = (float)(1 << KIVI_MF_PROD_SHIFT) * up;
    constexpr float W4 = BITS == 2 ? 0.0625f : 0.015625f;         // 2^-aexp of the registers 0, 1 (4-bit codes: 2^-6 like 2, 3)
    // per-lane dot sums -> LDS -> every lane gathers the four kb partials of the rows it needs
    if constexpr (HL) {
        // this lane's accumulator rows 4 kb + j = (channel group kb >> 1 of the row set, hi | lo = kb & 1, head j): the useful rows
        // of the channel chunks c = (kb >> 1) + 2 s.  The zero-point term was summed by the hi AND the lo lanes: half each.
#pragma unroll
        for (int s = 0; s < 2; s++)
            zl[64 * s + lane] = __builtin_fmaf(CF, A.c4[s] * W4 + A.c6[s] * 0.015625f, A.z4[s] * (0.5f * W4) + A.z6[s] * 0.0078125f);
        __builtin_amdgcn_wave_barrier();
        float* dhl = dst + (kb & 1) * R * 128;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            float br[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                br[r] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) br[r] += zl[64 * s + 4 * kb + r + 16 * k];
            }
            const int c = (kb >> 1) + 2 * s;
#pragma unroll
            for (int tile = 0; tile < 2; tile++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v = (kb >> 1) ? A.acc[2 * s + 1][tile][r] : A.acc[2 * s][tile][r];
                    dhl[r * 128 + 32 * c + 16 * tile + n] = __builtin_fmaf(v, PS, br[r]);
                }
            }
        }
    } else if constexpr (R == 1) {
        zl[lane] = A.z4[0] * W4 + A.z6[0] * 0.015625f;
        zl[64 + lane] = A.c4[0] * W4 + A.c6[0] * 0.015625f;
        __builtin_amdgcn_wave_barrier();
        // output lane (n, kb' = cg): sum p'' mn (rows 4 cg + 2) + CF * sum over the centring blocks of the operands hi + lo (rows 4 cg, 4 cg + 1)
        float zc = 0.f, zm = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            zc += zl[64 + 4 * kb + 16 * k] + zl[64 + 4 * kb + 1 + 16 * k];
            zm += zl[4 * kb + 2 + 16 * k];
        }
        const float br = __builtin_fmaf(CF, zc, zm);
#pragma unroll
        for (int tile = 0; tile < 2; tile++) {
            // (scalar selects: an `if (kb == c) v = acc[c]` chain over the vectors becomes a scratch array indexed by kb)
            float v0 = A.acc[0][tile][0], v1 = A.acc[0][tile][1];
#pragma unroll
            for (int c = 1; c < 4; c++) {
                v0 = (kb == c) ? A.acc[c][tile][0] : v0;
                v1 = (kb == c) ? A.acc[c][tile][1] : v1;
            }
            dst[32 * kb + 16 * tile + n] = __builtin_fmaf(v0 + v1, PS, br);
        }
    } else if constexpr (R == 4) {
        zl[lane] = __builtin_fmaf(CF, A.c4[0] * W4 + A.c6[0] * 0.015625f, A.z4[0] * W4 + A.z6[0] * 0.015625f);
        __builtin_amdgcn_wave_barrier();
        float br[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            br[r] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) br[r] += zl[4 * kb + r + 16 * k];
        }
#pragma unroll
        for (int tile = 0; tile < 2; tile++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = A.acc[0][tile][r];
#pragma unroll
                for (int c = 1; c < 4; c++) v = (kb == c) ? A.acc[c][tile][r] : v;
                dst[r * 128 + 32 * kb + 16 * tile + n] = __builtin_fmaf(v, PS, br[r]);
            }
        }
    } else {
        // R = 8: this lane's accumulator rows 4 kb + j = (channel group kb >> 1 of the row set, head 4 (kb & 1) + j); they are the
        // useful rows of the channel chunks c = (kb >> 1) + 2 s, s = 0, 1
#pragma unroll
        for (int s = 0; s < 2; s++)
            zl[64 * s + lane] = __builtin_fmaf(CF, A.c4[s] * W4 + A.c6[s] * 0.015625f, A.z4[s] * W4 + A.z6[s] * 0.015625f);
        __builtin_amdgcn_wave_barrier();
        const int hb = 4 * (kb & 1);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            float br[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                br[r] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) br[r] += zl[64 * s + 4 * kb + r + 16 * k];
            }
            const int c = (kb >> 1) + 2 * s;
#pragma unroll
            for (int tile = 0; tile < 2; tile++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v = (kb >> 1) ? A.acc[2 * s + 1][tile][r] : A.acc[2 * s][tile][r];
                    dst[(hb + r) * 128 + 32 * c + 16 * tile + n] = __builtin_fmaf(v, PS, br[r]);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Sp of a softmax row from its sum: the fp16 probabilities (<= 1 / sum) are scaled by 2^Sp, Sp = e = clamp(floor(log2 sum), 0, 14),
// so that p'' * scale stays a normal fp16 whatever the row length; plus the POSITIVE part of `rsh`, the range shift of the unit's V
// store (mf_range_shift, kivi_mfma_layout.h): 2^8 higher when all its scales are < 2^-8 (p'' <= 2^15), so that p'' * scale keeps a
// normal hi part.  A unit that holds a scale >= 256 (rsh < 0) needs the operand p'' * scale 2^KIVI_MF_BIG_SHIFT_V = 2^7 lower to stay
// finite for every finite scale.  As much of that as is LOSSLESS goes into p'' -- d = min(7, e + 4): p * 2^(e + (4 | 6) - d) is still
// p times a non-negative power of two, so not one probability is rounded; Sp = e - d, -4 .. 7 -- and only the rest, mf_ksh = 7 - d =
// max(0, 3 - e) bits, into the scales (MfVStream::run, ksh): nothing for a row whose sum is >= 8, at most 2^-3 for a peaked one
// (exact for every scale >= 2^-11).  (Round 6 before its last sessions: all 2^10 into p'', which rounded the small probabilities of
// a peaked row; then all 2^7 into the scales, which rounded scales below 2^-7 -- both found by tools/fuzz_decode.py.)
__device__ __forceinline__ int mf_sp_e(float sum) {
    const int e = (int)((__builtin_bit_cast(uint32_t, sum) >> 23) & 255u) - 127;
    return e < 0 ? 0 : (e > 14 ? 14 : e);
}
__device__ __forceinline__ int mf_big_d(int e) { return (e + 4 < KIVI_MF_BIG_SHIFT_V) ? e + 4 : KIVI_MF_BIG_SHIFT_V; }
__device__ __forceinline__ int mf_sp(float sum, int rsh) {
    const int e = mf_sp_e(sum);
    return rsh < 0 ? e - mf_big_d(e) : e + rsh;
}
// what is left for the scales of a row of a big-scale unit (0 for every other unit)
__device__ __forceinline__ int mf_ksh(float sum, int rsh) { return rsh < 0 ? KIVI_MF_BIG_SHIFT_V - mf_big_d(mf_sp_e(sum)) : 0; }
// The two exact power-of-two factors that take a fp16 probability p to p'' = p * 2^(Sp + 4 | 6): first 2^(4 | 6) (times 2^8 for
// a unit placed higher: p <= 1, the product is exact and <= 2^14), then 2^(Sp - max(rsh, 0)) = 2^-4 .. 2^14: both exact (a negative
// exponent only ever takes back part of the 2^(4 | 6): mf_sp).
__device__ __forceinline__ _Float16 mf_p_mul_a(bool reg23, int rsh) {       // reg23: registers 2, 3 of the operand (2^6), else 2^4
    return (_Float16)__builtin_ldexpf(1.0f, (reg23 ? 6 : 4) + (rsh > 0 ? rsh : 0));
}
__device__ __forceinline__ _Float16 mf_p_mul_sp(int sp, int rsh) { return (_Float16)__builtin_ldexpf(1.0f, sp - (rsh > 0 ? rsh : 0)); }
// fp16 p -> p'' for token t: 2^(Sp + 4) for (t & 7) < 4, 2^(Sp + 6) otherwise (the register i = (t & 7) >> 1 of the operand)
template <int BITS = 2>
__device__ __forceinline__ uint16_t mf_scale_p(uint16_t p, int sp, int t) {
    const int e = sp + ((BITS == 4 || (t & 4)) ? 6 : 4);
    return f2h_bits(__builtin_ldexpf(h2f_bits(p), e));
}

// ------------------------------------------------------------------------------------------------ softmax of a row held in LDS
// The row kernels keep the SCALED fp16 scores of a row in LDS (the qK^T sinks write fp16(fp16(score) * inv_scale),
// llama_kivi.py:339, and fold them into a per-lane running maximum `mx_lane`); halves [n, n_pad) hold fp16 -inf and
// n_pad >= n + 4.  This turns the row into p'' in place (fp16(exp(x - M) / sum), :364-375, times 2^(Sp + 4 | 6): mf_scale_p) for
// the packed prefix [0, Tv), zeros after it, and the fp16 probabilities of the window [Tv, n) into pw_row.  Returns Sp.
// With a mask the row is first rewritten with the mask row added (fp16, clamped at the fp16 minimum: :366-372) and the maximum
// taken again.  Two block barriers; sm_lds: 2 NW floats that nothing else touches between two calls.
// A thread owns 4 consecutive scores per chunk of 4 NTH: register resident, SMC chunks.  ~8 VALU issue slots per score
// (x - M by v_fma_mix_f32 straight from the packed halves, packed fp32 multiplies / adds, v_cvt_pk_f16_f32, two packed fp16
// multiplies for the exact power-of-two scalings, the exponential counted as 4): this was ~25 before and, with every block of
// the chip in its softmax at the same time (one round of blocks), 29 of the 105 us of the grouped-query row kernel.
__device__ __forceinline__ float mf_sub_lo(uint32_t hpair, float nmx) {       // float(low half) + nmx, exact
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpair), "v"(nmx));
    return d;
}
__device__ __forceinline__ float mf_sub_hi(uint32_t hpair, float nmx) {       // float(high half) + nmx, exact
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpair), "v"(nmx));
    return d;
}

// two fp32 values -> packed fp16 pair, round to nearest even (v_cvt_pk_f16_f32): lo = fp16(a), hi = fp16(b)
__device__ __forceinline__ uint32_t mf_cvt_pair(float a, float b) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2v){a, b}, h2v));
}

// fp16(float(h) * inv) of both halves of a packed pair (kivi_scaled_score without a mask: the fp32 product, then one
// rounding to fp16 -- v_fma_mixlo / mixhi_f16 do exactly that, one instruction per score)
__device__ __forceinline__ uint32_t mf_scale_pair(uint32_t hpair, float inv) {
    uint32_t d;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpair), "v"(inv));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hpair), "v"(inv));
    return d;
}

// rsh: the range shift of the unit's V store (mf_sp).  dump (KIVI_GQA_DUMP_SCORES, tests; a run-time pointer, null otherwise: the
// PRODUCT instantiation is the one the stage-A checks run on): the fp16 row as the softmax consumes it (scaled, mask added) also
// goes to this row of the caller's score buffer.  ksh (out): mf_ksh of the row.
template <int NTH, int SMC, int BITS = 2>
__device__ __forceinline__ int mf_row_softmax(uint16_t* row, int n, int n_pad, int Tv, float mx_lane, const uint16_t* mrow,
                                              uint16_t* pw_row, float* sm_lds, int rsh, int& ksh, uint16_t* dump = nullptr) {
    typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    constexpr int NW = NTH / 64, SCH = NTH * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = (n + SCH - 1) / SCH;                           // chunks that hold scores (block-uniform)
    float mx = mx_lane;
    if (mrow) {                                                    // masked rows: add the mask in place, take the maximum again
        mx = -__builtin_inff();
#pragma unroll 1
        for (int c = 0; c < nch; c++) {
            const int j0 = c * SCH + (int)threadIdx.x * 4;
            if (j0 < n) {
                u16x4 raw = *(const u16x4*)(row + j0);
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (j0 + e < n) {
                        float v = (float)(_Float16)(h2f_bits(raw[e]) + h2f_bits(mrow[j0 + e]));
                        if (v < -65504.0f) v = -65504.0f;
                        raw[e] = f2h_bits(v);
                        mx = __builtin_fmaxf(mx, v);
                    }
                *(u16x4*)(row + j0) = raw;                         // the same thread reads it back below
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) sm_lds[wave] = mx;
    __syncthreads();
    mx = sm_lds[0];
#pragma unroll
    for (int w = 1; w < NW; w++) mx = __builtin_fmaxf(mx, sm_lds[w]);
    const float nmx = -mx;
    const f2v l2e = {1.44269504088896340736f, 1.44269504088896340736f};
    f2v xe[SMC][2];
    f2v acc = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < SMC; c++) {
        xe[c][0] = xe[c][1] = (f2v){0.f, 0.f};
        if (c < nch) {
            const int j0 = c * SCH + (int)threadIdx.x * 4;
            const u32x2v raw = *(const u32x2v*)(row + (j0 < n_pad ? j0 : n_pad - 4));     // past the row: -inf -> exp = 0
            if (dump && j0 < n) *(u32x2v*)(dump + j0) = raw;        // (rows are padded to a multiple of 8 scores)
            const f2v d01 = (f2v){mf_sub_lo(raw[0], nmx), mf_sub_hi(raw[0], nmx)} * l2e;   // kivi_exp(x - M), two at a time
            const f2v d23 = (f2v){mf_sub_lo(raw[1], nmx), mf_sub_hi(raw[1], nmx)} * l2e;
            xe[c][0] = (f2v){__builtin_amdgcn_exp2f(d01[0]), __builtin_amdgcn_exp2f(d01[1])};
            xe[c][1] = (f2v){__builtin_amdgcn_exp2f(d23[0]), __builtin_amdgcn_exp2f(d23[1])};
            acc += xe[c][0];
            acc += xe[c][1];
        }
    }
    float sum = wave_sum(acc[0] + acc[1]);
    if (lane == 0) sm_lds[NW + wave] = sum;
    __syncthreads();
    sum = sm_lds[NW];
#pragma unroll
    for (int w = 1; w < NW; w++) sum += sm_lds[NW + w];
    const float inv = 1.0f / sum;
    const int sp = mf_sp(sum, rsh);
    ksh = mf_ksh(sum, rsh);                                      // (what the scales of a big-scale unit still have to take: MfVStream::run)
    const _Float16 m_sp = mf_p_mul_sp(sp, rsh);                   // 2^-4 .. 2^14 
    const _Float16 m_a4 = mf_p_mul_a(BITS == 4, rsh), m_a6 = mf_p_mul_a(true, rsh);
    const f2v inv2 = {inv, inv};
#pragma unroll
    for (int c = 0; c < SMC; c++) {
        const int j0 = c * SCH + (int)threadIdx.x * 4;
        if (j0 < n_pad) {
            u32x2v o = {0u, 0u};
            if (c < nch) {
                // p = fp16(e / sum) first (the reference's cast, :375), then the two power-of-two scalings (mf_p_mul_a, mf_p_mul_sp)
                const h2v p01 = __builtin_convertvector(xe[c][0] * inv2, h2v);
                const h2v p23 = __builtin_convertvector(xe[c][1] * inv2, h2v);
                if (j0 + 4 <= Tv) {
                    const _Float16 m_a = (j0 & 4) ? m_a6 : m_a4;
                    o[0] = __builtin_bit_cast(uint32_t, (p01 * (h2v){m_a, m_a}) * (h2v){m_sp, m_sp});
                    o[1] = __builtin_bit_cast(uint32_t, (p23 * (h2v){m_a, m_a}) * (h2v){m_sp, m_sp});
                } else {                                           // the chunk that holds the end of the packed prefix / the window
                    // (whole-register casts: __builtin_bit_cast applied directly to an ext-vector ELEMENT reads element 0 -- hipcc 7.2)
                    const uint32_t w01 = __builtin_bit_cast(uint32_t, p01), w23 = __builtin_bit_cast(uint32_t, p23);
                    const uint16_t pp[4] = {(uint16_t)(w01 & 0xFFFFu), (uint16_t)(w01 >> 16), (uint16_t)(w23 & 0xFFFFu), (uint16_t)(w23 >> 16)};
                    uint16_t q[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int j = j0 + e;
                        if (j >= Tv && j < n) pw_row[j - Tv] = pp[e];
                        q[e] = (j < Tv) ? mf_scale_p<BITS>(pp[e], sp, j) : (uint16_t)0;
                    }
                    o[0] = (uint32_t)q[0] | ((uint32_t)q[1] << 16);
                    o[1] = (uint32_t)q[2] | ((uint32_t)q[3] << 16);
                }
            }
            *(u32x2v*)(row + j0) = o;
        }
    }
    return sp;
}

// The same for ONE WAVE per row (mf_row4_kernel: wave r takes head r; the four rows of a unit at once instead of one after the
// other, nothing block-wide inside): three passes over the row in LDS -- [mask +] maximum, sum of exp(x - M), write of p'' -- each
// lane on 4 consecutive scores per 256-score chunk; the exponentials are computed twice rather than kept (a 9216-key row is 144
// scores per lane).  The block version above spends most of its ~4 us per row in two block barriers and dependent LDS round
// trips, four rows in sequence; profiles/r04_row4_levers.log.  Returns Sp (wave-uniform).
template <int BITS = 2>
__device__ __forceinline__ int mf_row_softmax_wave(uint16_t* row, int n, int n_pad, int Tv, const uint16_t* mrow, uint16_t* pw_row, int rsh,
                                                   int& ksh, uint16_t* dump) {
    typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    constexpr int NB = 4;                                          // chunks per batch: their LDS reads are issued before the first is used
    const int lane = threadIdx.x & 63;
    const int nch = (n + 255) >> 8;                                // 256-score chunks that hold scores
    const u32x2v ninf2 = {0xFC00FC00u, 0xFC00FC00u};
    auto load = [&](int c) -> u32x2v {                             // chunk c of this lane; past the padded row: -inf (exp = 0)
        const int j0 = c * 256 + lane * 4;
        return j0 < n_pad ? *(const u32x2v*)(row + j0) : ninf2;
    };
    // ---- pass A: [mask in place,] maximum
    if (mrow) {                                                    // masked rows: :366-372, fp16 add clamped at the fp16 minimum
        for (int c = 0; c < nch; c++) {
            const int j0 = c * 256 + lane * 4;
            if (j0 < n) {
                u16x4 rw = *(const u16x4*)(row + j0);
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (j0 + e < n) {
                        float v = (float)(_Float16)(h2f_bits(rw[e]) + h2f_bits(mrow[j0 + e]));
                        if (v < -65504.0f) v = -65504.0f;
                        rw[e] = f2h_bits(v);
                    }
                *(u16x4*)(row + j0) = rw;                          // the same lane reads it back below
            }
        }
    }
    h2v mx2 = {(_Float16)(-__builtin_inff()), (_Float16)(-__builtin_inff())};
    for (int c0 = 0; c0 < nch; c0 += NB) {
        u32x2v raw[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) raw[k] = load(c0 + k);
#pragma unroll
        for (int k = 0; k < NB; k++) {
            if (dump) {                                            // (KIVI_GQA_DUMP_SCORES: a run-time pointer, null in production)
                const int j0 = (c0 + k) * 256 + lane * 4;
                if (j0 < n) *(u32x2v*)(dump + j0) = raw[k];
            }
            const uint32_t w0 = raw[k][0], w1 = raw[k][1];
            mx2 = __builtin_elementwise_max(mx2, __builtin_elementwise_max(__builtin_bit_cast(h2v, w0), __builtin_bit_cast(h2v, w1)));
        }
    }
    const uint32_t mb = __builtin_bit_cast(uint32_t, mx2);
    const float mx = wave_max(__builtin_fmaxf(h2f_bits((uint16_t)(mb & 0xFFFFu)), h2f_bits((uint16_t)(mb >> 16))));
    const float nmx = -mx;
    const f2v l2e = {1.44269504088896340736f, 1.44269504088896340736f};
    auto exps = [&](const u32x2v& raw, f2v& e01, f2v& e23) {       // kivi_exp(x - M) of the four scores of a chunk
        const f2v d01 = (f2v){mf_sub_lo(raw[0], nmx), mf_sub_hi(raw[0], nmx)} * l2e;
        const f2v d23 = (f2v){mf_sub_lo(raw[1], nmx), mf_sub_hi(raw[1], nmx)} * l2e;
        e01 = (f2v){__builtin_amdgcn_exp2f(d01[0]), __builtin_amdgcn_exp2f(d01[1])};
        e23 = (f2v){__builtin_amdgcn_exp2f(d23[0]), __builtin_amdgcn_exp2f(d23[1])};
    };
    // ---- pass B: sum of exp
    f2v acc = {0.f, 0.f};
    for (int c0 = 0; c0 < nch; c0 += NB) {
        u32x2v raw[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) raw[k] = load(c0 + k);
#pragma unroll
        for (int k = 0; k < NB; k++) {
            f2v e01, e23;
            exps(raw[k], e01, e23);
            acc += e01;
            acc += e23;
        }
    }
    const float sum = wave_sum(acc[0] + acc[1]);
    const float inv = 1.0f / sum;
    const int sp = mf_sp(sum, rsh);
    ksh = mf_ksh(sum, rsh);                                      // (what the scales of a big-scale unit still have to take: MfVStream::run)
    const _Float16 m_sp = mf_p_mul_sp(sp, rsh);
    const _Float16 m_a4 = mf_p_mul_a(BITS == 4, rsh), m_a6 = mf_p_mul_a(true, rsh);
    const f2v inv2 = {inv, inv};
    // ---- pass C: p = fp16(e / sum) (:375), p'' back into the row (zeros from Tv on), the window's probabilities into pw_row
    const int nchp = (n_pad + 255) >> 8;
    for (int c0 = 0; c0 < nchp; c0 += NB) {
        u32x2v raw[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) raw[k] = load(c0 + k);
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int j0 = (c0 + k) * 256 + lane * 4;
            if (j0 >= n_pad) continue;
            u32x2v o = {0u, 0u};
            if (j0 < n) {
                f2v e01, e23;
                exps(raw[k], e01, e23);
                const h2v p01 = __builtin_convertvector(e01 * inv2, h2v);
                const h2v p23 = __builtin_convertvector(e23 * inv2, h2v);
                if (j0 + 4 <= Tv) {
                    const _Float16 m_a = (j0 & 4) ? m_a6 : m_a4;
                    o[0] = __builtin_bit_cast(uint32_t, (p01 * (h2v){m_a, m_a}) * (h2v){m_sp, m_sp});
                    o[1] = __builtin_bit_cast(uint32_t, (p23 * (h2v){m_a, m_a}) * (h2v){m_sp, m_sp});
                } else {                                           // the chunk that holds the end of the packed prefix / the window
                    const uint32_t w01 = __builtin_bit_cast(uint32_t, p01), w23 = __builtin_bit_cast(uint32_t, p23);
                    const uint16_t pp[4] = {(uint16_t)(w01 & 0xFFFFu), (uint16_t)(w01 >> 16), (uint16_t)(w23 & 0xFFFFu), (uint16_t)(w23 >> 16)};
                    uint16_t q[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int j = j0 + e;
                        if (j >= Tv && j < n) pw_row[j - Tv] = pp[e];
                        q[e] = (j < Tv) ? mf_scale_p<BITS>(pp[e], sp, j) : (uint16_t)0;
                    }
                    o[0] = (uint32_t)q[0] | ((uint32_t)q[1] << 16);
                    o[1] = (uint32_t)q[2] | ((uint32_t)q[3] << 16);
                }
            }
            *(u32x2v*)(row + j0) = o;
        }
    }
    return sp;
}

}  // namespace

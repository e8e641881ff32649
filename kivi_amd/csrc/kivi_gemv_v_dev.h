// Device side of the packed-V product and of the attend half of the decode step (see kivi_gemv_v.hip for the mapping):
// argument block, the operands requested ahead of time by the fused decode-row kernel, and the per-block row body.
#pragma once
#include <type_traits>

#include "kivi_common.h"
#include "kivi_gemv_k_dev.h"
#include "kivi_quant.h"

namespace {

constexpr int64_t KIVI_WS_COUNTERS = 16384;   // arrival counters at the head of the caller's workspace (one per row unit)

struct GemvVArgs {
    const uint16_t* a;
    int64_t a_sb, a_sh;
    const uint32_t* code;
    int64_t code_sb, code_sh, code_sr;
    const uint16_t* scale;
    const uint16_t* mn;
    int64_t sm_sb, sm_sh, sm_sr;
    uint16_t* out;
    int64_t out_sb, out_sh;
    int nh, ratio, D;
    int64_t Tv;
    int units_per_b;
    uint32_t code_extent, sm_extent, a_extent;
    bool extents_ok;
    // fused decode step (kivi_decode_output): fp16 V window (llama_kivi.py:377-399)
    int fused;                         // 0 = plain GEMV
    uint16_t* vres;                    // (B, nh_kv, W, D) window buffer
    int64_t vres_sb, vres_sh, vres_st;
    int win_start, res_len;            // live rows [win_start, win_start + res_len); the new token goes right after
    const uint16_t* vnew;              // (B, nh_kv, D) the new value
    int64_t vnew_sb, vnew_sh;
    int flush;                         // quantise the oldest window row into cache row Tv
    // softmax folded into this launch (kivi_decode_softmax_output): `a` then points at the PRE-softmax score rows
    // written by kivi_decode_scores; every block turns its R rows into fp16 probabilities in LDS first.
    int softmax;
    int n_scores, n_pad;               // row length kv_len, LDS pitch (halves)
    float inv_scale;
    const uint16_t* mask;              // (B, 1, 1, n) additive fp16 mask or null
    int64_t mask_sb;
    // residual scores folded in as well (kivi_decode_attend): q . [fp16 K residual | new key] is computed by the
    // block that owns the row, written at a[..., Tq:] and fed to the softmax; the new key is appended (:333-337)
    const uint16_t* rq;                // (B, nh, D) queries, null = scores are already complete
    int64_t rq_sb, rq_sh;
    uint16_t* rkres;                   // (B, nh_kv, R_k, D) fp16 K residual buffer
    int64_t rk_sb, rk_sh, rk_st;
    const uint16_t* rknew;             // (B, nh_kv, D) the new key
    int64_t rkn_sb, rkn_sh;
    int rk_len;                        // keys already in the residual
    int Tq;                            // packed K length = offset of the residual scores in a row
    // split-T (SPLIT kernels): nsplit blocks share one (b, head unit); partial sums meet in `ws`
    int nsplit, cps;                   // blocks per unit, chunks (of TPI tokens) per block
    float* ws;                         // [units][nsplit + 1][R * D] fp32 partials (+1: the window part)
    int* counters;                     // [units] arrival counters, zero between launches
    size_t ws_bytes;                   // bytes available at ws
    // fused decode row (decode_row_kernel): the packed qK^T of the row ran in this block just before, its fp16 scores
    // are already in `pl` (the dynamic LDS row) and are not written to memory
    int scores_lds;
    unsigned long long* dbg;           // phase time stamps, normally null (see kivi_stamp)
    // host only: the packed-K side of the step when the caller handed it over (kivi_decode_attend with K fields)
    const struct KSide* kside;
};

struct KSide {     // host only
    GemvKArgs args;                    // filled for the paged layout, R = 1 mapping
    bool fusable;                      // static conditions of decode_row_kernel hold for the K side
    int64_t page_tokens;
    int B, nh_kv, group_size, bits;
};

// Small operands of a row's step, requested ahead of time by the fused decode-row kernel (before its qK^T phase) so that
// their memory round trips are over when v_row_body needs them: the first window rows of every wave, the window token
// about to be quantised, and this thread's share of the residual keys / query.
template <int D, int NW = 4>
struct RowPre {
    static constexpr int NP = (D / 2 + 63) / 64;
    static constexpr int PWT = (NW == 8) ? 5 : 9;    // window tokens per wave requested ahead: NW * PWT - 1 >= 33
    static constexpr int NK = D / 64;                // 16-byte pieces of a thread's D/8 channels
    uint32_t vpre[PWT][NP];
    uint16_t xflush;
    u16x8 rk[NK], rq[NK];
};

template <int D, int NW>
__device__ __forceinline__ void row_prefetch(const GemvVArgs& a, RowPre<D, NW>& pre) {   // R = 1, not split
    constexpr int NP = RowPre<D, NW>::NP, PWT = RowPre<D, NW>::PWT, NK = RowPre<D, NW>::NK, CPL = D / 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = (int)blockIdx.x;
    const int b = unit / a.units_per_b;
    const int h0 = unit - b * a.units_per_b;
    const int hk = h0 / a.ratio;
    const bool owner = (h0 % a.ratio) == 0;
    const int Lw = a.res_len + 1;
    const uint16_t* vwin = a.vres + b * a.vres_sb + hk * a.vres_sh + (int64_t)a.win_start * a.vres_st;
    const uint16_t* vnew = a.vnew + b * a.vnew_sb + hk * a.vnew_sh;
#pragma unroll
    for (int k = 0; k < PWT; k++) {
        const int t = wave + NW * k;
        const uint16_t* vrow = (t < a.res_len) ? vwin + (int64_t)t * a.vres_st : vnew;
#pragma unroll
        for (int c = 0; c < NP; c++) {
            const int p = lane + 64 * c;
            pre.vpre[k][c] = (a.fused && t < Lw && p < D / 2) ? *(const uint32_t*)(vrow + 2 * p) : 0u;
        }
    }
    pre.xflush = 0;
    if (a.fused && a.flush && owner && (int)threadIdx.x < D) pre.xflush = vwin[threadIdx.x];
    const int L = a.rk_len + 1;
    const int idx = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NK; i++) pre.rk[i] = pre.rq[i] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (a.rq && idx < L * 8) {
        const int sub = idx & 7, t = idx >> 3;
        const uint16_t* kres = a.rkres + b * a.rk_sb + hk * a.rk_sh;
        const uint16_t* knew = a.rknew + b * a.rkn_sb + hk * a.rkn_sh;
        const uint16_t* krow = ((t < a.rk_len) ? kres + (int64_t)t * a.rk_st : knew) + sub * CPL;
        const uint16_t* qrow = a.rq + b * a.rq_sb + (int64_t)h0 * a.rq_sh + sub * CPL;
#pragma unroll
        for (int i = 0; i < NK; i++) {
            pre.rk[i] = *(const u16x8*)(krow + 8 * i);
            pre.rq[i] = *(const u16x8*)(qrow + 8 * i);
        }
    }
}

template <int BITS, int G, int DW, int WPL, int R, int U, int MODE, bool NT, bool SPLIT, bool PRE = false, bool DBG = false,
          bool FROW = false, int EARLY = 0, int NW = 4, int DEPTH = 2>
__device__ __forceinline__ void v_row_body(const GemvVArgs& a, const RowPre<DW * (32 / BITS), NW>* pre = nullptr) {
    constexpr int NTH = NW * 64;                // threads of the block
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per block");
    constexpr int FPI = 32 / BITS;
    // R > 1 (grouped queries) and SPLIT kernels always get finished probabilities from the row-softmax launch (v_run):
    // the in-block softmax is compiled out of them
    constexpr bool CAN_SOFTMAX = (R == 1) && !SPLIT;
    constexpr int LPR = DW / WPL;               // lanes per token row
    static_assert(LPR >= 1 && LPR <= 16 && (LPR & (LPR - 1)) == 0, "D/fpi must be 4, 8, 16 or 32 words");
    typedef typename std::conditional<WPL == 4, u32x4, typename std::conditional<WPL == 2, u32x2, uint32_t>::type>::type WV;
    constexpr int TPI = 64 / LPR;               // tokens per wave-iteration
    constexpr int EPL = WPL * FPI;              // channels per lane
    constexpr int D = DW * FPI;
    constexpr int NGL = (EPL >= G) ? (EPL / G) : 1;
    static_assert(NGL == 1 || NGL == 2, "lane spans at most two groups");
    constexpr int NFIN = D / 64;                // channels per lane after the butterfly
    static_assert(NFIN >= 1, "head_dim >= 64");
    typedef typename std::conditional<NGL == 1, uint16_t, uint32_t>::type SV;

    __shared__ float red[NW][R][D];
    __shared__ float resl[NW][R][D];   // fused decode step: per-wave partial sums over the fp16 V window
    __shared__ float sm_lds[NW];
    extern __shared__ uint16_t pl[];  // [R][n_pad] fp16 probabilities when the softmax is folded in
    constexpr int RSMAX = 136;        // residual keys per row handled in LDS (R_k <= 128, + the new one)
    __shared__ uint16_t rs_lds[R][RSMAX];
    __shared__ int last_flag;
    auto sum_waves = [](const float (*buf)[R][D], int r, int d) {   // fixed tree over the per-wave partial sums
        const float lo = (buf[0][r][d] + buf[1][r][d]) + (buf[2][r][d] + buf[3][r][d]);
        if constexpr (NW == 8) return lo + ((buf[4][r][d] + buf[5][r][d]) + (buf[6][r][d] + buf[7][r][d]));
        else return lo;
    };

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // SPLIT: nsplit consecutive blocks share a (b, head unit), each takes a contiguous range of token chunks
    const int nsplit = SPLIT ? a.nsplit : 1;
    const int unit = SPLIT ? (int)blockIdx.x / nsplit : (int)blockIdx.x;
    const int split = SPLIT ? (int)blockIdx.x - unit * nsplit : 0;
    const int b = unit / a.units_per_b;
    const int hu = unit - b * a.units_per_b;
    const int h0 = hu * R;
    const int hk = h0 / a.ratio;
    const int lr = lane % LPR;                  // which 4-word slice of the row
    const int lt = lane / LPR;                  // token inside the iteration

    const rsrc_t rc = make_rsrc(a.code + b * a.code_sb + hk * a.code_sh, a.code_extent);
    const rsrc_t rs = make_rsrc(a.scale + b * a.sm_sb + hk * a.sm_sh, a.sm_extent);
    const rsrc_t rm = make_rsrc(a.mn + b * a.sm_sb + hk * a.sm_sh, a.sm_extent);
    // chunk c = TPI tokens; a SPLIT block owns chunks [c_begin, c_end) of its row
    const int nchunk = (int)((a.Tv + TPI - 1) / TPI);
    const int c_begin = SPLIT ? split * a.cps : 0;
    const int c_end = SPLIT ? ((c_begin + a.cps < nchunk) ? c_begin + a.cps : nchunk) : nchunk;
    // the probability rows are bounded at the END OF THIS BLOCK'S RANGE: a wave's last batch may reach into the next
    // block's chunks, those tokens then read probability 0 (hardware range check) and contribute nothing
    const uint32_t a_ext = SPLIT ? (uint32_t)__builtin_amdgcn_readfirstlane(
                                       (int)(((int64_t)c_end * TPI * 2 < (int64_t)a.a_extent) ? (int64_t)c_end * TPI * 2 : (int64_t)a.a_extent))
                                 : a.a_extent;
    rsrc_t ra[R];
#pragma unroll
    for (int r = 0; r < R; r++) ra[r] = make_rsrc(a.a + b * a.a_sb + (int64_t)(h0 + r) * a.a_sh, a_ext);

    const int gi0 = (lr * EPL) / G;             // first group index of this lane inside a row
    const uint32_t coff = (uint32_t)(((int64_t)lt * a.code_sr + lr * WPL) * 4);
    const uint32_t soff = (uint32_t)(((int64_t)lt * a.sm_sr + gi0) * 2);
    const uint32_t aoff = (uint32_t)(lt * 2);
    const uint32_t cstep = (uint32_t)(a.code_sr * 4 * TPI);  // bytes per chunk of TPI tokens
    const uint32_t sstep = (uint32_t)(a.sm_sr * 2 * TPI);
    const uint32_t astep = (uint32_t)(2 * TPI);

    float acc[R][EPL];
    float zacc[R][NGL];
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
        for (int i = 0; i < EPL; i++) acc[r][i] = 0.f;
#pragma unroll
        for (int g = 0; g < NGL; g++) zacc[r][g] = 0.f;
    }

    auto tok = [&](const WV& w, SV sraw, SV mraw, const uint16_t* av) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t ab = av[r];                      // fp16 bits of a[t] for this lane's token
            float as[NGL];
            as[0] = mul_hh_vv(ab, (uint32_t)sraw, false);   // exact fp16 x fp16 product, one instruction
            zacc[r][0] = fma_hh_vv(ab, (uint32_t)mraw, zacc[r][0], false);
            if constexpr (NGL == 2) {
                as[1] = mul_hh_vv(ab, (uint32_t)sraw, true);
                zacc[r][1] = fma_hh_vv(ab, (uint32_t)mraw, zacc[r][1], true);
            }
            if constexpr (qs_factor<MODE>() != 1.0f) {
#pragma unroll
                for (int g = 0; g < NGL; g++) as[g] *= qs_factor<MODE>();
            }
#pragma unroll
            for (int j = 0; j < WPL; j++) {
                const int g = (NGL == 1) ? 0 : (j * FPI) / G;
                uint32_t wj;
                if constexpr (WPL == 1) wj = w;
                else wj = w[j];
                accum_word<BITS, MODE>(wj, as[g], &acc[r][j * FPI]);
            }
        }
    };

    // wave w owns chunks w, w+NW, ... of the block's range; batch = U chunks of this wave
    const int nloc = c_end > c_begin ? c_end - c_begin : 0;
    const int my_chunks = (nloc > wave) ? (nloc - wave + NW - 1) / NW : 0;
    const int nbatch = (my_chunks + U - 1) / U;  // out-of-range chunks read zeros (bounds check)

    // The chunk offset goes into the (bounds-checked) per-lane voffset: soffset is excluded
    // from the hardware range check, and the tail relies on out-of-range rows reading 0.
    auto load_wsm = [&](int bi, WV* wb, SV* sb, SV* mb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t c = (uint32_t)(c_begin + (bi * U + u) * NW + wave);
            wb[u] = buf_load<WV, NT>(rc, coff + c * cstep, 0);
            sb[u] = buf_load<SV, NT>(rs, soff + c * sstep, 0);
            mb[u] = buf_load<SV, NT>(rm, soff + c * sstep, 0);
        }
    };
    auto load_a = [&](int bi, uint16_t (*ab)[R]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t c = (uint32_t)(c_begin + (bi * U + u) * NW + wave);
#pragma unroll
            for (int r = 0; r < R; r++) {
                if constexpr (FROW) {             // fused row (R = 1): always from this block's LDS row, 32-bit index, no branch
                    const int t = (int)c * TPI + lt;
                    ab[u][r] = (t < (int)a.Tv) ? pl[t] : (uint16_t)0;
                } else if (CAN_SOFTMAX && a.softmax) {   // probabilities produced by this block, in LDS
                    const int64_t t = (int64_t)c * TPI + lt;
                    ab[u][r] = (t < a.Tv) ? pl[(size_t)r * a.n_pad + t] : (uint16_t)0;
                } else {
                    ab[u][r] = buf_load<uint16_t, false>(ra[r], aoff + c * astep, 0);
                }
            }
        }
    };
    auto compute_batch = [&](const WV* wb, const SV* sb, const SV* mb, const uint16_t (*ab)[R]) {
#pragma unroll
        for (int u = 0; u < U; u++) tok(wb[u], sb[u], mb[u], ab[u]);
    };

    WV wA[U], wB[U];
    SV sA[U], sB[U], mA[U], mB[U];
    uint16_t aA[U][R], aB[U][R];

    // Everything small that the step needs besides the packed stream is REQUESTED first, in one go, so that none of
    // it waits behind the stream's own loads (the memory system is saturated once the stream runs, a dependent
    // round trip then costs several us): the score row(s), the fp16 V window, the window token about to be
    // quantised.  The consumers come later, in the order the data is needed.
    typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
    constexpr int SCH = NTH * 4;                // scores per softmax chunk (4 per thread)
    constexpr int SMC = 8192 / SCH;             // chunks of a row of <= 8192 scores
    const bool owner = (h0 % a.ratio) == 0;   // the first head unit of a kv head owns its cache writes
    const bool do_win = a.fused && split == 0;
    const bool do_flush = do_win && a.flush && owner;
    const bool reg_softmax = CAN_SOFTMAX && a.softmax;
    const int n_sc = a.n_scores;
    const int nch_sc = (n_sc + SCH - 1) / SCH;
    // (a) row 0 of the register-resident softmax: the part of the score row that is already in memory
    auto load_raw = [&](int r, u16x4* raw) {
        const uint16_t* srow = a.a + b * a.a_sb + (int64_t)(h0 + r) * a.a_sh;
        const int lim = a.rq ? (a.Tq < n_sc ? a.Tq : n_sc) : n_sc;   // scores from `lim` on are produced by this block
#pragma unroll
        for (int c = 0; c < SMC; c++) {
            const int j0 = c * SCH + (int)threadIdx.x * 4;
            raw[c] = u16x4{0, 0, 0, 0};
            if (c < nch_sc) {
                // fused row: the packed scores are in LDS already (FROW: known at compile time, so no global load -- and no
                // s_waitcnt vmcnt(0) that would drain the V batches requested ahead -- is left in the softmax)
                const uint16_t* src = (FROW || a.scores_lds) ? pl : srow;
                if (j0 + 4 <= lim) {
                    raw[c] = *(const u16x4*)(src + j0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (j0 + e < lim) raw[c][e] = src[j0 + e];
                }
            }
        }
    };
    // EARLY: the first (two) batch(es) of packed V are requested before anything else of this half: they fly during the
    // residual scores, the softmax and the window part
    if constexpr (EARLY >= 1) { if (nbatch > 0) load_wsm(0, wA, sA, mA); }
    if constexpr (EARLY >= 2) { if (nbatch > 1) load_wsm(1, wB, sB, mB); }
    u16x4 raw0[SMC];
    if (reg_softmax) load_raw(0, raw0);
    // (b) the first PWT window tokens of this wave (covers a window of 4*PWT-1 = 35 tokens; longer ones loop below)
    constexpr int NP = (D / 2 + 63) / 64;            // channel pairs per lane
    constexpr int PWT = RowPre<D, NW>::PWT;
    const int Lw = a.res_len + 1;
    uint32_t vpre[PWT][NP];
    if constexpr (PRE) {
#pragma unroll
        for (int k = 0; k < PWT; k++)
#pragma unroll
            for (int c = 0; c < NP; c++) vpre[k][c] = pre->vpre[k][c];
    } else if (do_win) {
        const uint16_t* vwin = a.vres + b * a.vres_sb + hk * a.vres_sh + (int64_t)a.win_start * a.vres_st;
        const uint16_t* vnew = a.vnew + b * a.vnew_sb + hk * a.vnew_sh;
#pragma unroll
        for (int k = 0; k < PWT; k++) {
            const int t = wave + NW * k;
            const uint16_t* vrow = (t < a.res_len) ? vwin + (int64_t)t * a.vres_st : vnew;
#pragma unroll
            for (int c = 0; c < NP; c++) {
                const int p = lane + 64 * c;
                vpre[k][c] = (t < Lw && p < D / 2) ? *(const uint32_t*)(vrow + 2 * p) : 0u;
            }
        }
    }
    // (c) the oldest window token, quantised below
    uint16_t xflush = 0;
    if constexpr (PRE) xflush = pre->xflush;
    else if (do_flush && (int)threadIdx.x < D)
        xflush = a.vres[b * a.vres_sb + hk * a.vres_sh + (int64_t)a.win_start * a.vres_st + threadIdx.x];

    if (reg_softmax) {
        // scale + mask + softmax of this block's R score rows, the arithmetic of kivi_softmax_scaled (same element ->
        // thread assignment and reduction tree, so the probabilities are bit-identical to the stand-alone kernel).
        // The whole row (<= 8192 scores) is fetched with up to 8 independent 8-byte loads per thread: one L2 round trip.
        const int n = n_sc;
        const int nch = nch_sc;
        const uint16_t* mrow = a.mask ? a.mask + b * a.mask_sb : nullptr;
        const bool owner_k = owner;
        if (a.rq) {
            // q . k over the fp16 residual keys and the new key: one thread per (head, key), 16-byte loads,
            // fp32 accumulate, one rounding (the reference's fp16 torch.matmul, :337); kept in LDS for the softmax and
            // also written to the score row
            const int L = a.rk_len + 1;
            const uint16_t* knew = a.rknew + b * a.rkn_sb + hk * a.rkn_sh;
            uint16_t* kres = a.rkres + b * a.rk_sb + hk * a.rk_sh;
            // 8 lanes per (head, key): each takes D/8 channels with 16-byte loads, then a 3-step shuffle reduction
            constexpr int CPL = D / 8;                       // channels per lane (D % 64 == 0)
            for (int idx = threadIdx.x; idx < R * L * 8; idx += NTH) {
                const int sub = idx & 7, rt = idx >> 3;
                const int r = rt / L, t = rt - r * L;
                const uint16_t* krow = ((t < a.rk_len) ? kres + (int64_t)t * a.rk_st : knew) + sub * CPL;
                const uint16_t* qrow = a.rq + b * a.rq_sb + (int64_t)(h0 + r) * a.rq_sh + sub * CPL;
                const bool append = (t == a.rk_len) && owner_k && r == 0 && split == 0;
                float sc = 0.f;
#pragma unroll
                for (int d = 0; d < CPL; d += 8) {
                    u16x8 kv, qv;
                    if (PRE && idx == (int)threadIdx.x) {   // this thread's first item was requested before the qK^T phase
                        kv = pre->rk[d / 8];
                        qv = pre->rq[d / 8];
                    } else {
                        kv = *(const u16x8*)(krow + d);
                        qv = *(const u16x8*)(qrow + d);
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++) sc = __builtin_fmaf(h2f_bits(qv[e]), h2f_bits(kv[e]), sc);
                    if (append) *(u16x8*)(kres + (int64_t)t * a.rk_st + sub * CPL + d) = kv;
                }
                sc += __shfl_xor(sc, 1);
                sc += __shfl_xor(sc, 2);
                sc += __shfl_xor(sc, 4);
                if (sub == 0) {
                    const uint16_t hs = f2h_bits(sc);
                    rs_lds[r][t] = hs;
                    if (!FROW && !a.scores_lds) const_cast<uint16_t*>(a.a)[b * a.a_sb + (int64_t)(h0 + r) * a.a_sh + a.Tq + t] = hs;
                }
            }
        }
        // the first batch of packed V is requested before the softmax arithmetic so the stream is already moving
        if constexpr (EARLY == 0) { if (nbatch > 0) load_wsm(0, wA, sA, mA); }
        kivi_stamp<DBG>(a.dbg, 6);
        if (a.rq) __syncthreads();
        {
#pragma unroll 1
        for (int r = 0; r < R; r++) {
            uint16_t* prow = pl + (size_t)r * a.n_pad;
            u16x4 raw[SMC];
            if (r == 0) {
#pragma unroll
                for (int c = 0; c < SMC; c++) raw[c] = raw0[c];
            } else {
                load_raw(r, raw);
            }
            if (a.rq) {   // the residual range was produced by this block: take it from LDS
#pragma unroll
                for (int c = 0; c < SMC; c++) {
                    const int j0 = c * SCH + (int)threadIdx.x * 4;
                    if (c < nch && j0 + 4 > a.Tq) {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (j0 + e >= a.Tq && j0 + e < n) raw[c][e] = rs_lds[r][j0 + e - a.Tq];
                    }
                }
            }
            float x[SMC][4];
            float mx = -__builtin_inff();
#pragma unroll
            for (int c = 0; c < SMC; c++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int j = c * SCH + (int)threadIdx.x * 4 + e;
                    float v = -__builtin_inff();
                    if (c < nch && j < n)
                        v = h2f_bits(kivi_scaled_score(raw[c][e], a.inv_scale, mrow != nullptr, mrow ? mrow[j] : 0));
                    x[c][e] = v;
                    mx = __builtin_fmaxf(mx, v);
                }
            mx = kivi_block_reduce<NW>(mx, true, sm_lds);
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < SMC; c++)
                if (c < nch)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        x[c][e] = kivi_exp(x[c][e] - mx);
                        sum += x[c][e];
                    }
            sum = kivi_block_reduce<NW>(sum, false, sm_lds);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int c = 0; c < SMC; c++)
                if (c < nch) {
                    const int j0 = c * SCH + (int)threadIdx.x * 4;
                    u16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = f2h_bits(x[c][e] * inv);
                    if (j0 < a.n_pad) *(u16x4*)(prow + j0) = o;   // n_pad is a multiple of 8: whole vectors stay inside the row
                }
        }
        }
        __syncthreads();
        kivi_stamp<DBG>(a.dbg, 7);
    } else {
        if constexpr (EARLY == 0) { if (nbatch > 0) load_wsm(0, wA, sA, mA); }
    }

    if (do_win) {
        // probs[..., -L:] @ V_window (llama_kivi.py:384): the <= R+1 fp16 window tokens (the last one is the new
        // value, appended here, :377) are spread over the 4 waves, a lane owns channel pairs.  Done BEFORE the
        // stream from the rows requested at the top, so nothing but the packed sum is left for the tail.
        float racc[R][NP][2];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int c = 0; c < NP; c++) racc[r][c][0] = racc[r][c][1] = 0.f;
        const uint16_t* vwin = a.vres + b * a.vres_sb + hk * a.vres_sh + (int64_t)a.win_start * a.vres_st;
        const uint16_t* vnew = a.vnew + b * a.vnew_sb + hk * a.vnew_sh;
        auto win_tok = [&](int t, const uint32_t* vv_c) {
            float at[R];
#pragma unroll
            for (int r = 0; r < R; r++)
                at[r] = h2f_bits(!reg_softmax ? a.a[b * a.a_sb + (int64_t)(h0 + r) * a.a_sh + a.Tv + t]
                                 : pl[(size_t)r * a.n_pad + a.Tv + t]);
#pragma unroll
            for (int c = 0; c < NP; c++) {
                const int p = lane + 64 * c;
                if (p < D / 2) {
                    const uint32_t vv = vv_c[c];
                    const float v0 = h2f_bits((uint16_t)(vv & 0xFFFFu)), v1 = h2f_bits((uint16_t)(vv >> 16));
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        racc[r][c][0] = __builtin_fmaf(at[r], v0, racc[r][c][0]);
                        racc[r][c][1] = __builtin_fmaf(at[r], v1, racc[r][c][1]);
                    }
                    if (t == a.res_len && owner)
                        *(uint32_t*)(a.vres + b * a.vres_sb + hk * a.vres_sh + (int64_t)(a.win_start + t) * a.vres_st + 2 * p) = vv;
                }
            }
        };
#pragma unroll
        for (int k = 0; k < PWT; k++) {
            const int t = wave + NW * k;
            if (t < Lw) win_tok(t, vpre[k]);
        }
        // longer windows (residual_length 64 / 128): PWT rows per round, all their loads in flight together
        constexpr int PW2 = (R >= 4) ? 4 : PWT;   // the R x EPL accumulators of the grouped-query kernels leave fewer registers
        for (int k0 = PWT; wave + NW * k0 < Lw; k0 += PW2) {
            uint32_t vb[PW2][NP];
#pragma unroll
            for (int k = 0; k < PW2; k++) {
                const int t = wave + NW * (k0 + k);
                const uint16_t* vrow = (t < a.res_len) ? vwin + (int64_t)t * a.vres_st : vnew;
#pragma unroll
                for (int c = 0; c < NP; c++) {
                    const int p = lane + 64 * c;
                    vb[k][c] = (t < Lw && p < D / 2) ? *(const uint32_t*)(vrow + 2 * p) : 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < PW2; k++) {
                const int t = wave + NW * (k0 + k);
                if (t < Lw) win_tok(t, vb[k]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int c = 0; c < NP; c++) {
                const int p = lane + 64 * c;
                if (p < D / 2) {
                    resl[wave][r][2 * p] = racc[r][c][0];
                    resl[wave][r][2 * p + 1] = racc[r][c][1];
                }
            }
    }
    if (do_flush) {
        // the window now holds R+1 tokens: quantise the OLDEST one into cache row Tv (:386-399), bit-identical
        // to the stand-alone pack kernel (shared quantiser).  Thread d owns channel d; group min / max and the word
        // assembly go through lane shuffles (G <= 64: a group never leaves a wave), so no barrier is spent here.
        const int d = threadIdx.x;
        if constexpr (G <= 64) {
            if (d < D) {   // wave-uniform: D is a multiple of 64
                const uint32_t key = h_key(xflush);
                uint32_t kmin = key, kmax = key;
#pragma unroll
                for (int m = 1; m < G; m <<= 1) {
                    const uint32_t o1 = (uint32_t)__shfl_xor((int)kmin, m), o2 = (uint32_t)__shfl_xor((int)kmax, m);
                    kmin = o1 < kmin ? o1 : kmin;
                    kmax = o2 > kmax ? o2 : kmax;
                }
                const GroupQ gq = make_group(kmin, kmax, (1 << BITS) - 1);
                uint32_t word = quant_one<BITS>(xflush, gq) << (BITS * (d % FPI));
#pragma unroll
                for (int m = 1; m < FPI; m <<= 1) word |= (uint32_t)__shfl_xor((int)word, m);
                if ((d % FPI) == 0)
                    const_cast<uint32_t*>(a.code)[b * a.code_sb + hk * a.code_sh + a.Tv * a.code_sr + d / FPI] = word;
                if ((d % G) == 0) {
                    const int64_t so = b * a.sm_sb + hk * a.sm_sh + a.Tv * a.sm_sr + d / G;
                    const_cast<uint16_t*>(a.scale)[so] = gq.scale;
                    const_cast<uint16_t*>(a.mn)[so] = gq.mn;
                }
            }
        } else {
            uint32_t* lds = reinterpret_cast<uint32_t*>(&red[0][0][0]);   // D keys, then D codes (red is still unused)
            if (d < D) lds[d] = h_key(xflush);
            __syncthreads();
            GroupQ gq;
            uint32_t c = 0;
            if (d < D) {
                uint32_t kmin = 0xFFFFu, kmax = 0u;
                const int g0 = (d / G) * G;
                for (int i = 0; i < G; i++) {
                    const uint32_t k = lds[g0 + i];
                    kmin = k < kmin ? k : kmin;
                    kmax = k > kmax ? k : kmax;
                }
                gq = make_group(kmin, kmax, (1 << BITS) - 1);
                c = quant_one<BITS>(xflush, gq);
            }
            __syncthreads();
            if (d < D) lds[d] = c;
            __syncthreads();
            if (d < DW) {
                uint32_t word = 0;
#pragma unroll
                for (int i = 0; i < FPI; i++) word |= lds[d * FPI + i] << (BITS * i);
                const_cast<uint32_t*>(a.code)[b * a.code_sb + hk * a.code_sh + a.Tv * a.code_sr + d] = word;
            }
            if (d < D && (d % G) == 0) {
                const int64_t so = b * a.sm_sb + hk * a.sm_sh + a.Tv * a.sm_sr + d / G;
                const_cast<uint16_t*>(a.scale)[so] = gq.scale;
                const_cast<uint16_t*>(a.mn)[so] = gq.mn;
            }
            __syncthreads();
        }
    }

    kivi_stamp<DBG>(a.dbg, 8);
    if constexpr (DEPTH == 3) {
        // Two batches ahead of the one being consumed.  A wave that streams ALONE (the last blocks of a launch: co-resident
        // blocks are served oldest first) is bound by one memory round trip per batch; the third buffer halves that.
        // Loads past the row's end are dropped by the buffer range check (zeros, no traffic).
        static_assert(!SPLIT && EARLY == 0, "three-deep ring: unsplit rows only");
        WV wC[U];
        SV sC[U], mC[U];
        uint16_t aC[U][R];
        load_wsm(1, wB, sB, mB);
        int it = 0;
        for (; it + 3 <= nbatch; it += 3) {
            load_wsm(it + 2, wC, sC, mC);
            load_a(it, aA);
            compute_batch(wA, sA, mA, aA);
            load_wsm(it + 3, wA, sA, mA);
            load_a(it + 1, aB);
            compute_batch(wB, sB, mB, aB);
            load_wsm(it + 4, wB, sB, mB);
            load_a(it + 2, aC);
            compute_batch(wC, sC, mC, aC);
        }
        if (it < nbatch) {
            load_a(it, aA);
            compute_batch(wA, sA, mA, aA);
        }
        if (it + 1 < nbatch) {
            load_a(it + 1, aB);
            compute_batch(wB, sB, mB, aB);
        }
    } else {
        if (nbatch > 0) load_a(0, aA);
        int it = 0;
        for (; it + 2 <= nbatch; it += 2) {
            if (EARLY < 2 || it > 0) load_wsm(it + 1, wB, sB, mB);
            load_a(it + 1, aB);
            compute_batch(wA, sA, mA, aA);
            if (it + 2 < nbatch) {
                load_wsm(it + 2, wA, sA, mA);
                load_a(it + 2, aA);
            }
            compute_batch(wB, sB, mB, aB);
        }
        if (it < nbatch) compute_batch(wA, sA, mA, aA);
    }
    kivi_stamp<DBG>(a.dbg, 9);

    // undo the positional power-of-two factors, then combine the TPI lanes that share `lr`
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
        for (int i = 0; i < EPL; i++) acc[r][i] *= post_scale<BITS, MODE>(i % FPI);
#pragma unroll
        for (int g = 0; g < NGL; g++) {
#pragma unroll
            for (int m = LPR; m < 64; m <<= 1) zacc[r][g] += __shfl_xor(zacc[r][g], m);
        }
    }
    int eoff = 0;  // first surviving channel (inside this lane's EPL slice)
#pragma unroll
    for (int r = 0; r < R; r++) {
        int n = EPL;
        int off = 0;
#pragma unroll
        for (int m = LPR; m < 64; m <<= 1) {
            const int half = n / 2;
            const bool upper = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < half; i++) {
                const float send = upper ? acc[r][i] : acc[r][i + half];
                const float keep = upper ? acc[r][i + half] : acc[r][i];
                acc[r][i] = keep + __shfl_xor(send, m);
            }
            off += upper ? half : 0;
            n = half;
        }
        eoff = off;
    }
    // lane now holds NFIN channels: d = lr*EPL + eoff + i
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int i = 0; i < NFIN; i++) {
            const int e = eoff + i;
            const int g = (NGL == 1) ? 0 : (e / G);
            // select without dynamic register indexing
            float z = zacc[r][0];
            if constexpr (NGL == 2) z = (g == 1) ? zacc[r][1] : zacc[r][0];
            red[wave][r][lr * EPL + e] = acc[r][i] + z;
        }
    __syncthreads();
    kivi_stamp<DBG>(a.dbg, 10);
    if constexpr (!SPLIT) {
        for (int i = threadIdx.x; i < R * D; i += NTH) {
            const int r = i / D, d = i - r * D;
            const float s = sum_waves(red, r, d);
            uint16_t o = f2h_bits(s);
            if (a.fused) {
                const float res = sum_waves(resl, r, d);
                // fp16(quantised part) + fp16(window part), rounded: the reference's `attn_output += matmul(...)` (:382-384);
                // only the window part exists before anything is quantised (:380)
                o = (a.Tv > 0) ? f2h_bits(h2f_bits(o) + h2f_bits(f2h_bits(res))) : f2h_bits(res);
            }
            a.out[b * a.out_sb + (int64_t)(h0 + r) * a.out_sh + d] = o;
        }
    } else {
        // partial sums of this block -> workspace; the block that arrives last adds them up in split order and
        // writes the row.  Hand-off without fences (cdna_hip_programming.md G16, "write-through payload"): the partials
        // are stored write-through (agent-scope relaxed atomic stores = sc1), every wave drains its stores, one lane
        // bumps the arrival counter; the last arriver reads them with agent-scope (sc1, L1-bypassing) loads.
        uint32_t* part = reinterpret_cast<uint32_t*>(a.ws + ((size_t)unit * (nsplit + 1) + split) * (R * D));
        uint32_t* winp = reinterpret_cast<uint32_t*>(a.ws + ((size_t)unit * (nsplit + 1) + nsplit) * (R * D));
        for (int i = threadIdx.x; i < R * D; i += NTH) {
            const int r = i / D, d = i - r * D;
            const float qs = sum_waves(red, r, d);
            __hip_atomic_store(part + i, __builtin_bit_cast(uint32_t, qs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.fused && split == 0) {
                const float ws_ = sum_waves(resl, r, d);
                __hip_atomic_store(winp + i, __builtin_bit_cast(uint32_t, ws_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(a.counters + unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (old == nsplit - 1);
            if (last) __hip_atomic_store(a.counters + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
            last_flag = last;
        }
        __syncthreads();
        if (last_flag) {
            const uint32_t* p0 = reinterpret_cast<const uint32_t*>(a.ws + (size_t)unit * (nsplit + 1) * (R * D));
            for (int i = threadIdx.x; i < R * D; i += NTH) {
                const int r = i / D, d = i - r * D;
                float s = 0.f;
                for (int sp0 = 0; sp0 < nsplit; sp0 += 8) {   // 8 independent loads in flight, added in split order
                    uint32_t v8[8];
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        v8[k] = (sp0 + k < nsplit) ? __hip_atomic_load(p0 + (size_t)(sp0 + k) * (R * D) + i, __ATOMIC_RELAXED,
                                                                       __HIP_MEMORY_SCOPE_AGENT)
                                                   : 0u;
#pragma unroll
                    for (int k = 0; k < 8; k++) s += __builtin_bit_cast(float, v8[k]);
                }
                uint16_t o = f2h_bits(s);
                if (a.fused) {
                    const float res = __builtin_bit_cast(float, __hip_atomic_load(p0 + (size_t)nsplit * (R * D) + i,
                                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    o = (a.Tv > 0) ? f2h_bits(h2f_bits(o) + h2f_bits(f2h_bits(res))) : f2h_bits(res);
                }
                a.out[b * a.out_sb + (int64_t)(h0 + r) * a.out_sh + d] = o;
            }
        }
    }
}


}  // namespace

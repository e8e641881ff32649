// sV over the packed per-token V cache (hook-state layout), gfx950.
//
// Replaces cuda_bmm_fA_qB_outer + bgemv{2,4}_kernel_outer_dim of the reference
// (quant/matmul.py:178-219, quant/csrc/gemv_cuda.cu:265-427) for the call at
// models/llama_kivi.py:382, reading V_code (B,nh_kv,Tv,D/fpi) directly.
//
// Mapping (wave64).  Here the dot axis t runs ACROSS lanes: one wave-instruction
// reads 64 x 16 contiguous bytes = TPI = 64/LPR whole token rows of codes (LPR
// lanes per row, 4 words per lane), plus the matching scale / mn / a entries.
// A lane accumulates its EPL = 4*fpi channels over every TPI-th token in fp32
// (one mask per two codes + one v_fma_mix_f32 per code, a*scale folded once per
// (token, group), zero-point term hoisted), then the 64/LPR lanes that own the
// same channels are combined with a halving butterfly (EPL-EPL/TPI shuffles
// instead of EPL*log2(TPI)) and the 4 waves of the block through LDS.
#include <stdlib.h>
#include <string.h>

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
template <int D>
struct RowPre {
    static constexpr int NP = (D / 2 + 63) / 64;
    static constexpr int PWT = 9;
    static constexpr int NK = D / 64;                // 16-byte pieces of a thread's D/8 channels
    uint32_t vpre[PWT][NP];
    uint16_t xflush;
    u16x8 rk[NK], rq[NK];
};

template <int D>
__device__ __forceinline__ void row_prefetch(const GemvVArgs& a, RowPre<D>& pre) {   // R = 1, not split
    constexpr int NP = RowPre<D>::NP, PWT = RowPre<D>::PWT, NK = RowPre<D>::NK, CPL = D / 8;
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
        const int t = wave + 4 * k;
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

template <int BITS, int G, int DW, int WPL, int R, int U, int MODE, bool NT, bool SPLIT, bool PRE = false>
__device__ __forceinline__ void v_row_body(const GemvVArgs& a, const RowPre<DW * (32 / BITS)>* pre = nullptr) {
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

    __shared__ float red[4][R][D];
    __shared__ float resl[4][R][D];   // fused decode step: per-wave partial sums over the fp16 V window
    __shared__ float sm_lds[4];
    extern __shared__ uint16_t pl[];  // [R][n_pad] fp16 probabilities when the softmax is folded in
    constexpr int RSMAX = 136;        // residual keys per row handled in LDS (R_k <= 128, + the new one)
    __shared__ uint16_t rs_lds[R][RSMAX];
    __shared__ int last_flag;

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

    // wave w owns chunks w, w+4, ... of the block's range; batch = U chunks of this wave
    const int nloc = c_end > c_begin ? c_end - c_begin : 0;
    const int my_chunks = (nloc > wave) ? (nloc - wave + 3) / 4 : 0;
    const int nbatch = (my_chunks + U - 1) / U;  // out-of-range chunks read zeros (bounds check)

    // The chunk offset goes into the (bounds-checked) per-lane voffset: soffset is excluded
    // from the hardware range check, and the tail relies on out-of-range rows reading 0.
    auto load_wsm = [&](int bi, WV* wb, SV* sb, SV* mb) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t c = (uint32_t)(c_begin + (bi * U + u) * 4 + wave);
            wb[u] = buf_load<WV, NT>(rc, coff + c * cstep, 0);
            sb[u] = buf_load<SV, NT>(rs, soff + c * sstep, 0);
            mb[u] = buf_load<SV, NT>(rm, soff + c * sstep, 0);
        }
    };
    auto load_a = [&](int bi, uint16_t (*ab)[R]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t c = (uint32_t)(c_begin + (bi * U + u) * 4 + wave);
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (CAN_SOFTMAX && a.softmax) {   // probabilities produced by this block, in LDS
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
    constexpr int SMC = 8;
    const bool owner = (h0 % a.ratio) == 0;   // the first head unit of a kv head owns its cache writes
    const bool do_win = a.fused && split == 0;
    const bool do_flush = do_win && a.flush && owner;
    const bool reg_softmax = CAN_SOFTMAX && a.softmax;
    const int n_sc = a.n_scores;
    const int nch_sc = (n_sc + 1023) / 1024;
    // (a) row 0 of the register-resident softmax: the part of the score row that is already in memory
    auto load_raw = [&](int r, u16x4* raw) {
        const uint16_t* srow = a.a + b * a.a_sb + (int64_t)(h0 + r) * a.a_sh;
        const int lim = a.rq ? (a.Tq < n_sc ? a.Tq : n_sc) : n_sc;   // scores from `lim` on are produced by this block
#pragma unroll
        for (int c = 0; c < SMC; c++) {
            const int j0 = c * 1024 + (int)threadIdx.x * 4;
            raw[c] = u16x4{0, 0, 0, 0};
            if (c < nch_sc) {
                const uint16_t* src = a.scores_lds ? pl : srow;   // fused row: the packed scores are in LDS already
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
    u16x4 raw0[SMC];
    if (reg_softmax) load_raw(0, raw0);
    // (b) the first PWT window tokens of this wave (covers a window of 4*PWT-1 = 35 tokens; longer ones loop below)
    constexpr int NP = (D / 2 + 63) / 64;            // channel pairs per lane
    constexpr int PWT = 9;
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
            const int t = wave + 4 * k;
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
            for (int idx = threadIdx.x; idx < R * L * 8; idx += 256) {
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
                    if (!a.scores_lds) const_cast<uint16_t*>(a.a)[b * a.a_sb + (int64_t)(h0 + r) * a.a_sh + a.Tq + t] = hs;
                }
            }
        }
        // the first batch of packed V is requested before the softmax arithmetic so the stream is already moving
        if (nbatch > 0) load_wsm(0, wA, sA, mA);
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
                    const int j0 = c * 1024 + (int)threadIdx.x * 4;
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
                    const int j = c * 1024 + (int)threadIdx.x * 4 + e;
                    float v = -__builtin_inff();
                    if (c < nch && j < n)
                        v = h2f_bits(kivi_scaled_score(raw[c][e], a.inv_scale, mrow != nullptr, mrow ? mrow[j] : 0));
                    x[c][e] = v;
                    mx = __builtin_fmaxf(mx, v);
                }
            mx = kivi_block_reduce(mx, true, sm_lds);
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < SMC; c++)
                if (c < nch)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        x[c][e] = kivi_exp(x[c][e] - mx);
                        sum += x[c][e];
                    }
            sum = kivi_block_reduce(sum, false, sm_lds);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int c = 0; c < SMC; c++)
                if (c < nch) {
                    const int j0 = c * 1024 + (int)threadIdx.x * 4;
                    u16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = f2h_bits(x[c][e] * inv);
                    if (j0 < a.n_pad) *(u16x4*)(prow + j0) = o;   // n_pad is a multiple of 8: whole vectors stay inside the row
                }
        }
        }
        __syncthreads();
    } else {
        if (nbatch > 0) load_wsm(0, wA, sA, mA);
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
            const int t = wave + 4 * k;
            if (t < Lw) win_tok(t, vpre[k]);
        }
        // longer windows (residual_length 64 / 128): PWT rows per round, all their loads in flight together
        constexpr int PW2 = (R >= 4) ? 4 : PWT;   // the R x EPL accumulators of the grouped-query kernels leave fewer registers
        for (int k0 = PWT; wave + 4 * k0 < Lw; k0 += PW2) {
            uint32_t vb[PW2][NP];
#pragma unroll
            for (int k = 0; k < PW2; k++) {
                const int t = wave + 4 * (k0 + k);
                const uint16_t* vrow = (t < a.res_len) ? vwin + (int64_t)t * a.vres_st : vnew;
#pragma unroll
                for (int c = 0; c < NP; c++) {
                    const int p = lane + 64 * c;
                    vb[k][c] = (t < Lw && p < D / 2) ? *(const uint32_t*)(vrow + 2 * p) : 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < PW2; k++) {
                const int t = wave + 4 * (k0 + k);
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

    {
        if (nbatch > 0) load_a(0, aA);
        int it = 0;
        for (; it + 2 <= nbatch; it += 2) {
            load_wsm(it + 1, wB, sB, mB);
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
    if constexpr (!SPLIT) {
        for (int i = threadIdx.x; i < R * D; i += 256) {
            const int r = i / D, d = i - r * D;
            const float s = (red[0][r][d] + red[1][r][d]) + (red[2][r][d] + red[3][r][d]);
            uint16_t o = f2h_bits(s);
            if (a.fused) {
                const float res = (resl[0][r][d] + resl[1][r][d]) + (resl[2][r][d] + resl[3][r][d]);
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
        for (int i = threadIdx.x; i < R * D; i += 256) {
            const int r = i / D, d = i - r * D;
            const float qs = (red[0][r][d] + red[1][r][d]) + (red[2][r][d] + red[3][r][d]);
            __hip_atomic_store(part + i, __builtin_bit_cast(uint32_t, qs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.fused && split == 0) {
                const float ws_ = (resl[0][r][d] + resl[1][r][d]) + (resl[2][r][d] + resl[3][r][d]);
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
            for (int i = threadIdx.x; i < R * D; i += 256) {
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

template <int BITS, int G, int DW, int WPL, int R, int U, int MODE, bool NT, bool SPLIT>
__global__ __launch_bounds__(256) void gemv_v_kernel(const GemvVArgs a) {
    v_row_body<BITS, G, DW, WPL, R, U, MODE, NT, SPLIT>(a);
}

// The whole decode step of one (b, head) row in ONE block (MHA, rows <= 8192 keys): the packed qK^T of the row tile
// by tile (k_tile_body, the scores go to the LDS row instead of memory), then everything v_row_body does with them
// (residual scores, softmax, window, packed sV).  Against the two-launch form this drops the 2 x 8 MB score round
// trip through HBM, one launch ramp/drain and the cold start of the second kernel.
template <int BITS, int G, int DW, int KWPL, int KDS, int KU, int VWPL, int VU, bool PRE = true>
__global__ __launch_bounds__(256) void decode_row_kernel(const GemvKArgs ak, const GemvVArgs av) {
    extern __shared__ uint16_t pl_row[];
    const int unit = (int)blockIdx.x;
    RowPre<DW * (32 / BITS)> pre;
    if constexpr (PRE) row_prefetch(av, pre);   // in flight during the whole qK^T phase (26 VGPRs, where they fit)
    for (int tb = 0; tb < ak.tile_blocks; tb++) {
        k_tile_body<BITS, G, KWPL, KDS, 1, KU, KIVI_UNPACK_MIX, true>(ak, unit * ak.tile_blocks + tb, pl_row);
        __syncthreads();   // the exchange buffer is reused by the next tile; the scores must be visible below
    }
    v_row_body<BITS, G, DW, VWPL, 1, VU, KIVI_UNPACK_MIX, true, false, PRE>(av, &pre);
}

// Stand-alone row softmax of the decode step, used when the block-prologue softmax of gemv_v_kernel does not pay
// (grouped queries: R rows per block; rows longer than the register-resident form; rows split over blocks): one
// block per (b, h) score row computes the residual scores q . [fp16 K residual | new key] (+ the K append) when a
// query is given, then scale + mask + softmax with exactly the element order and reduction tree of
// kivi_softmax_scaled, and overwrites the score row with the fp16 probabilities (llama_kivi.py:339, :364-375).
struct RowSoftmaxArgs {
    uint16_t* scores;
    int64_t s_sb, s_sh;
    int n, Tq;
    float inv_scale;
    const uint16_t* mask;
    int64_t mask_sb;
    const uint16_t* q;                 // null: the score rows are complete (no residual part to compute)
    int64_t q_sb, q_sh;
    uint16_t* kres;
    int64_t k_sb, k_sh, k_st;
    const uint16_t* knew;
    int64_t kn_sb, kn_sh;
    int rk_len, ratio, nh, D;
    // long rows / few rows: P blocks per row, each owning `chunk` scores (the last one the rest, incl. the residual
    // part); launch 1 leaves (max, sum exp) of every chunk in `partial`, launch 2 combines them and normalises.
    int P, chunk;
    float* partial;                    // [rows][P][2]
};

// MODE 0: one block per row does everything.  MODE 1: chunk statistics.  MODE 2: combine + normalise the chunk.
template <int MODE>
__global__ __launch_bounds__(256) void row_softmax_kernel(const RowSoftmaxArgs p) {
    typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
    __shared__ float sm_lds[4];
    __shared__ uint16_t rs_lds[136];
    const int row = (MODE == 0) ? (int)blockIdx.x : (int)blockIdx.x / p.P;
    const int c = (MODE == 0) ? 0 : (int)blockIdx.x - row * p.P;
    const int P = (MODE == 0) ? 1 : p.P;
    const int b = row / p.nh, h = row - b * p.nh;
    const int hk = h / p.ratio;
    uint16_t* srow = p.scores + b * p.s_sb + (int64_t)h * p.s_sh;
    const uint16_t* mrow = p.mask ? p.mask + b * p.mask_sb : nullptr;
    const int n = p.n;
    const int lo = c * p.chunk;                               // multiple of 1024: the 8-byte loads stay aligned
    const int hi = (c == P - 1) ? n : lo + p.chunk;
    const bool res_here = (MODE != 2) && p.q != nullptr && c == P - 1;   // the last chunk contains [Tq, n)
    if (res_here) {
        const int L = p.rk_len + 1;
        const uint16_t* knew = p.knew + b * p.kn_sb + hk * p.kn_sh;
        uint16_t* kres = p.kres + b * p.k_sb + hk * p.k_sh;
        const int cpl = p.D / 8;
        for (int idx = threadIdx.x; idx < L * 8; idx += 256) {
            const int sub = idx & 7, t = idx >> 3;
            const uint16_t* krow = ((t < p.rk_len) ? kres + (int64_t)t * p.k_st : knew) + sub * cpl;
            const uint16_t* qrow = p.q + b * p.q_sb + (int64_t)h * p.q_sh + sub * cpl;
            const bool append = (t == p.rk_len) && (h % p.ratio) == 0;
            float sc = 0.f;
            for (int d = 0; d < cpl; d += 8) {
                const u16x8 kv = *(const u16x8*)(krow + d);
                const u16x8 qv = *(const u16x8*)(qrow + d);
#pragma unroll
                for (int e = 0; e < 8; e++) sc = __builtin_fmaf(h2f_bits(qv[e]), h2f_bits(kv[e]), sc);
                if (append) *(u16x8*)(kres + (int64_t)t * p.k_st + sub * cpl + d) = kv;
            }
            sc += __shfl_xor(sc, 1);
            sc += __shfl_xor(sc, 2);
            sc += __shfl_xor(sc, 4);
            if (sub == 0) {
                const uint16_t hs = f2h_bits(sc);
                rs_lds[t] = hs;
                srow[p.Tq + t] = hs;
            }
        }
        __syncthreads();
    }
    auto sval = [&](int j) {
        const uint16_t raw1 = (res_here && j >= p.Tq) ? rs_lds[j - p.Tq] : srow[j];
        return h2f_bits(kivi_scaled_score(raw1, p.inv_scale, mrow != nullptr, mrow ? mrow[j] : 0));
    };
    const int nvec = (res_here ? p.Tq : n) & ~3;   // scores below this index come straight from memory, 4 at a time
    float mx = -__builtin_inff();
    float sum = 0.f;
    if constexpr (MODE != 2) {
        // ONE pass over the chunk: every thread keeps a running (max, sum of exp(x - max)) of its scores and rescales
        // the sum when the max grows; the 256 pairs are then merged the same way (wave shuffles, 4 values of LDS).
        for (int j0 = lo + threadIdx.x * 4; j0 < hi; j0 += 1024) {
            float x[4];
            if (j0 + 4 <= nvec && j0 + 4 <= hi) {
                const u16x4 v4 = *(const u16x4*)(srow + j0);
#pragma unroll
                for (int e = 0; e < 4; e++)
                    x[e] = h2f_bits(kivi_scaled_score(v4[e], p.inv_scale, mrow != nullptr, mrow ? mrow[j0 + e] : 0));
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) x[e] = (j0 + e < hi) ? sval(j0 + e) : -__builtin_inff();
            }
            const float m4 = __builtin_fmaxf(__builtin_fmaxf(x[0], x[1]), __builtin_fmaxf(x[2], x[3]));
            const float mn = __builtin_fmaxf(mx, m4);   // finite: every score is a finite fp16 (masked ones sit at -65504)
            sum = sum * kivi_exp(mx - mn) + ((kivi_exp(x[0] - mn) + kivi_exp(x[1] - mn)) + (kivi_exp(x[2] - mn) + kivi_exp(x[3] - mn)));
            mx = mn;
        }
        auto merge = [](float& m, float& l, float m2, float l2) {
            const float mn = __builtin_fmaxf(m, m2);
            const float a = (m == mn) ? 1.0f : kivi_exp(m - mn);      // also covers -inf - -inf (an empty side)
            const float b = (m2 == mn) ? 1.0f : kivi_exp(m2 - mn);
            l = l * a + l2 * b;
            m = mn;
        };
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) merge(mx, sum, __shfl_xor(mx, k), __shfl_xor(sum, k));
        __shared__ float ml_lds[8];
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            ml_lds[threadIdx.x >> 6] = mx;
            ml_lds[4 + (threadIdx.x >> 6)] = sum;
        }
        __syncthreads();
        mx = ml_lds[0];
        sum = ml_lds[4];
#pragma unroll
        for (int w = 1; w < 4; w++) merge(mx, sum, ml_lds[w], ml_lds[4 + w]);
    }
    if constexpr (MODE == 1) {
        if (threadIdx.x == 0) {
            p.partial[2 * ((int64_t)row * P + c)] = mx;
            p.partial[2 * ((int64_t)row * P + c) + 1] = sum;
        }
        return;
    }
    if constexpr (MODE == 2) {   // every thread combines the P chunk statistics the same way (chunk order)
        const float* pp = p.partial + 2 * (int64_t)row * P;
        for (int i = 0; i < P; i++) mx = __builtin_fmaxf(mx, pp[2 * i]);
        for (int i = 0; i < P; i++) sum += pp[2 * i + 1] * kivi_exp(pp[2 * i] - mx);
    }
    // every thread rewrites exactly the elements it read (the reductions above are barriers), so in place is safe
    const float inv = 1.0f / sum;
    for (int j0 = lo + threadIdx.x * 4; j0 < hi; j0 += 1024) {
        if (j0 + 4 <= nvec && j0 + 4 <= hi) {
            const u16x4 v4 = *(const u16x4*)(srow + j0);
            u16x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = f2h_bits(kivi_exp(h2f_bits(kivi_scaled_score(v4[e], p.inv_scale, mrow != nullptr,
                                                                         mrow ? mrow[j0 + e] : 0)) - mx) * inv);
            *(u16x4*)(srow + j0) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (j0 + e < hi) srow[j0 + e] = f2h_bits(kivi_exp(sval(j0 + e) - mx) * inv);
        }
    }
}

template <int BITS>
__global__ __launch_bounds__(64) void gemv_v_generic(const GemvVArgs a, int G, int Dw) {
    constexpr int FPI = 32 / BITS;
    const int w = blockIdx.y * 64 + threadIdx.x;
    const int bh = blockIdx.x;
    const int b = bh / a.nh, h = bh - b * a.nh;
    const int hk = h / a.ratio;
    if (w >= Dw) return;
    const int g = (w * FPI) / G;
    const uint32_t* cp = a.code + b * a.code_sb + hk * a.code_sh + w;
    const uint16_t* sp = a.scale + b * a.sm_sb + hk * a.sm_sh + g;
    const uint16_t* mp = a.mn + b * a.sm_sb + hk * a.sm_sh + g;
    const uint16_t* ap = a.a + b * a.a_sb + (int64_t)h * a.a_sh;
    float acc[FPI];
#pragma unroll
    for (int p = 0; p < FPI; p++) acc[p] = 0.f;
    float z = 0.f;
    for (int64_t t = 0; t < a.Tv; t++) {
        const float at = h2f_bits(ap[t]);
        const float as = at * h2f_bits(sp[t * a.sm_sr]);
        z = __builtin_fmaf(at, h2f_bits(mp[t * a.sm_sr]), z);
        accum_word<BITS, KIVI_UNPACK_BFE>(cp[t * a.code_sr], as, acc);
    }
    uint16_t* op = a.out + b * a.out_sb + (int64_t)h * a.out_sh + (int64_t)w * FPI;
#pragma unroll
    for (int p = 0; p < FPI; p++) op[p] = f2h_bits(acc[p] + z);
}

// ------------------------------------------------------------------ host side

typedef void (*VLaunch)(const GemvVArgs&, dim3, hipStream_t);

template <int BITS, int G, int DW, int WPL, int R, int U, int MODE, bool NT>
void launch_v(const GemvVArgs& a, dim3 grid, hipStream_t s) {
    const size_t lds = a.softmax ? (size_t)R * a.n_pad * sizeof(uint16_t) : 0;
    if (a.nsplit > 1)
        KIVI_LAUNCH_LDS((gemv_v_kernel<BITS, G, DW, WPL, R, U, MODE, NT, true>), grid, dim3(256), lds, s, a);
    else
        KIVI_LAUNCH_LDS((gemv_v_kernel<BITS, G, DW, WPL, R, U, MODE, NT, false>), grid, dim3(256), lds, s, a);
}

struct VVariant {
    const char* name;
    int bits, G, dw, wpl, R, U, mode, nt;
    VLaunch fn;
};

#define VV(BITS, G, DW, WPL, R, U, MODE, NT)                                                               \
    {"v_b" #BITS "_g" #G "_dw" #DW "_w" #WPL "_r" #R "_u" #U "_m" #MODE "_nt" #NT, BITS, G, DW, WPL, R, U, MODE, \
     NT, launch_v<BITS, G, DW, WPL, R, U, MODE, (NT != 0)>}

const VVariant v_variants[] = {
    // ---- 2-bit, D=128 (DW=8), g=32, MHA: table order = dispatch preference (measured, profiles/)
    VV(2, 32, 8, 4, 1, 1, 2, 1),
    VV(2, 32, 8, 2, 1, 4, 2, 1),
    VV(2, 32, 8, 4, 1, 2, 2, 0),
    VV(2, 32, 8, 4, 1, 2, 2, 1),
    VV(2, 32, 8, 4, 1, 4, 2, 0),
    VV(2, 32, 8, 4, 1, 4, 2, 1),
    VV(2, 32, 8, 4, 1, 1, 2, 0),
    VV(2, 32, 8, 2, 1, 4, 2, 0),
    VV(2, 32, 8, 2, 1, 8, 2, 1),
    VV(2, 32, 8, 4, 1, 2, 0, 0),
    VV(2, 32, 8, 4, 1, 2, 1, 0),
    VV(2, 32, 8, 4, 1, 2, 4, 1),
    VV(2, 32, 8, 2, 1, 4, 4, 1),
    VV(2, 32, 8, 4, 1, 2, 3, 1),   // diagnostic: memory-side ceiling
    VV(2, 32, 8, 2, 1, 4, 3, 1),
    // other group sizes / head dims
    VV(2, 64, 8, 2, 1, 4, 2, 1),
    VV(2, 128, 8, 2, 1, 4, 2, 1),
    VV(2, 64, 8, 4, 1, 2, 2, 0),
    VV(2, 128, 8, 4, 1, 2, 2, 0),
    VV(2, 32, 4, 4, 1, 2, 2, 0),
    VV(2, 64, 4, 4, 1, 2, 2, 0),
    VV(2, 32, 16, 4, 1, 2, 2, 0),
    VV(2, 64, 16, 4, 1, 2, 2, 0),
    VV(2, 128, 16, 4, 1, 2, 2, 0),
    // ---- 4-bit (D=128 -> DW=16; D=64 -> DW=8)
    VV(4, 32, 16, 4, 1, 4, 2, 1),
    VV(4, 64, 16, 4, 1, 4, 2, 1),
    VV(4, 128, 16, 4, 1, 4, 2, 1),
    VV(4, 32, 8, 4, 1, 4, 2, 1),
    VV(4, 64, 8, 4, 1, 4, 2, 1),
    VV(4, 32, 16, 4, 1, 4, 0, 0),
    // ---- GQA (R heads share the unpack; 2 words per lane keeps R*EPL accumulators in registers)
    VV(2, 32, 8, 2, 4, 2, 4, 1),
    VV(2, 32, 8, 1, 4, 2, 4, 1),
    VV(2, 32, 8, 1, 4, 1, 4, 1),
    VV(2, 32, 8, 2, 4, 1, 4, 1),
    VV(2, 32, 8, 1, 4, 4, 4, 1),
    VV(2, 32, 8, 1, 8, 4, 4, 1),
    VV(2, 32, 8, 1, 4, 8, 4, 1),
    VV(2, 32, 8, 2, 2, 4, 4, 1),
    VV(2, 64, 8, 2, 4, 2, 4, 1),
    VV(2, 128, 8, 2, 4, 2, 4, 1),
    VV(4, 32, 16, 4, 4, 2, 4, 1),
    VV(2, 32, 8, 2, 4, 2, 2, 0),
    VV(2, 32, 8, 2, 2, 4, 2, 0),
    VV(2, 64, 8, 2, 4, 2, 2, 0),
    VV(2, 128, 8, 2, 4, 2, 2, 0),
    VV(4, 32, 16, 4, 4, 2, 2, 0),
    VV(2, 32, 8, 2, 8, 1, 2, 0),
};
constexpr int v_nvariants = sizeof(v_variants) / sizeof(v_variants[0]);

bool v_variant_fits(const VVariant& v, const GemvVArgs& a, int bits, int G) {
    if (v.bits != bits || v.G != G) return false;
    const int fpi = 32 / bits;
    if (a.D != v.dw * fpi) return false;
    if (a.ratio % v.R) return false;
    if (!a.extents_ok) return false;
    if (!a.fused && a.Tv == 0) return false;
    if (a.softmax) {   // vector loads of the score rows need aligned rows; the LDS budget is checked in v_run
        if ((a.a_sh % 4) || (a.a_sb % 4) || ((uintptr_t)a.a % 8)) return false;
    }
    const int epl = v.wpl * fpi;
    const int ngl = epl >= G ? epl / G : 1;
    if ((a.code_sr % v.wpl) || (a.code_sh % v.wpl) || (a.code_sb % v.wpl) || ((uintptr_t)a.code % (4 * v.wpl)))
        return false;
    if ((a.sm_sr % ngl) || (a.sm_sh % ngl) || (a.sm_sb % ngl)) return false;
    if (((uintptr_t)a.scale % (2 * ngl)) || ((uintptr_t)a.mn % (2 * ngl))) return false;
    // per-lane byte offsets (incl. the <= 40 chunks a wave may overshoot the tail by) must not wrap
    if ((a.Tv + 64 * 41) * a.code_sr * 4 >= (int64_t)0xFFFFFFFFll) return false;
    return true;
}

int v_run(int variant, GemvVArgs a, int B, int G, int bits, hipStream_t s) {
    if (variant >= 0) {
        KIVI_REQUIRE(variant < v_nvariants, KIVI_EINVAL, "kivi_gemv_v_variant: no variant %d", variant);
        const VVariant& v = v_variants[variant];
        KIVI_REQUIRE(v_variant_fits(v, a, bits, G), KIVI_EINVAL,
                     "kivi_gemv_v_variant: %s does not fit this problem (bits=%d g=%d D=%d ratio=%d)", v.name, bits, G,
                     a.D, a.ratio);
        a.units_per_b = a.nh / v.R;
        const int64_t units = (int64_t)B * a.units_per_b;
        const int tpi = 64 / (v.dw / v.wpl);
        const int64_t nchunk = (a.Tv + tpi - 1) / tpi;
        // split-T: with few (b, head unit) rows a block per row cannot fill 256 CUs; S blocks share a row and meet in
        // the caller's workspace.  Only when a workspace was supplied (kivi_decode_attend).
        // split-T: with few (b, head unit) rows a block per row cannot fill 256 CUs; S blocks share a row and meet in
        // the caller's workspace (kivi_decode_attend): ~3 blocks per CU in total, few enough that the per-block
        // epilogue (butterfly, workspace hand-off) stays small next to the streamed range.
        int S = 1;
        // (measured at T=4k: 128 rows gain 6 % from the split, 256 rows lose 5 % to its extra launches; long rows gain)
        if (a.ws && nchunk >= 32 && (units < 256 || (units < 512 && nchunk >= 256)) && units <= KIVI_WS_COUNTERS) {
            // R >= 4 variants hold R x EPL accumulators: 2 blocks per CU are resident, so 512 blocks = one full round
            const int64_t target = v.R >= 4 ? 512 : 768;
            S = (int)((target + units - 1) / units);
            static const char* forced_split = getenv("KIVI_V_SPLIT");   // tuning aid
            if (forced_split) S = atoi(forced_split);
            if (S > nchunk / 32) S = (int)(nchunk / 32);
            if (S > 64) S = 64;
            const size_t need = (size_t)units * (S + 1) * v.R * a.D * sizeof(float);
            if (S < 2 || need > a.ws_bytes) S = 1;
        }
        // Where the softmax runs.  In the prologue of the block that owns the row: one query head per block, a row
        // of <= 8192 keys held in registers, nothing split (the MHA decode shape: no extra launch, the probabilities
        // never leave the CU).  Otherwise (grouped queries = R rows per block, longer rows, split rows) that prologue
        // would serialise R x n exps per block while the memory system idles (measured +150 us at B=64 / 8 kv heads /
        // 8k keys): the well-parallel row-softmax launch turns the score rows into probabilities in place first.
        // The caller may have handed over the packed-K side of the step too (kivi_decode_attend with K fields): one
        // launch for the whole row when the shape is the tuned MHA one (in-block softmax, nothing split), otherwise the
        // stand-alone qK^T launch goes first.
        bool fuse_row = false;
        if (a.kside) {
            const KSide& ks = *a.kside;
            static const char* nofuse = getenv("KIVI_NO_ROW_FUSION");   // tuning aid
            fuse_row = !nofuse && ks.fusable && a.softmax && S == 1 && v.R == 1 && a.n_scores <= 8192 && a.rq != nullptr &&
                       ((bits == 2 && (G == 32 || G == 64 || G == 128)) || (bits == 4 && G == 32)) && a.D == 128 &&
                       v.mode == KIVI_UNPACK_MIX;
            if (!fuse_row && ks.args.T > 0) {
                const GemvKArgs& k = ks.args;
                const int rc = kivi_gemv_k_paged(-1, ks.page_tokens, k.code_sp, k.sm_sp, k.q, k.q_sb, k.q_sh, k.code, k.code_sb,
                                                 k.code_sh, k.code_sr, k.scale, k.mn, k.sm_sb, k.sm_sh, k.sm_sr, k.out, k.out_sb,
                                                 k.out_sh, ks.B, k.nh, ks.nh_kv, k.D, k.T, ks.group_size, ks.bits, (kivi_stream_t)s);
                if (rc) return rc;
            }
        }
        if (a.softmax && (v.R > 1 || S > 1 || a.n_scores > 8192)) {
            RowSoftmaxArgs rp;
            rp.scores = const_cast<uint16_t*>(a.a); rp.s_sb = a.a_sb; rp.s_sh = a.a_sh;
            rp.n = a.n_scores; rp.Tq = a.Tq; rp.inv_scale = a.inv_scale; rp.mask = a.mask; rp.mask_sb = a.mask_sb;
            rp.q = a.rq; rp.q_sb = a.rq_sb; rp.q_sh = a.rq_sh;
            rp.kres = a.rkres; rp.k_sb = a.rk_sb; rp.k_sh = a.rk_sh; rp.k_st = a.rk_st;
            rp.knew = a.rknew; rp.kn_sb = a.rkn_sb; rp.kn_sh = a.rkn_sh;
            rp.rk_len = a.rk_len; rp.ratio = a.ratio; rp.nh = a.nh; rp.D = a.D;
            // few rows: P blocks per row so that ~2048 blocks of >= 2048 scores share the work (measured: 2048 rows of
            // 8k scores run best as one launch of whole rows, 512 rows of 32k as 4 chunks, 32 rows as 16)
            const int64_t rows = (int64_t)B * a.nh;
            int P = (int)((2048 + rows - 1) / rows);
            if (P > a.n_scores / 2048) P = a.n_scores / 2048;
            if (P > 64) P = 64;
            static const char* fp = getenv("KIVI_SOFTMAX_P");   // tuning aid: blocks per row of the row softmax
            if (fp) P = atoi(fp);
            const size_t part_bytes = ((size_t)rows * (P > 0 ? P : 1) * 2 * sizeof(float) + 255) / 256 * 256;
            if (P >= 2 && a.ws && part_bytes + (size_t)units * (S + 1) * v.R * a.D * sizeof(float) <= a.ws_bytes) {
                rp.P = P;
                rp.chunk = (a.n_scores / P) / 1024 * 1024;
                rp.partial = a.ws;
                a.ws = (float*)((char*)a.ws + part_bytes);
                a.ws_bytes -= part_bytes;
                hipLaunchKernelGGL(row_softmax_kernel<1>, dim3((unsigned)(rows * P)), dim3(256), 0, s, rp);
                hipLaunchKernelGGL(row_softmax_kernel<2>, dim3((unsigned)(rows * P)), dim3(256), 0, s, rp);
            } else {
                rp.P = 1; rp.chunk = a.n_scores; rp.partial = nullptr;
                hipLaunchKernelGGL(row_softmax_kernel<0>, dim3((unsigned)rows), dim3(256), 0, s, rp);
            }
            a.softmax = 0;     // the sV blocks read finished probabilities
            a.rq = nullptr;
        }
        a.nsplit = S;
        a.cps = (int)((nchunk + S - 1) / S);
        if (a.softmax) {
            a.n_pad = (int)((a.n_scores + 7) / 8 * 8);
            KIVI_REQUIRE((size_t)v.R * a.n_pad * 2 <= 96 * 1024, KIVI_EUNSUPPORTED,
                         "kivi_decode_attend: %d probability rows of %d do not fit the LDS", v.R, a.n_pad);
        }
        if (fuse_row) {
            GemvKArgs ak = a.kside->args;
            ak.units_per_b = a.nh;
            if (bits == 4) {   // 4-bit: 8 codes per word, 4 words per lane = the same 2048-token tile; sV over 16 words per row
                ak.tile_blocks = (int)(((ak.Tw + 255) / 256 + 1) / 2);
                ak.res_blocks = 0;
                a.scores_lds = 1;
                KIVI_LAUNCH_LDS((decode_row_kernel<4, 32, 16, 4, 2, 4, 4, 2, false>), dim3((unsigned)units), dim3(256),
                                (size_t)a.n_pad * sizeof(uint16_t), s, ak, a);
                return kivi_launch_status("decode_row");
            }
            const int tiles = (int)((ak.Tw + 127) / 128);
            ak.res_blocks = 0;
            a.scores_lds = 1;
            static const char* xl = getenv("KIVI_ROW_EXTRA_LDS");   // diagnostic: fewer co-resident blocks per CU
            const size_t lds = (size_t)a.n_pad * sizeof(uint16_t) + (xl ? (size_t)atoi(xl) : 0);
            static const char* rv = getenv("KIVI_ROW_VARIANT");   // tuning aid: K-phase shape "ds<DSPLIT>u<U>"
            static const char* rvv = getenv("KIVI_ROW_V");        // tuning aid: sV-phase shape "w<WPL>u<U>"
            const int sel = !rv ? 1 : !strcmp(rv, "ds4u4") ? 0 : !strcmp(rv, "ds2u4") ? 1 : !strcmp(rv, "ds2u8") ? 2 : !strcmp(rv, "ds4u8") ? 3 : 1;
            const int selv = (rvv && !strcmp(rvv, "w2u4")) ? 1 : 0;
            ak.tile_blocks = (G != 32 || selv == 1 || sel == 1 || sel == 2) ? (tiles + 1) / 2 : tiles;   // DSPLIT = 2: two tiles per pass
            const dim3 grid((unsigned)units);
            if (G == 64) KIVI_LAUNCH_LDS((decode_row_kernel<2, 64, 8, 2, 2, 4, 4, 1>), grid, dim3(256), lds, s, ak, a);
            else if (G == 128) KIVI_LAUNCH_LDS((decode_row_kernel<2, 128, 8, 2, 2, 4, 4, 1>), grid, dim3(256), lds, s, ak, a);
            else if (selv == 1) KIVI_LAUNCH_LDS((decode_row_kernel<2, 32, 8, 2, 2, 4, 2, 4>), grid, dim3(256), lds, s, ak, a);
            else if (sel == 0) KIVI_LAUNCH_LDS((decode_row_kernel<2, 32, 8, 2, 4, 4, 4, 1>), grid, dim3(256), lds, s, ak, a);
            else if (sel == 1) KIVI_LAUNCH_LDS((decode_row_kernel<2, 32, 8, 2, 2, 4, 4, 1>), grid, dim3(256), lds, s, ak, a);
            else if (sel == 2) KIVI_LAUNCH_LDS((decode_row_kernel<2, 32, 8, 2, 2, 8, 4, 1>), grid, dim3(256), lds, s, ak, a);
            else KIVI_LAUNCH_LDS((decode_row_kernel<2, 32, 8, 2, 4, 8, 4, 1>), grid, dim3(256), lds, s, ak, a);
            return kivi_launch_status("decode_row");
        }
        v.fn(a, dim3((unsigned)(units * S)), s);
        return kivi_launch_status(v.name);
    }
    int best = -1;
    static const char* forced = getenv("KIVI_GEMV_V_VARIANT");   // tuning aid: force a variant by name if it fits
    if (forced)
        for (int i = 0; i < v_nvariants; i++)
            if (!strcmp(forced, v_variants[i].name) && v_variant_fits(v_variants[i], a, bits, G)) return v_run(i, a, B, G, bits, s);
    // GQA: R query heads of a kv head share every unpacked code.  Measured (B=64 / nh_kv=8 / T=8k, probabilities from
    // memory, fp16 window of 128): R=4 108 us, R=2 143 us, R=1 (every head re-reads its kv head) 162 us.
    const int want_r = (a.ratio % 4 == 0) ? 4 : (a.ratio % 2 == 0) ? 2 : 1;
    for (int pass = 0; pass < 2 && best < 0; pass++)
        for (int i = 0; i < v_nvariants; i++) {
            const VVariant& v = v_variants[i];
            if (!(v.mode == KIVI_UNPACK_MIX || (v.mode == KIVI_UNPACK_DEN32 && v.R > 1))) continue;
            if (pass == 0 && v.R != want_r) continue;
            if (!v_variant_fits(v, a, bits, G)) continue;
            best = i;
            break;
        }
    if (best >= 0) return v_run(best, a, B, G, bits, s);
    KIVI_REQUIRE(!a.fused, KIVI_EUNSUPPORTED,
                 "kivi_decode_output: no tuned kernel for this shape (bits=%d g=%d D=%d); use the unfused path", bits, G, a.D);
    const int fpi = 32 / bits;
    const int Dw = a.D / fpi;
    dim3 grid((unsigned)(B * a.nh), (unsigned)((Dw + 63) / 64));
    if (bits == 2) hipLaunchKernelGGL(gemv_v_generic<2>, grid, dim3(64), 0, s, a, G, Dw);
    else hipLaunchKernelGGL(gemv_v_generic<4>, grid, dim3(64), 0, s, a, G, Dw);
    return kivi_launch_status("gemv_v_generic");
}

}  // namespace

extern "C" int kivi_gemv_v_num_variants(void) { return v_nvariants; }
extern "C" const char* kivi_gemv_v_variant_name(int v) {
    return (v >= 0 && v < v_nvariants) ? v_variants[v].name : "";
}

static int v_fill(GemvVArgs& a, const char* who, const void* av, int64_t a_sb, int64_t a_sh, const void* code,
                  int64_t code_sb, int64_t code_sh, int64_t code_sr, const void* scale, const void* mn, int64_t sm_sb,
                  int64_t sm_sh, int64_t sm_sr, void* out, int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv,
                  int64_t Tv, int D, int group_size, int bits) {
    KIVI_REQUIRE(bits == 2 || bits == 4, KIVI_EINVAL, "%s: bits must be 2 or 4 (matmul.py:215), got %d", who, bits);
    KIVI_REQUIRE(nh_kv > 0 && nh > 0 && nh % nh_kv == 0, KIVI_EINVAL, "%s: nh %% nh_kv != 0 (matmul.py:216): nh=%d nh_kv=%d",
                 who, nh, nh_kv);
    const int fpi = 32 / bits;
    KIVI_REQUIRE(group_size > 0 && group_size % fpi == 0, KIVI_EINVAL, "%s: group_size %d must be a positive multiple of %d",
                 who, group_size, fpi);
    KIVI_REQUIRE(D > 0 && D % fpi == 0 && D % group_size == 0, KIVI_EINVAL,
                 "%s: head_dim=%d must be a multiple of group_size=%d", who, D, group_size);
    KIVI_REQUIRE(B > 0 && Tv >= 0, KIVI_EINVAL, "%s: empty batch", who);
    KIVI_REQUIRE((int64_t)B * nh < ((int64_t)1 << 31), KIVI_EINVAL, "%s: B*nh too large", who);
    a.a = (const uint16_t*)av; a.a_sb = a_sb; a.a_sh = a_sh;
    a.code = (const uint32_t*)code; a.code_sb = code_sb; a.code_sh = code_sh; a.code_sr = code_sr;
    a.scale = (const uint16_t*)scale; a.mn = (const uint16_t*)mn;
    a.sm_sb = sm_sb; a.sm_sh = sm_sh; a.sm_sr = sm_sr;
    a.out = (uint16_t*)out; a.out_sb = out_sb; a.out_sh = out_sh;
    a.nh = nh; a.ratio = nh / nh_kv; a.D = D; a.Tv = Tv;
    a.units_per_b = nh;
    // extents: exactly Tv token rows, so chunk tails past Tv read zeros (hardware bounds check)
    const int64_t ce = Tv > 0 ? ((Tv - 1) * code_sr + D / fpi) * 4 : 0;
    const int64_t se = Tv > 0 ? ((Tv - 1) * sm_sr + D / group_size) * 2 : 0;
    const int64_t ae = Tv * 2;
    a.extents_ok = ce < (int64_t)0xFFFFFFFFll && se < (int64_t)0xFFFFFFFFll && ae < (int64_t)0xFFFFFFFFll;
    a.code_extent = (uint32_t)(a.extents_ok ? ce : 0);
    a.sm_extent = (uint32_t)(a.extents_ok ? se : 0);
    a.a_extent = (uint32_t)(a.extents_ok ? ae : 0);
    a.softmax = 0; a.n_scores = 0; a.n_pad = 0; a.inv_scale = 1.0f; a.mask = nullptr; a.mask_sb = 0;
    a.nsplit = 1; a.cps = 0; a.ws = nullptr; a.counters = nullptr; a.ws_bytes = 0;
    a.rq = nullptr; a.rkres = nullptr; a.rknew = nullptr; a.rk_len = 0; a.Tq = 0;
    a.rq_sb = a.rq_sh = a.rk_sb = a.rk_sh = a.rk_st = a.rkn_sb = a.rkn_sh = 0;
    a.fused = 0; a.vres = nullptr; a.vnew = nullptr; a.flush = 0; a.win_start = 0; a.res_len = 0;
    a.scores_lds = 0; a.kside = nullptr;
    a.vres_sb = a.vres_sh = a.vres_st = a.vnew_sb = a.vnew_sh = 0;
    return 0;
}

extern "C" int kivi_gemv_v_variant(int variant, const void* av, int64_t a_sb, int64_t a_sh, const void* code,
                                   int64_t code_sb, int64_t code_sh, int64_t code_sr, const void* scale,
                                   const void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr, void* out,
                                   int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv, int64_t Tv, int D,
                                   int group_size, int bits, kivi_stream_t stream) {
    GemvVArgs a;
    int rc = v_fill(a, "kivi_gemv_v", av, a_sb, a_sh, code, code_sb, code_sh, code_sr, scale, mn, sm_sb, sm_sh, sm_sr, out,
                    out_sb, out_sh, B, nh, nh_kv, Tv, D, group_size, bits);
    if (rc) return rc;
    return v_run(variant, a, B, group_size, bits, (hipStream_t)stream);
}

extern "C" int kivi_gemv_v(const void* av, int64_t a_sb, int64_t a_sh, const void* code, int64_t code_sb,
                           int64_t code_sh, int64_t code_sr, const void* scale, const void* mn, int64_t sm_sb,
                           int64_t sm_sh, int64_t sm_sr, void* out, int64_t out_sb, int64_t out_sh, int B, int nh,
                           int nh_kv, int64_t Tv, int D, int group_size, int bits, kivi_stream_t stream) {
    return kivi_gemv_v_variant(-1, av, a_sb, a_sh, code, code_sb, code_sh, code_sr, scale, mn, sm_sb, sm_sh, sm_sr, out,
                               out_sb, out_sh, B, nh, nh_kv, Tv, D, group_size, bits, stream);
}

struct ResidualK {
    const void* q;
    int64_t q_sb, q_sh;
    void* kres;
    int64_t kres_sb, kres_sh, kres_st;
    const void* knew;
    int64_t knew_sb, knew_sh;
    int res_len;
    int64_t Tq;
    void* workspace;
    size_t workspace_bytes;
    const KSide* kside;                // packed-K side of the step, or null
};

static int decode_output_impl(int softmax, float inv_scale, const void* mask, int64_t mask_sb, const void* probs,
                              int64_t a_sb, int64_t a_sh, void* code, int64_t code_sb, int64_t code_sh, int64_t code_sr,
                              void* scale, void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr, void* vres,
                              int64_t vres_sb, int64_t vres_sh, int64_t vres_st, int win_start, int res_len,
                              const void* vnew, int64_t vnew_sb, int64_t vnew_sh, int flush, void* out, int64_t out_sb,
                              int64_t out_sh, int B, int nh, int nh_kv, int64_t Tv, int D, int group_size, int bits,
                              kivi_stream_t stream, const ResidualK* rkp = nullptr);

extern "C" int kivi_decode_output(const void* probs, int64_t a_sb, int64_t a_sh, void* code, int64_t code_sb,
                                  int64_t code_sh, int64_t code_sr, void* scale, void* mn, int64_t sm_sb, int64_t sm_sh,
                                  int64_t sm_sr, void* vres, int64_t vres_sb, int64_t vres_sh, int64_t vres_st,
                                  int win_start, int res_len, const void* vnew, int64_t vnew_sb, int64_t vnew_sh,
                                  int flush, void* out, int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv,
                                  int64_t Tv, int D, int group_size, int bits, kivi_stream_t stream) {
    return decode_output_impl(0, 1.0f, nullptr, 0, probs, a_sb, a_sh, code, code_sb, code_sh, code_sr, scale, mn, sm_sb, sm_sh,
                              sm_sr, vres, vres_sb, vres_sh, vres_st, win_start, res_len, vnew, vnew_sb, vnew_sh, flush,
                              out, out_sb, out_sh, B, nh, nh_kv, Tv, D, group_size, bits, stream);
}

extern "C" int kivi_decode_softmax_output(const void* scores, int64_t a_sb, int64_t a_sh, float inv_scale,
                                          const void* mask, int64_t mask_sb, void* code, int64_t code_sb,
                                          int64_t code_sh, int64_t code_sr, void* scale, void* mn, int64_t sm_sb,
                                          int64_t sm_sh, int64_t sm_sr, void* vres, int64_t vres_sb, int64_t vres_sh,
                                          int64_t vres_st, int win_start, int res_len, const void* vnew, int64_t vnew_sb,
                                          int64_t vnew_sh, int flush, void* out, int64_t out_sb, int64_t out_sh, int B,
                                          int nh, int nh_kv, int64_t Tv, int D, int group_size, int bits,
                                          kivi_stream_t stream) {
    return decode_output_impl(1, inv_scale, mask, mask_sb, scores, a_sb, a_sh, code, code_sb, code_sh, code_sr, scale, mn,
                              sm_sb, sm_sh, sm_sr, vres, vres_sb, vres_sh, vres_st, win_start, res_len, vnew, vnew_sb,
                              vnew_sh, flush, out, out_sb, out_sh, B, nh, nh_kv, Tv, D, group_size, bits, stream);
}

static int decode_output_impl(int softmax, float inv_scale, const void* mask, int64_t mask_sb, const void* probs,
                              int64_t a_sb, int64_t a_sh, void* code, int64_t code_sb, int64_t code_sh, int64_t code_sr,
                              void* scale, void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr, void* vres,
                              int64_t vres_sb, int64_t vres_sh, int64_t vres_st, int win_start, int res_len,
                              const void* vnew, int64_t vnew_sb, int64_t vnew_sh, int flush, void* out, int64_t out_sb,
                              int64_t out_sh, int B, int nh, int nh_kv, int64_t Tv, int D, int group_size, int bits,
                              kivi_stream_t stream, const ResidualK* rkp) {
    GemvVArgs a;
    int rc = v_fill(a, "kivi_decode_output", probs, a_sb, a_sh, code, code_sb, code_sh, code_sr, scale, mn, sm_sb, sm_sh,
                    sm_sr, out, out_sb, out_sh, B, nh, nh_kv, Tv, D, group_size, bits);
    if (rc) return rc;
    KIVI_REQUIRE(vres && vnew && res_len >= 0 && win_start >= 0, KIVI_EINVAL, "kivi_decode_output: window buffers missing");
    KIVI_REQUIRE(D % 2 == 0 && vres_sb % 2 == 0 && vres_sh % 2 == 0 && vres_st % 2 == 0 && vnew_sb % 2 == 0 &&
                     vnew_sh % 2 == 0 && (uintptr_t)vres % 4 == 0 && (uintptr_t)vnew % 4 == 0,
                 KIVI_EALIGN, "kivi_decode_output: window rows must be 4-byte aligned");
    KIVI_REQUIRE(D <= 256, KIVI_EUNSUPPORTED, "kivi_decode_output: head_dim %d > 256", D);
    a.fused = 1;
    a.vres = (uint16_t*)vres; a.vres_sb = vres_sb; a.vres_sh = vres_sh; a.vres_st = vres_st;
    a.win_start = win_start; a.res_len = res_len;
    a.vnew = (const uint16_t*)vnew; a.vnew_sb = vnew_sb; a.vnew_sh = vnew_sh;
    a.flush = flush ? 1 : 0;
    if (softmax) {
        const int64_t n = Tv + res_len + 1;
        KIVI_REQUIRE(n < ((int64_t)1 << 30), KIVI_EINVAL, "kivi_decode_softmax_output: row too long");
        a.softmax = 1;
        a.n_scores = (int)n;
        a.n_pad = (int)((n + 7) / 8 * 8);
        a.inv_scale = inv_scale;
        a.mask = (const uint16_t*)mask;
        a.mask_sb = mask_sb;
        if (rkp) {
            const ResidualK& rk = *rkp;
            KIVI_REQUIRE(rk.res_len + 1 <= 136 && rk.Tq + rk.res_len + 1 == n, KIVI_EUNSUPPORTED,
                         "kivi_decode_attend: residual of %d keys does not fit / lengths disagree", rk.res_len);
            KIVI_REQUIRE(D % 64 == 0 && rk.q_sb % 8 == 0 && rk.q_sh % 8 == 0 && rk.kres_sb % 8 == 0 && rk.kres_sh % 8 == 0 &&
                             rk.kres_st % 8 == 0 && rk.knew_sb % 8 == 0 && rk.knew_sh % 8 == 0 &&
                             (uintptr_t)rk.q % 16 == 0 && (uintptr_t)rk.kres % 16 == 0 && (uintptr_t)rk.knew % 16 == 0,
                         KIVI_EUNSUPPORTED, "kivi_decode_attend: rows are not 16-byte aligned");
            a.rq = (const uint16_t*)rk.q; a.rq_sb = rk.q_sb; a.rq_sh = rk.q_sh;
            a.rkres = (uint16_t*)rk.kres; a.rk_sb = rk.kres_sb; a.rk_sh = rk.kres_sh; a.rk_st = rk.kres_st;
            a.rknew = (const uint16_t*)rk.knew; a.rkn_sb = rk.knew_sb; a.rkn_sh = rk.knew_sh;
            a.rk_len = rk.res_len;
            a.Tq = (int)rk.Tq;
            a.kside = rk.kside;
            if (rk.workspace && rk.workspace_bytes > KIVI_WS_COUNTERS * sizeof(int)) {   // arrival counters, then fp32 partials
                a.counters = (int*)rk.workspace;
                a.ws = (float*)((char*)rk.workspace + KIVI_WS_COUNTERS * sizeof(int));
                a.ws_bytes = rk.workspace_bytes - KIVI_WS_COUNTERS * sizeof(int);
            }
        }
    }
    return v_run(-1, a, B, group_size, bits, (hipStream_t)stream);
}

extern "C" int kivi_decode_attend(const kivi_decode_attend_args* p, kivi_stream_t stream) {
    KIVI_REQUIRE(p != nullptr, KIVI_EINVAL, "kivi_decode_attend: null arguments");
    ResidualK rk = {p->q, p->q_sb, p->q_sh, p->kres, p->kres_sb, p->kres_sh, p->kres_st, p->knew, p->knew_sb, p->knew_sh,
                    p->k_res_len, p->Tq, p->workspace, (size_t)p->workspace_bytes, nullptr};
    KSide ks;
    if (p->k_code) {
        int rc = k_check_and_fill(ks.args, p->q, p->q_sb, p->q_sh, p->k_code, p->kc_sb, p->kc_sh, p->kc_sr, p->k_scale, p->k_mn,
                                  p->ks_sb, p->ks_sh, p->ks_sr, p->scores, p->s_sb, p->s_sh, p->B, p->nh, p->nh_kv, p->D, p->Tq,
                                  p->group_size, p->k_bits);
        if (rc) return rc;
        const int fpi = 32 / p->k_bits;
        KIVI_REQUIRE(p->k_page_tokens > 0 && p->k_page_tokens % p->group_size == 0 && p->k_page_tokens % fpi == 0, KIVI_EINVAL,
                     "kivi_decode_attend: k_page_tokens=%lld must be a positive multiple of group_size=%d",
                     (long long)p->k_page_tokens, p->group_size);
        GemvKArgs& k = ks.args;
        k.page_words = p->k_page_tokens / fpi;
        k.page_groups = p->k_page_tokens / p->group_size;
        k.code_sp = p->kc_sp;
        k.sm_sp = p->ks_sp;
        ks.page_tokens = p->k_page_tokens; ks.B = p->B; ks.nh_kv = p->nh_kv; ks.group_size = p->group_size; ks.bits = p->k_bits;
        // decode_row_kernel runs the qK^T mapping {2-bit, g=32, 2 words per lane, 4 waves split D, 4-row batches}
        const int kw = p->k_bits == 2 ? 2 : 4;   // words per lane of the qK^T mapping
        ks.fusable = (p->k_bits == 2 || p->k_bits == 4) && p->v_bits == p->k_bits &&
                     (p->group_size == 32 || (p->k_bits == 2 && (p->group_size == 64 || p->group_size == 128))) &&
                     p->nh == p->nh_kv && p->D == 128 &&
                     k.page_words % (64 * kw) == 0 && k.code_sp % kw == 0 && k.q_sh % 2 == 0 && k.q_sb % 2 == 0 &&
                     (uintptr_t)k.q % 4 == 0 && k.Tw % kw == 0 && k.code_sr % kw == 0 && k.code_sh % kw == 0 &&
                     k.code_sb % kw == 0 && (uintptr_t)k.code % (4 * kw) == 0 && (uintptr_t)k.scale % 2 == 0 &&
                     (uintptr_t)k.mn % 2 == 0 && (int64_t)k.D * k.code_sr * 4 < ((int64_t)1 << 31) &&
                     (int64_t)k.D * k.sm_sr * 2 < ((int64_t)1 << 31);
        rk.kside = &ks;
    }
    const int rc = decode_output_impl(1, p->inv_scale, p->mask, p->mask_sb, p->scores, p->s_sb, p->s_sh, p->v_code, p->vc_sb,
                                      p->vc_sh, p->vc_sr, p->v_scale, p->v_mn, p->vs_sb, p->vs_sh, p->vs_sr, p->vres,
                                      p->vres_sb, p->vres_sh, p->vres_st, p->v_win_start, p->v_res_len, p->vnew, p->vnew_sb,
                                      p->vnew_sh, p->v_flush, p->out, p->out_sb, p->out_sh, p->B, p->nh, p->nh_kv, p->Tv, p->D,
                                      p->group_size, p->v_bits, stream, &rk);
    return rc;
}

// Device side of the packed-K qK^T product (see kivi_gemv_k.hip for the mapping): argument block, the residual-score
// role of the fused decode step, and the per-block tile body.  Shared by the stand-alone kernel (kivi_gemv_k.hip) and
// the fused decode-row kernel (kivi_gemv_v.hip), which runs it tile by tile inside the block that owns a (b, head) row.
#pragma once
#include <type_traits>

#include "kivi_common.h"

namespace {

struct GemvKArgs {
    const uint16_t* q;
    int64_t q_sb, q_sh;
    const uint32_t* code;
    int64_t code_sb, code_sh, code_sr;
    const uint16_t* scale;
    const uint16_t* mn;
    int64_t sm_sb, sm_sh, sm_sr;
    uint16_t* out;
    int64_t out_sb, out_sh;
    int nh, ratio, D;
    int64_t T, Tw;
    int units_per_b;   // nh / R
    int tile_blocks;   // blocks per (b, head unit)
    // paged K storage (kivi_gemv_k_paged): a page = page_tokens tokens of every channel, stored as its own
    // contiguous (D, page_tokens/fpi) block.  page_words == 0: plain hook-state layout (one "page").
    int64_t page_words, page_groups;   // words / quant groups of one channel row inside a page
    int64_t code_sp, sm_sp;            // page strides
    // fused decode step (kivi_decode_scores): blocks >= main_blocks score the fp16 K residual
    // (models/llama_kivi.py:333-337) and append the new token to it.
    int main_blocks;                   // >= 0: fused decode step (residual role on); -1 = plain GEMV
    int res_blocks;                    // the FIRST res_blocks blocks of the grid do the residual (they are short and
                                       // latency-bound: started first, they hide under the streaming blocks)
    const uint16_t* kres;              // (B, nh_kv, R, D) fp16 residual buffer
    int64_t kres_sb, kres_sh, kres_st;
    const uint16_t* knew;              // (B, nh_kv, D) the new key
    int64_t knew_sb, knew_sh;
    int res_len;                       // tokens already in the residual; the new one becomes index res_len
    unsigned long long* dbg;           // phase time stamps (kivi_debug_set_stamps; tools/row_phases.py), normally null
};

// phase time stamp of this wave: slot i of its 16-entry record (s_memtime = shader clock); compiled only into the
// diagnostic instantiation of decode_row_kernel (DBG)
template <bool DBG>
__device__ __forceinline__ void kivi_stamp(unsigned long long* dbg, int i) {
    if constexpr (DBG) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) dbg[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + i] = t;
    }
}

template <int N> struct WordVec;
template <> struct WordVec<1> { typedef uint32_t type; };
template <> struct WordVec<2> { typedef u32x2 type; };
template <> struct WordVec<4> { typedef u32x4 type; };

template <int N, typename V>
__device__ __forceinline__ uint32_t vec_get(const V& v, int j) {
    if constexpr (N == 1) return v;
    else return v[j];
}

constexpr int KQ_MAXD = 1 << 20;  // head_dim bound of the tuned kernels (q is read row by row)

// Residual role of the fused decode step: one block per (b, head unit) computes q . k for the <= R fp16 residual
// keys plus the new one (out[b, h, T + t], fp32 accumulate, one rounding: what the reference's fp16 torch.matmul
// does at llama_kivi.py:337) and the first unit of every kv head appends the new key (:333-336).  Pure latency
// work on L2-resident data: q is loaded once, a wave takes 4 tokens per pass with all their loads in flight
// together, and these blocks are scheduled FIRST so they hide under the streaming blocks.
template <int R>
__device__ __forceinline__ void k_residual_role(const GemvKArgs& a, int unit) {
    constexpr int TB = 4;                      // tokens per wave per pass
    constexpr int NC = 2;                      // channel pairs per lane: D <= 256
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = unit / a.units_per_b;
    const int hu = unit - b * a.units_per_b;
    const int h0 = hu * R;
    const int hk = h0 / a.ratio;
    const int L = a.res_len + 1;
    const bool owner = (h0 % a.ratio) == 0;
    const uint16_t* knew = a.knew + b * a.knew_sb + hk * a.knew_sh;
    uint16_t* kres = const_cast<uint16_t*>(a.kres) + b * a.kres_sb + hk * a.kres_sh;
    float q0[R][NC], q1[R][NC];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int d = lane * 2 + 128 * c;
            uint32_t qq = 0;
            if (d < a.D) qq = *(const uint32_t*)(a.q + b * a.q_sb + (int64_t)(h0 + r) * a.q_sh + d);
            q0[r][c] = h2f_bits((uint16_t)(qq & 0xFFFFu));
            q1[r][c] = h2f_bits((uint16_t)(qq >> 16));
        }
    for (int t0 = wave; t0 < L; t0 += 4 * TB) {
        uint32_t kk[TB][NC];
#pragma unroll
        for (int u = 0; u < TB; u++) {
            const int t = t0 + 4 * u;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const int d = lane * 2 + 128 * c;
                kk[u][c] = 0;
                if (t < L && d < a.D) {
                    const uint16_t* krow = (t < a.res_len) ? kres + (int64_t)t * a.kres_st : knew;
                    kk[u][c] = *(const uint32_t*)(krow + d);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < TB; u++) {
            const int t = t0 + 4 * u;
            float s[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                s[r] = 0.f;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    s[r] = __builtin_fmaf(q0[r][c], h2f_bits((uint16_t)(kk[u][c] & 0xFFFFu)), s[r]);
                    s[r] = __builtin_fmaf(q1[r][c], h2f_bits((uint16_t)(kk[u][c] >> 16)), s[r]);
                }
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) s[r] += __shfl_xor(s[r], m);
            }
            if (t < L) {
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < R; r++) a.out[b * a.out_sb + (int64_t)(h0 + r) * a.out_sh + a.T + t] = f2h_bits(s[r]);
                }
                if (t == a.res_len && owner) {   // append the new key (:333-336)
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const int d = lane * 2 + 128 * c;
                        if (d < a.D) *(uint32_t*)(kres + (int64_t)t * a.kres_st + d) = kk[u][c];
                    }
                }
            }
        }
    }
}

// One block's share of the product: TILES_PER_BLOCK tiles of 64*WPL words of one (b, head unit), `bid` = index among
// the streaming blocks.  `lds_out` == nullptr: results go to a.out (the stand-alone kernel).  Otherwise (fused decode
// row, R == 1): the fp16 scores of the tile's tokens are written to lds_out[token] and nothing goes to memory.
// NW = waves of the block (4; 8 in the eight-wave decode-row kernel).
template <int BITS, int G, int WPL, int DSPLIT, int R, int U, int MODE, bool NT, bool DBG = false, int NW = 4>
__device__ __forceinline__ void k_tile_body(const GemvKArgs& a, const int bid, uint16_t* lds_out) {
    constexpr int FPI = 32 / BITS;
    constexpr int TPL = WPL * FPI;                   // tokens per lane
    constexpr int NGL = (TPL >= G) ? (TPL / G) : 1;  // quant groups per lane
    static_assert(NGL == 1 || NGL == 2, "lane spans at most two groups");
    static_assert(G % FPI == 0, "a word never straddles two groups");
    constexpr int NACC = TPL;
    constexpr int TILES_PER_BLOCK = NW / DSPLIT;
    constexpr int Q = NACC / DSPLIT;                 // tokens per lane each wave finalises
    static_assert(Q % 4 == 0, "finalisation stores 8 or 16 bytes per lane");
    typedef typename WordVec<WPL>::type WV;
    typedef typename std::conditional<NGL == 1, uint16_t, uint32_t>::type SV;

    // cross-wave exchange: every wave keeps 1/DSPLIT of its accumulators and hands the rest over
    __shared__ float red[DSPLIT > 1 ? NW * (DSPLIT - 1) * R * Q * 64 : 1];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform -> SGPR
    const int unit = bid / a.tile_blocks;            // (b, head unit)
    const int tb = bid - unit * a.tile_blocks;
    const int b = unit / a.units_per_b;
    const int hu = unit - b * a.units_per_b;
    const int h0 = hu * R;                            // first query head of the unit
    const int hk = h0 / a.ratio;                      // kv head (gemv_cuda.cu:361-365)
    const int tile = tb * TILES_PER_BLOCK + wave / DSPLIT;
    const int tib = wave / DSPLIT;                    // tile index inside the block
    const int dpart = wave % DSPLIT;

    constexpr int TILE_W = 64 * WPL;                  // words of one channel row covered by a wave
    const int64_t tile_w0 = (int64_t)tile * TILE_W;    // first word (global token order) of the tile
    const int64_t word0 = tile_w0 + lane * WPL;
    const bool valid = word0 < a.Tw;
    // where the tile's first word lives: page index + word offset inside the page row
    const int64_t page = a.page_words ? tile_w0 / a.page_words : 0;
    const int64_t win = a.page_words ? tile_w0 - page * a.page_words : tile_w0;
    const int64_t left = a.Tw - tile_w0;               // words of the row that exist from here on
    const uint32_t tile_words = (uint32_t)(left < TILE_W ? (left > 0 ? left : 0) : TILE_W);

    const int DP = (a.D + DSPLIT - 1) / DSPLIT;
    const int d0 = dpart * DP;
    const int d1 = (d0 + DP < a.D) ? d0 + DP : a.D;
    const int nrows = d1 - d0;

    // wave-uniform buffer descriptors over this (b, kv head) slab; per-row scalar offsets,
    // 32-bit per-lane byte offsets.  Lanes past the row end are masked by `valid`.
    // descriptors start at the tile's first word of channel 0 and span its D rows
    const uint32_t tile_groups = (tile_words * FPI + G - 1) / G;
    const uint32_t c_ext = (uint32_t)(((int64_t)(a.D - 1) * a.code_sr + tile_words) * 4);
    const uint32_t s_ext = (uint32_t)(((int64_t)(a.D - 1) * a.sm_sr + tile_groups) * 2);
    const int64_t sm_base = b * a.sm_sb + hk * a.sm_sh + page * a.sm_sp + (win * FPI) / G;
    const rsrc_t rc = make_rsrc(a.code + b * a.code_sb + hk * a.code_sh + page * a.code_sp + win, c_ext);
    const rsrc_t rs = make_rsrc(a.scale + sm_base, s_ext);
    const rsrc_t rm = make_rsrc(a.mn + sm_base, s_ext);
    const uint32_t coff = (uint32_t)(lane * WPL * 4);
    const uint32_t soff = (uint32_t)(((lane * WPL * FPI) / G) * 2);
    const uint32_t cstep = (uint32_t)(a.code_sr * 4), sstep = (uint32_t)(a.sm_sr * 2);

    float acc[R][NACC];
    float zacc[R][NGL];
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[r][i] = 0.f;
#pragma unroll
        for (int g = 0; g < NGL; g++) zacc[r][g] = 0.f;
    }

    // q is wave-uniform per row: it is read with SCALAR loads, two channels (one dword) at a time, and
    // enters the arithmetic as an SGPR fp16 operand of v_fma_mix_f32 -- no LDS, no block barrier, no VGPRs.
    const uint32_t* qrow[R];
#pragma unroll
    for (int r = 0; r < R; r++)
        qrow[r] = (const uint32_t*)(a.q + b * a.q_sb + (int64_t)(h0 + r) * a.q_sh);   // 4-byte aligned (k_variant_fits)

    // channel d: w = packed codes, sraw/mraw = raw fp16 scale / zero-point bits; `odd` = d & 1 (folds to a
    // constant in the unrolled batches)
    // qp[r] = the fp16 pair (q[d & ~1], q[d | 1]) of head r, fetched one batch ahead together with the codes
    auto row = [&](const uint32_t* qp, bool odd, const WV& w, SV sraw, SV mraw) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t qb = qp[r];
            float qs[NGL];
            qs[0] = mul_hh_s(qb, odd, (uint32_t)sraw, false);
            zacc[r][0] = fma_hh_s(qb, odd, (uint32_t)mraw, false, zacc[r][0]);
            if constexpr (NGL == 2) {
                qs[1] = mul_hh_s(qb, odd, (uint32_t)sraw, true);
                zacc[r][1] = fma_hh_s(qb, odd, (uint32_t)mraw, true, zacc[r][1]);
            }
            if constexpr (qs_factor<MODE>() != 1.0f) {
#pragma unroll
                for (int g = 0; g < NGL; g++) qs[g] *= qs_factor<MODE>();
            }
#pragma unroll
            for (int j = 0; j < WPL; j++) {
                const int g = (NGL == 1) ? 0 : (j * FPI) / G;
                accum_word<BITS, MODE>(vec_get<WPL>(w, j), qs[g], &acc[r][j * FPI]);
            }
        }
    };
    // batch = U consecutive channels starting at `dr` (even: d0 and U are even)
    auto load_batch = [&](int dr, WV* wb, SV* sb, SV* mb, uint32_t (*qb)[R]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            wb[u] = buf_load<WV, NT>(rc, coff, (uint32_t)(dr + u) * cstep);
            sb[u] = buf_load<SV, NT>(rs, soff, (uint32_t)(dr + u) * sstep);
            mb[u] = buf_load<SV, NT>(rm, soff, (uint32_t)(dr + u) * sstep);
        }
#pragma unroll
        for (int u2 = 0; u2 < U / 2; u2++)
#pragma unroll
            for (int r = 0; r < R; r++) qb[u2][r] = qrow[r][(dr >> 1) + u2];   // s_load_dword, a batch ahead of its use
    };
    auto compute_batch = [&](const WV* wb, const SV* sb, const SV* mb, const uint32_t (*qb)[R]) {
#pragma unroll
        for (int u = 0; u < U; u++) row(qb[u / 2], (u & 1) != 0, wb[u], sb[u], mb[u]);
    };
    static_assert(U % 2 == 0, "batches start on even channels");

    if (valid) {
        // ping-pong register buffers: batch n+1 is in flight while batch n is consumed
        WV wA[U], wB[U];
        SV sA[U], sB[U], mA[U], mB[U];
        uint32_t qA[U / 2][R], qB[U / 2][R];
        const int nfull = nrows / U;
        if (nfull > 0) load_batch(d0, wA, sA, mA, qA);
        int it = 0;
        for (; it + 2 <= nfull; it += 2) {
            load_batch(d0 + (it + 1) * U, wB, sB, mB, qB);
            compute_batch(wA, sA, mA, qA);
            if (it + 2 < nfull) load_batch(d0 + (it + 2) * U, wA, sA, mA, qA);
            compute_batch(wB, sB, mB, qB);
        }
        if (it < nfull) compute_batch(wA, sA, mA, qA);
        kivi_stamp<DBG>(a.dbg, 3);
        for (int d = d0 + nfull * U; d < d1; d++) {   // channel tail (rows per wave not a multiple of U)
            WV w = buf_load<WV, NT>(rc, coff, (uint32_t)d * cstep);
            SV sv = buf_load<SV, NT>(rs, soff, (uint32_t)d * sstep);
            SV mv = buf_load<SV, NT>(rm, soff, (uint32_t)d * sstep);
            uint32_t qp[R];
#pragma unroll
            for (int r = 0; r < R; r++) qp[r] = qrow[r][d >> 1];
            if (d & 1) row(qp, true, w, sv, mv);
            else row(qp, false, w, sv, mv);
        }
    }

    // acc -> sum_d q*scale*code + sum_d q*mn   (partial over this wave's channels)
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            const int g = (NGL == 1) ? 0 : (i / G);
            acc[r][i] = __builtin_fmaf(acc[r][i], post_scale<BITS, MODE>(i % FPI), zacc[r][g]);
        }

    if constexpr (DSPLIT == 1) {
        if (valid && lds_out) {
#pragma unroll
            for (int i = 0; i < NACC; i++) lds_out[word0 * FPI + i] = f2h_bits(acc[0][i]);
        } else if (valid) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint16_t* op = a.out + b * a.out_sb + (int64_t)(h0 + r) * a.out_sh + word0 * FPI;
#pragma unroll
                for (int c = 0; c < NACC / 8; c++) {
                    u16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = f2h_bits(acc[r][c * 8 + e]);
                    *(u16x8*)(op + c * 8) = v;
                }
            }
        }
    } else {
        // red[tib][dst dpart][slot][r][i][lane]; slot = source dpart with dst skipped
        float keep[R][Q];
#pragma unroll
        for (int dq = 0; dq < DSPLIT; dq++) {
            if (dq == dpart) {
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int i = 0; i < Q; i++) keep[r][i] = acc[r][dq * Q + i];
            } else if (valid) {
                const int slot = dpart < dq ? dpart : dpart - 1;
                float* dst = red + (size_t)(((tib * DSPLIT + dq) * (DSPLIT - 1) + slot) * R * Q) * 64 + lane;
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int i = 0; i < Q; i++) dst[(r * Q + i) * 64] = acc[r][dq * Q + i];
            }
        }
        __syncthreads();
        if (valid) {
            const float* src = red + (size_t)((tib * DSPLIT + dpart) * (DSPLIT - 1) * R * Q) * 64 + lane;
#pragma unroll
            for (int r = 0; r < R; r++) {
#pragma unroll
                for (int i = 0; i < Q; i++)
#pragma unroll
                    for (int sl = 0; sl < DSPLIT - 1; sl++) keep[r][i] += src[((sl * R + r) * Q + i) * 64];
                if (lds_out) {
                    if (r == 0) {
#pragma unroll
                        for (int i = 0; i < Q; i++) lds_out[word0 * FPI + dpart * Q + i] = f2h_bits(keep[0][i]);
                    }
                    continue;
                }
                uint16_t* op = a.out + b * a.out_sb + (int64_t)(h0 + r) * a.out_sh + word0 * FPI + dpart * Q;
                if constexpr (Q % 8 == 0) {
#pragma unroll
                    for (int c = 0; c < Q / 8; c++) {
                        u16x8 v;
#pragma unroll
                        for (int e = 0; e < 8; e++) v[e] = f2h_bits(keep[r][c * 8 + e]);
                        *(u16x8*)(op + c * 8) = v;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < Q / 4; c++) {
                        typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
                        u16x4 v;
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = f2h_bits(keep[r][c * 4 + e]);
                        *(u16x4*)(op + c * 4) = v;
                    }
                }
            }
        }
    }
}


// host: argument checks shared by every entry point that runs the packed-K product
int k_check_and_fill(GemvKArgs& a, const void* q, int64_t q_sb, int64_t q_sh, const void* code, int64_t code_sb,
                     int64_t code_sh, int64_t code_sr, const void* scale, const void* mn, int64_t sm_sb,
                     int64_t sm_sh, int64_t sm_sr, void* out, int64_t out_sb, int64_t out_sh, int B, int nh,
                     int nh_kv, int D, int64_t T, int group_size, int bits) {
    KIVI_REQUIRE(bits == 2 || bits == 4, KIVI_EINVAL, "kivi_gemv_k: bits must be 2 or 4 (matmul.py:215), got %d", bits);
    KIVI_REQUIRE(nh_kv > 0 && nh > 0 && nh % nh_kv == 0, KIVI_EINVAL,
                 "kivi_gemv_k: nh %% nh_kv != 0 (matmul.py:216): nh=%d nh_kv=%d", nh, nh_kv);
    const int fpi = 32 / bits;
    KIVI_REQUIRE(group_size > 0 && group_size % fpi == 0, KIVI_EINVAL,
                 "kivi_gemv_k: group_size %d must be a positive multiple of %d", group_size, fpi);
    KIVI_REQUIRE(T >= 0 && T % fpi == 0 && T % group_size == 0, KIVI_EINVAL,
                 "kivi_gemv_k: T=%lld must be a multiple of group_size=%d", (long long)T, group_size);
    KIVI_REQUIRE(B > 0 && D > 0, KIVI_EINVAL, "kivi_gemv_k: empty batch or head_dim");
    a.q = (const uint16_t*)q; a.q_sb = q_sb; a.q_sh = q_sh;
    a.code = (const uint32_t*)code; a.code_sb = code_sb; a.code_sh = code_sh; a.code_sr = code_sr;
    a.scale = (const uint16_t*)scale; a.mn = (const uint16_t*)mn;
    a.sm_sb = sm_sb; a.sm_sh = sm_sh; a.sm_sr = sm_sr;
    a.out = (uint16_t*)out; a.out_sb = out_sb; a.out_sh = out_sh;
    a.nh = nh; a.ratio = nh / nh_kv; a.D = D; a.T = T; a.Tw = T / fpi;
    a.units_per_b = nh; a.tile_blocks = 1;
    a.page_words = 0; a.page_groups = 0; a.code_sp = 0; a.sm_sp = 0;
    a.main_blocks = -1; a.res_blocks = 0; a.kres = nullptr; a.knew = nullptr; a.res_len = 0; a.dbg = nullptr;
    a.kres_sb = a.kres_sh = a.kres_st = a.knew_sb = a.knew_sh = 0;
    return 0;
}


}  // namespace

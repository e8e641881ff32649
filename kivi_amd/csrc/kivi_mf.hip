// Matrix-pipe decode kernels over the KT / VT cache layouts (kivi_mfma_layout.h) for nh / nh_kv = R in {1, 4, 8}:
// the packed qK^T and sV of models/llama_kivi.py:324 / :382 (kernel quant/csrc/gemv_cuda.cu:348-427, head mapping
// :361-365) and the whole decode step around them (:314-399).  Device bodies and the reasoning: kivi_mf_dev.h.
// Every kernel takes the range flag of its unit's store(s) (kivi_mfma_layout.h: where q'' / p'' are placed so that the fp16
// operands stay finite for any finite scale) and, optionally, the step's lengths from device memory (MfStep: hipGraph replay).
//
//   mf_k_kernel    one wave per super-block: scores (raw, or scaled + masked with softmax statistics per segment) + the
//                  fp16 K-residual role (q . K_full, K append) in short blocks at the tail of the grid
//   mf_v_kernel    slices of a unit's packed V: probabilities from the score rows + segment statistics (or given fp16
//                  probabilities: kivi_gqa_output), packed sV, fp16 window + V append + quantisation of the token leaving
//                  the window, partial sums meet in the workspace
//   mf_row_kernel  R = 1 (MHA), rows <= 8192 keys: the whole step of a (batch row, head) in ONE block -- packed qK^T ->
//                  LDS scores -> residual scores -> softmax -> window -> packed sV -> output; nothing but the output
//                  and the cache appends goes to memory
//   mf_row4_kernel the same for the four query heads of a kv head (R = 4, rows <= 9216 keys)
#include <stdlib.h>
#include <string.h>

#include "kivi_common.h"
#include "kivi_gqa_dev.h"
#include "kivi_quant.h"
#include "kivi_gqa_roles.h"
#include "kivi_mf_dev.h"

namespace {

// Tuning builds, KIVI_MF_XCD=1: the blocks of the two-launch form are renumbered so that consecutive LOGICAL blocks -- the
// chunks / slices of one (batch row, kv head) -- run on ONE XCD (the hardware deals consecutive workgroup ids round-robin over
// the 8 XCDs) and a unit sits on the same XCD in the qK^T and the sV launch: the score rows and statistics written by the first
// are then read through the L2 that wrote them.  Measured, not adopted: profiles/r04_xcd_affinity.log.
#ifdef KIVI_TUNING
__device__ int mf_tune_xcd = 0;
void mf_tune_sync() {
    static int done = 0;
    if (done) return;
    const char* e = KIVI_TUNE_ENV("KIVI_MF_XCD");
    const int v = e ? atoi(e) : 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(mf_tune_xcd), &v, sizeof(int));
    done = 1;
}
__device__ __forceinline__ int mf_block_id(int nblk) {
    const int bid = (int)blockIdx.x;
    if (!mf_tune_xcd) return bid;
    const int nb8 = nblk & ~7;
    return bid < nb8 ? (bid & 7) * (nb8 >> 3) + (bid >> 3) : bid;
}
#else
__device__ __forceinline__ int mf_block_id(int) { return (int)blockIdx.x; }
#endif

// ------------------------------------------------------------------------------------------------ qK^T launch

// per-wave LDS words of mf_k_kernel: [scale of the super-block: 1024 words, R = 4 only | R x 512 fp16 scores]
template <int R>
constexpr int mf_k_lds_words() { return (R == 1 ? 64 : 0) + R * 256; }   // R = 1: 64 words for the q operand

// DIAG (tuning builds, wrong results): 1 = nothing leaves the LDS (no flush), 2 = scores stored without the statistics
template <int R, int W, int RING, int DIAG = 0, int BITS = 2>
__global__ __launch_bounds__(64 * W) void mf_k_kernel(const GqaKArgs a, int spw) {
    static_assert(BITS == 2 || R == 4 || R == 1, "4-bit codes: nh / nh_kv in {1, 4}");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    // the step's lengths: by value, or device-resident (a.dyn).  (Only these two scalars: a mutable copy of the whole argument block
    // cost 3-8 % of the raw-score launch -- the flush lambda then reads its fields from a local object instead of the kernarg segment.)
    const long long Tq = a.dyn ? a.dyn->Tq : (long long)a.Tq;
    const int main_blocks = (int)gridDim.x - a.res_blocks;
    if ((int)blockIdx.x >= main_blocks) {                            // short residual blocks at the tail of the grid
        gqa_k_residual<R>(a, (int)blockIdx.x - main_blocks, Tq, a.dyn ? a.dyn->k_res_len : a.res_len);
        return;
    }
    const int bid = mf_block_id(main_blocks);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* lds_w = lds_all + wave * mf_k_lds_words<R>();
    uint16_t* lds_o = (uint16_t*)(lds_w + (R == 1 ? 64 : 0));
    const int unit = bid / a.sb_blocks;
    const int sb0 = ((bid - unit * a.sb_blocks) * W + wave) * spw;  // this wave: super-blocks sb0 .. sb0 + spw - 1
    if (sb0 >= a.nsb) return;
    const int b = unit / a.nh_kv, hk = unit - b * a.nh_kv;
    const int h0 = hk * a.ratio;

    // scores of one finished super-block (R x 512 fp16 in lds_o) -> memory: one 16-byte store per lane and head; the decode
    // step scales + masks them as the reference feeds its softmax (llama_kivi.py:339, :364-372) and leaves (max, sum exp)
    auto flush_sb = [&](int sb, int ng) {
        if constexpr (DIAG == 1) return;
        __builtin_amdgcn_wave_barrier();
        const bool valid = lane * 8 < ng * 32;
        const uint16_t* mrow = a.mask ? a.mask + b * a.mask_sb + (int64_t)sb * KIVI_MF_SB_TOKENS + lane * 8 : nullptr;
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            u32x4 v = valid ? *(const u32x4*)(lds_o + rr * 512 + lane * 8) : u32x4{0, 0, 0, 0};
            if (a.stats && DIAG != 2) {
                float m;
                if (mrow) {                                        // masked rows: element-wise (:366-372)
                    m = -__builtin_inff();
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint16_t lo = kivi_scaled_score((uint16_t)(v[i] & 0xFFFFu), a.inv_scale, true, valid ? mrow[2 * i] : 0);
                        const uint16_t hi = kivi_scaled_score((uint16_t)(v[i] >> 16), a.inv_scale, true, valid ? mrow[2 * i + 1] : 0);
                        v[i] = (uint32_t)lo | ((uint32_t)hi << 16);
                        m = __builtin_fmaxf(m, __builtin_fmaxf(h2f_bits(lo), h2f_bits(hi)));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = mf_scale_pair(v[i], a.inv_scale);
                    typedef _Float16 hp2 __attribute__((ext_vector_type(2)));
                    // (scalar copies: __builtin_bit_cast applied directly to an element of an ext-vector reads element 0)
                    const uint32_t w0 = v[0], w1 = v[1], w2 = v[2], w3 = v[3];
                    const hp2 m2 = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_bit_cast(hp2, w0), __builtin_bit_cast(hp2, w1)),
                                                             __builtin_elementwise_max(__builtin_bit_cast(hp2, w2), __builtin_bit_cast(hp2, w3)));
                    const uint32_t mb = __builtin_bit_cast(uint32_t, m2);
                    m = __builtin_fmaxf(h2f_bits((uint16_t)(mb & 0xFFFFu)), h2f_bits((uint16_t)(mb >> 16)));
                }
                m = wave_max(valid ? m : -__builtin_inff());
                typedef float fp2 __attribute__((ext_vector_type(2)));
                const fp2 l2e = {1.44269504088896340736f, 1.44269504088896340736f};
                fp2 acc = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const fp2 d = (fp2){mf_sub_lo(v[i], -m), mf_sub_hi(v[i], -m)} * l2e;       // kivi_exp(x - m)
                    acc += (fp2){__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
                }
                const float l = wave_sum(valid ? acc[0] + acc[1] : 0.f);
                if (lane == 0) {
                    float* st = a.stats + (((int64_t)b * a.nh + h0 + rr) * a.nseg + sb) * 2;
                    st[0] = m;
                    st[1] = l;
                }
            }
            if (valid)
                *(u32x4*)(a.out + b * a.out_sb + (int64_t)(h0 + rr) * a.out_sh + (int64_t)sb * KIVI_MF_SB_TOKENS + lane * 8) = v;
        }
        __builtin_amdgcn_wave_barrier();
    };

    const rsrc_t rk = make_rsrc(mf_sb(a.kt, b, hk, 0), (uint32_t)((int64_t)a.nsb * a.kt.sb_s * 4));
    const int rsh = mf_range_shift(__builtin_amdgcn_readfirstlane(a.range[unit]));  // range shift of the unit's K store (kivi_mfma_layout.h)
    MfKSeq seq;
    seq.sb_bytes = (uint32_t)(a.kt.sb_s * 4);
    seq.sb_first = sb0;
    seq.sb_stride = 1;
    seq.n_sb = (sb0 + spw <= a.nsb) ? spw : a.nsb - sb0;
    const int64_t tok_end = (int64_t)(sb0 + seq.n_sb) * KIVI_MF_SB_TOKENS;
    seq.ng_total = (int)(((Tq < tok_end ? Tq : tok_end) - (int64_t)sb0 * KIVI_MF_SB_TOKENS) / 32);
    if constexpr (R == 1) {
        mf_k_seq1<RING, BITS>(rk, seq, a.q + b * a.q_sb + (int64_t)h0 * a.q_sh, lds_w, rsh,
                        [&](int, int tt, float v) { lds_o[tt] = f2h_bits(v); }, flush_sb);
    } else {
        // R = 4 / 8: the same continuous walk (mf_k_seqR: scale requested a round ahead, the code ring runs across super-blocks)
        const int hb = (4 * (lane >> 4)) % R;                       // heads hb .. hb + 3 sit in this lane's result registers
        mf_k_seqR<R, RING, BITS>(rk, seq, a.q + b * a.q_sb + (int64_t)h0 * a.q_sh, a.q_sh, rsh,
                           [&](int, int tt, int r, float v0, float v1) {
                               const uint32_t hp = mf_cvt_pair(v0, v1);
                               lds_o[(hb + r) * 512 + tt] = (uint16_t)(hp & 0xFFFFu);
                               lds_o[(hb + r) * 512 + tt + 16] = (uint16_t)(hp >> 16);
                           }, flush_sb);
    }
}

template <int R, int W, int RING, int DIAG = 0, int BITS = 2>
void launch_mf_k(const GqaKArgs& a, int units, int spw, hipStream_t s) {
    const size_t lds = (size_t)W * mf_k_lds_words<R>() * 4;
    KIVI_LAUNCH_LDS((mf_k_kernel<R, W, RING, DIAG, BITS>), dim3((unsigned)(a.res_blocks + units * a.sb_blocks)), dim3(64 * W), lds, s, a, spw);
}

int run_mf_k(GqaKArgs& a, int units, int bits, hipStream_t s) {
#ifdef KIVI_TUNING
    mf_tune_sync();
#endif
    // a wave walks `spw` consecutive super-blocks of its unit (the next one's operands are requested while the current one is
    // multiplied): 2 when that still leaves >= 4 waves per SIMD, else 1; few super-blocks: one wave per block spreads them
    const int64_t total = (int64_t)units * a.nsb;
    int spw = (total >= 8192 && a.nsb >= 2) ? 2 : 1;
    if (a.ratio >= 4) {
        // R = 4 / 8 hold ~190 registers (two waves per SIMD = 2048 resident waves): as many super-blocks per wave as keeps the
        // launch in one round; the streams stay deep enough at that occupancy (cf. the qK^T phase of mf_row4_kernel)
        spw = (int)((total + 2047) / 2048);
        spw = spw < 1 ? 1 : (spw > 8 ? 8 : spw);
        if (spw > a.nsb) spw = a.nsb > 0 ? a.nsb : 1;
    }
    static const char* fs = KIVI_TUNE_ENV("KIVI_MF_SPW");                 // tuning aid
    if (fs) spw = atoi(fs) > 0 ? atoi(fs) : 1;
    const int chunks = (a.nsb + spw - 1) / spw;                     // waves per unit
    const int W = ((int64_t)units * chunks >= 2048) ? 4 : 1;
    a.sb_blocks = (chunks + W - 1) / W;
    if ((int64_t)a.res_blocks + (int64_t)units * a.sb_blocks == 0) return 0;
    KIVI_REQUIRE(bits == 2 || (bits == 4 && (a.ratio == 4 || a.ratio == 1)), KIVI_EUNSUPPORTED, "mf_k: %d-bit codes with nh / nh_kv = %d have no matrix-pipe kernel", bits, a.ratio);
    if (bits == 4) {                                                // 4-bit codes: nh / nh_kv in {1, 4}
        if (a.ratio == 1) { if (W == 4) launch_mf_k<1, 4, 2, 0, 4>(a, units, spw, s); else launch_mf_k<1, 1, 2, 0, 4>(a, units, spw, s); }
        else if (W == 4) launch_mf_k<4, 4, 4, 0, 4>(a, units, spw, s); else launch_mf_k<4, 1, 4, 0, 4>(a, units, spw, s);
        return kivi_launch_status("mf_k");
    }
    // two code blocks in flight per wave: 39.2 us per BASELINE configs[1] launch against 40.8 with four (fewer registers, the
    // same bytes in flight per SIMD); -DKIVI_TUNING builds keep the four-deep ring for A/B
#ifdef KIVI_TUNING
    static const char* fd = KIVI_TUNE_ENV("KIVI_MF_K_DIAG");             // 1 / 2: see mf_k_kernel (R = 4, four-wave blocks)
    if (fd && a.ratio == 4 && W == 4) {
        if (atoi(fd) == 1) launch_mf_k<4, 4, 4, 1>(a, units, spw, s); else launch_mf_k<4, 4, 4, 2>(a, units, spw, s);
        return kivi_launch_status("mf_k");
    }
    static const char* fr = KIVI_TUNE_ENV("KIVI_MF_RING");
    if (fr && atoi(fr) == 4 && a.ratio != 8) {
        if (a.ratio == 1) { if (W == 4) launch_mf_k<1, 4, 4>(a, units, spw, s); else launch_mf_k<1, 1, 4>(a, units, spw, s); }
        else { if (W == 4) launch_mf_k<4, 4, 4>(a, units, spw, s); else launch_mf_k<4, 1, 4>(a, units, spw, s); }
        return kivi_launch_status("mf_k");
    }
    if (fr && atoi(fr) == 2 && a.ratio == 4) {
        if (W == 4) launch_mf_k<4, 4, 2>(a, units, spw, s); else launch_mf_k<4, 1, 2>(a, units, spw, s);
        return kivi_launch_status("mf_k");
    }
    if (fr && atoi(fr) == 8 && a.ratio == 4) {
        if (W == 4) launch_mf_k<4, 4, 8>(a, units, spw, s); else launch_mf_k<4, 1, 8>(a, units, spw, s);
        return kivi_launch_status("mf_k");
    }
#endif
    if (a.ratio == 1) { if (W == 4) launch_mf_k<1, 4, 2>(a, units, spw, s); else launch_mf_k<1, 1, 2>(a, units, spw, s); }
    else if (a.ratio == 4) { if (W == 4) launch_mf_k<4, 4, 4>(a, units, spw, s); else launch_mf_k<4, 1, 4>(a, units, spw, s); }
    else { if (W == 4) launch_mf_k<8, 4, 4>(a, units, spw, s); else launch_mf_k<8, 1, 4>(a, units, spw, s); }
    return kivi_launch_status("mf_k");
}

// ------------------------------------------------------------------------------------------------ sV launch

// One super-block's R x 512 scaled probabilities p'' into this wave's LDS rows (pitch 512 halves): lane l owns tokens
// 8 l .. 8 l + 7 of every head.  mf_probs_request issues the loads of the scores (or, PROB, of given fp16 probabilities:
// kivi_gqa_output) -- one 16-byte load per lane and head, issued a super-block ahead of its use -- and mf_probs_store turns
// them into p = fp16(exp(x - M) / sum) exactly as the reference casts them (llama_kivi.py:375) times 2^(Sp + 4 | 6)
// (mf_scale_p), packed math as in mf_row_softmax.  Tokens at or past Tv get 0.
template <int R>
__device__ __forceinline__ void mf_probs_request(rsrc_t rx, uint32_t x_row_bytes, int64_t tok0, u32x4* xv) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < R; r++) xv[r] = buf_load<u32x4, false>(rx, (uint32_t)(r * x_row_bytes + (tok0 + lane * 8) * 2), 0);
}
template <int R, bool PROB, int BITS = 2>
__device__ __forceinline__ void mf_probs_store(const u32x4* xv, int64_t tok0, int64_t Tv, const float* M, const float* invS,
                                               const int* sp, int rsh, uint16_t* lds_p) {
    typedef _Float16 hp2 __attribute__((ext_vector_type(2)));
    typedef float fp2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int64_t left = Tv - tok0 - lane * 8;                      // tokens e < left are inside the packed prefix
    const fp2 l2e = {1.44269504088896340736f, 1.44269504088896340736f};
#pragma unroll
    for (int r = 0; r < R; r++) {
        const _Float16 m_sp = mf_p_mul_sp(sp[r], rsh);             // 2^-4 .. 2^14 
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            hp2 pp;
            const uint32_t xw = xv[r][i];                          // (scalar copy before the cast: see flush_sb)
            if constexpr (PROB) pp = __builtin_bit_cast(hp2, xw);
            else {
                const fp2 d = (fp2){mf_sub_lo(xw, -M[r]), mf_sub_hi(xw, -M[r])} * l2e;                 // kivi_exp(x - M)
                const fp2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
                pp = __builtin_convertvector(e * (fp2){invS[r], invS[r]}, hp2);
            }
            const _Float16 m_a = mf_p_mul_a(BITS == 4 || i >= 2, rsh);                                 // tokens (e & 4): 2^6, else 2^4 (4-bit codes: 2^6)
            o[i] = __builtin_bit_cast(uint32_t, (pp * (hp2){m_a, m_a}) * (hp2){m_sp, m_sp});   // (order: see mf_p_mul_a)
        }
        if (left < 8) {                                            // the end of the packed prefix falls into this lane's eight
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = (2 * i >= left) ? 0u : ((2 * i + 1 >= left) ? (o[i] & 0xFFFFu) : o[i]);
        }
        *(u32x4*)(lds_p + r * 512 + lane * 8) = o;
    }
}

// PW: window probabilities per head kept in LDS
constexpr int MF_PW = 200;                                       // >= NW * TW of GqaWindow (4 x 40, 8 x 24), rows 16-byte aligned

// HL (R = 4): hi / lo of p'' * scale in MFMA rows (MfVStream<4, ., true>), as in mf_row4_kernel
template <int R, int RING, bool PROB, bool HL = false, int BITS = 2>
__global__ __launch_bounds__(256, R == 8 ? 2 : 1) void mf_v_kernel(const GqaVArgs a_in) {   // (R = 8: two blocks per CU, gqa_v_slices)
    static_assert(BITS == 2 || R == 4 || R == 1, "4-bit codes: nh / nh_kv in {1, 4}");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];                          // 4 waves x (R x 256 words of p'' | 128 words of dot sums)
    GqaVArgs a = a_in;
    a.take_dyn();
    __shared__ uint16_t pw[R][MF_PW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WW = R * 256 + 128;                              // per-wave words
    uint16_t* lds_p = (uint16_t*)(lds_all + wave * WW);
    float* zl = (float*)(lds_all + wave * WW + R * 256);
    const int nstream = a.units * a.S;
    const int bid = mf_block_id(nstream);                          // (window blocks, at the tail, keep their ids)
    const bool win_role = bid >= nstream;
    const int unit = win_role ? bid - nstream : bid / a.S;
    const int slice = win_role ? a.S : bid - unit * a.S;           // = the block's partial-sum slot
    const int b = unit / a.nh_kv, hk = unit - b * a.nh_kv;
    const int h0 = hk * a.ratio;

    float M[R], invS[R];
    int sp[R];
    int ksh = 0;                                                   // what the unit's scales take of the big-scale shift (mf_ksh), the largest over the R rows
    // range shift of the unit's V store: the blocks of a unit may read different words in the step whose V flush marks it (the token
    // the mark is for is not part of this step's packed prefix); every block undoes its own 2^Sp before the hand-off
    const int rsh = mf_range_shift(__builtin_amdgcn_readfirstlane(a.range[unit]));
    if constexpr (PROB) {
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            M[rr] = 0.f;
            invS[rr] = 1.f;
            // (mf_row_sp_kernel read the same word: nothing marks a store between the two launches of kivi_gqa_output)
            const int v = __builtin_amdgcn_readfirstlane(a.sp_rows[(int64_t)b * a.nh + h0 + rr]);      // Sp * 8 + ksh
            sp[rr] = v >> 3;
            ksh = (v & 7) > ksh ? (v & 7) : ksh;
        }
    } else {
        gqa_row_consts<R>(a, b, h0, M, invS);
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            sp[rr] = mf_sp(1.0f / invS[rr], rsh);
            const int k = mf_ksh(1.0f / invS[rr], rsh);
            ksh = k > ksh ? k : ksh;
        }
    }

    const rsrc_t rx = make_rsrc(a.x + b * a.x_sb + (int64_t)h0 * a.x_sh, (uint32_t)((R - 1) * a.x_sh * 2 + ((a.Tv + 7) & ~(int64_t)7) * 2));
    const rsrc_t rv = make_rsrc(mf_sb(a.vt, b, hk, 0), (uint32_t)((int64_t)a.nsb * a.vt.sb_s * 4));
    const uint32_t sb_bytes = (uint32_t)(a.vt.sb_s * 4);

    MfVAcc<R, HL> A;
    mf_v_init(A);
    const int sb_begin = win_role ? 0 : slice * a.spb;
    const int sb_end = win_role ? 0 : ((sb_begin + a.spb < a.nsb) ? sb_begin + a.spb : a.nsb);
    // wave w streams super-blocks sb_begin + w, + 4, ... of the slice as ONE stream: the code ring runs across the super-blocks
    // (the probabilities of the next one are made while its first blocks are already in flight)
    static_assert(16 % RING == 0, "a super-block is a whole number of ring rounds");
    const int sb_w0 = sb_begin + wave;
    const int n_my = sb_end > sb_w0 ? (sb_end - sb_w0 + 3) / 4 : 0;
    if (n_my > 0) {
        const int last_sb = sb_w0 + 4 * (n_my - 1);
        int nb_last = (int)((a.Tv - (int64_t)last_sb * KIVI_MF_SB_TOKENS + 31) / 32);
        nb_last = nb_last > 16 ? 16 : nb_last;
        u32x4 xv[R];
        mf_probs_request<R>(rx, (uint32_t)(a.x_sh * 2), (int64_t)sb_w0 * KIVI_MF_SB_TOKENS, xv);
        MfVStream<R, RING, HL, BITS> vs;
        vs.prime(rv, sb_bytes, 0, 16 * (n_my - 1) + nb_last, sb_w0, 4);
        for (int i = 0; i < n_my; i++) {
            const int64_t tok0 = (int64_t)(sb_w0 + 4 * i) * KIVI_MF_SB_TOKENS;
            const int nb = (i == n_my - 1) ? nb_last : 16;
            __builtin_amdgcn_wave_barrier();                       // the previous super-block's LDS reads are over
            mf_probs_store<R, PROB, BITS>(xv, tok0, a.Tv, M, invS, sp, rsh, lds_p);
            // the next super-block's scores fly during this one's stream
            if (i + 1 < n_my) mf_probs_request<R>(rx, (uint32_t)(a.x_sh * 2), tok0 + 4 * KIVI_MF_SB_TOKENS, xv);
            __builtin_amdgcn_wave_barrier();
            vs.run(A, rv, 16 * i, 16 * i + nb, lds_p, 512, 16 * i * 32, ksh);
        }
    }

    // ---- fp16 window (+ V append + flush): the window block of the unit (tail of the grid), or shares inside the stream blocks
    float ow[R][2];
    {
        const int Lw = a.res_len + 1;
        const int wchunk = a.win_blocks ? Lw : (Lw + a.S - 1) / a.S;
        const int w0 = a.win_blocks ? 0 : slice * wchunk;
        const int w1 = a.win_blocks ? (win_role ? Lw : 0) : ((w0 + wchunk < Lw) ? w0 + wchunk : Lw);
        const bool flusher = a.flush && (a.win_blocks ? win_role : slice == 0);
        const int nwt = w1 > w0 ? w1 - w0 : 0;
        if (a.vres) {
            for (int idx = threadIdx.x; idx < R * nwt; idx += 256) {
                const int rr = idx / nwt, t = idx - rr * nwt;
                float Mr = M[0], Ir = invS[0];
#pragma unroll
                for (int q = 1; q < R; q++)
                    if (rr == q) { Mr = M[q]; Ir = invS[q]; }
                const uint16_t xw = a.x[b * a.x_sb + (int64_t)(h0 + rr) * a.x_sh + a.Tv + w0 + t];
                pw[rr][t] = PROB ? xw : f2h_bits(kivi_exp(h2f_bits(xw) - Mr) * Ir);
            }
            __syncthreads();
            gqa_window_part<R, 256, MF_PW, BITS>(a, b, hk, w0, w1, flusher, pw, ow);
        } else {
#pragma unroll
            for (int rr = 0; rr < R; rr++) ow[rr][0] = ow[rr][1] = 0.f;
        }
    }

    // per-wave [quantised part (R x 128) | window part (R x 128)] -> the block's sum -> workspace hand-off
    __syncthreads();                                               // every wave is done with its p'' rows
    float* Lf = (float*)(lds_all + wave * WW);
    mf_v_finish<R, RING, HL, BITS>(A, zl, Lf, (float)(1 << ksh));   // Lf[r * 128 + d], before 2^-Sp (HL: hi part, lo part behind it)
    if constexpr (HL) {
        for (int i = lane; i < R * 128; i += 64) Lf[i] += Lf[R * 128 + i];
        __builtin_amdgcn_wave_barrier();
    }
    // the p'' region of a wave holds R x 256 words = R x 128 floats twice: quantised part first, window part second
#pragma unroll
    for (int rr = 0; rr < R; rr++) {
        Lf[R * 128 + rr * 128 + 2 * lane] = ow[rr][0];
        Lf[R * 128 + rr * 128 + 2 * lane + 1] = ow[rr][1];
    }
    __syncthreads();
    float* lf = (float*)lds_all;
    constexpr int NT = (2 * R * 128 + 255) / 256;
    float tot[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int i = threadIdx.x + 256 * k;
        float v = 0.f;
        if (i < 2 * R * 128) {
            v = (lf[i] + lf[WW + i]) + (lf[2 * WW + i] + lf[3 * WW + i]);
            if (i < R * 128) {                                     // quantised part: undo the 2^Sp of its head
                int s_ = sp[0];
#pragma unroll
                for (int q = 1; q < R; q++)
                    if ((i >> 7) == q) s_ = sp[q];
                v = __builtin_ldexpf(v, -s_);
            }
        }
        tot[k] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int i = threadIdx.x + 256 * k;
        if (i < 2 * R * 128) lf[i] = tot[k];
    }
    __syncthreads();
    gqa_arrive_and_combine<R>(a, unit, slice, lf, b, h0);
}

// row maximum of given fp16 probabilities -> Sp of the row (kivi_gqa_output): 2^Sp * max p in [1, 2) (clamped to [0, 14])
__global__ __launch_bounds__(256) void mf_row_sp_kernel(const uint16_t* p, int64_t p_sb, int64_t p_sh, int nh, int nh_kv, int64_t T,
                                                        const int* range, int* sp) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int b = row / nh, h = row - b * nh;
    const uint16_t* pr = p + b * p_sb + (int64_t)h * p_sh;
    float m = 0.f;
    for (int64_t t = threadIdx.x; t < T; t += 256) m = __builtin_fmaxf(m, __builtin_fabsf(h2f_bits(pr[t])));
    m = kivi_block_reduce<4>(m, true, red);
    if (threadIdx.x == 0) {
        int e = 0;
        if (m > 0.f && m < __builtin_inff()) e = -((int)((__builtin_bit_cast(uint32_t, m) >> 23) & 255u) - 127);
        // + the range shift of the (batch row, kv head) the row reads, as mf_sp / mf_ksh split it; stored as Sp * 8 + ksh
        const int rsh = mf_range_shift(range[b * nh_kv + h / (nh / nh_kv)]);
        e = e < 0 ? 0 : (e > 14 ? 14 : e);
        const int d = rsh < 0 ? mf_big_d(e) : 0;
        sp[row] = (rsh < 0 ? e - d : e + rsh) * 8 + (rsh < 0 ? KIVI_MF_BIG_SHIFT_V - d : 0);
    }
}

// ------------------------------------------------------------------------------------------------ fused row (R = 1)

// The whole decode step of one (batch row, head) in one block of NW waves.  Dynamic LDS: the score / p'' row (n_pad halves).
// DBG (tools/mf_row_phases.py): every wave stamps the shader clock at its phase boundaries into av.dbg
// ak.dump (KIVI_GQA_DUMP_SCORES, tests): the fp16 row the softmax consumes (scaled, mask added) also goes to ak.out
// OCC: waves per SIMD the register budget allows (4: 128 registers; 2: the few-rows instantiation with rings of 8 -- at most one
// block per CU is resident anyway, so a wave may hold a half super-block of K and 8 blocks of V in flight)
// BITS = 4 (round 6): 4-bit K / V of a multi-head model (Llama-2-7B / LongChat-7B with KIVI-4) over the 4-bit super-blocks
template <int KRING, int VRING, int NW, bool DBG = false, bool PRIO = true, int OCC = 4, int BITS = 2>
__global__ __launch_bounds__(NW * 64, OCC) void mf_row_kernel(const GqaKArgs ak_in, const GqaVArgs av_in, int n_pad) {
    constexpr int NTH = NW * 64;
    GqaKArgs ak = ak_in;
    GqaVArgs av = av_in;
    ak.take_dyn();
    av.take_dyn();
    extern __shared__ __attribute__((aligned(16))) uint16_t row[];  // [n_pad] fp16 scores, then p''
    __shared__ float red[NW][128], resl[NW][128];
    __shared__ float zl[NW][128];
    __shared__ uint32_t q_lds[NW][64];
    __shared__ uint16_t pw[1][MF_PW];
    __shared__ float sm_lds[2 * NW];
    const int unit = (int)blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto stamp = [&](int i) {
        if constexpr (DBG) {
            const unsigned long long tck = __builtin_amdgcn_s_memtime();
            if (lane == 0) av.dbg[((size_t)blockIdx.x * NW + wave) * 16 + i] = tck;
        }
    };
    stamp(0);
    if (DBG && lane == 0) av.dbg[((size_t)blockIdx.x * NW + wave) * 16 + 1] = __builtin_amdgcn_s_memrealtime();
    const int b = unit / ak.nh_kv, hk = unit - b * ak.nh_kv;       // nh == nh_kv
    const int Tq = (int)ak.Tq, Tv = (int)av.Tv;
    const int L = ak.res_len + 1;                                  // residual keys incl. the new one
    const int n = Tq + L;                                          // row length
    // range flags of the unit's stores (kivi_mfma_layout.h): where q'' / p'' are placed; read before this step's V flush can set one
    const int krsh = mf_range_shift(__builtin_amdgcn_readfirstlane(ak.range[unit])), vrsh = mf_range_shift(__builtin_amdgcn_readfirstlane(av.range[unit]));

    const uint16_t* qrow = ak.q + b * ak.q_sb + (int64_t)hk * ak.q_sh;
    uint16_t* kres = ak.kres + b * ak.kres_sb + hk * ak.kres_sh;
    const uint16_t* knew = ak.knew + b * ak.knew_sb + hk * ak.knew_sh;

    // ---- packed qK^T: wave w walks super-blocks w, w + NW, ... of the row (mf_k_seq1: one memory round trip up front, the next
    // half's operands requested while the current one is multiplied).  The row in LDS holds the SCALED scores (:339); every
    // lane keeps the maximum of what it wrote (the softmax below starts from it: no separate pass for the maximum)
    float mxl = -__builtin_inff();
    {
        const rsrc_t rk = make_rsrc(mf_sb(ak.kt, b, hk, 0), (uint32_t)((int64_t)ak.nsb * ak.kt.sb_s * 4));
        MfKSeq seq;
        seq.sb_bytes = (uint32_t)(ak.kt.sb_s * 4);
        seq.sb_first = wave;
        seq.sb_stride = NW;
        seq.n_sb = ak.nsb > wave ? (ak.nsb - wave + NW - 1) / NW : 0;
        const int last = wave + (seq.n_sb - 1) * NW;                // this wave's last super-block
        const int NG = Tq >> 5;
        seq.ng_total = seq.n_sb > 0 ? 16 * (seq.n_sb - 1) + ((NG - 16 * last) < 16 ? (NG - 16 * last) : 16) : 0;
        mf_k_seq1<KRING, BITS>(rk, seq, qrow, q_lds[wave], krsh,
                         [&](int sb, int tt, float v) {
                             const uint16_t h = kivi_scaled_score(f2h_bits(v), ak.inv_scale, false, 0);
                             row[sb * KIVI_MF_SB_TOKENS + tt] = h;
                             mxl = __builtin_fmaxf(mxl, h2f_bits(h));
                         },
                         [](int, int) {});
    }
    stamp(3);
    // the latency-bound middle of the step (residual scores, softmax, window: ~15 us per block when it competes with the
    // streams of the three older blocks of its CU -- the SIMD arbiter is oldest-first) runs at raised priority
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
    // the first packed V blocks of this wave are requested now: they fly during the residual scores, the softmax and the window
    const rsrc_t rv = make_rsrc(mf_sb(av.vt, b, hk, 0), (uint32_t)((int64_t)av.nsb * av.vt.sb_s * 4));
    const int NB = (Tv + 31) >> 5;
    const int nbw = (NB + NW - 1) / NW;
    const int b_lo = wave * nbw;
    const int b_hi = (b_lo + nbw < NB) ? b_lo + nbw : NB;
    MfVStream<1, VRING, false, BITS> vs;
    vs.prime(rv, (uint32_t)(av.vt.sb_s * 4), b_lo, b_hi);
    // ---- residual scores q . [K_full | k_new] (fp32 accumulate, one rounding: the reference's fp16 matmul, :337) + K append
    for (int idx = threadIdx.x; idx < L * 8; idx += NTH) {
        const int sub = idx & 7, t = idx >> 3;
        const uint16_t* krow = ((t < ak.res_len) ? kres + (int64_t)t * ak.kres_st : knew) + sub * 16;
        const u16x8 k0 = *(const u16x8*)krow, k1 = *(const u16x8*)(krow + 8);
        const u16x8 q0 = *(const u16x8*)(qrow + sub * 16), q1 = *(const u16x8*)(qrow + sub * 16 + 8);
        float sc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) sc = __builtin_fmaf(h2f_bits(q0[e]), h2f_bits(k0[e]), sc);
#pragma unroll
        for (int e = 0; e < 8; e++) sc = __builtin_fmaf(h2f_bits(q1[e]), h2f_bits(k1[e]), sc);
        if (t == ak.res_len) {                                      // append the new key (:333-336)
            *(u16x8*)(kres + (int64_t)t * ak.kres_st + sub * 16) = k0;
            *(u16x8*)(kres + (int64_t)t * ak.kres_st + sub * 16 + 8) = k1;
        }
        sc += __shfl_xor(sc, 1);
        sc += __shfl_xor(sc, 2);
        sc += __shfl_xor(sc, 4);
        if (sub == 0) {
            const uint16_t h = kivi_scaled_score(f2h_bits(sc), ak.inv_scale, false, 0);
            row[Tq + t] = h;
            mxl = __builtin_fmaxf(mxl, h2f_bits(h));
        }
    }
    for (int j = n + (int)threadIdx.x; j < n_pad; j += NTH) row[j] = 0xFC00u;      // -inf past the row
    stamp(4);
    __syncthreads();
    stamp(5);

    // the fp16 window rows (and the token leaving it) are requested before the softmax and used after it
    GqaWindow<1, NTH, MF_PW, (NW == 8 ? 8 : 16), BITS> win;              // (prefetched tokens per wave: the whole share of a 33- / 65-token window)
    win.request(av, b, hk, 0, av.res_len + 1, av.flush != 0);
    for (int j = threadIdx.x; j < MF_PW; j += NTH) pw[0][j] = 0;    // (the window walk reads whole 8-token groups: zeros past the window;
                                                                   //  the softmax writes the probabilities behind its two barriers)
    // ---- [mask +] fp32 softmax of the row (llama_kivi.py:364-375): the probabilities of the packed prefix go back into the
    // row as p'', the window's into pw
    const uint16_t* mrow = ak.mask ? ak.mask + b * ak.mask_sb : nullptr;
    int ksh;                                                       // (mf_ksh of the row: the scales' share of a big-scale unit's shift)
    const int sp = mf_row_softmax<NTH, (8192 + 128 + NTH * 4 - 1) / (NTH * 4), BITS>(row, n, n_pad, Tv, mxl, mrow, pw[0], sm_lds, vrsh, ksh,
                                                        (ak.dump & 1) ? ak.out + b * ak.out_sb + (int64_t)hk * ak.out_sh : nullptr);
    __syncthreads();
    stamp(7);

    // ---- fp16 window: probs[-Lw:] . V_full, V append, quantisation of the token leaving the window (:377-399)
    float ow[1][2];
    win.finish(av, b, hk, 0, av.res_len + 1, av.flush != 0, pw, ow);
    resl[wave][2 * lane] = ow[0][0];
    resl[wave][2 * lane + 1] = ow[0][1];
    stamp(8);

    // ---- packed sV: contiguous block ranges per wave
    {
        MfVAcc<1> A;
        mf_v_init(A);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        vs.run(A, rv, b_lo, b_hi, row, 0, 0, ksh);
        stamp(9);
        mf_v_finish<1, VRING, false, BITS>(A, zl[wave], red[wave], (float)(1 << ksh));
    }
    __syncthreads();
    stamp(10);
    if (threadIdx.x < 128) {
        const int d = threadIdx.x;
        float qs = (red[0][d] + red[1][d]) + (red[2][d] + red[3][d]);
        float ws = (resl[0][d] + resl[1][d]) + (resl[2][d] + resl[3][d]);
        if constexpr (NW == 8) {
            qs += (red[4][d] + red[5][d]) + (red[6][d] + red[7][d]);
            ws += (resl[4][d] + resl[5][d]) + (resl[6][d] + resl[7][d]);
        }
        qs = __builtin_ldexpf(qs, -sp);
        // fp16(quantised part) + fp16(window part), rounded: the reference's `attn_output += matmul(...)` (llama_kivi.py:382-384);
        // only the window part exists before anything is quantised (:380)
        const uint16_t o = (Tv > 0) ? f2h_bits(h2f_bits(f2h_bits(qs)) + h2f_bits(f2h_bits(ws))) : f2h_bits(ws);
        av.out[b * av.out_sb + (int64_t)hk * av.out_sh + d] = o;
    }
    stamp(11);
    if (DBG && lane == 0) {
        unsigned long long* rec = av.dbg + ((size_t)blockIdx.x * NW + wave) * 16;
        rec[12] = __builtin_amdgcn_s_memrealtime();
        rec[13] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_ID: wave slot, SIMD, CU, SE
        rec[14] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // XCC_ID
    }
}

// ------------------------------------------------------------------------------------------------ fused row, R = 4 / 8
// The whole decode step of one (batch row, kv head) with its R = 4 (or 8) query heads in ONE launch: no score / statistics round
// trip through memory, no second launch.  Round 5: the softmax is no longer a phase of its own --
//   * its statistics come out of the K walk: when a wave has finished a 512-token super-block it takes (max, sum exp(x - max)) of
//     that segment of the R rows from the LDS (the arithmetic of mf_k_kernel's flush: [+ mask,] one exponential per score) while
//     its code ring keeps flying;
//   * after ONE block barrier every wave merges the segments (+ the fp16 residual's scores) into (M, 1 / sum) of the R rows;
//   * the probabilities p = fp16(exp(x - M) / sum) (the reference's cast, llama_kivi.py:375) times 2^(Sp + 4 | 6) are made in place,
//     15-16 blocks at a time, by the wave that streams those blocks of the packed V (mf_probs_inplace), under its own ring.
// (Round 4 ran three passes over the rows between two barriers, 16 of the launch's ~98 us at BASELINE config 4 with nothing streaming.)
// Rows that do not fit the LDS (or too few units to fill the chip) are cut into S SLICES of whole super-blocks, one block each: a
// slice's block walks its own keys and values; the slices of a unit exchange their (max, sum exp) through memory (one arrival
// counter; blocks WAIT for each other: every block of the grid is resident at once, or the block ids are handed out by a ticket
// counter in the order the blocks start, so a waiting block's partners have started or will start as soon as any older unit
// finishes), form the same p as one block would, and their partial outputs meet in the workspace (gqa_arrive_and_combine).  The
// last slice holds the tokens from the super-block of token Tv on -- the fp16 residual, the window, the appends and the V flush.
// Dynamic LDS: [R][n_pad] fp16 scores -> p''; reused for the per-wave partial sums at the end.
// VHL: hi / lo of p'' * scale in MFMA rows (MfVStream<4, ., true>: 8 instead of 16 matrix instructions per block)

__device__ __forceinline__ uint16_t mf_add_mask(uint16_t h, uint16_t m) {      // fp16(x + mask), clamped at the fp16 minimum (:366-372)
    float v = (float)(_Float16)(h2f_bits(h) + h2f_bits(m));
    if (v < -65504.0f) v = -65504.0f;
    return f2h_bits(v);
}

// p'' of the tokens [t0, t0 + ntok) (indices into the block's rows; t0 a multiple of 32) of ONE row, in place: lane l owns tokens
// 8 l .. 8 l + 7 (mf_probs_store on LDS-resident scores).  `left0`: tokens of the piece inside the packed prefix.
template <int BITS>
__device__ __forceinline__ void mf_probs_inplace_row(uint16_t* row, int t0, int ntok, int left0, float M, float invS, int sp, int rsh) {
    typedef _Float16 hp2 __attribute__((ext_vector_type(2)));
    typedef float fp2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    if (lane * 8 >= ntok) return;
    const int left = left0 - lane * 8;
    const fp2 l2e = {1.44269504088896340736f, 1.44269504088896340736f};
    uint16_t* p = row + t0 + lane * 8;
    const u32x4 xv = *(const u32x4*)p;
    const _Float16 m_sp = mf_p_mul_sp(sp, rsh);                    // 2^-4 .. 2^14 
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t xw = xv[i];
        const fp2 d = (fp2){mf_sub_lo(xw, -M), mf_sub_hi(xw, -M)} * l2e;                               // kivi_exp(x - M)
        const fp2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
        const hp2 pp = __builtin_convertvector(e * (fp2){invS, invS}, hp2);
        const _Float16 m_a = mf_p_mul_a(BITS == 4 || i >= 2, rsh);                                     // tokens (e & 4): 2^6, else 2^4 (4-bit codes: 2^6)
        o[i] = __builtin_bit_cast(uint32_t, (pp * (hp2){m_a, m_a}) * (hp2){m_sp, m_sp});
    }
    if (left < 8) {                                                // the end of the packed prefix falls into (or before) this lane's eight
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = (2 * i >= left) ? 0u : ((2 * i + 1 >= left) ? (o[i] & 0xFFFFu) : o[i]);
    }
    *(u32x4*)p = o;
}

// In-stream statistics are carried per LANE through the K walk (a running (max, sum exp) of the lane's eight scores per segment and
// head, rescaled when the maximum moves) and reduced across the wave ONCE at the end of the walk; two wave reductions per segment
// and head (each a chain of ~10 dependent DPP / readlane operations) measured 1-2 % slower (profiles/r05_row4_flows.log, "2434").
// R = 1 (nh == nh_kv; OCC = 4: four blocks per CU in <= 128 registers): the same block over mf_k_seq1 / MfVStream<1> -- the sliced form of
// multi-head rows (mf_row_kernel keeps the unsliced one)
// PSM ("phase softmax", unsliced rows only): the round-4 flow -- the K walk only writes the scores, every wave then takes whole rows
// through the three-pass softmax (mf_row_softmax_wave) between the two barriers, the V stream reads finished p''.  Kept because it
// measures FASTER than the in-stream form for a block that holds a whole row (BASELINE config 4, one box, alternating: 97.8 us
// against 104.0-104.8: the exponentials cost the same wherever they run, and inside the streams they delay a wave's next request
// -- profiles/r05_row4_flows.log); the in-stream form is what makes slices possible (their statistics must exist before the exchange).
template <int KRING, int VRING, int NW, bool DBG = false, bool VHL = true, int R = 4, int BITS = 2, int OCC = 2, bool PSM = false>
__global__ __launch_bounds__(NW * 64, OCC) void mf_row4_kernel(const GqaKArgs ak_in, const GqaVArgs av_in, int n_pad, int S) {
    constexpr int NTH = NW * 64;
    // NW = 6 (round 6, a block per row only): SIX waves per block, two blocks per CU = three waves per SIMD in <= 168 registers -- the
    // LDS (four score rows of up to 9344 keys: 73 KiB) allows two blocks per CU whatever their size, so the third wave per SIMD the
    // latency-bound streams of this kernel lacked comes from wider blocks, not from more of them.
    static_assert(NW == 4 || NW == 6, "four or six waves");
    static_assert((R == 4) || ((R == 8 || R == 1) && !VHL), "R = 1 / 8: chained hi / lo sV");
    static_assert(BITS == 2 || R == 4 || R == 1, "4-bit codes: nh / nh_kv in {1, 4}");
    GqaKArgs ak = ak_in;
    GqaVArgs av = av_in;
    ak.take_dyn();
    av.take_dyn();
    extern __shared__ __attribute__((aligned(16))) uint16_t rows[];   // [R][n_pad]
    __shared__ float zl[NW][128];
    __shared__ uint16_t pw[R][MF_PW];
    __shared__ float st_lds[R][NW][2];                             // (max, sum exp) of every wave's segments of the R rows
    __shared__ int sp_lds[2 * R];                                  // Sp of the R rows | their mf_ksh (PSM: wave r holds row r's)
    __shared__ int bid_lds;
    __shared__ uint32_t q_lds[R == 1 ? NW : 1][64];                // R = 1: the normalised q operand of mf_k_seq1
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = (int)blockIdx.x;
    if (ak.ticket) {                                               // block ids in the order the blocks start (see above)
        if (threadIdx.x == 0) {
            // EIGHT counters, chosen by blockIdx % 8: ids c, c + 8, c + 16, ... in the start order of the blocks of class c.  One counter
            // for the whole grid serialises its atomics on one address -- 512 blocks starting together: the last ticket ~7 us after
            // the first, +13 % on the 70B-like slice once round 6 made the ticket unconditional (profiles/r06_confirm.log) --; a class has
            // an eighth of the blocks and its own address.  Consecutive ids (the slices of a unit) sit in different classes with the same
            // or neighbouring ticket numbers; blocks are dispatched in blockIdx order, so the classes advance together (on a chip whose
            // XCDs take blockIdx round-robin a class IS an XCD's share: the XCD furthest behind finds all its partners started).
            int c = (int)blockIdx.x & (KIVI_GQA_TICKETS - 1);
            int nc = ((int)gridDim.x - c + KIVI_GQA_TICKETS - 1) / KIVI_GQA_TICKETS;                // blocks of this class
            int step = KIVI_GQA_TICKETS;
#ifdef KIVI_TUNING
            if (ak.dump & 4) { c = 0; nc = (int)gridDim.x; step = 1; }                               // A/B: ONE counter for the whole grid (KIVI_MF_ONE_TICKET=1)
#endif
            const int t = __hip_atomic_fetch_add(ak.ticket - c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == nc - 1) __hip_atomic_store(ak.ticket - c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
            bid_lds = c + step * t;
        }
        __syncthreads();
        bid = __builtin_amdgcn_readfirstlane(bid_lds);
    }
    auto stamp = [&](int i) {                                      // tools/mf_row_phases.py, slots as in mf_row_kernel
        if constexpr (DBG) {
            const unsigned long long tck = __builtin_amdgcn_s_memtime();
            if (lane == 0) av.dbg[((size_t)bid * NW + wave) * 16 + i] = tck;
        }
    };
    stamp(0);
    if (DBG && lane == 0) av.dbg[((size_t)bid * NW + wave) * 16 + 1] = __builtin_amdgcn_s_memrealtime();
    const int unit = bid / S, slice = bid - unit * S;
    const int b = unit / ak.nh_kv, hk = unit - b * ak.nh_kv;
    const int h0 = hk * R;
    const int Tq = (int)ak.Tq, Tv = (int)av.Tv;
    const int L = ak.res_len + 1;
    const int krsh = mf_range_shift(__builtin_amdgcn_readfirstlane(ak.range[unit])), vrsh = mf_range_shift(__builtin_amdgcn_readfirstlane(av.range[unit]));   // range shifts: see mf_row_kernel
    const uint16_t* q_h0 = ak.q + b * ak.q_sb + (int64_t)h0 * ak.q_sh;
    uint16_t* kres = ak.kres + b * ak.kres_sb + hk * ak.kres_sh;
    const uint16_t* knew = ak.knew + b * ak.knew_sb + hk * ak.knew_sh;
    const uint16_t* mrow = ak.mask ? ak.mask + b * ak.mask_sb : nullptr;
    uint16_t* dump0 = (ak.dump & 1) ? ak.out + b * ak.out_sb + (int64_t)h0 * ak.out_sh : nullptr;

    // ---- the slice: super-blocks [sb_lo, sb_hi) of the unit's packed keys, the packed values of the same tokens; `last`: the slice
    // that also owns the fp16 residual, the window, the appends and the V flush -- it starts no later than the super-block of token
    // Tv (the window's probabilities come from scores of packed keys when Tv < Tq), so it may be up to one super-block longer
    int sb_lo = 0, sb_hi = ak.nsb;
    bool last = true;
    if (S > 1) {
        const int spb = (ak.nsb + S - 1) / S;
        int last_start = (S - 1) * spb;
        if ((Tv >> 9) < last_start) last_start = Tv >> 9;
        last = slice == S - 1;
        const int lo = slice * spb, hi = lo + spb;
        sb_lo = last ? last_start : (lo < last_start ? lo : last_start);
        sb_hi = last ? ak.nsb : (hi < last_start ? hi : last_start);
    }
    const int tok0 = sb_lo * KIVI_MF_SB_TOKENS;                   // the rows are indexed by token - tok0

    // ---- packed qK^T: wave w walks super-blocks sb_lo + w, sb_lo + w + NW, ...; the rows hold the SCALED scores
    // fp16(fp16(s) * inv_scale) (:339; = kivi_scaled_score): two at a time -- one packed conversion, two v_fma_mix
    const int hb = (4 * (lane >> 4)) % R;
    float lm[R], ll[R];                                            // this lane's running (max, sum exp(x - max)) of every head
#pragma unroll
    for (int rr = 0; rr < R; rr++) { lm[rr] = -__builtin_inff(); ll[rr] = 0.f; }
    {
        const rsrc_t rk = make_rsrc(mf_sb(ak.kt, b, hk, 0), (uint32_t)((int64_t)ak.nsb * ak.kt.sb_s * 4));
        MfKSeq seq;
        seq.sb_bytes = (uint32_t)(ak.kt.sb_s * 4);
        seq.sb_first = sb_lo + wave;
        seq.sb_stride = NW;
        seq.n_sb = sb_hi > sb_lo + wave ? (sb_hi - sb_lo - wave + NW - 1) / NW : 0;
        const int lastsb = sb_lo + wave + (seq.n_sb - 1) * NW;
        const int NG = Tq >> 5;
        seq.ng_total = seq.n_sb > 0 ? 16 * (seq.n_sb - 1) + ((NG - 16 * lastsb) < 16 ? (NG - 16 * lastsb) : 16) : 0;
        if constexpr (PSM && NW == 6 && R == 4) {
            // six waves, a block per row (the K walk only writes scores): the row's groups are dealt in CONTIGUOUS runs of whole ring
            // rounds (4 groups) -- 16 super-blocks over six waves would be 3, 3, 3, 3, 2, 2 (the block waits for 48 groups), the runs
            // are 44, 44, 44, 40, 40, 40
            const int nu = (NG + 3) >> 2;
            const int base = nu / NW, rem = nu - base * NW;
            const int u0 = wave * base + (wave < rem ? wave : rem), un = base + (wave < rem ? 1 : 0);
            const int g0 = 4 * u0;
            int g1 = 4 * (u0 + un);
            g1 = g1 < NG ? g1 : NG;
            seq.sb_first = g0 >> 4;
            seq.sb_stride = 1;
            seq.g_first = g0 & 15;
            seq.n_sb = g1 > g0 ? ((g1 + 15) >> 4) - seq.sb_first : 0;
            seq.ng_total = g1 > g0 ? g1 - 16 * seq.sb_first : 0;
        }
        // a finished super-block: [mask in place (:366-372),] (max, sum exp(x - max)) of the segment of every row
        auto seg_done = [&](int sb, int ng) {
            typedef _Float16 hp2 __attribute__((ext_vector_type(2)));
            typedef float fp2 __attribute__((ext_vector_type(2)));
            if constexpr (PSM) return;
            __builtin_amdgcn_wave_barrier();
            const int seg = sb - sb_lo;
            const bool valid = lane * 8 < ng * 32;
            uint16_t mk[8];
            if (mrow) {
#pragma unroll
                for (int e = 0; e < 8; e++) mk[e] = valid ? mrow[(int64_t)sb * KIVI_MF_SB_TOKENS + lane * 8 + e] : (uint16_t)0;
            }
#pragma unroll
            for (int rr = 0; rr < R; rr++) {
                uint16_t* p = rows + rr * n_pad + seg * KIVI_MF_SB_TOKENS + lane * 8;
                u32x4 v = valid ? *(const u32x4*)p : u32x4{0xFC00FC00u, 0xFC00FC00u, 0xFC00FC00u, 0xFC00FC00u};
                float m;
                if (mrow) {
                    m = -__builtin_inff();
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint16_t lo = mf_add_mask((uint16_t)(v[i] & 0xFFFFu), mk[2 * i]);
                        const uint16_t hi = mf_add_mask((uint16_t)(v[i] >> 16), mk[2 * i + 1]);
                        v[i] = (uint32_t)lo | ((uint32_t)hi << 16);
                        m = __builtin_fmaxf(m, __builtin_fmaxf(h2f_bits(lo), h2f_bits(hi)));
                    }
                    if (valid) *(u32x4*)p = v;
                } else {
                    // (scalar copies: __builtin_bit_cast applied directly to an element of an ext-vector reads element 0)
                    const uint32_t w0 = v[0], w1 = v[1], w2 = v[2], w3 = v[3];
                    const hp2 m2 = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_bit_cast(hp2, w0), __builtin_bit_cast(hp2, w1)),
                                                             __builtin_elementwise_max(__builtin_bit_cast(hp2, w2), __builtin_bit_cast(hp2, w3)));
                    const uint32_t mb = __builtin_bit_cast(uint32_t, m2);
                    m = __builtin_fmaxf(h2f_bits((uint16_t)(mb & 0xFFFFu)), h2f_bits((uint16_t)(mb >> 16)));
                }
                if (dump0 && valid) *(u32x4*)(dump0 + (int64_t)rr * ak.out_sh + (int64_t)sb * KIVI_MF_SB_TOKENS + lane * 8) = v;
                const fp2 l2e = {1.44269504088896340736f, 1.44269504088896340736f};
                if (valid) {                                       // (no cross-lane step inside: lanes past the segment's end just skip it)
                    const float mo = lm[rr];
                    const float mn_ = __builtin_fmaxf(mo, m);
                    const float ms = mn_ == -__builtin_inff() ? 0.f : mn_;                 // (a lane whose scores are all -inf so far)
                    fp2 acc = {ll[rr] * kivi_exp(mo - ms), 0.f};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t xw = v[i];
                        const fp2 d = (fp2){mf_sub_lo(xw, -ms), mf_sub_hi(xw, -ms)} * l2e;       // kivi_exp(x - m)
                        acc += (fp2){__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
                    }
                    lm[rr] = mn_;
                    ll[rr] = acc[0] + acc[1];
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        if constexpr (R == 1) {
            mf_k_seq1<KRING, BITS>(rk, seq, q_h0, q_lds[wave], krsh, [&](int sb, int tt, float v) {
                rows[(sb - sb_lo) * KIVI_MF_SB_TOKENS + tt] = kivi_scaled_score(f2h_bits(v), ak.inv_scale, false, 0);
            }, seg_done);
        } else {
            mf_k_seqR<R, KRING, BITS, (PSM && NW == 6 && R == 4)>(rk, seq, q_h0, ak.q_sh, krsh, [&](int sb, int tt, int r, float v0, float v1) {
                const uint32_t hs = mf_scale_pair(mf_cvt_pair(v0, v1), ak.inv_scale);
                uint16_t* dst = rows + (hb + r) * n_pad + (sb - sb_lo) * KIVI_MF_SB_TOKENS + tt;
                dst[0] = (uint16_t)(hs & 0xFFFFu);                 // head hb + r at tokens tt, tt + 16
                dst[16] = (uint16_t)(hs >> 16);
            }, seg_done);
        }
    }
    if constexpr (!PSM) {                                          // the wave's (max, sum exp) of every head: one entry per wave
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            const float m = wave_max(lm[rr]);
            const float ms = m == -__builtin_inff() ? 0.f : m;
            const float l = wave_sum(ll[rr] * kivi_exp(lm[rr] - ms));
            if (lane == 0) {
                st_lds[rr][wave][0] = m;
                st_lds[rr][wave][1] = l;
            }
        }
    }
    stamp(3);
    __builtin_amdgcn_s_setprio(3);                                  // the latency-bound middle of the step (see mf_row_kernel)
    // the first packed V blocks of this wave are requested now: they fly during the residual scores, the statistics and the window
    const rsrc_t rv = make_rsrc(mf_sb(av.vt, b, hk, 0), (uint32_t)((int64_t)av.nsb * av.vt.sb_s * 4));
    const int vb_lo = tok0 >> 5;
    int vb_hi = ((sb_hi * KIVI_MF_SB_TOKENS < Tv ? sb_hi * KIVI_MF_SB_TOKENS : Tv) + 31) >> 5;
    if (vb_hi < vb_lo) vb_hi = vb_lo;
    const int nbw = (vb_hi - vb_lo + NW - 1) / NW;
    const int b_lo = vb_lo + wave * nbw;
    const int b_hi = (b_lo + nbw < vb_hi) ? b_lo + nbw : vb_hi;
    MfVStream<R, VRING, VHL, BITS> vs;
    vs.prime(rv, (uint32_t)(av.vt.sb_s * 4), b_lo, b_hi);
    // ---- residual scores q . [K_full | k_new] of the R heads (:337: fp32 accumulate, one rounding = the reference's fp16 matmul)
    // [+ mask] + K append (:333-336): the last slice.  Eight lanes per key (16 channels each) take the key against ALL R heads: a
    // key row is read once, every load of the phase is issued before the first product -- one memory round trip for up to 129 keys
    // (round 4 walked (head, key) pairs, 16 dependent round trips at residual_length 128)
    if (last) {
        constexpr int TPP = NTH / 8, KP = (129 + TPP - 1) / TPP;   // keys per pass, passes
        constexpr int RH = R > 4 ? 4 : R;                          // heads per q load (R = 8: two rounds over the key registers)
        const int sub = threadIdx.x & 7, ts = threadIdx.x >> 3;
        u16x8 ka[KP], kc[KP];
#pragma unroll
        for (int p = 0; p < KP; p++) {
            const int t = ts + p * TPP;
            ka[p] = kc[p] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (t < L) {
                const uint16_t* krow = ((t < ak.res_len) ? kres + (int64_t)t * ak.kres_st : knew) + sub * 16;
                ka[p] = *(const u16x8*)krow;
                kc[p] = *(const u16x8*)(krow + 8);
            }
        }
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += RH) {
            u16x8 qa[RH], qb[RH];
#pragma unroll
            for (int r = 0; r < RH; r++) {
                const uint16_t* qrow = q_h0 + (int64_t)(r0 + r) * ak.q_sh + sub * 16;
                qa[r] = *(const u16x8*)qrow;
                qb[r] = *(const u16x8*)(qrow + 8);
            }
#pragma unroll
            for (int p = 0; p < KP; p++) {
                const int t = ts + p * TPP;
                if (t < L) {                                       // (the eight lanes of a key agree)
                    if (r0 == 0 && t == ak.res_len) {
                        *(u16x8*)(kres + (int64_t)t * ak.kres_st + sub * 16) = ka[p];
                        *(u16x8*)(kres + (int64_t)t * ak.kres_st + sub * 16 + 8) = kc[p];
                    }
                    const bool rmask = !PSM && mrow != nullptr;    // (PSM: the softmax adds the mask over the whole row)
                    const uint16_t mk = rmask ? mrow[Tq + t] : (uint16_t)0;
#pragma unroll
                    for (int r = 0; r < RH; r++) {
                        float sc = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; e++) sc = __builtin_fmaf(h2f_bits(qa[r][e]), h2f_bits(ka[p][e]), sc);
#pragma unroll
                        for (int e = 0; e < 8; e++) sc = __builtin_fmaf(h2f_bits(qb[r][e]), h2f_bits(kc[p][e]), sc);
                        sc += dpp_f<0xB1>(sc);                     // the eight lanes of the key (order of the additions as before:
                        sc += dpp_f<0x4E>(sc);                     // neighbours, pairs, the two quads)
                        sc += dpp_f<0x141>(sc);
                        if (sub == 0) {
                            const uint16_t h = kivi_scaled_score(f2h_bits(sc), ak.inv_scale, rmask, mk);
                            rows[(r0 + r) * n_pad + (Tq - tok0) + t] = h;
                            if (!PSM && dump0) dump0[(int64_t)(r0 + r) * ak.out_sh + Tq + t] = h;
                        }
                    }
                }
            }
        }
    }
    if constexpr (PSM) {                                           // -inf past the rows (the softmax reads whole 4-score chunks)
        const int n = Tq + L;
        for (int j = (int)threadIdx.x; j < R * (n_pad - n); j += NTH) rows[(j / (n_pad - n)) * n_pad + n + j % (n_pad - n)] = 0xFC00u;
    }
    stamp(4);
    // the fp16 window rows (and the token leaving it) are requested before the barrier and used after the statistics
    // (prefetched window rows per wave: all 40 where the registers allow it)
    GqaWindow<R, NTH, MF_PW, (R == 8 ? 16 : (BITS == 4 ? 24 : 40)), BITS> win;
    if (last) win.request(av, b, hk, 0, av.res_len + 1, av.flush != 0);
    kivi_lds_barrier();                                            // (the V ring and the window rows stay in flight)
    stamp(5);

    // ---- (M, sum exp(x - M)) of the R rows: every wave merges the slice's segments + (last slice) the residual scores; S > 1: the
    // slices of the unit exchange theirs
    float M[R], invS[R];
    int sp[R];
    int ksh = 0;                                                   // the scales' share of a big-scale unit's shift (mf_ksh), the largest over the R rows
    if constexpr (PSM) {
        // ---- [mask +] softmax of the rows wave, wave + NW, ... (fp32, cast to fp16: :364-375): p'' in place, the window's into pw
#pragma unroll
        for (int rr = 0; rr < R; rr++) { M[rr] = 0.f; invS[rr] = 1.f; sp[rr] = 0; }
#pragma unroll 1
        for (int r = wave; r < R; r += NW) {
            for (int j = lane; j < NW * decltype(win)::TW; j += 64) pw[r][j] = 0;      // (zeros past the window: its walk reads whole 8-token groups)
            __builtin_amdgcn_wave_barrier();
            int kr;
            const int spr = mf_row_softmax_wave<BITS>(rows + r * n_pad, Tq + L, n_pad, Tv, mrow, pw[r], vrsh, kr,
                                                      dump0 ? dump0 + (int64_t)r * ak.out_sh : nullptr);
            if (lane == 0) {
                sp_lds[r] = spr;
                sp_lds[R + r] = kr;
            }
        }
    } else {
        const int nseg_loc = NW;                                   // entries of st_lds: one per wave
        float Ls[R];
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            const float sm = lane < nseg_loc ? st_lds[rr][lane][0] : -__builtin_inff();
            const float sl = lane < nseg_loc ? st_lds[rr][lane][1] : 0.f;
            float x0 = -__builtin_inff(), x1 = -__builtin_inff(), x2 = -__builtin_inff();
            if (last) {
                const uint16_t* rp = rows + rr * n_pad + (Tq - tok0);
                if (lane < L) x0 = h2f_bits(rp[lane]);
                if (lane + 64 < L) x1 = h2f_bits(rp[lane + 64]);
                if (lane + 128 < L) x2 = h2f_bits(rp[lane + 128]);
            }
            const float m = wave_max(__builtin_fmaxf(__builtin_fmaxf(sm, x0), __builtin_fmaxf(x1, x2)));
            const float ms = m == -__builtin_inff() ? 0.f : m;     // (an empty slice: no exp(-inf + inf))
            M[rr] = m;
            Ls[rr] = wave_sum(sl * kivi_exp(sm - ms) + kivi_exp(x0 - ms) + (kivi_exp(x1 - ms) + kivi_exp(x2 - ms)));
        }
        if (S > 1) {
            __shared__ int xok_lds;
            uint32_t* xs = reinterpret_cast<uint32_t*>(ak.stats) + (size_t)unit * S * R * 2;
            if (wave == 0) {
                if (lane == 0) {
#pragma unroll
                    for (int rr = 0; rr < R; rr++) {
                        __hip_atomic_store(xs + (slice * R + rr) * 2, __builtin_bit_cast(uint32_t, M[rr]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(xs + (slice * R + rr) * 2 + 1, __builtin_bit_cast(uint32_t, Ls[rr]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (payload written through before the arrival: cdna_hip_programming.md G16)
                if (lane == 0) {
                    bool arrive = true;
#ifdef KIVI_TUNING
                    if ((ak.dump & 2) && unit == 0 && slice == 0) arrive = false;     // fault injection (tests/test_mfma_gpu.py): a partner that never arrives
#endif
                    if (arrive) __hip_atomic_fetch_add(ak.xcount + unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // bounded: the ticket order makes a missing partner impossible for an ordinary launch (see above); if one never
                    // arrives all the same (~1 s of polls), the unit's output is poisoned with NaN instead of hanging the device AND
                    // the step is reported: KIVI_ETIMEOUT in the workspace's error word and in the process's host-visible one, which
                    // the next decode call returns (kivi_device_error).  The launch's last blocks still reset every counter.
                    int it = 0;
                    while (__hip_atomic_load(ak.xcount + unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < S && it < (1 << 21)) {
                        __builtin_amdgcn_s_sleep(16);
                        it++;
                    }
                    const bool ok_ = it < (1 << 21);
                    xok_lds = ok_;
                    if (!ok_) {
                        if (ak.err_ws) __hip_atomic_store(ak.err_ws, -KIVI_ETIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ak.err_host) {
                            __hip_atomic_store(ak.err_host + 1, unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store(ak.err_host, -KIVI_ETIMEOUT, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                }
            }
            __syncthreads();
            const bool ok = xok_lds != 0;
#pragma unroll
            for (int rr = 0; rr < R; rr++) {
                float sm = -__builtin_inff(), sl = 0.f;
                if (lane < S) {
                    sm = __builtin_bit_cast(float, __hip_atomic_load(xs + (lane * R + rr) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    sl = __builtin_bit_cast(float, __hip_atomic_load(xs + (lane * R + rr) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                }
                const float m = wave_max(sm);
                const float ms = sm == -__builtin_inff() ? m : sm; // (an empty slice's term: 0 * exp(0))
                M[rr] = ok ? m : __builtin_nanf("");
                Ls[rr] = wave_sum(sl * kivi_exp(ms - m));
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            invS[rr] = 1.0f / Ls[rr];
            sp[rr] = mf_sp(Ls[rr], vrsh);
            const int k = mf_ksh(Ls[rr], vrsh);
            ksh = k > ksh ? k : ksh;
        }
    }
    stamp(6);
    // the window's probabilities fp16(exp(x - M) / sum) (:375) of the rows wave, wave + NW, ...
    if (!PSM && last) {
        const int Lw = av.res_len + 1;
#pragma unroll 1
        for (int rr = wave; rr < R; rr += NW) {
            float Mr = M[0], Ir = invS[0];
#pragma unroll
            for (int q = 1; q < R; q++)
                if (rr == q) { Mr = M[q]; Ir = invS[q]; }
            const uint16_t* rp = rows + rr * n_pad + (Tv - tok0);
            for (int j = lane; j < NW * decltype(win)::TW; j += 64)     // (zeros past the window: its walk reads whole 8-token groups)
                pw[rr][j] = j < Lw ? f2h_bits(kivi_exp(h2f_bits(rp[j]) - Mr) * Ir) : (uint16_t)0;
        }
    }
    if (!PSM && threadIdx.x == 0) {
#pragma unroll
        for (int rr = 0; rr < R; rr++) sp_lds[rr] = sp[rr];
    }
    kivi_lds_barrier();                                            // pw complete; nobody reads scores past Tv any more
    if constexpr (PSM) {
#pragma unroll
        for (int rr = 0; rr < R; rr++) ksh = sp_lds[R + rr] > ksh ? sp_lds[R + rr] : ksh;
    }
    ksh = __builtin_amdgcn_readfirstlane(ksh);
    stamp(7);

    // ---- fp16 window of the R heads, V append, quantisation of the token leaving the window (:377-399)
    float ow[R][2];
    if (last) win.finish(av, b, hk, 0, av.res_len + 1, av.flush != 0, pw, ow);
    else {
#pragma unroll
        for (int rr = 0; rr < R; rr++) ow[rr][0] = ow[rr][1] = 0.f;
    }
    stamp(8);

    // ---- packed sV: this wave's blocks, PB at a time: their p'' in place, then the stream over them
    MfVAcc<R, VHL> A;
    mf_v_init(A);
    __builtin_amdgcn_s_setprio(0);
    // pieces of PB blocks (whole ring rounds, <= 512 tokens: 8 per lane).  The first piece's p'' are made before the stream starts;
    // from then on ONE row of the next piece is converted after every ring round of the current one, under the loads that round has
    // just requested: converting a whole piece (R rows) at once let the ring run dry five times per wave (+8 us on the sV phase at
    // BASELINE config 4, profiles/r05_row4_phases.log)
    constexpr int PB = (16 / VRING) * VRING, NR = PB / VRING;
    static_assert(R <= NR, "a row of the next piece per ring round");
    if constexpr (PSM) {
        vs.run(A, rv, b_lo, b_hi, rows, n_pad, 0, ksh);            // (the rows hold finished p'')
    } else {
        const int nb0 = (b_hi - b_lo) < PB ? (b_hi - b_lo) : PB;
        if (nb0 > 0) {
#pragma unroll
            for (int rr = 0; rr < R; rr++)
                mf_probs_inplace_row<BITS>(rows + rr * n_pad, b_lo * 32 - tok0, nb0 * 32, Tv - b_lo * 32, M[rr], invS[rr], sp[rr], vrsh);
        }
        __builtin_amdgcn_wave_barrier();
        for (int bp = b_lo; bp < b_hi; bp += PB) {
            const int pe = (bp + PB < b_hi) ? bp + PB : b_hi;      // end of this piece
            const int nbn = (b_hi - pe) < PB ? (b_hi - pe) : PB;   // blocks of the next one (<= 0: none)
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int b0 = bp + r * VRING;
                if (b0 < pe) vs.run(A, rv, b0, (b0 + VRING < pe) ? b0 + VRING : pe, rows, n_pad, tok0, ksh);
                if (r < R && nbn > 0)
                    mf_probs_inplace_row<BITS>(rows + r * n_pad, pe * 32 - tok0, nbn * 32, Tv - pe * 32, M[r], invS[r], sp[r], vrsh);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    stamp(9);
    __syncthreads();                                               // every wave is done with the p'' rows: their memory is reused
    constexpr int NP = VHL ? 2 * NW : NW;                          // partial results of the quantised part (VHL: hi and lo of every wave)
    float* red = reinterpret_cast<float*>(rows);                   // [NP][R * 128] quantised part | [NW][R * 128] window part | [2][R * 128]
    float* resl = red + NP * R * 128;
    float* lf = resl + NW * R * 128;                               // the block's sums (hand-off between slices)
    mf_v_finish<R, VRING, VHL, BITS>(A, zl[wave], red + wave * (NP / NW) * R * 128, (float)(1 << ksh));
#pragma unroll
    for (int rr = 0; rr < R; rr++) {
        resl[wave * R * 128 + rr * 128 + 2 * lane] = ow[rr][0];
        resl[wave * R * 128 + rr * 128 + 2 * lane + 1] = ow[rr][1];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R * 128; i += NTH) {
        const int rr = i >> 7, d = i & 127;
        float qs = 0.f, ws = 0.f;
#pragma unroll
        for (int w = 0; w < NP; w++) qs += red[w * R * 128 + i];
#pragma unroll
        for (int w = 0; w < NW; w++) ws += resl[w * R * 128 + i];
        qs = __builtin_ldexpf(qs, -sp_lds[rr]);
        if (S > 1) {
            lf[i] = qs;
            lf[R * 128 + i] = ws;
        } else {
            // fp16(quantised part) + fp16(window part), rounded: the reference's `attn_output += matmul(...)` (llama_kivi.py:382-384)
            const uint16_t o = (Tv > 0) ? f2h_bits(h2f_bits(f2h_bits(qs)) + h2f_bits(f2h_bits(ws))) : f2h_bits(ws);
            av.out[b * av.out_sb + (int64_t)(h0 + rr) * av.out_sh + d] = o;
        }
    }
    if (S > 1) {
        __syncthreads();
        gqa_arrive_and_combine<R, NTH>(av, unit, slice, lf, b, h0, ak.xcount + unit);
    }
    stamp(11);
    if (DBG && lane == 0) {
        unsigned long long* rec = av.dbg + ((size_t)bid * NW + wave) * 16;
        rec[10] = rec[9];                                          // (no separate stamp between the barrier and the final sum)
        rec[12] = __builtin_amdgcn_s_memrealtime();
        rec[13] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        rec[14] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host side (called from kivi_gqa.hip)

// qK^T launch of the decode step / kivi_gqa_scores for nh / nh_kv in {1, 4}
// (the argument blocks live in an anonymous namespace of a shared header: they cross the translation-unit boundary as void*)
int kivi_mf_run_k(void* k_args, int units, int bits, hipStream_t s) { return run_mf_k(*(GqaKArgs*)k_args, units, bits, s); }

// sV launch of the decode step (prob == 0) or of kivi_gqa_output (prob != 0: a.x rows hold fp16 probabilities, a.sp_rows
// their exponents)
int kivi_mf_run_v(const void* v_args, int prob, int bits, hipStream_t s) {
    const GqaVArgs& a = *(const GqaVArgs*)v_args;
    const int R = a.ratio;
#ifdef KIVI_TUNING
    mf_tune_sync();
#endif
    const dim3 grid((unsigned)(a.units * a.S + a.win_blocks));
    const size_t lds = (size_t)4 * (R * 256 + 128) * 4;
    KIVI_REQUIRE(bits == 2 || (bits == 4 && (R == 4 || R == 1)), KIVI_EUNSUPPORTED, "mf_v: %d-bit codes with nh / nh_kv = %d have no matrix-pipe kernel", bits, R);
    if (bits == 4) {                                                // 4-bit codes: nh / nh_kv in {1, 4}
        if (R == 1) {
            if (prob) KIVI_LAUNCH_LDS((mf_v_kernel<1, 2, true, false, 4>), grid, dim3(256), lds, s, a);
            else KIVI_LAUNCH_LDS((mf_v_kernel<1, 2, false, false, 4>), grid, dim3(256), lds, s, a);
        } else if (prob) KIVI_LAUNCH_LDS((mf_v_kernel<4, 4, true, false, 4>), grid, dim3(256), lds, s, a);
        else KIVI_LAUNCH_LDS((mf_v_kernel<4, 4, false, false, 4>), grid, dim3(256), lds, s, a);
        return kivi_launch_status("mf_v");
    }
#define KIVI_MV(RR, RG, PB) KIVI_LAUNCH_LDS((mf_v_kernel<RR, RG, PB>), grid, dim3(256), lds, s, a)
#ifdef KIVI_TUNING
    static const char* fr0 = KIVI_TUNE_ENV("KIVI_MF_RING");
    static const char* fr1 = KIVI_TUNE_ENV("KIVI_MF_VRING");             // the sV ring alone
    const char* fr = fr1 ? fr1 : fr0;
    if (fr && atoi(fr) == 4 && R == 1) {
        if (prob) KIVI_MV(1, 4, true); else KIVI_MV(1, 4, false);
        return kivi_launch_status("mf_v");
    }
    if (fr && atoi(fr) == 2 && R == 4) {
        if (prob) KIVI_MV(4, 2, true); else KIVI_MV(4, 2, false);
        return kivi_launch_status("mf_v");
    }
#endif
    // R = 4 runs two blocks per CU (gqa_v_slices): four code blocks in flight per wave, 57.4 us per launch at the 70B-like
    // slice against 60.5 with two (profiles/r03_gqa_split_restructure.log); R = 1 keeps four waves per SIMD with two
    // R = 8: two row sets of scale / zero points per block in flight (20 registers per ring slot): two blocks
#ifdef KIVI_TUNING
    static const char* hl = KIVI_TUNE_ENV("KIVI_MF_VHL");                // A/B: hi / lo rows in the two-launch sV (R = 4), ring 4 or 2
    if (hl && R == 4 && !prob) {
        if (atoi(hl) == 4) KIVI_LAUNCH_LDS((mf_v_kernel<4, 4, false, true>), grid, dim3(256), lds, s, a);
        else KIVI_LAUNCH_LDS((mf_v_kernel<4, 2, false, true>), grid, dim3(256), lds, s, a);
        return kivi_launch_status("mf_v");
    }
    if (fr && atoi(fr) == 2 && R == 8) {
        if (prob) KIVI_MV(8, 2, true); else KIVI_MV(8, 2, false);
        return kivi_launch_status("mf_v");
    }
#endif
    if (R == 1) { if (prob) KIVI_MV(1, 2, true); else KIVI_MV(1, 2, false); }
    else if (R == 4) { if (prob) KIVI_MV(4, 4, true); else KIVI_MV(4, 4, false); }
    else { if (prob) KIVI_MV(8, 4, true); else KIVI_MV(8, 4, false); }
#undef KIVI_MV
    return kivi_launch_status("mf_v");
}

int kivi_mf_run_row_sp(const void* p, int64_t p_sb, int64_t p_sh, int B, int nh, int nh_kv, int64_t T, const int* range, int* sp,
                       hipStream_t s) {
    hipLaunchKernelGGL(mf_row_sp_kernel, dim3((unsigned)(B * nh)), dim3(256), 0, s, (const uint16_t*)p, p_sb, p_sh, nh, nh_kv, T, range, sp);
    return kivi_launch_status("mf_row_sp");
}

// > 64 KiB of dynamic LDS needs an opt-in per kernel AND per device (the attribute belongs to the function object of the
// device that is current): tracked per device ordinal, return code checked.
template <typename K>
static int mf_lds_opt_in(K kernel, unsigned long long* done_mask, const char* what) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { kivi_set_error("%s: hipGetDevice: %s", what, hipGetErrorString(e)); return (int)e; }
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return 0;
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e != hipSuccess) {
        kivi_set_error("%s: opting in to 80 KiB of dynamic LDS failed on device %d: %s", what, dev, hipGetErrorString(e));
        return (int)e;
    }
    __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
    return 0;
}

// compute units of the current device (0 if the query fails: then every sliced launch takes its block ids from the ticket counter)
static int mf_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int& c = cus[dev & 63];
    if (c == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) c = v;
    }
    return c;
}

// The whole step in one launch: nh == nh_kv (rows <= 8192 keys: mf_row_kernel, 4 blocks of 4 waves per CU, 8 waves per row
// for <= 512 rows) or nh / nh_kv in {4, 8} (mf_row4_kernel: the R score rows of a unit -- or, S > 1, of one of the S slices of its
// row -- in one block of 4 waves, 2 blocks per CU).  KIVI_EUNSUPPORTED (with a message) when the shape does not qualify.
// dump != 0 (KIVI_GQA_DUMP_SCORES, tests): the rows the softmax (statistics) are taken from also go to the score buffer -- a run-time
// pointer in the PRODUCT instantiations (round 4 used separate ones).
// n_rows: the longest row the launch must hold (= Tq + k_res_len + 1, or the bound of the step's geometry class when the lengths
// are device-resident).  S: slices per row; res_cap: residual_length (the fp16 keys a last slice may hold); slice_kernel: nh == nh_kv
// with S = 1 through mf_row4_kernel<R = 1> instead of mf_row_kernel (KIVI_GQA_SLICES(1): A/B).
int kivi_mf_run_row(void* k_args, const void* v_args, int units, int64_t n_rows, int dump, int bits, int S, int res_cap, int slice_kernel, hipStream_t s) {
    GqaKArgs& k = *(GqaKArgs*)k_args;
    const GqaVArgs& v = *(const GqaVArgs*)v_args;
    const int64_t n = n_rows;
    KIVI_REQUIRE(bits == 2 || (bits == 4 && (k.ratio == 4 || k.ratio == 1)), KIVI_EUNSUPPORTED, "mf_row: %d-bit codes with nh / nh_kv = %d have no matrix-pipe kernel", bits, k.ratio);
    KIVI_REQUIRE(S == 1 || k.ratio == 1 || k.ratio == 4 || k.ratio == 8, KIVI_EUNSUPPORTED, "mf_row: no sliced form for nh / nh_kv = %d", k.ratio);
    if (k.ratio == 4 || k.ratio == 8 || (k.ratio == 1 && (S > 1 || slice_kernel))) {
        const int R = k.ratio;
        const int64_t cap = R == 4 ? 9216 + 128 : (R == 8 ? 4608 : 8192);    // keys whose R score rows fit the LDS of a block (mf_plan, kivi_gqa.hip: R = 4: 73 KiB, two blocks per CU)
        KIVI_REQUIRE(S >= 1 && S <= 64 && (S == 1 || S <= k.nsb), KIVI_EINVAL, "mf_row%d: %d slices for %d super-blocks", R, S, k.nsb);
        // the longest row of a block: the whole row, or (S > 1) max(ceil(nsb / S), 2) super-blocks + the residual (mf_row4_kernel)
        int64_t n_blk = n;
        if (S > 1) {
            const int spb = (k.nsb + S - 1) / S;
            n_blk = (int64_t)(spb > 2 ? spb : 2) * KIVI_MF_SB_TOKENS + res_cap + 1;
        }
        KIVI_REQUIRE(n_blk <= cap, KIVI_EUNSUPPORTED, "mf_row%d: rows of %lld keys do not fit the LDS (<= %lld)", R, (long long)n_blk, (long long)cap);
        const int n_pad = (int)((n_blk + 4 + 31) / 32 * 32);
        size_t lds = (size_t)R * n_pad * 2;
        // the per-wave partial sums + the block's sums reuse the rows: [NP][R * 128] quantised part (hi and lo of every wave for R = 4) |
        // [NW][R * 128] window part | [2][R * 128] hand-off.  (Sized for the FOUR waves of the product blocks: a blanket six-wave size cost
        // nh / nh_kv = 8 its second block per CU -- 80 KB instead of 56 -- and 50 % of its speed for a session.)
        auto fin_bytes = [&](int nw) { return (size_t)((R == 4 ? 2 * nw : nw) + nw + 2) * R * 128 * 4; };
        const size_t fin = fin_bytes(4);
        const int occ = R == 1 ? 4 : 2;                             // blocks per CU
        if (lds < fin) lds = fin;
        const dim3 grid((unsigned)((int64_t)units * S));
        // blocks that wait for each other (S > 1) must not wait for blocks that cannot start: their ids ALWAYS come from the ticket
        // counter (start order).  (Round 5 skipped it when the grid fitted 2 blocks per CU -- but an ordinary launch does not own the
        // chip: other streams or processes, a CU mask, a second sliced launch can keep a waiting block's partners from starting.)
        k.dump = dump ? 1 : 0;
        (void)occ;
#ifdef KIVI_TUNING
        static const char* ft = KIVI_TUNE_ENV("KIVI_MF_FAULT_DROP_ARRIVAL");       // fault injection: see mf_row4_kernel
        if (ft && atoi(ft) && S > 1) k.dump |= 2;
        static const char* f1t = KIVI_TUNE_ENV("KIVI_MF_ONE_TICKET");              // A/B: one ticket counter instead of eight
        if (f1t && atoi(f1t) && S > 1) k.dump |= 4;
        static const char* fnt = KIVI_TUNE_ENV("KIVI_MF_NO_TICKET");               // A/B: round 5's rule -- static block ids when the grid fits 2 (4) blocks per CU
        if (fnt && atoi(fnt) && S > 1 && (int64_t)units * S <= occ * (int64_t)mf_cu_count()) k.ticket = nullptr;
#endif
        if (S == 1) { k.ticket = nullptr; k.xcount = nullptr; k.err_ws = nullptr; k.err_host = nullptr; }
        static unsigned long long opt8 = 0, opt4 = 0, opt44 = 0, opt1 = 0, opt14 = 0, opt8p = 0, opt4p = 0, opt44p = 0;
        // a block per row (S = 1): the phase-softmax flow (PSM, see mf_row4_kernel); slices: the in-stream flow
        bool psm = S == 1 && R != 1;
#ifdef KIVI_TUNING
        static const char* fl = KIVI_TUNE_ENV("KIVI_MF_ROW4_FLOW");      // A/B: "stream" = the in-stream flow for unsliced rows too
        if (fl && !strcmp(fl, "stream")) psm = false;
#endif
#define KIVI_ROW4_LAUNCH_T(OPT, THREADS, ...)                                                      \
    do {                                                                                           \
        const int rc = mf_lds_opt_in(mf_row4_kernel<__VA_ARGS__>, &OPT, "mf_row4");                \
        if (rc) return rc;                                                                         \
        KIVI_LAUNCH_LDS((mf_row4_kernel<__VA_ARGS__>), grid, dim3(THREADS), lds, s, k, v, n_pad, S); \
        return kivi_launch_status("mf_row4");                                                      \
    } while (0)
#define KIVI_ROW4_LAUNCH(OPT, ...) KIVI_ROW4_LAUNCH_T(OPT, 256, __VA_ARGS__)
        if (R == 1 && bits == 4) KIVI_ROW4_LAUNCH(opt14, 2, 3, 4, false, false, 1, 4, 4);
        if (R == 1) KIVI_ROW4_LAUNCH(opt1, 2, 3, 4, false, false, 1, 2, 4);
        if (R == 8 && psm) KIVI_ROW4_LAUNCH(opt8p, 4, 2, 4, false, false, 8, 2, 2, true);
        if (R == 8) KIVI_ROW4_LAUNCH(opt8, 4, 2, 4, false, false, 8);
#ifdef KIVI_TUNING
        static unsigned long long opt446[3] = {0};
        static const char* f46 = KIVI_TUNE_ENV("KIVI_MF_ROW4_46");       // 4-bit codes, six waves per block: "<K ring><V ring>" (A/B; not the product)
        if (bits == 4 && psm && f46 && lds < fin_bytes(6)) lds = fin_bytes(6);
        if (bits == 4 && psm && f46 && atoi(f46) == 22) KIVI_ROW4_LAUNCH_T(opt446[0], 384, 2, 2, 6, false, true, 4, 4, 3, true);
        if (bits == 4 && psm && f46 && atoi(f46) == 23) KIVI_ROW4_LAUNCH_T(opt446[1], 384, 2, 3, 6, false, true, 4, 4, 3, true);
        if (bits == 4 && psm && f46 && atoi(f46) == 43) KIVI_ROW4_LAUNCH_T(opt446[2], 384, 4, 3, 6, false, true, 4, 4, 3, true);
#endif
        if (bits == 4 && psm) KIVI_ROW4_LAUNCH(opt44p, 4, 3, 4, false, true, 4, 4, 2, true);
        if (bits == 4) KIVI_ROW4_LAUNCH(opt44, 4, 3, 4, false, true, 4, 4);
#ifdef KIVI_TUNING
        static unsigned long long opt_t[16] = {0};
        static const char* fr4 = KIVI_TUNE_ENV("KIVI_MF_ROW4");          // "<K ring><V ring><waves>"; + 1000: chained hi / lo in the sV phase
        const int cfg = fr4 ? atoi(fr4) : 434;
        if (v.dbg && psm) KIVI_ROW4_LAUNCH(opt_t[11], 4, 3, 4, true, true, 4, 2, 2, true);
        if (v.dbg && cfg == 844) KIVI_ROW4_LAUNCH(opt_t[9], 8, 4, 4, true);
        if (v.dbg) KIVI_ROW4_LAUNCH(opt_t[0], 4, 3, 4, true);
        if (cfg == 234) KIVI_ROW4_LAUNCH(opt_t[1], 2, 3, 4);
        if (cfg == 834) KIVI_ROW4_LAUNCH(opt_t[2], 8, 3, 4);
        if (cfg == 444) KIVI_ROW4_LAUNCH(opt_t[3], 4, 4, 4);
        if (cfg == 844) KIVI_ROW4_LAUNCH(opt_t[4], 8, 4, 4);
        if (cfg == 424) KIVI_ROW4_LAUNCH(opt_t[5], 4, 2, 4);
        if (cfg == 1434) KIVI_ROW4_LAUNCH(opt_t[6], 4, 3, 4, false, false);
#endif
        // a block per row, 2 bits, nh / nh_kv = 4.  Round 6 built the "third wave per SIMD" the last review asked for and measured it (profiles/
        // r06_six_wave.log): SIX waves per block (two blocks per CU by LDS -> three waves per SIMD in 168 registers, 36-56 bytes spilled) are
        // 24 % SLOWER at BASELINE config 4 (122.3 against 98.6 us on one box, every ring combination 119.7-127.7), 39-43 % slower on 4k / 2k
        // rows, equal with one block per CU (66.3 / 64.8) -- six waves on four SIMDs sit 2, 2, 1, 1: a SIMD carries a THIRD of the block's
        // work instead of a quarter, which eats what the extra wave hides.  (A BALANCED third wave -- three four-wave blocks per CU, below --
        // is worth 4-9 % where the LDS allows it: config 4's 8k rows do not.)  The product keeps four waves per block; the six-wave
        // instantiations live in the tuning build: KIVI_MF_ROW4_NW=6 [KIVI_MF_ROW4_6=<K ring><V ring>].
#ifdef KIVI_TUNING
        static unsigned long long opt4p6 = 0;
        static const char* fnw = KIVI_TUNE_ENV("KIVI_MF_ROW4_NW");
        const bool six = fnw && atoi(fnw) == 6;
        static unsigned long long opt6[4] = {0};
        static const char* fr6 = KIVI_TUNE_ENV("KIVI_MF_ROW4_6");        // "<K ring><V ring>" of the six-wave block
        const int c6 = fr6 ? atoi(fr6) : 43;
        if (psm && six && lds < fin_bytes(6)) lds = fin_bytes(6);
        if (psm && six && c6 == 23) KIVI_ROW4_LAUNCH_T(opt6[0], 384, 2, 3, 6, false, true, 4, 2, 3, true);
        if (psm && six && c6 == 42) KIVI_ROW4_LAUNCH_T(opt6[1], 384, 4, 2, 6, false, true, 4, 2, 3, true);
        if (psm && six && c6 == 22) KIVI_ROW4_LAUNCH_T(opt6[2], 384, 2, 2, 6, false, true, 4, 2, 3, true);
        if (psm && six) KIVI_ROW4_LAUNCH_T(opt4p6, 384, 4, 3, 6, false, true, 4, 2, 3, true);
#endif
        // THREE blocks per CU (round 6): where the four score rows of the geometry class's longest row leave room for a third block in
        // the LDS (<= ~6.3k keys) AND the launch has at least three blocks per CU to place, the instantiation compiled for three waves
        // per SIMD (167 registers, rings 2 / 2, no spill) runs 4-9 % faster (B=96 x 6k keys: 109.8 against 119.9 us; B=128 x 2k: 68.2
        // against 71.1; with only two blocks per CU it is 1-2 % slower: profiles/r06_three_blocks.log).  The criterion uses the CLASS
        // bound (nsb * 512 + residual_length keys), not the step's own row, so that eager and replayed steps pick the same instantiation
        // (their V rings centre differently: not bit-identical to each other).
        static unsigned long long opt4o3 = 0;
        {
            const int64_t n_class = (int64_t)k.nsb * KIVI_MF_SB_TOKENS + res_cap;
            const size_t lds_class = (size_t)R * (size_t)((n_class + 4 + 31) / 32 * 32) * 2;
            const int cus = mf_cu_count();
            bool three = R == 4 && bits == 2 && psm && cus > 0 && lds_class + 3696 + 240 <= (160 * 1024) / 3 && units >= 3 * cus;
#ifdef KIVI_TUNING
            static const char* fo3 = KIVI_TUNE_ENV("KIVI_MF_ROW4_OCC3");     // 0 / 1: never / whenever the rows fit (A/B)
            if (fo3) three = R == 4 && bits == 2 && psm && atoi(fo3) != 0;
#endif
            if (three) KIVI_ROW4_LAUNCH(opt4o3, 2, 2, 4, false, true, 4, 2, 3, true);
        }
        if (psm) KIVI_ROW4_LAUNCH(opt4p, 4, 3, 4, false, true, 4, 2, 2, true);
#ifdef KIVI_TUNING
        static unsigned long long opt4s3 = 0;
        static const char* fs3 = KIVI_TUNE_ENV("KIVI_MF_ROW4_SOCC3");    // A/B: slices (in-stream flow) compiled for three waves per SIMD (use with more, shorter slices)
        if (fs3 && atoi(fs3)) KIVI_ROW4_LAUNCH(opt4s3, 2, 2, 4, false, true, 4, 2, 3, false);
#endif
        KIVI_ROW4_LAUNCH(opt4, 4, 3, 4);
#undef KIVI_ROW4_LAUNCH
#undef KIVI_ROW4_LAUNCH_T
    }
    const int n_pad = (int)((n + 4 + 31) / 32 * 32);            // >= 4 halves of -inf behind every row (mf_row_softmax)
    const dim3 grid((unsigned)units);
    KIVI_REQUIRE(k.ratio == 1 && n <= 8192 + 128, KIVI_EUNSUPPORTED, "mf_row: nh / nh_kv = %d with rows of %lld keys has no one-launch kernel",
                 k.ratio, (long long)n);
    const size_t lds = (size_t)n_pad * 2;
    // (K ring, V ring) = (2, 3) code blocks in flight: 76.2 us per launch at the bench shape against 77.2 (2, 2), 76.7 (2, 4),
    // 78.4 (4, 2), 78.2 (4, 3) -- profiles/r03_row_rings.log
#ifdef KIVI_TUNING
    static const char* fr = KIVI_TUNE_ENV("KIVI_MF_ROW_RINGS");          // "<K ring><V ring>", e.g. 42
    const int rings = fr ? atoi(fr) : 23;
    static const char* np = KIVI_TUNE_ENV("KIVI_MF_ROW_NOPRIO");         // A/B of the raised priority
    if (v.dbg && np) { KIVI_LAUNCH_LDS((mf_row_kernel<2, 3, 4, true, false>), grid, dim3(256), lds, s, k, v, n_pad); return kivi_launch_status("mf_row"); }
    if (v.dbg) { KIVI_LAUNCH_LDS((mf_row_kernel<2, 3, 4, true>), grid, dim3(256), lds, s, k, v, n_pad); return kivi_launch_status("mf_row"); }
    if (np) { KIVI_LAUNCH_LDS((mf_row_kernel<2, 3, 4, false, false>), grid, dim3(256), lds, s, k, v, n_pad); return kivi_launch_status("mf_row"); }
    if (rings == 24) { KIVI_LAUNCH_LDS((mf_row_kernel<2, 4, 4>), grid, dim3(256), lds, s, k, v, n_pad); return kivi_launch_status("mf_row"); }
    if (rings == 22) { KIVI_LAUNCH_LDS((mf_row_kernel<2, 2, 4>), grid, dim3(256), lds, s, k, v, n_pad); return kivi_launch_status("mf_row"); }
    if (rings == 43) { KIVI_LAUNCH_LDS((mf_row_kernel<4, 3, 4>), grid, dim3(256), lds, s, k, v, n_pad); return kivi_launch_status("mf_row"); }
    if (rings == 42) { KIVI_LAUNCH_LDS((mf_row_kernel<4, 2, 4>), grid, dim3(256), lds, s, k, v, n_pad); return kivi_launch_status("mf_row"); }
#endif
    // few rows (under ~2 four-wave blocks per CU): eight waves per row, the row's own waves hide the latency
    static const char* f8 = KIVI_TUNE_ENV("KIVI_MF_ROW_NW8");            // tuning builds: 0 / 1 forces either
    const bool nw8 = f8 ? atoi(f8) != 0 : units <= 512;         // 256 rows: 30.3 -> 26.8 us, 384: 39.7 -> 37.0, 512: 45.8 -> 44.0, 768: 61.1 vs 65.4 (profiles/r03_other_shapes.log)
    k.dump = dump ? 1 : 0;
    // at most one block per CU (<= 256 rows): a row's waves are alone on their SIMDs and each is bound by the round trips of its own
    // ring (2 KiB of K / 3 KiB of V in flight stream ~3 GB/s per wave): rings of 8 blocks, 256 registers per wave
    static const char* fdp = KIVI_TUNE_ENV("KIVI_MF_ROW_DEEP");          // tuning builds: 0 / 1 forces either
    const bool deep = fdp ? atoi(fdp) != 0 : units <= 256;
    if (bits == 4) {                                               // 4-bit multi-head rows (round 6): a ring slot is 2 x 16 bytes -- rings of (2, 3) fit four waves per SIMD all the same
        if (deep) KIVI_LAUNCH_LDS((mf_row_kernel<4, 4, 8, false, true, 2, 4>), grid, dim3(512), lds, s, k, v, n_pad);
        else if (nw8) KIVI_LAUNCH_LDS((mf_row_kernel<2, 3, 8, false, true, 4, 4>), grid, dim3(512), lds, s, k, v, n_pad);
        else KIVI_LAUNCH_LDS((mf_row_kernel<2, 3, 4, false, true, 4, 4>), grid, dim3(256), lds, s, k, v, n_pad);
        return kivi_launch_status("mf_row");
    }
    if (deep) KIVI_LAUNCH_LDS((mf_row_kernel<8, 8, 8, false, true, 2>), grid, dim3(512), lds, s, k, v, n_pad);
    else if (nw8) KIVI_LAUNCH_LDS((mf_row_kernel<2, 3, 8>), grid, dim3(512), lds, s, k, v, n_pad);
    else KIVI_LAUNCH_LDS((mf_row_kernel<2, 3, 4>), grid, dim3(256), lds, s, k, v, n_pad);
    return kivi_launch_status("mf_row");
}

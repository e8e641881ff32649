// Grouped-query decode on the matrix pipe (gfx950): the packed qK^T and sV products of models/llama_kivi.py:324 / :382
// (models/mistral_kivi.py:381-385 / :441-445; kernel quant/csrc/gemv_cuda.cu:348-427 with its head mapping :361-365)
// for nh / nh_kv = R in {4, 8} query heads per kv head, over the MFMA-friendly cache layouts of kivi_mfma_layout.h.
//
//   qK^T:  S[r, t] = sum_d  q[r, d] * (scale[d, G(t)] * code[d, t] + mn[d, G(t)])
//     one wave = one super-block (512 tokens) of one (batch row, kv head); per 32-token group:
//       A (16 x 32 per 32-channel chunk) = q * scale * 2^(Sq + 6 - 2 i) as fp16, rows 0..R-1 the rounded product ("hi"),
//         rows R..2R-1 the exact remainder ("lo": v_pk_fma_f16(q, s, -hi)), so hi + lo is the exact 22-bit product
//       B (32 x 16)  = one masked code word per register: fp16 subnormals code * 4^i * 2^-24, no conversion at all
//       8 x v_mfma_f32_16x16x32_f16 (4 channel chunks x 2 token tiles), fp32 accumulate
//     the zero-point term sum_d q * mn is one more MFMA set per super-block (columns = its 16 groups).
//   sV:    O[r, d] = sum_t  p[r, t] * (scale[t, G(d)] * code[t, d] + mn[t, G(d)])
//     same structure with the roles of tokens and channels exchanged; accumulators live across the whole token range.
// VALU work per code is 1/2 mask + 1/8 (A build) independent of R; the old shared-unpack kernels spent 1/2 + R FMAs.
#include <stdlib.h>
#include <string.h>

#include "kivi_common.h"
#include "kivi_gqa_dev.h"
#include "kivi_quant.h"
#include "kivi_gqa_roles.h"

#include <type_traits>

namespace {


// ------------------------------------------------------------------------------------------------ pack / relayout

// Per-channel K quantise + pack of whole 32-token blocks straight into the KT layout (prompt pass, models/
// llama_kivi.py:436, and the flush of the fp16 residual every R tokens, :343-356).  Reference arithmetic through the
// shared quantiser (kivi_quant.h = new_pack.py:236-241 op for op); group_size == 32 == the block.
// One wave per block; lane l owns channels 2l, 2l+1 over the 32 tokens (min / max without any cross-lane step).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_or(uint32_t v) {     // the value of the quad neighbour (quad_perm), for an OR across a quad
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__global__ __launch_bounds__(256) void kt_pack_kernel(const uint16_t* k, int64_t k_sb, int64_t k_sh, int64_t k_st, MfStore st,
                                                      int* range, int64_t blk0, int nblk, int nh_kv, int64_t ntile) {
    // four waves per block, one 32-token block each (single-wave workgroups: 131 072 of them per GiB)
    const int wave = threadIdx.x >> 6;
    int64_t tile_id = (int64_t)blockIdx.x * 4 + wave;
    const bool live_tile = tile_id < ntile;
    if (!live_tile) tile_id = ntile - 1;                   // (keeps the barrier below uniform; its stores are skipped)
    const int unit = (int)(tile_id / nblk), bi = (int)(tile_id - (int64_t)unit * nblk);
    const int b = unit / nh_kv, hk = unit - b * nh_kv;
    const int lane = threadIdx.x & 63;
    // the 32 x 128 tile through LDS: eight 16-byte loads per lane (four whole rows per instruction) instead of 32 four-byte ones,
    // then the lane reads its channel pair down the 32 tokens (consecutive lanes, consecutive banks)
    constexpr int PITCH = 68;
    __shared__ uint32_t stage[4][32 * PITCH];
    uint32_t* stw = stage[wave];
    {
        const uint16_t* tb = k + b * k_sb + hk * k_sh + (int64_t)bi * 32 * k_st;
        u32x4 in[8];
#pragma unroll
        for (int j = 0; j < 8; j++) in[j] = __builtin_nontemporal_load((const u32x4*)(tb + (int64_t)(4 * j + (lane >> 4)) * k_st + 8 * (lane & 15)));
#pragma unroll
        for (int j = 0; j < 8; j++) *(u32x4*)(stw + (4 * j + (lane >> 4)) * PITCH + 4 * (lane & 15)) = in[j];
    }
    __syncthreads();
    uint32_t x[32];
#pragma unroll
    for (int t = 0; t < 32; t++) x[t] = stw[t * PITCH + lane];
    // both channels of the lane at once on packed 16-bit math (kivi_quant.h): cq[t] = code(2l) | code(2l + 1) << 16
    uint32_t cq[32], scale2, mn2;
    pk16_pair_quant2<32>(x, cq, scale2, mn2);
    const int i = lane & 3;
    const int p0 = mf_pos(0, i), p1 = mf_pos(1, i);
    uint32_t pw[16];
#pragma unroll
    for (int n = 0; n < 16; n++) {
        uint32_t w = (cq[n] << p0) | (cq[n + 16] << p1);     // tokens n (tile 0) and n + 16 (tile 1); even channel low half, odd high
        w |= dpp_or<0xB1>(w);                                // the 4 lanes of a quad hold the 4 channel pairs of one word
        w |= dpp_or<0x4E>(w);
        pw[n] = w;
    }
    const int c = lane >> 4, kb = (lane >> 2) & 3;
    const int64_t blk = blk0 + bi;
    uint32_t* sb = mf_sb(st, b, hk, blk >> 4);
    uint32_t* cw = sb + (blk & 15) * KIVI_MF_BLOCK_WORDS;
    // the 256 code words of the block meet in LDS and leave as ONE 1 KiB store (16 bytes per lane): written straight from the
    // quads they were 16 four-byte stores per lane group, 64 partial-line transactions per block
    __shared__ uint32_t tiles[4][256];
    uint32_t* tile = tiles[wave];
#pragma unroll
    for (int n = 0; n < 16; n++)
        if ((n >> 2) == i) tile[(n + 16 * kb) * 4 + c] = pw[n];
    __syncthreads();
    if (!live_tile) return;
    *(u32x4*)(cw + lane * 4) = *(const u32x4*)(tile + lane * 4);
    const int hidx = kt_sm_half((int)(blk & 15), 2 * lane);    // channel 2l (even): the pair (2l, 2l+1) is one word
    (sb + KIVI_MF_SB_SCALE_WORD0)[hidx >> 1] = scale2;
    (sb + KIVI_MF_SB_MN_WORD0)[hidx >> 1] = mn2;
    // range flag of the unit (kivi_mfma_layout.h): sticky, every writer stores the same value
    if ((scale2 & 0xFFFFu) >= KIVI_MF_BIG_SCALE_BITS || (scale2 >> 16) >= KIVI_MF_BIG_SCALE_BITS) range[unit] = 1;
}

// Per-token V quantise + pack of a prompt straight into the VT layout (prompt pass, models/llama_kivi.py:441-448: the
// reference quantises value_states[:, :, :-R] along the channel axis, new_pack.py:217-252); replaces the last-dim pack
// into hook-state tensors followed by kivi_vt_relayout (two passes, the second one all bit moves).
// One wave per 32-token block.  Lane (kb, c, ee) = 16 kb + 4 c + ee owns the token PAIR (8 kb + 2 ee, + 1) of channel
// group c: even token in the low halves, odd token in the high halves of x[j], j = channel inside the group -- exactly
// the VT word's halves -- so the pair quantiser (kivi_quant.h) gives both groups at once, the 4 lanes of a quad (ee)
// complete a word with two DPP ORs, and the lane index is the index of the pair's scale / zero-point word.  Tokens at
// or past T read as zeros (constant group: scale 0, zero point 0, codes 0 = never-written storage).
__global__ __launch_bounds__(256) void vt_pack_kernel(const uint16_t* v, int64_t v_sb, int64_t v_sh, int64_t v_st, MfStore st,
                                                      int* range, int64_t T, int nblk, int nh_kv, int64_t ntile) {
    const int wave = threadIdx.x >> 6;                     // four waves per block, one 32-token block each (cf. kt_pack_kernel)
    int64_t tile_id = (int64_t)blockIdx.x * 4 + wave;
    const bool live_tile = tile_id < ntile;
    if (!live_tile) tile_id = ntile - 1;
    const int unit = (int)(tile_id / nblk), bi = (int)(tile_id - (int64_t)unit * nblk);
    const int b = unit / nh_kv, hk = unit - b * nh_kv;
    const int lane = threadIdx.x & 63;
    const int kb = lane >> 4, c = (lane >> 2) & 3, ee = lane & 3;
    // the 32 x 128 tile comes in through LDS: eight loads of 16 bytes per lane that each cover four whole rows (lane l of load j:
    // row 4 j + l / 16, bytes 16 (l % 16) ...), then every lane picks the 2 x 64 bytes of its token pair and channel group.
    // Read straight from memory the lanes of one instruction took 16-byte pieces out of 32 different cache lines.
    constexpr int PITCH = 68;                              // words per staged row (256 bytes + 16: rows start 4 banks apart)
    __shared__ uint32_t stage[4][32 * PITCH];
    uint32_t* stw = stage[wave];
    {
        const uint16_t* tb = v + b * v_sb + hk * v_sh + (int64_t)bi * 32 * v_st;
        u32x4 in[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = 4 * j + (lane >> 4);
            in[j] = ((int64_t)bi * 32 + r < T) ? __builtin_nontemporal_load((const u32x4*)(tb + (int64_t)r * v_st + 8 * (lane & 15))) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < 8; j++) *(u32x4*)(stw + (4 * j + (lane >> 4)) * PITCH + 4 * (lane & 15)) = in[j];
    }
    __syncthreads();
    u32x4 ra[4], rb[4];                                   // 32 channels of the even / odd token
    {
        const uint32_t* pa = stw + (8 * kb + 2 * ee) * PITCH + 16 * c;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            ra[i] = *(const u32x4*)(pa + 4 * i);
            rb[i] = *(const u32x4*)(pa + PITCH + 4 * i);
        }
    }
    uint32_t x[32];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t a = ra[i][k], bq = rb[i][k];   // channels 8 i + 2 k, + 1
            x[8 * i + 2 * k] = __builtin_amdgcn_perm(bq, a, 0x05040100u);       // (a.lo, b.lo)
            x[8 * i + 2 * k + 1] = __builtin_amdgcn_perm(bq, a, 0x07060302u);   // (a.hi, b.hi)
        }
    uint32_t cq[32], scale2, mn2;
    pk16_pair_quant2<32>(x, cq, scale2, mn2);
    const int p0 = mf_pos(0, ee), p1 = mf_pos(1, ee);
    uint32_t* sb = mf_sb(st, b, hk, bi >> 4);
    uint32_t* cw = sb + (bi & 15) * KIVI_MF_BLOCK_WORDS;
    __shared__ uint32_t tiles[4][256];                     // as in kt_pack_kernel: one 1 KiB store for the block's code words
    uint32_t* tile = tiles[wave];
#pragma unroll
    for (int n = 0; n < 16; n++) {
        uint32_t w = (cq[n] << p0) | (cq[n + 16] << p1);   // channels n (tile 0) and 16 + n (tile 1) of the group
        w |= dpp_or<0xB1>(w);
        w |= dpp_or<0x4E>(w);
        if ((n >> 2) == ee) tile[(n + 16 * kb) * 4 + c] = w;
    }
    __syncthreads();
    if (!live_tile) return;
    *(u32x4*)(cw + lane * 4) = *(const u32x4*)(tile + lane * 4);
    (sb + KIVI_MF_SB_SCALE_WORD0 + (bi & 15) * 64)[lane] = scale2;    // vt_half(8 kb + 2 ee, c) / 2 == lane
    (sb + KIVI_MF_SB_MN_WORD0 + (bi & 15) * 64)[lane] = mn2;
    if ((scale2 & 0xFFFFu) >= KIVI_MF_BIG_SCALE_BITS || (scale2 >> 16) >= KIVI_MF_BIG_SCALE_BITS) range[unit] = 1;   // range flag (kt_pack_kernel)
}

// KT <-> reference layout K_code_T (B, nh_kv, D, T/16), K_scale_T / K_mn_T (B, nh_kv, D, T/32) (llama_kivi.py:454-455).
// Pure bit moves; one 128-thread block per 32-token block (thread = channel).
template <bool TO_REF>
__global__ __launch_bounds__(128) void kt_relayout_kernel(MfStore st, uint32_t* code, int64_t code_sb, int64_t code_sh,
                                                          int64_t code_sr, uint16_t* scale, uint16_t* mn, int64_t sm_sb,
                                                          int64_t sm_sh, int64_t sm_sr, int nblk, int nh_kv, int* range) {
    __shared__ uint32_t lds[256];
    const int unit = blockIdx.x / nblk, blk = blockIdx.x - unit * nblk;
    const int b = unit / nh_kv, hk = unit - b * nh_kv;
    const int d = threadIdx.x;
    uint32_t* sb = mf_sb(st, b, hk, blk >> 4);
    uint32_t* cw = sb + (blk & 15) * KIVI_MF_BLOCK_WORDS;
    uint16_t* ks = (uint16_t*)(sb + KIVI_MF_SB_SCALE_WORD0);
    uint16_t* km = (uint16_t*)(sb + KIVI_MF_SB_MN_WORD0);
    const int gsb = blk & 15;
    uint32_t* cref = code + b * code_sb + hk * code_sh + (int64_t)d * code_sr + (int64_t)blk * 2;
    const int64_t sidx = b * sm_sb + hk * sm_sh + (int64_t)d * sm_sr + blk;
    if constexpr (TO_REF) {
        lds[d] = cw[d];
        lds[d + 128] = cw[d + 128];
        __syncthreads();
#pragma unroll
        for (int tile = 0; tile < 2; tile++) {
            uint32_t w = 0;
#pragma unroll
            for (int n = 0; n < 16; n++) w |= ((lds[kt_word(n + 16 * tile, d)] >> kt_bit(n + 16 * tile, d)) & 3u) << (2 * n);
            cref[tile] = w;
        }
        scale[sidx] = ks[kt_sm_half(gsb, d)];
        mn[sidx] = km[kt_sm_half(gsb, d)];
    } else {
        lds[2 * d] = cref[0];
        lds[2 * d + 1] = cref[1];
        __syncthreads();
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            const int wi = d + 128 * rep, l = wi >> 2, c = wi & 3, n = l & 15, kb = l >> 4;
            uint32_t w = 0;
#pragma unroll
            for (int e = 0; e < 8; e++)
#pragma unroll
                for (int tile = 0; tile < 2; tile++) {
                    const int ch = 32 * c + 8 * kb + e;
                    w |= ((lds[2 * ch + tile] >> (2 * n)) & 3u) << kt_bit(n + 16 * tile, ch);
                }
            cw[wi] = w;
        }
        const uint16_t sc = scale[sidx];
        ks[kt_sm_half(gsb, d)] = sc;
        km[kt_sm_half(gsb, d)] = mn[sidx];
        if (sc >= KIVI_MF_BIG_SCALE_BITS) range[unit] = 1;      // range flag of the unit (kivi_mfma_layout.h)
    }
}

// VT <-> reference layout V_code (B, nh_kv, T, D/16), V_scale / V_mn (B, nh_kv, T, D/32); tokens [0, T), the slots of
// the last block past T are written as zeros (the kernels rely on never-written slots being zero).
template <bool TO_REF>
__global__ __launch_bounds__(256) void vt_relayout_kernel(MfStore st, uint32_t* code, int64_t code_sb, int64_t code_sh,
                                                          int64_t code_sr, uint16_t* scale, uint16_t* mn, int64_t sm_sb,
                                                          int64_t sm_sh, int64_t sm_sr, int64_t T, int nblk, int nh_kv, int* range) {
    __shared__ uint32_t lds[256];
    const int unit = blockIdx.x / nblk, blk = blockIdx.x - unit * nblk;
    const int b = unit / nh_kv, hk = unit - b * nh_kv;
    const int tid = threadIdx.x;
    uint32_t* sb = mf_sb(st, b, hk, blk >> 4);
    uint32_t* cw = sb + (blk & 15) * KIVI_MF_BLOCK_WORDS;
    uint16_t* vs = (uint16_t*)(sb + KIVI_MF_SB_SCALE_WORD0) + (blk & 15) * 128;
    uint16_t* vm = (uint16_t*)(sb + KIVI_MF_SB_MN_WORD0) + (blk & 15) * 128;
    const int tt = tid >> 3, wi = tid & 7;
    const int64_t t = (int64_t)blk * 32 + tt;
    uint32_t* cref = code + b * code_sb + hk * code_sh + t * code_sr + wi;
    if constexpr (TO_REF) {
        lds[tid] = cw[tid];
        __syncthreads();
        if (t < T) {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) w |= ((lds[vt_word(tt, 16 * wi + j)] >> vt_bit(tt, 16 * wi + j)) & 3u) << (2 * j);
            *cref = w;
            if (wi < 4) {
                scale[b * sm_sb + hk * sm_sh + t * sm_sr + wi] = vs[vt_half(tt, wi)];
                mn[b * sm_sb + hk * sm_sh + t * sm_sr + wi] = vm[vt_half(tt, wi)];
            }
        }
    } else {
        lds[tid] = (t < T) ? *cref : 0u;
        __syncthreads();
        const int l = tid >> 2, c = tid & 3, n = l & 15, kb = l >> 4;
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 8; e++)
#pragma unroll
            for (int tile = 0; tile < 2; tile++) {
                const int t2 = 8 * kb + e, d = 32 * c + 16 * tile + n;
                w |= ((lds[t2 * 8 + (d >> 4)] >> (2 * (d & 15))) & 3u) << vt_bit(t2, d);
            }
        cw[tid] = w;
        if (wi < 4) {
            const uint16_t sc = (t < T) ? scale[b * sm_sb + hk * sm_sh + t * sm_sr + wi] : (uint16_t)0;
            vs[vt_half(tt, wi)] = sc;
            if (sc >= KIVI_MF_BIG_SCALE_BITS) range[unit] = 1;  // range flag of the unit (kivi_mfma_layout.h)
            vm[vt_half(tt, wi)] = (t < T) ? mn[b * sm_sb + hk * sm_sh + t * sm_sr + wi] : (uint16_t)0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ qK^T


// W waves per thread block, one super-block each; nothing is shared between the waves of a block.
// RING = code blocks requested ahead of the one being multiplied (1 KiB per wave each).
template <int R, int W, bool HILO, int RING>
__global__ __launch_bounds__(64 * W) void gqa_k_kernel(const GqaKArgs a) {
    extern __shared__ uint32_t lds_all[];
    // the short residual blocks come LAST or FIRST in the grid (res_first): last, they fill the slots the streaming blocks
    // free up instead of delaying their start
    const int main_blocks = (int)gridDim.x - a.res_blocks;
    if (a.res_first ? (int)blockIdx.x < a.res_blocks : (int)blockIdx.x >= main_blocks) {
        gqa_k_residual<R>(a, a.res_first ? (int)blockIdx.x : (int)blockIdx.x - main_blocks);
        return;
    }
    const int bid = a.res_first ? (int)blockIdx.x - a.res_blocks : (int)blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* lds_s = lds_all + wave * (1024 + R * 256);          // scale of the super-block: 16 groups x 64 words
    uint16_t* lds_o = (uint16_t*)(lds_s + 1024);                  // R x 512 fp16 scores
    const int unit = bid / a.sb_blocks;
    const int sb = (bid - unit * a.sb_blocks) * W + wave;
    if (sb >= a.nsb) return;
    const int b = unit / a.nh_kv, hk = unit - b * a.nh_kv;
    const int h0 = hk * a.ratio;
    const int n = lane & 15, kb = lane >> 4;
    const int r = n % R;                                          // this lane's A row: head r, hi (n / R even) or lo
    const bool lo_row = ((n / R) & 1) != 0;
    int ng = (int)((a.Tq - (int64_t)sb * KIVI_MF_SB_TOKENS) / 32);
    ng = ng > 16 ? 16 : ng;

    const uint32_t* sbp = mf_sb(a.kt, b, hk, sb);
    const rsrc_t rk = make_rsrc(sbp, KIVI_MF_SB_WORDS * 4);

    // requests first: scale of the whole super-block (-> LDS), zero points (B operand of the zero-point MFMAs: lane
    // (n, kb) takes group n), the first RING code blocks
    u32x4 sreg[4], zreg[4];
#pragma unroll
    for (int j = 0; j < 4; j++) sreg[j] = buf_load<u32x4, true>(rk, (uint32_t)(KIVI_MF_SB_SCALE_WORD0 * 4 + (j * 64 + lane) * 16), 0);
#pragma unroll
    for (int c = 0; c < 4; c++) zreg[c] = buf_load<u32x4, true>(rk, (uint32_t)(KIVI_MF_SB_MN_WORD0 * 4 + kt_sm_word4(n, kb, c) * 4), 0);
    u32x4 wr[RING];
#pragma unroll
    for (int i = 0; i < RING; i++) wr[i] = buf_load<u32x4, true>(rk, (uint32_t)(i * 1024 + lane * 16), 0);

    // q of this lane's row: channels 32 c + 8 kb + e, normalised to max |q| in [1, 2) (Sq) and pre-multiplied by
    // 2^(6 - 2 i) per channel pair i, so that A = q'' * scale stays a normal fp16 for any realistic scale
    const uint16_t* qrow = a.q + b * a.q_sb + (int64_t)(h0 + r) * a.q_sh + 8 * kb;
    u16x8 qv[4];
#pragma unroll
    for (int c = 0; c < 4; c++) qv[c] = *(const u16x8*)(qrow + 32 * c);
    uint32_t amax = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t m = qv[c][e] & 0x7FFFu;
            amax = m > amax ? m : amax;
        }
    amax = max(amax, (uint32_t)__shfl_xor((int)amax, 16));
    amax = max(amax, (uint32_t)__shfl_xor((int)amax, 32));
    const int ex = (int)(amax >> 10);                             // biased exponent of the row maximum (0: zero / subnormal)
    const int sq = amax >= 0x7C00u ? 0 : 15 - (ex ? ex : 1);      // inf / nan rows: no scaling (they poison the row anyway)
    uint32_t qq[4][4], qf[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float f0 = __builtin_ldexpf(h2f_bits(qv[c][2 * i]), sq + aexp(i));
            const float f1 = __builtin_ldexpf(h2f_bits(qv[c][2 * i + 1]), sq + aexp(i));
            qq[c][i] = (uint32_t)f2h_bits(f0) | ((uint32_t)f2h_bits(f1) << 16);
            qf[c][i] = (lo_row == HILO) ? qq[c][i] : 0u;           // see a_elem
        }
    // per output register j of a lane: which head, its 2^(12 - Sq) and 2^-Sq
    float cmul[4], zmul[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int rj = (R == 4) ? j : 4 * ((lane >> 4) & 1) + j;  // head of output register j after fold_rows
        const int sqj = __shfl(sq, rj);                           // lane rj (kb = 0, n = rj) holds head rj's exponent
        cmul[j] = __builtin_ldexpf(1.0f, KIVI_MF_PROD_SHIFT - sqj);
        zmul[j] = __builtin_ldexpf(1.0f, -sqj);
    }

    // scale -> LDS (the wave's own region; same-wave LDS traffic is in order)
#pragma unroll
    for (int j = 0; j < 4; j++) *(u32x4*)(lds_s + (j * 64 + lane) * 4) = sreg[j];

    // zero-point term for the 16 groups of the super-block: Z[row, G] = 2^Sq * sum_d q * mn
    f4 zacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const h8 bz = as_h8(pk_mul(zreg[c][0], zfac(0)), pk_mul(zreg[c][1], zfac(1)), pk_mul(zreg[c][2], zfac(2)),
                            pk_mul(zreg[c][3], zfac(3)));
        zacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(qq[c][0], qq[c][1], qq[c][2], qq[c][3]), bz, zacc, 0, 0, 0);
    }
    // rows 0..R-1 (the "hi" rows carry plain q''): R = 4 -> lanes 0-15 reg j = head j; R = 8 -> lanes 0-31
    // (the 2^-Sq of the head is applied here, in the lane that holds the head's row: for R = 8 lanes 16-31 hold heads 4-7)
    int zz[4];
#pragma unroll
    for (int j = 0; j < 4; j++) zz[j] = __builtin_bit_cast(int, zacc[j] * zmul[j]);
    __builtin_amdgcn_wave_barrier();

    auto group = [&](int g, const u32x4& w) {
        u32x4 s[4];
#pragma unroll
        for (int c = 0; c < 4; c++) s[c] = *(const u32x4*)(lds_s + kt_sm_word4(g, kb, c));
        f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            uint32_t A[4];
#pragma unroll
            for (int i = 0; i < 4; i++) A[i] = a_elem<HILO>(qq[c][i], qf[c][i], s[c][i]);
            mfma_pair(A, w[c], acc0, acc1);
        }
        // lane l < 32 (R = 4): token l of the group, register j = head j;  R = 8: token (l & 15) + 16 (l >> 5), head 4 ((l >> 4) & 1) + j
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float zj;
            if constexpr (R == 4) {
                zj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(zz[j], g));
            } else {
                const int zlo = __builtin_amdgcn_readlane(zz[j], g), zhi = __builtin_amdgcn_readlane(zz[j], g + 16);
                zj = __builtin_bit_cast(float, (lane & 16) ? zhi : zlo);
            }
            const float v = __builtin_fmaf(fold_rows<R>(acc0[j], acc1[j]), cmul[j], zj);
            const int tok = (R == 4) ? lane : (lane & 15) + 16 * (lane >> 5);
            const int head = (R == 4) ? j : 4 * ((lane >> 4) & 1) + j;
            if (R == 8 || lane < 32) lds_o[head * 512 + g * 32 + tok] = f2h_bits(v);
        }
    };

    // ring of code blocks: RING groups in flight ahead of the one being multiplied (loads past the last group of a
    // partial super-block read the zero-filled rest of it: harmless)
    // (rolled over rounds of RING groups, unrolled inside a round: the ring slots are static registers.  No branch
    // inside a round -- with one, hipcc stages every load through one temporary and waits vmcnt(0) per group; the
    // groups past `ng` of the last round read the zero-filled rest of the super-block and their scores are not stored)
    static_assert(16 % RING == 0, "rounds must not leave the super-block");
    const int ngr = (ng + RING - 1) / RING * RING;
    for (int g0 = 0; g0 < ngr; g0 += RING) {
#pragma unroll
        for (int j = 0; j < RING; j++) {
            group(g0 + j, wr[j]);
            // reload AFTER the last use: the slot's register is dead here, so the load lands in it directly (requested at
            // the top of the group it would need a second register and a copy -- i.e. a vmcnt(0) -- at the loop edge)
            wr[j] = buf_load<u32x4, true>(rk, (uint32_t)((g0 + j + RING) * 1024 + lane * 16), 0);   // past the codes: never used
        }
    }
    __builtin_amdgcn_wave_barrier();
    // 512 tokens x R heads of fp16 scores: one 16-byte store per lane and head
    const bool valid = lane * 8 < ng * 32;
    const uint16_t* mrow = a.mask ? a.mask + b * a.mask_sb + (int64_t)sb * KIVI_MF_SB_TOKENS + lane * 8 : nullptr;
#pragma unroll
    for (int rr = 0; rr < R; rr++) {
        u16x8 v = valid ? *(const u16x8*)(lds_o + rr * 512 + lane * 8) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (a.stats) {
            float x[8], m = -__builtin_inff();
#pragma unroll
            for (int e = 0; e < 8; e++) {
                v[e] = kivi_scaled_score(v[e], a.inv_scale, mrow != nullptr, (mrow && valid) ? mrow[e] : 0);
                x[e] = h2f_bits(v[e]);
                m = __builtin_fmaxf(m, x[e]);
            }
            m = wave_max(valid ? m : -__builtin_inff());
            float l = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) l += kivi_exp(x[e] - m);
            l = wave_sum(valid ? l : 0.f);
            if (lane == 0) {
                float* st = a.stats + (((int64_t)b * a.nh + h0 + rr) * a.nseg + sb) * 2;
                st[0] = m;
                st[1] = l;
            }
        }
        if (valid)
            *(u16x8*)(a.out + b * a.out_sb + (int64_t)(h0 + rr) * a.out_sh + (int64_t)sb * KIVI_MF_SB_TOKENS + lane * 8) = v;
    }
}

template <int R, int W, bool HILO, int RING>
void launch_gqa_k(const GqaKArgs& a, int units, hipStream_t s) {
    const size_t lds = (size_t)W * (1024 + R * 256) * 4;
    KIVI_LAUNCH_LDS((gqa_k_kernel<R, W, HILO, RING>), dim3((unsigned)(a.res_blocks + units * a.sb_blocks)), dim3(64 * W), lds, s, a);
}

// shared by kivi_gqa_scores and kivi_gqa_decode
int run_gqa_k(GqaKArgs& a, int units, hipStream_t s) {
    static const char* nohilo = KIVI_TUNE_ENV("KIVI_GQA_NO_HILO");      // tuning aid: fp16-rounded q * scale (no remainder rows)
    static const char* fw = KIVI_TUNE_ENV("KIVI_GQA_K_WAVES");          // tuning aid: waves per block (1 or 4)
    static const char* fr = KIVI_TUNE_ENV("KIVI_GQA_K_RING");           // tuning aid: code blocks in flight (2, 4 or 8)
    int W = ((int64_t)units * a.nsb >= 2048) ? 4 : 1;            // few super-blocks: one wave per block spreads them over the CUs
    if (fw) W = atoi(fw) == 1 ? 1 : 4;
    const int ring = fr ? atoi(fr) : 4;
    a.sb_blocks = (a.nsb + W - 1) / W;
    if ((int64_t)a.res_blocks + (int64_t)units * a.sb_blocks == 0) return 0;
#ifdef KIVI_TUNING
#define KIVI_GK(RR, WW, HL)                                       \
    do {                                                          \
        if (ring == 2) launch_gqa_k<RR, WW, HL, 2>(a, units, s);  \
        else if (ring == 8) launch_gqa_k<RR, WW, HL, 8>(a, units, s); \
        else launch_gqa_k<RR, WW, HL, 4>(a, units, s);            \
    } while (0)
    if (a.ratio == 4) {
        if (nohilo) { if (W == 4) KIVI_GK(4, 4, false); else KIVI_GK(4, 1, false); }
        else { if (W == 4) KIVI_GK(4, 4, true); else KIVI_GK(4, 1, true); }
    } else {
        if (nohilo) { if (W == 4) KIVI_GK(8, 4, false); else KIVI_GK(8, 1, false); }
        else { if (W == 4) KIVI_GK(8, 4, true); else KIVI_GK(8, 1, true); }
    }
#else
    // product build: the round-2 kernels serve nh / nh_kv = 8 only (4 and 1 run the round-3 kernels of kivi_mf.hip)
    (void)nohilo; (void)ring;
    KIVI_REQUIRE(a.ratio == 8, KIVI_EUNSUPPORTED, "gqa_k: nh / nh_kv = %d has no round-2 kernel in this build", a.ratio);
    if (W == 4) launch_gqa_k<8, 4, true, 4>(a, units, s);
    else launch_gqa_k<8, 1, true, 4>(a, units, s);
#define KIVI_GK(RR, WW, HL)
#endif
#undef KIVI_GK
    return kivi_launch_status("gqa_k");
}

// ------------------------------------------------------------------------------------------------ sV (+ softmax, window)


// Stream role: block (unit, slice) takes `spb` consecutive super-blocks of the unit's packed V, one per wave at a time.
// DIAG (tools only, wrong results): 1 = the ring is never reloaded (no memory traffic in the loop), 2 = no MFMA / no
// accumulate, 3 = no probability chain (constant p), 4 = no A build, 5 = no ds_swizzle, 6 = no LDS reads in the loop, 7 = 5 + 6
// V2 (R == 4): MFMA row = (channel group c, head r) instead of (hi / lo, head).  A lane then needs the scale / zero points of
// ONE channel group (2 LDS reads per block instead of 5: the ablation's lever), builds hi and lo operands for it (16
// packed ops instead of 36), and every (channel group c', tile) takes two chained MFMAs (hi, lo) whose rows are useful
// where c == c' -- 18 MFMAs per block instead of 10 on a matrix pipe that is 8 % busy.  Rows (c, r) of column n land in
// lane (n, kb = c), register r: the output fold needs no transposition.
template <int R, bool HILO, int RING, bool DBG = false, int OCC = 4, int DIAG = 0, bool V2 = false>
__global__ __launch_bounds__(256, OCC) void gqa_v_kernel(const GqaVArgs a) {
    static_assert(!V2 || (R == 4 && HILO), "row = (channel group, head) needs 4 x 4 rows");
    extern __shared__ uint32_t lds_all[];                          // 4 waves x 2048 words (scale | mn of the super-block)
    gstamp<DBG>(a.dbg, 0);
    if (DBG && (threadIdx.x & 63) == 0) a.dbg[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + 1] = __builtin_amdgcn_s_memrealtime();
    const int bid = (int)blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* lds_s = lds_all + wave * 2048;
    uint32_t* lds_m = lds_s + 1024;
    // window role (a.win_blocks): the last `units` blocks of the grid stream nothing; they run when the CUs drain, next to
    // the youngest stream blocks (co-resident blocks are served oldest first), instead of 8 us at the end of EVERY stream block
    const int nstream = a.units * a.S;
    const bool win_role = bid >= nstream;
    const int unit = win_role ? bid - nstream : bid / a.S;
    const int slice = win_role ? a.S : bid - unit * a.S;           // = the block's partial-sum slot
    const int b = unit / a.nh_kv, hk = unit - b * a.nh_kv;
    const int h0 = hk * a.ratio;
    const int n = lane & 15, kb = lane >> 4;
    const int r = n % R;
    const bool lo_row = ((n / R) & 1) != 0;
    const uint32_t lomask = (lo_row == HILO) ? 0xFFFFFFFFu : 0u;   // see a_elem

    float M[R], invS[R];
    gqa_row_consts<R>(a, b, h0, M, invS);
    gstamp<DBG>(a.dbg, 2);
    // per head: Sp = floor(log2(sum)) (<= 14): the fp16 probabilities (<= 1 / sum) are scaled by 2^Sp before they enter
    // the A operand, so that p * scale * 2^(6 - 2 i) stays a normal fp16 whatever the row length
    int sp[R];
#pragma unroll
    for (int rr = 0; rr < R; rr++) {
        const float sum = 1.0f / invS[rr];
        int e = (int)((__builtin_bit_cast(uint32_t, sum) >> 23) & 255u) - 127;
        sp[rr] = e < 0 ? 0 : (e > 14 ? 14 : e);
    }
    float myM = M[0], myInv = invS[0];
    int mySp = sp[0];
#pragma unroll
    for (int rr = 1; rr < R; rr++)
        if (r == rr) { myM = M[rr]; myInv = invS[rr]; mySp = sp[rr]; }
    const uint32_t c1h = (uint32_t)(mySp + 15) << 10;              // fp16 2^Sp
    const uint32_t c1 = c1h | (c1h << 16);
    // The 16 / R lanes that hold the same A row (head r: hi, lo and their duplicates) need the same 8 probabilities of a
    // block: each computes 8 R / 16 of them (one token pair for R = 4, two for R = 8) and they exchange the packed
    // results with ds_swizzle -- 2-4 v_exp per lane and block instead of 8.
    constexpr int NCOPY = 16 / R, PPL = 4 / NCOPY;
    const int q4 = n / R;

    f4 acc[4][2], zacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c][0] = acc[c][1] = f4{0.f, 0.f, 0.f, 0.f};

    // score rows of the unit's R heads as one buffer: lane (n, kb) reads 8 scores of its head r per 32-token block
    // (extent rounded up to whole 16-byte loads: the rows are padded to a multiple of 8 scores)
    const rsrc_t rx = make_rsrc(a.x + b * a.x_sb + (int64_t)h0 * a.x_sh, (uint32_t)((R - 1) * a.x_sh * 2 + ((a.Tv + 7) & ~(int64_t)7) * 2));
    const uint32_t xoff = (uint32_t)((r * a.x_sh + 8 * kb + 2 * PPL * q4) * 2);
    typedef typename std::conditional<PPL == 1, uint32_t, u32x2>::type XV;

    const int sb_begin = win_role ? 0 : slice * a.spb;
    const int sb_end = win_role ? 0 : ((sb_begin + a.spb < a.nsb) ? sb_begin + a.spb : a.nsb);
    for (int sb = sb_begin + wave; sb < sb_end; sb += 4) {
        const int64_t tok0 = (int64_t)sb * KIVI_MF_SB_TOKENS;
        int ng = (int)((a.Tv - tok0 + 31) / 32);
        ng = ng > 16 ? 16 : ng;
        const rsrc_t rv = make_rsrc(mf_sb(a.vt, b, hk, sb), KIVI_MF_SB_WORDS * 4);
        u32x4 sreg[4], mreg[4];
#pragma unroll
        for (int j = 0; j < 4; j++) sreg[j] = buf_load<u32x4, true>(rv, (uint32_t)(KIVI_MF_SB_SCALE_WORD0 * 4 + (j * 64 + lane) * 16), 0);
#pragma unroll
        for (int j = 0; j < 4; j++) mreg[j] = buf_load<u32x4, true>(rv, (uint32_t)(KIVI_MF_SB_MN_WORD0 * 4 + (j * 64 + lane) * 16), 0);
        u32x4 wr[RING];
        XV xr[RING];
#pragma unroll
        for (int i = 0; i < RING; i++) {
            wr[i] = buf_load<u32x4, true>(rv, (uint32_t)(i * 1024 + lane * 16), 0);
            xr[i] = buf_load<XV, false>(rx, xoff + (uint32_t)((tok0 + i * 32) * 2), 0);
        }
        __builtin_amdgcn_wave_barrier();                           // the previous super-block's LDS reads are over
#pragma unroll
        for (int j = 0; j < 4; j++) {
            *(u32x4*)(lds_s + (j * 64 + lane) * 4) = sreg[j];
            *(u32x4*)(lds_m + (j * 64 + lane) * 4) = mreg[j];
        }
        __builtin_amdgcn_wave_barrier();
        // rounds of RING blocks without a branch inside (see gqa_k_kernel); blocks past `ng` of the last round lie inside
        // the super-block, hold zeros (never-written slots) and get zero probabilities
        static_assert(16 % RING == 0, "rounds must not leave the super-block");
        const int ngr = (ng + RING - 1) / RING * RING;
        const int64_t left = a.Tv - tok0;
        const int lim = (int)(left > 512 ? 512 : left) - (8 * kb + 2 * PPL * q4);   // this lane's share: token offset < lim is valid
        for (int g0 = 0; g0 < ngr; g0 += RING) {
#pragma unroll
            for (int j = 0; j < RING; j++) {
                const int g = g0 + j;
                const u32x4& w = wr[j];
                const XV& xv = xr[j];
                // scale of channel group 0 (V2: of the lane's own channel group): in flight during the exps
                const u32x4 s_first = *(const u32x4*)(lds_s + g * 64 + kb * 16 + (V2 ? (n >> 2) * 4 : 0));
                // this lane's share of the probabilities: fp16(exp(x - M) / sum) as the reference casts them
                // (llama_kivi.py:375), then the exact power-of-two scaling by 2^Sp; slots past the packed prefix get 0
                uint32_t own[PPL];
#pragma unroll
                for (int jj = 0; jj < PPL; jj++) {
                    uint32_t xw;
                    if constexpr (PPL == 1) xw = xv;
                    else xw = xv[jj];
                    float p0 = kivi_exp(h2f_bits((uint16_t)(xw & 0xFFFFu)) - myM) * myInv;
                    float p1 = kivi_exp(h2f_bits((uint16_t)(xw >> 16)) - myM) * myInv;
                    if constexpr (DIAG == 3) { p0 = myInv; p1 = myInv + __builtin_bit_cast(float, xw & 1u); }
                    const int t0 = g * 32 + 2 * jj;
                    p0 = (t0 < lim) ? p0 : 0.f;
                    p1 = (t0 + 1 < lim) ? p1 : 0.f;
                    own[jj] = pk_mul((uint32_t)f2h_bits(p0) | ((uint32_t)f2h_bits(p1) << 16), c1);
                }
                // pz[i] <- the lane (same head, same kb) whose share holds pair i: lane' = (lane & 0x13) | (copy << 2)
                uint32_t pz[4], pp[4];
                constexpr int AND = (R == 4) ? 0x13 : 0x17;            // or_mask = copy index * R (sets the n / R bits)
                if constexpr (DIAG == 5 || DIAG == 7) {
                    pz[0] = own[0]; pz[1] = own[0] + 1u; pz[2] = own[0] + 2u; pz[3] = own[0] + 3u;
                } else {
                    pz[0] = swz<AND | (((0 / PPL) * R) << 5)>(own[0 % PPL]);
                    pz[1] = swz<AND | (((1 / PPL) * R) << 5)>(own[1 % PPL]);
                    pz[2] = swz<AND | (((2 / PPL) * R) << 5)>(own[2 % PPL]);
                    pz[3] = swz<AND | (((3 / PPL) * R) << 5)>(own[3 % PPL]);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) pp[i] = pk_mul(pz[i], afac(i));
                // Every MFMA starts from a ZERO accumulator and its result is added to the fp32 running sums on the VALU:
                // the matrix pipe aligns the 32 products and C to the largest exponent and truncates what falls below
                // ~2^-23 of it (tools/mfma_prec_probe.hip), so a long chain of same-sign products (codes >= 0, zero
                // points < 0) through C loses ~2^-19 |C| per step -- 2e-3 of the output after the two sums cancel.
                const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (V2) {
                    const int cg = n >> 2;                                    // this lane's row: channel group cg, head n & 3
                    const u32x4 m4 = *(const u32x4*)(lds_m + g * 64 + kb * 16 + cg * 4);
                    uint32_t Ah[4], Al[4], Zh[4], Zl[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        Ah[i] = pk_mul(pp[i], s_first[i]);
                        Al[i] = pk_fms(pp[i], s_first[i], Ah[i]);
                    }
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        f4 d0 = zero4, d1 = zero4;
                        mfma_pair(Ah, w[c], d0, d1);
                        mfma_pair(Al, w[c], d0, d1);
                        acc[c][0] += d0;
                        acc[c][1] += d1;
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        Zh[i] = pk_mul(pz[i], m4[i]);
                        Zl[i] = pk_fms(pz[i], m4[i], Zh[i]);
                    }
                    const h8 ones = as_h8(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);
                    f4 z = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(Zh[0], Zh[1], Zh[2], Zh[3]), ones, zero4, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(Zl[0], Zl[1], Zl[2], Zl[3]), ones, z, 0, 0, 0);
                    zacc += z;
                } else {
                // the LDS operands (scale of channel group c + 1, then the zero points) are requested one step ahead of their use
                u32x4 s_next = s_first;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const u32x4 s = s_next;
                    if constexpr (DIAG == 6 || DIAG == 7) s_next = s_first + (uint32_t)c;
                    else if (c < 3) s_next = *(const u32x4*)(lds_s + g * 64 + kb * 16 + (c + 1) * 4);
                    else s_next = *(const u32x4*)(lds_m + g * 64 + kb * 16 + (n & 3) * 4);
                    uint32_t A[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) A[i] = (DIAG == 4) ? (pp[i] ^ s[i]) : a_elem<HILO>(pp[i], pp[i] & lomask, s[i]);
                    if constexpr (DIAG == 2) {
                        acc[c][0][0] += __builtin_bit_cast(float, (A[0] ^ A[1] ^ A[2] ^ A[3] ^ w[c]) & 0x3FFFFFFFu);
                    } else {
                        f4 d0 = zero4, d1 = zero4;
                        mfma_pair(A, w[c], d0, d1);
                        acc[c][0] += d0;
                        acc[c][1] += d1;
                    }
                }
                // zero-point term: Z[row, c] += sum_t p' * mn[t, c]  (columns n -> channel group n & 3)
                const u32x4 bz = s_next;
                zacc += __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(pz[0], pz[1], pz[2], pz[3]), as_h8(bz[0], bz[1], bz[2], bz[3]), zero4, 0, 0, 0);
                }
                // reload AFTER the last use (see gqa_k_kernel)
                if constexpr (DIAG != 1) {
                    wr[j] = buf_load<u32x4, true>(rv, (uint32_t)((g + RING) * 1024 + lane * 16), 0);   // past the codes: never used
                    xr[j] = buf_load<XV, false>(rx, xoff + (uint32_t)((tok0 + (g + RING) * 32) * 2), 0);
                }
            }
        }
    }

    gstamp<DBG>(a.dbg, 3);
    // ---- this slice's share of the fp16 window: probs[..., Tv + t] * V_window[t] for t in [w0, w1) (llama_kivi.py:384), the
    // V append (:377) by the slice that holds the new token, the quantisation of the token leaving the window (:386-399)
    // by slice 0.  A lane owns two channels, wave w the tokens w0 + w, w0 + w + 4, ...; all loads of a batch in flight.
    constexpr int PW = 136, WB = 12;
    __shared__ uint16_t pw[R][PW];
    const int Lw = a.res_len + 1;
    const int wchunk = a.win_blocks ? Lw : (Lw + a.S - 1) / a.S;
    const int w0 = a.win_blocks ? 0 : slice * wchunk;
    const int w1 = a.win_blocks ? (win_role ? Lw : 0) : ((w0 + wchunk < Lw) ? w0 + wchunk : Lw);
    const bool flusher = a.flush && (a.win_blocks ? win_role : slice == 0);
    const int nwt = w1 > w0 ? w1 - w0 : 0;
    uint16_t* vwin = a.vres + b * a.vres_sb + hk * a.vres_sh + (int64_t)a.win_start * a.vres_st;
    const uint16_t* vnew = a.vnew + b * a.vnew_sb + hk * a.vnew_sh;
    uint16_t xflush = 0;
    if (flusher && threadIdx.x < 128) xflush = vwin[threadIdx.x];      // requested early, used last
    for (int idx = threadIdx.x; idx < R * nwt; idx += 256) {
        const int rr = idx / nwt, t = idx - rr * nwt;
        float Mr = M[0], Ir = invS[0];
#pragma unroll
        for (int q = 1; q < R; q++)
            if (rr == q) { Mr = M[q]; Ir = invS[q]; }
        const float xw = h2f_bits(a.x[b * a.x_sb + (int64_t)(h0 + rr) * a.x_sh + a.Tv + w0 + t]);
        pw[rr][t] = f2h_bits(kivi_exp(xw - Mr) * Ir);
    }
    __syncthreads();
    float ow[R][2];
#pragma unroll
    for (int rr = 0; rr < R; rr++) ow[rr][0] = ow[rr][1] = 0.f;
    for (int tb = wave; tb < nwt; tb += 4 * WB) {
        uint32_t vv[WB];
#pragma unroll
        for (int u = 0; u < WB; u++) {
            const int t = w0 + tb + 4 * u;
            const uint16_t* vrow = (t < a.res_len) ? vwin + (int64_t)t * a.vres_st : vnew;
            vv[u] = (t < w1) ? *(const uint32_t*)(vrow + 2 * lane) : 0u;
        }
#pragma unroll
        for (int u = 0; u < WB; u++) {
            const int t = w0 + tb + 4 * u;
            if (t < w1) {
                const float v0 = h2f_bits((uint16_t)(vv[u] & 0xFFFFu)), v1 = h2f_bits((uint16_t)(vv[u] >> 16));
#pragma unroll
                for (int rr = 0; rr < R; rr++) {
                    const float p = h2f_bits(pw[rr][t - w0]);
                    ow[rr][0] = __builtin_fmaf(p, v0, ow[rr][0]);
                    ow[rr][1] = __builtin_fmaf(p, v1, ow[rr][1]);
                }
                if (t == a.res_len) *(uint32_t*)(vwin + (int64_t)t * a.vres_st + 2 * lane) = vv[u];   // V append
            }
        }
    }
    if (flusher && threadIdx.x < 128) {   // waves 0 and 1 (wave-uniform)
        const int d = threadIdx.x;
        const uint32_t key = h_key(xflush);
        uint32_t kmin = key, kmax = key;
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
            const uint32_t o1 = (uint32_t)__shfl_xor((int)kmin, m), o2 = (uint32_t)__shfl_xor((int)kmax, m);
            kmin = o1 < kmin ? o1 : kmin;
            kmax = o2 > kmax ? o2 : kmax;
        }
        const GroupQ gq = make_group(kmin, kmax, 3);
        const uint32_t code = quant_one<2>(xflush, gq);
        const int tt = (int)(a.Tv & 31), blk = (int)((a.Tv >> 5) & 15);
        const int e = tt & 7, kbq = tt >> 3;
        const int c = d >> 5, tile = (d >> 4) & 1, nn = d & 15;
        uint32_t val = code << (mf_pos(tile, e >> 1) + 16 * (e & 1));
        val |= (uint32_t)__shfl_xor((int)val, 16);
        uint32_t* sbp = mf_sb(a.vt, b, hk, a.Tv >> 9);
        if (tile == 0) {
            uint32_t* wp = sbp + blk * KIVI_MF_BLOCK_WORDS + (nn + 16 * kbq) * 4 + c;
            *wp = *wp | val;                 // the slot of a token is written once, on zero-initialised storage
        }
        if ((d & 31) == 0) {
            const int hidx = blk * 128 + kbq * 32 + c * 8 + e;
            ((uint16_t*)(sbp + KIVI_MF_SB_SCALE_WORD0))[hidx] = gq.scale;
            ((uint16_t*)(sbp + KIVI_MF_SB_MN_WORD0))[hidx] = gq.mn;
        }
    }

    float* Lf = reinterpret_cast<float*>(lds_s);                   // the wave's own 8 KiB
    if constexpr (V2) {
        // rows (c, r) of column n sit in lane (n, kb = c), register r: O[r, 32 kb + 16 tile + n] = 2^-Sp (2^12 acc[kb] + Z[r, kb])
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tile = 0; tile < 2; tile++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = acc[0][tile][j];
                if (kb == 1) v = acc[1][tile][j];
                if (kb == 2) v = acc[2][tile][j];
                if (kb == 3) v = acc[3][tile][j];
                Lf[j * 128 + 32 * kb + 16 * tile + n] = __builtin_ldexpf(__builtin_fmaf(v, (float)(1 << KIVI_MF_PROD_SHIFT), zacc[j]), -sp[j]);
            }
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            Lf[R * 128 + rr * 128 + 2 * lane] = ow[rr][0];
            Lf[R * 128 + rr * 128 + 2 * lane + 1] = ow[rr][1];
        }
    } else {
    // fold: O[r, d] = 2^-Sp * (2^12 * (hi + lo rows) + Z[r, d >> 5]); lane takes d = lane and lane + 64
    float zsel[R][2];
#pragma unroll
    for (int rr = 0; rr < R; rr++)
#pragma unroll
        for (int half = 0; half < 2; half++)
            zsel[rr][half] = __shfl(zacc[rr % 4], (rr / 4) * 16 + (lane >> 5) + 2 * half);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int tile = 0; tile < 2; tile++)
#pragma unroll
            for (int j = 0; j < 4; j++) Lf[(4 * kb + j) * 128 + 32 * c + 16 * tile + n] = acc[c][tile][j];
    __builtin_amdgcn_wave_barrier();
    float o[R][2];
#pragma unroll
    for (int rr = 0; rr < R; rr++)
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int d = lane + 64 * half;
            const float v = Lf[rr * 128 + d] + Lf[(rr + R) * 128 + d];
            o[rr][half] = __builtin_ldexpf(__builtin_fmaf(v, (float)(1 << KIVI_MF_PROD_SHIFT), zsel[rr][half]), -sp[rr]);
        }
    __builtin_amdgcn_wave_barrier();
    // per-wave [quantised part (R x 128) | window part (R x 128)] at the start of the wave's region, then the 4 waves
#pragma unroll
    for (int rr = 0; rr < R; rr++) {
        Lf[rr * 128 + lane] = o[rr][0];
        Lf[rr * 128 + lane + 64] = o[rr][1];
        Lf[R * 128 + rr * 128 + 2 * lane] = ow[rr][0];
        Lf[R * 128 + rr * 128 + 2 * lane + 1] = ow[rr][1];
    }
    }
    __syncthreads();
    float* lf = reinterpret_cast<float*>(lds_all);
    constexpr int NT = (2 * R * 128 + 255) / 256;
    float tot[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int i = threadIdx.x + 256 * k;
        tot[k] = (i < 2 * R * 128) ? (lf[i] + lf[2048 + i]) + (lf[4096 + i] + lf[6144 + i]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int i = threadIdx.x + 256 * k;
        if (i < 2 * R * 128) lf[i] = tot[k];
    }
    __syncthreads();
    gstamp<DBG>(a.dbg, 4);
    gqa_arrive_and_combine<R>(a, unit, slice, lf, b, h0);
    gstamp<DBG>(a.dbg, 5);
    if (DBG && (threadIdx.x & 63) == 0) a.dbg[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + 12] = __builtin_amdgcn_s_memrealtime();
}

// Largest byte extent a buffer descriptor over a unit's store may have: requests past the end of a wave's stream carry the
// per-lane offset MF_DEAD_OFF (kivi_mf_dev.h) and must fall OUTSIDE the descriptor's range (zeros, no memory access)
constexpr uint32_t MF_DESC_LIMIT = 0xFFFE0000u;

bool mf_store_ok(const void* base, int64_t sb_b, int64_t sb_h, int64_t sb_s) {
    return base && (uintptr_t)base % 16 == 0 && sb_b % 4 == 0 && sb_h % 4 == 0 && sb_s % 4 == 0 && sb_s >= KIVI_MF_SB_WORDS;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI

#define KIVI_MF_SHAPE_CHECK(who)                                                                                       \
    KIVI_REQUIRE(bits == 2 && group_size == 32 && D == 128, KIVI_EUNSUPPORTED,                                         \
                 who ": the MFMA cache layout covers 2-bit codes, group_size 32, head_dim 128 (got %d / %d / %d)", bits, \
                 group_size, D);                                                                                       \
    KIVI_REQUIRE(B > 0 && nh_kv > 0, KIVI_EINVAL, who ": empty batch");                                                \
    KIVI_REQUIRE((int64_t)B * nh_kv * ((T + 31) / 32) < ((int64_t)1 << 31), KIVI_EINVAL, who ": grid too large")

extern "C" int kivi_kt_pack(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_st, void* kt, int64_t kt_sb, int64_t kt_sh,
                            int64_t kt_ss, void* kt_range, int64_t token_offset, int B, int nh_kv, int64_t T, int D, int group_size,
                            int bits, kivi_stream_t stream) {
    KIVI_MF_SHAPE_CHECK("kivi_kt_pack");
    KIVI_REQUIRE(kt_range && (uintptr_t)kt_range % 4 == 0, KIVI_EINVAL, "kivi_kt_pack: null / misaligned range flags");
    KIVI_REQUIRE(T >= 0 && T % 32 == 0 && token_offset >= 0 && token_offset % 32 == 0, KIVI_EINVAL,
                 "kivi_kt_pack: T=%lld and token_offset=%lld must be multiples of the 32-token block", (long long)T,
                 (long long)token_offset);
    KIVI_REQUIRE(mf_store_ok(kt, kt_sb, kt_sh, kt_ss), KIVI_EALIGN, "kivi_kt_pack: cache storage must be 16-byte aligned super-blocks");
    KIVI_REQUIRE(k && (uintptr_t)k % 16 == 0 && k_sb % 8 == 0 && k_sh % 8 == 0 && k_st % 8 == 0, KIVI_EALIGN,
                 "kivi_kt_pack: key rows must be 16-byte aligned");
    if (T == 0) return 0;
    const int nblk = (int)(T / 32);
    const MfStore st = {(uint32_t*)kt, kt_sb, kt_sh, kt_ss};
    const int64_t ntile = (int64_t)B * nh_kv * nblk;
    hipLaunchKernelGGL(kt_pack_kernel, dim3((unsigned)((ntile + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)k, k_sb, k_sh, k_st, st, (int*)kt_range, token_offset / 32, nblk, nh_kv, ntile);
    return kivi_launch_status("kt_pack");
}

extern "C" int kivi_vt_pack(const void* v, int64_t v_sb, int64_t v_sh, int64_t v_st, void* vt, int64_t vt_sb, int64_t vt_sh,
                            int64_t vt_ss, void* vt_range, int B, int nh_kv, int64_t T, int D, int group_size, int bits,
                            kivi_stream_t stream) {
    KIVI_MF_SHAPE_CHECK("kivi_vt_pack");
    KIVI_REQUIRE(vt_range && (uintptr_t)vt_range % 4 == 0, KIVI_EINVAL, "kivi_vt_pack: null / misaligned range flags");
    KIVI_REQUIRE(T >= 0, KIVI_EINVAL, "kivi_vt_pack: negative length");
    KIVI_REQUIRE(mf_store_ok(vt, vt_sb, vt_sh, vt_ss), KIVI_EALIGN, "kivi_vt_pack: cache storage must be 16-byte aligned super-blocks");
    KIVI_REQUIRE(v && (uintptr_t)v % 16 == 0 && v_sb % 8 == 0 && v_sh % 8 == 0 && v_st % 8 == 0, KIVI_EALIGN,
                 "kivi_vt_pack: value rows must be 16-byte aligned");
    if (T == 0) return 0;
    const int nblk = (int)((T + 31) / 32);
    const MfStore st = {(uint32_t*)vt, vt_sb, vt_sh, vt_ss};
    const int64_t ntile = (int64_t)B * nh_kv * nblk;
    hipLaunchKernelGGL(vt_pack_kernel, dim3((unsigned)((ntile + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)v, v_sb, v_sh, v_st, st, (int*)vt_range, T, nblk, nh_kv, ntile);
    return kivi_launch_status("vt_pack");
}

extern "C" int kivi_kt_relayout(int to_ref, void* kt, int64_t kt_sb, int64_t kt_sh, int64_t kt_ss, void* kt_range, void* code,
                                int64_t code_sb, int64_t code_sh, int64_t code_sr, void* scale, void* mn, int64_t sm_sb,
                                int64_t sm_sh, int64_t sm_sr, int B, int nh_kv, int64_t T, int D, int group_size, int bits,
                                kivi_stream_t stream) {
    KIVI_MF_SHAPE_CHECK("kivi_kt_relayout");
    KIVI_REQUIRE(T >= 0 && T % 32 == 0, KIVI_EINVAL, "kivi_kt_relayout: T=%lld must be a multiple of 32", (long long)T);
    KIVI_REQUIRE(mf_store_ok(kt, kt_sb, kt_sh, kt_ss) && code && scale && mn, KIVI_EALIGN, "kivi_kt_relayout: bad buffers");
    KIVI_REQUIRE(to_ref || (kt_range && (uintptr_t)kt_range % 4 == 0), KIVI_EINVAL, "kivi_kt_relayout: writing a store needs its range flags");
    if (T == 0) return 0;
    const int nblk = (int)(T / 32);
    const MfStore st = {(uint32_t*)kt, kt_sb, kt_sh, kt_ss};
    const dim3 grid((unsigned)((int64_t)B * nh_kv * nblk));
    if (to_ref)
        hipLaunchKernelGGL(kt_relayout_kernel<true>, grid, dim3(128), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                           code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, nblk, nh_kv, (int*)kt_range);
    else
        hipLaunchKernelGGL(kt_relayout_kernel<false>, grid, dim3(128), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                           code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, nblk, nh_kv, (int*)kt_range);
    return kivi_launch_status("kt_relayout");
}

extern "C" int kivi_vt_relayout(int to_ref, void* vt, int64_t vt_sb, int64_t vt_sh, int64_t vt_ss, void* vt_range, void* code,
                                int64_t code_sb, int64_t code_sh, int64_t code_sr, void* scale, void* mn, int64_t sm_sb,
                                int64_t sm_sh, int64_t sm_sr, int B, int nh_kv, int64_t T, int D, int group_size, int bits,
                                kivi_stream_t stream) {
    KIVI_MF_SHAPE_CHECK("kivi_vt_relayout");
    KIVI_REQUIRE(T >= 0, KIVI_EINVAL, "kivi_vt_relayout: negative length");
    KIVI_REQUIRE(mf_store_ok(vt, vt_sb, vt_sh, vt_ss) && code && scale && mn, KIVI_EALIGN, "kivi_vt_relayout: bad buffers");
    KIVI_REQUIRE(to_ref || (vt_range && (uintptr_t)vt_range % 4 == 0), KIVI_EINVAL, "kivi_vt_relayout: writing a store needs its range flags");
    if (T == 0) return 0;
    const int nblk = (int)((T + 31) / 32);
    const MfStore st = {(uint32_t*)vt, vt_sb, vt_sh, vt_ss};
    const dim3 grid((unsigned)((int64_t)B * nh_kv * nblk));
    if (to_ref)
        hipLaunchKernelGGL(vt_relayout_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                           code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, T, nblk, nh_kv, (int*)vt_range);
    else
        hipLaunchKernelGGL(vt_relayout_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                           code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, T, nblk, nh_kv, (int*)vt_range);
    return kivi_launch_status("vt_relayout");
}

// round-3 kernels for nh / nh_kv in {1, 4} (kivi_mf.hip); the argument blocks cross the translation-unit boundary as void*
int kivi_mf_run_k(void* k_args, int units, hipStream_t s);
int kivi_mf_run_v(const void* v_args, int prob, hipStream_t s);
int kivi_mf_run_row_sp(const void* p, int64_t p_sb, int64_t p_sh, int B, int nh, int nh_kv, int64_t T, const int* range, int* sp,
                       hipStream_t s);
int kivi_mf_run_row(const void* k_args, const void* v_args, int units, int dump, hipStream_t s);

static bool mf_new_path(int ratio) {
    static const char* old = KIVI_TUNE_ENV("KIVI_MF_OLD");              // tuning builds (A/B): nh / nh_kv = 4 on the round-2 kernels
    return ratio == 1 || (ratio == 4 && !(old && atoi(old)));
}

extern "C" int kivi_gqa_scores(const void* q, int64_t q_sb, int64_t q_sh, const void* kt, int64_t kt_sb, int64_t kt_sh,
                               int64_t kt_ss, const void* kt_range, void* out, int64_t out_sb, int64_t out_sh, int B, int nh,
                               int nh_kv, int D, int64_t T, int group_size, int bits, kivi_stream_t stream) {
    KIVI_MF_SHAPE_CHECK("kivi_gqa_scores");
    KIVI_REQUIRE(nh > 0 && nh % nh_kv == 0 && (nh / nh_kv == 1 || nh / nh_kv == 4 || nh / nh_kv == 8), KIVI_EUNSUPPORTED,
                 "kivi_gqa_scores: nh / nh_kv must be 1, 4 or 8 (got %d / %d)", nh, nh_kv);
    KIVI_REQUIRE(T >= 0 && T % 32 == 0, KIVI_EINVAL, "kivi_gqa_scores: T=%lld must be a multiple of 32", (long long)T);
    KIVI_REQUIRE(mf_store_ok(kt, kt_sb, kt_sh, kt_ss), KIVI_EALIGN, "kivi_gqa_scores: cache storage must be 16-byte aligned super-blocks");
    KIVI_REQUIRE(kt_range && (uintptr_t)kt_range % 4 == 0, KIVI_EINVAL, "kivi_gqa_scores: null / misaligned range flags");
    KIVI_REQUIRE(((T + KIVI_MF_SB_TOKENS - 1) / KIVI_MF_SB_TOKENS) * kt_ss * 4 <= (int64_t)MF_DESC_LIMIT, KIVI_EINVAL,
                 "kivi_gqa_scores: store too large for one descriptor");
    KIVI_REQUIRE(q && (uintptr_t)q % 16 == 0 && q_sb % 8 == 0 && q_sh % 8 == 0, KIVI_EALIGN, "kivi_gqa_scores: q rows must be 16-byte aligned");
    KIVI_REQUIRE(out && (uintptr_t)out % 16 == 0 && out_sb % 8 == 0 && out_sh % 8 == 0, KIVI_EALIGN,
                 "kivi_gqa_scores: score rows must be 16-byte aligned");
    if (T == 0) return 0;
    GqaKArgs a;
    a.q = (const uint16_t*)q; a.q_sb = q_sb; a.q_sh = q_sh;
    a.kt = {(uint32_t*)kt, kt_sb, kt_sh, kt_ss};
    a.out = (uint16_t*)out; a.out_sb = out_sb; a.out_sh = out_sh;
    a.nh_kv = nh_kv; a.ratio = nh / nh_kv; a.Tq = T;
    a.nsb = (int)((T + KIVI_MF_SB_TOKENS - 1) / KIVI_MF_SB_TOKENS);
    a.nh = nh;
    a.stats = nullptr; a.nseg = 0; a.inv_scale = 1.0f; a.mask = nullptr; a.mask_sb = 0;
    a.res_blocks = 0; a.res_first = 0; a.kres = nullptr; a.knew = nullptr; a.res_len = 0;
    a.kres_sb = a.kres_sh = a.kres_st = a.knew_sb = a.knew_sh = 0;
    a.range = (const int*)kt_range;
    if (mf_new_path(a.ratio)) return kivi_mf_run_k(&a, B * nh_kv, (hipStream_t)stream);
    return run_gqa_k(a, B * nh_kv, (hipStream_t)stream);
}

// slices of the sV launch: ~1024 stream blocks (4 per CU) of 4 waves, a wave then streams 1-2 super-blocks (24 KiB each)
static void gqa_v_slices(int units, int nsbv, int R, int& S, int& spb) {
    S = 1; spb = 0;
    if (nsbv <= 0) return;
    static const char* fs = KIVI_TUNE_ENV("KIVI_GQA_V_BLOCKS");         // tuning aid: target number of stream blocks
    // R = 4 blocks carry four heads each: 2 per CU measured best
    // (config 4: 512 blocks 59.7 us, 1024 63.5; the 70B-like slice 63.8 vs 68.6)
    const int target = fs ? atoi(fs) : (R == 4 ? 512 : 1024);
    S = (target + units - 1) / units;
    S = S < 1 ? 1 : (S > nsbv ? nsbv : S);
    spb = (nsbv + S - 1) / S;
    // the four waves of a block take super-blocks w, w + 4, ... of its slice: slices of fewer than 4 leave waves idle
    if (spb < 4 && nsbv >= 4 && (int64_t)units * ((nsbv + 3) / 4) >= 256) spb = 4;
    S = (nsbv + spb - 1) / spb;
}

extern "C" int kivi_gqa_output(const void* probs, int64_t p_sb, int64_t p_sh, const void* vt, int64_t vt_sb, int64_t vt_sh,
                               int64_t vt_ss, const void* vt_range, void* out, int64_t out_sb, int64_t out_sh, int B, int nh,
                               int nh_kv, int D, int64_t T, int group_size, int bits, void* workspace, int64_t workspace_bytes,
                               kivi_stream_t stream) {
    KIVI_MF_SHAPE_CHECK("kivi_gqa_output");
    KIVI_REQUIRE(nh > 0 && nh % nh_kv == 0 && (nh / nh_kv == 1 || nh / nh_kv == 4), KIVI_EUNSUPPORTED,
                 "kivi_gqa_output: nh / nh_kv must be 1 or 4 (got %d / %d)", nh, nh_kv);
    KIVI_REQUIRE(T >= 0, KIVI_EINVAL, "kivi_gqa_output: negative length");
    KIVI_REQUIRE(mf_store_ok(vt, vt_sb, vt_sh, vt_ss), KIVI_EALIGN, "kivi_gqa_output: cache storage must be 16-byte aligned super-blocks");
    KIVI_REQUIRE(probs && (uintptr_t)probs % 16 == 0 && p_sb % 8 == 0 && p_sh % 8 == 0 && p_sh >= ((T + 7) & ~(int64_t)7), KIVI_EALIGN,
                 "kivi_gqa_output: probability rows must be 16-byte aligned and padded to a multiple of 8");
    KIVI_REQUIRE(out != nullptr, KIVI_EINVAL, "kivi_gqa_output: null output");
    const int R = nh / nh_kv, units = B * nh_kv;
    KIVI_REQUIRE((int64_t)(R - 1) * p_sh * 2 + T * 2 + 16 < ((int64_t)1 << 32), KIVI_EINVAL, "kivi_gqa_output: rows too long");
    KIVI_REQUIRE(units <= KIVI_GQA_WS_COUNTERS, KIVI_EUNSUPPORTED, "kivi_gqa_output: more than %d (batch row, kv head) units", KIVI_GQA_WS_COUNTERS);
    const int nsbv = (int)((T + KIVI_MF_SB_TOKENS - 1) / KIVI_MF_SB_TOKENS);
    KIVI_REQUIRE((int64_t)nsbv * vt_ss * 4 <= (int64_t)MF_DESC_LIMIT, KIVI_EINVAL, "kivi_gqa_output: store too large for one descriptor");
    KIVI_REQUIRE(vt_range && (uintptr_t)vt_range % 4 == 0, KIVI_EINVAL, "kivi_gqa_output: null / misaligned range flags");
    int S, spb;
    gqa_v_slices(units, nsbv, R, S, spb);
    const int64_t sp_bytes = ((int64_t)B * nh * 4 + 255) / 256 * 256;
    const int64_t need = (int64_t)KIVI_GQA_WS_COUNTERS * 4 + sp_bytes + (int64_t)units * S * 2 * R * 128 * 4;
    KIVI_REQUIRE(workspace && (uintptr_t)workspace % 16 == 0 && workspace_bytes >= need, KIVI_EINVAL,
                 "kivi_gqa_output: workspace too small (%lld bytes needed)", (long long)need);
    hipStream_t s = (hipStream_t)stream;
    GqaVArgs v;
    memset(&v, 0, sizeof(v));
    v.x = (const uint16_t*)probs; v.x_sb = p_sb; v.x_sh = p_sh;
    v.vt = {(uint32_t*)vt, vt_sb, vt_sh, vt_ss};
    v.nh_kv = nh_kv; v.ratio = R; v.nh = nh; v.Tv = T; v.nsb = nsbv; v.S = S; v.spb = spb;
    v.units = units; v.win_blocks = 0; v.nslot = S;
    v.out = (uint16_t*)out; v.out_sb = out_sb; v.out_sh = out_sh;
    v.counters = (int*)workspace;
    v.sp_rows = (const int*)((char*)workspace + (size_t)KIVI_GQA_WS_COUNTERS * 4);
    v.ws = (float*)((char*)workspace + (size_t)KIVI_GQA_WS_COUNTERS * 4 + sp_bytes);
    v.range = (int*)vt_range;        // (read only here: no V flush in this launch)
    int rc = kivi_mf_run_row_sp(probs, p_sb, p_sh, B, nh, nh_kv, T, (const int*)vt_range, (int*)v.sp_rows, s);
    if (rc) return rc;
    return kivi_mf_run_v(&v, 1, s);
}

#ifdef KIVI_TUNING
template <int R, bool HILO, int RING>
static void launch_gqa_v(const GqaVArgs& a, int units, hipStream_t s) {
    static const char* occ = KIVI_TUNE_ENV("KIVI_GQA_V_OCC");            // tuning aid: 3 = let the kernel use up to 168 registers
    if (occ && atoi(occ) == 3 && HILO && RING == 4) {
        KIVI_LAUNCH_LDS((gqa_v_kernel<R, true, 4, false, 3>), dim3((unsigned)(units * a.S + a.win_blocks)), dim3(256), 4 * 2048 * 4 + (R > 4 ? 8192 : 0), s, a);
        return;
    }
    static const char* v2 = KIVI_TUNE_ENV("KIVI_GQA_V2");                // tuning aid: 1 / 0 = row = (channel group, head) mapping on / off
    if (R == 4 && HILO && (v2 ? atoi(v2) != 0 : true)) {
        const dim3 grid((unsigned)(units * a.S + a.win_blocks));
        static const char* occ2 = KIVI_TUNE_ENV("KIVI_GQA_V_OCC");
        if (a.dbg) KIVI_LAUNCH_LDS((gqa_v_kernel<4, true, 4, true, 4, 0, true>), grid, dim3(256), 4 * 2048 * 4, s, a);
        else if (occ2 && atoi(occ2) == 3) KIVI_LAUNCH_LDS((gqa_v_kernel<4, true, 4, false, 3, 0, true>), grid, dim3(256), 4 * 2048 * 4, s, a);
        else if (RING == 2) KIVI_LAUNCH_LDS((gqa_v_kernel<4, true, 2, false, 4, 0, true>), grid, dim3(256), 4 * 2048 * 4, s, a);
        else KIVI_LAUNCH_LDS((gqa_v_kernel<4, true, 4, false, 4, 0, true>), grid, dim3(256), 4 * 2048 * 4, s, a);
        return;
    }
    static const char* dg = KIVI_TUNE_ENV("KIVI_GQA_V_DIAG");
    if (dg && HILO && RING == 4 && R == 4) {
        const dim3 grid((unsigned)(units * a.S + a.win_blocks));
        const size_t lds = 4 * 2048 * 4;
        switch (atoi(dg)) {
            case 1: KIVI_LAUNCH_LDS((gqa_v_kernel<4, true, 4, false, 4, 1>), grid, dim3(256), lds, s, a); return;
            case 2: KIVI_LAUNCH_LDS((gqa_v_kernel<4, true, 4, false, 4, 2>), grid, dim3(256), lds, s, a); return;
            case 3: KIVI_LAUNCH_LDS((gqa_v_kernel<4, true, 4, false, 4, 3>), grid, dim3(256), lds, s, a); return;
            case 4: KIVI_LAUNCH_LDS((gqa_v_kernel<4, true, 4, false, 4, 4>), grid, dim3(256), lds, s, a); return;
            case 5: KIVI_LAUNCH_LDS((gqa_v_kernel<4, true, 4, false, 4, 5>), grid, dim3(256), lds, s, a); return;
            case 6: KIVI_LAUNCH_LDS((gqa_v_kernel<4, true, 4, false, 4, 6>), grid, dim3(256), lds, s, a); return;
            case 7: KIVI_LAUNCH_LDS((gqa_v_kernel<4, true, 4, false, 4, 7>), grid, dim3(256), lds, s, a); return;
            default: break;
        }
    }
    if (occ && atoi(occ) == 2 && HILO && RING == 4) {
        KIVI_LAUNCH_LDS((gqa_v_kernel<R, true, 4, false, 2>), dim3((unsigned)(units * a.S + a.win_blocks)), dim3(256), 4 * 2048 * 4 + (R > 4 ? 8192 : 0), s, a);
        return;
    }
    if (a.dbg && HILO && RING <= 4) {
        KIVI_LAUNCH_LDS((gqa_v_kernel<R, true, RING, true>), dim3((unsigned)(units * a.S + a.win_blocks)), dim3(256), 4 * 2048 * 4 + (R > 4 ? 8192 : 0), s, a);
        return;
    }
    KIVI_LAUNCH_LDS((gqa_v_kernel<R, HILO, RING>), dim3((unsigned)(units * a.S + a.win_blocks)), dim3(256), 4 * 2048 * 4 + (R > 4 ? 8192 : 0), s, a);
}
#endif

extern "C" int kivi_gqa_decode(const kivi_gqa_decode_args* p, kivi_stream_t stream) {
    KIVI_REQUIRE(p != nullptr, KIVI_EINVAL, "kivi_gqa_decode: null arguments");
    const int B = p->B, nh = p->nh, nh_kv = p->nh_kv, D = p->D, group_size = p->group_size, bits = p->bits;
    const int64_t T = p->Tq > p->Tv ? p->Tq : p->Tv;
    KIVI_MF_SHAPE_CHECK("kivi_gqa_decode");
    KIVI_REQUIRE(nh > 0 && nh % nh_kv == 0 && (nh / nh_kv == 1 || nh / nh_kv == 4 || nh / nh_kv == 8), KIVI_EUNSUPPORTED,
                 "kivi_gqa_decode: nh / nh_kv must be 1, 4 or 8 (got %d / %d)", nh, nh_kv);
    const int R = nh / nh_kv;
    const int units = B * nh_kv;
    KIVI_REQUIRE(p->Tq >= 0 && p->Tq % 32 == 0 && p->Tv >= 0 && p->k_res_len >= 0 && p->k_res_len <= 128 && p->v_res_len >= 0 &&
                     p->v_res_len <= 128 && p->Tq + p->k_res_len == p->Tv + p->v_res_len,
                 KIVI_EINVAL, "kivi_gqa_decode: inconsistent lengths (Tq=%lld k_res=%d Tv=%lld v_res=%d)", (long long)p->Tq,
                 p->k_res_len, (long long)p->Tv, p->v_res_len);
    const int64_t n = p->Tq + p->k_res_len + 1;
    // what the step WRITES: the K append (row k_res_len of kres), the V append (row v_win_start + v_res_len of vres), the
    // slot of the token leaving the window (token Tv of the VT store)
    KIVI_REQUIRE(p->residual_length > 0 && p->residual_length <= 128 && p->k_res_len < p->residual_length &&
                     p->v_res_len <= p->residual_length,
                 KIVI_EINVAL, "kivi_gqa_decode: residual of %d keys / window of %d values do not fit residual_length %d",
                 p->k_res_len, p->v_res_len, p->residual_length);
    const bool win_ring = (p->flags & KIVI_GQA_WINDOW_RING) != 0;
    KIVI_REQUIRE(!win_ring || nh / nh_kv != 8, KIVI_EUNSUPPORTED, "kivi_gqa_decode: the ring window needs the round-3 kernels (nh / nh_kv in {1, 4})");
    KIVI_REQUIRE(p->v_win_start >= 0 &&
                     (win_ring ? (p->v_win_start < p->v_window_rows && p->v_res_len + 1 <= p->v_window_rows)
                           : (int64_t)p->v_win_start + p->v_res_len + 1 <= p->v_window_rows),
                 KIVI_EINVAL, "kivi_gqa_decode: window rows [%d, %d] exceed the %lld rows of the buffer", p->v_win_start,
                 p->v_win_start + p->v_res_len, (long long)p->v_window_rows);
    KIVI_REQUIRE(p->Tq <= p->kt_superblocks * KIVI_MF_SB_TOKENS && p->Tv + (p->v_flush ? 1 : 0) <= p->vt_superblocks * KIVI_MF_SB_TOKENS,
                 KIVI_EINVAL, "kivi_gqa_decode: Tq=%lld / Tv=%lld exceed the stores (%lld / %lld super-blocks)", (long long)p->Tq,
                 (long long)p->Tv, (long long)p->kt_superblocks, (long long)p->vt_superblocks);
    KIVI_REQUIRE(!p->v_flush || p->v_res_len == p->residual_length, KIVI_EINVAL,
                 "kivi_gqa_decode: v_flush with a window of %d values (residual_length %d)", p->v_res_len, p->residual_length);
    KIVI_REQUIRE(mf_store_ok(p->kt, p->kt_sb, p->kt_sh, p->kt_ss) && mf_store_ok(p->vt, p->vt_sb, p->vt_sh, p->vt_ss), KIVI_EALIGN,
                 "kivi_gqa_decode: cache storage must be 16-byte aligned super-blocks");
    KIVI_REQUIRE(p->q && (uintptr_t)p->q % 16 == 0 && p->q_sb % 8 == 0 && p->q_sh % 8 == 0, KIVI_EALIGN,
                 "kivi_gqa_decode: q rows must be 16-byte aligned");
    KIVI_REQUIRE(p->kres && p->knew && (uintptr_t)p->kres % 16 == 0 && (uintptr_t)p->knew % 16 == 0 && p->kres_sb % 8 == 0 &&
                     p->kres_sh % 8 == 0 && p->kres_st % 8 == 0 && p->knew_sb % 8 == 0 && p->knew_sh % 8 == 0,
                 KIVI_EALIGN, "kivi_gqa_decode: key rows must be 16-byte aligned");
    KIVI_REQUIRE(p->vres && p->vnew && (uintptr_t)p->vres % 16 == 0 && (uintptr_t)p->vnew % 16 == 0 && p->vres_sb % 8 == 0 &&
                     p->vres_sh % 8 == 0 && p->vres_st % 8 == 0 && p->vnew_sb % 8 == 0 && p->vnew_sh % 8 == 0,
                 KIVI_EALIGN, "kivi_gqa_decode: value rows must be 16-byte aligned");
    KIVI_REQUIRE(p->scores && (uintptr_t)p->scores % 16 == 0 && p->s_sb % 8 == 0 && p->s_sh % 8 == 0 && p->s_sh >= ((n + 7) & ~(int64_t)7),
                 KIVI_EALIGN, "kivi_gqa_decode: score rows must be 16-byte aligned and hold %lld scores", (long long)n);
    KIVI_REQUIRE((int64_t)(R - 1) * p->s_sh * 2 + n * 2 + 16 < ((int64_t)1 << 32), KIVI_EINVAL, "kivi_gqa_decode: score rows too long");
    KIVI_REQUIRE(p->out != nullptr, KIVI_EINVAL, "kivi_gqa_decode: null output");
    const int nsbk = (int)((p->Tq + KIVI_MF_SB_TOKENS - 1) / KIVI_MF_SB_TOKENS);
    const int nseg = nsbk + KIVI_GQA_RES_SEGS;
    KIVI_REQUIRE(p->stats && (uintptr_t)p->stats % 8 == 0 && p->stats_bytes >= (int64_t)B * nh * nseg * 2 * (int64_t)sizeof(float), KIVI_EINVAL,
                 "kivi_gqa_decode: statistics buffer too small (%lld bytes for %d segments)", (long long)p->stats_bytes, nseg);
    const int nsbv = (int)((p->Tv + KIVI_MF_SB_TOKENS - 1) / KIVI_MF_SB_TOKENS);
    int S, spb;
    gqa_v_slices(units, nsbv, R, S, spb);
    KIVI_REQUIRE((int64_t)(nsbv > nsbk ? nsbv : nsbk) * (p->kt_ss > p->vt_ss ? p->kt_ss : p->vt_ss) * 4 <= (int64_t)MF_DESC_LIMIT, KIVI_EINVAL,
                 "kivi_gqa_decode: store too large for one descriptor");
    KIVI_REQUIRE(p->kt_range && p->vt_range && (uintptr_t)p->kt_range % 4 == 0 && (uintptr_t)p->vt_range % 4 == 0, KIVI_EINVAL,
                 "kivi_gqa_decode: null / misaligned range flags");
    KIVI_REQUIRE(units <= KIVI_GQA_WS_COUNTERS, KIVI_EUNSUPPORTED, "kivi_gqa_decode: more than %d (batch row, kv head) units", KIVI_GQA_WS_COUNTERS);
    static const char* wt = KIVI_TUNE_ENV("KIVI_GQA_WIN_TAIL");         // tuning aid: 0 = window shares inside the stream blocks
    const int win_blocks = (nsbv > 0 && !(wt && atoi(wt) == 0)) ? units : 0;
    const int nslot = S + (win_blocks ? 1 : 0);
    const int64_t need = (int64_t)KIVI_GQA_WS_COUNTERS * 4 + (int64_t)units * nslot * 2 * R * 128 * 4;
    KIVI_REQUIRE(p->workspace && (uintptr_t)p->workspace % 16 == 0 && p->workspace_bytes >= need, KIVI_EINVAL,
                 "kivi_gqa_decode: workspace too small (%lld bytes needed)", (long long)need);
    hipStream_t s = (hipStream_t)stream;

    GqaKArgs k;
    k.q = (const uint16_t*)p->q; k.q_sb = p->q_sb; k.q_sh = p->q_sh;
    k.kt = {(uint32_t*)p->kt, p->kt_sb, p->kt_sh, p->kt_ss};
    k.out = (uint16_t*)p->scores; k.out_sb = p->s_sb; k.out_sh = p->s_sh;
    k.nh_kv = nh_kv; k.ratio = R; k.nh = nh; k.Tq = p->Tq; k.nsb = nsbk; k.sb_blocks = 0;
    k.stats = (float*)p->stats; k.nseg = nseg; k.inv_scale = p->inv_scale;
    k.mask = (const uint16_t*)p->mask; k.mask_sb = p->mask_sb;
    k.res_blocks = units * KIVI_GQA_RES_SEGS;
    static const char* rf = KIVI_TUNE_ENV("KIVI_GQA_RES_FIRST");        // tuning aid
    k.res_first = rf ? atoi(rf) : 0;
    k.kres = (uint16_t*)p->kres; k.kres_sb = p->kres_sb; k.kres_sh = p->kres_sh; k.kres_st = p->kres_st;
    k.knew = (const uint16_t*)p->knew; k.knew_sb = p->knew_sb; k.knew_sh = p->knew_sh; k.res_len = p->k_res_len;
    k.range = (const int*)p->kt_range;
    static const char* skipk = KIVI_TUNE_ENV("KIVI_GQA_SKIP_K");       // diagnostic (tools/mf_stage_error.py): the caller filled scores / stats
    static const char* timev = KIVI_TUNE_ENV("KIVI_GQA_TIME_V");       // tuning aid: a pending event pair brackets the sV launch instead
    KiviLaunchEvents held = {nullptr, nullptr};
    if (timev) held = kivi_take_launch_events();
    const bool newp = mf_new_path(R);
    GqaVArgs v;
    memset(&v, 0, sizeof(v));
    v.x = (const uint16_t*)p->scores; v.x_sb = p->s_sb; v.x_sh = p->s_sh;
    v.stats = (const float*)p->stats; v.nseg = nseg;
    v.vt = {(uint32_t*)p->vt, p->vt_sb, p->vt_sh, p->vt_ss};
    v.nh_kv = nh_kv; v.ratio = R; v.nh = nh; v.Tv = p->Tv; v.nsb = nsbv; v.S = S; v.spb = spb;
    v.units = units; v.win_blocks = win_blocks; v.nslot = nslot;
    v.vres = (uint16_t*)p->vres; v.vres_sb = p->vres_sb; v.vres_sh = p->vres_sh; v.vres_st = p->vres_st;
    v.win_start = p->v_win_start; v.res_len = p->v_res_len;
    v.win_rows = win_ring ? (int)p->v_window_rows : 0;
    v.vnew = (const uint16_t*)p->vnew; v.vnew_sb = p->vnew_sb; v.vnew_sh = p->vnew_sh; v.flush = p->v_flush ? 1 : 0;
    v.out = (uint16_t*)p->out; v.out_sb = p->out_sb; v.out_sh = p->out_sh;
    v.dbg = kivi_debug_stamps();
    v.counters = (int*)p->workspace;
    v.ws = (float*)((char*)p->workspace + (size_t)KIVI_GQA_WS_COUNTERS * 4);
    v.range = (int*)p->vt_range;
    if (newp && ((R == 1 && n <= 8192) || (R == 4 && n <= 9216))) {
        // rows that fit the LDS: the whole step of a (batch row, kv head) in one launch (nh == nh_kv: 4 blocks of 4 waves per CU;
        // nh / nh_kv == 4: the four score rows of a unit in one block, 2 blocks of 8 waves per CU)
        static const char* norow = KIVI_TUNE_ENV("KIVI_MF_NO_ROW");     // tuning aid: keep the two-launch form
        const bool split = (p->flags & KIVI_GQA_FORCE_SPLIT) || (norow && atoi(norow));
        // too few units: the split two-launch form fills the chip better -- unless the rows are short enough for the eight waves
        // of a row block to take one super-block each (nh == nh_kv, <= 4096 packed keys): then one launch beats two whatever
        // the batch (32-160 rows: 0.64-0.68 ms per 32-layer step against 0.68-0.86; at 8000 keys 1.02 against 0.79,
        // profiles/r03_other_shapes.log)
        const int min_units = R == 1 ? 192 : 128;
        if (!split && (units >= min_units || (R == 1 && nsbk <= 8) || (p->flags & KIVI_GQA_FORCE_ROW))) return kivi_mf_run_row(&k, &v, units, (p->flags & KIVI_GQA_DUMP_SCORES) != 0, s);
    }
    int rc = skipk ? 0 : (newp ? kivi_mf_run_k(&k, units, s) : run_gqa_k(k, units, s));
    if (rc) return rc;
    if (timev) kivi_set_launch_events(held.start, held.stop);
    if (newp) return kivi_mf_run_v(&v, 0, s);

#ifdef KIVI_TUNING
    static const char* nohilo = KIVI_TUNE_ENV("KIVI_GQA_NO_HILO");
    static const char* fr = KIVI_TUNE_ENV("KIVI_GQA_V_RING");            // tuning aid: blocks in flight (2, 4 or 8)
    const int ring = fr ? atoi(fr) : (R == 4 ? 4 : 2);           // R = 8 spills at 4
#define KIVI_GV(RR, HL)                                          \
    do {                                                         \
        if (ring == 2) launch_gqa_v<RR, HL, 2>(v, units, s);     \
        else if (ring == 8) launch_gqa_v<RR, HL, 8>(v, units, s); \
        else launch_gqa_v<RR, HL, 4>(v, units, s);               \
    } while (0)
    if (R == 4) { if (nohilo) KIVI_GV(4, false); else KIVI_GV(4, true); }
    else { if (nohilo) KIVI_GV(8, false); else KIVI_GV(8, true); }
#undef KIVI_GV
#else
    KIVI_REQUIRE(R == 8, KIVI_EUNSUPPORTED, "kivi_gqa_decode: nh / nh_kv = %d has no round-2 kernel in this build", R);
    KIVI_LAUNCH_LDS((gqa_v_kernel<8, true, 2>), dim3((unsigned)(units * v.S + v.win_blocks)), dim3(256), 4 * 2048 * 4 + 8192, s, v);
#endif
    return kivi_launch_status("gqa_v");
}

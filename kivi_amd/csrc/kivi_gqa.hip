// The KT / VT cache layouts of the matrix-pipe decode kernels (kivi_mfma_layout.h; gfx950): packers (per-channel K and per-token
// V quantisation straight into the layout: models/llama_kivi.py:436 / :441-448 / :343-356), the bit-exact relayouts to and from
// the hook-state tensors (:454-455), and the C ABI of the packed products and of the layer step over them -- kivi_gqa_scores
// (cuda_bmm_fA_qB_outer at :324), kivi_gqa_output (:382), kivi_gqa_decode (:314-399; models/mistral_kivi.py:381-445; head
// mapping quant/csrc/gemv_cuda.cu:361-365) for nh / nh_kv in {1, 4, 8}.  The kernels behind them live in kivi_mf.hip /
// kivi_mf_dev.h (round 3; the round-2 kernels that used to be here served nh / nh_kv = 8 until round 4 and are gone).
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "kivi_common.h"
#include "kivi_gqa_dev.h"
#include "kivi_quant.h"
#include "kivi_gqa_roles.h"


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
    // range marks of the unit (kivi_mfma_layout.h): sticky, every writer stores the same bytes
    mf_range_mark(range + unit, scale2 & 0xFFFFu, scale2 >> 16);
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
    mf_range_mark(range + unit, scale2 & 0xFFFFu, scale2 >> 16);          // range marks (kt_pack_kernel)
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
        mf_range_mark(range + unit, sc);                         // range marks of the unit (kivi_mfma_layout.h)
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
            mf_range_mark(range + unit, sc);                     // range marks of the unit (kivi_mfma_layout.h)
            vm[vt_half(tt, wi)] = (t < T) ? mn[b * sm_sb + hk * sm_sh + t * sm_sr + wi] : (uint16_t)0;
        }
    }
}

// ---- 4-bit codes (KT4 / VT4, kivi_mfma_layout.h): the same four jobs.  Packers: the packed 16-bit reciprocal quantiser of the
// hook-layout 4-bit packers (pk16_pair_quantN, kivi_quant.h).  Relayouts: one block per 32-token block, whole words, 4 (K) or 2
// (V) per thread.
// word `wi` of a block -> (tile, n, kb, c)
struct Mf4Word { int tile, n, kb, c; };
__device__ __forceinline__ Mf4Word mf4_word_of(int wi) {
    const int l = (wi & 255) >> 2;
    return {wi >> 8, l & 15, l >> 4, wi & 3};
}

// OR across the four lanes of a quad as a reduce-scatter: v[j] (32 per lane, already shifted to the lane's bit positions) -> out[jj] =
// OR over the quad of v[8 (lane & 3) + jj].  Two exchange steps (xor 1 on bit 3 of j, xor 2 on bit 4), each lane keeping the half it
// will own and handing the other half to its partner: 72 instructions for the 32 words of a quad, every lane ending with the 8 words it
// stores -- the all-lanes form (two full ORs per word + a store predicated on the owner) was 128 + 32 exec-mask regions.
__device__ __forceinline__ void quad_or_scatter32(const uint32_t (&v)[32], uint32_t (&out)[8]) {
    const bool b0 = threadIdx.x & 1, b1 = threadIdx.x & 2;
    uint32_t a[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int jlo = (q & 7) | ((q >> 3) << 4), jhi = jlo | 8;
        uint32_t lo = v[jlo], hi = v[jhi];
        asm("" : "+v"(lo), "+v"(hi));                      // (values, not addresses: hipcc otherwise selects the INDEX and walks a 32-way chain)
        const uint32_t keep = b0 ? hi : lo, send = b0 ? lo : hi;
        a[q] = keep | dpp_or<0xB1>(send);
    }
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        uint32_t lo = a[jj], hi = a[jj + 8];
        asm("" : "+v"(lo), "+v"(hi));
        const uint32_t keep = b1 ? hi : lo, send = b1 ? lo : hi;
        out[jj] = keep | dpp_or<0x4E>(send);
    }
}

// Round 5: the structure of the 2-bit packers (four waves per workgroup, one 32-token block each; the 32 x 128 tile comes in through
// LDS with 16-byte loads, packed 16-bit statistics and quantiser on the lane's two channels / tokens at once, the block's 512 code
// words meet in LDS and leave as two 1 KiB stores) -- the round-4 kernels (one 128-thread block per 32 tokens, scalar keys, codes
// as bytes through LDS) ran at 0.59 / 0.49 of the HBM roofline against 0.78 / 0.72 for the 2-bit ones.
__global__ __launch_bounds__(256) void kt_pack4_kernel(const uint16_t* k, int64_t k_sb, int64_t k_sh, int64_t k_st, MfStore st,
                                                       int* range, int64_t blk0, int nblk, int nh_kv, int64_t ntile) {
    const int wave = threadIdx.x >> 6;
    int64_t tile_id = (int64_t)blockIdx.x * 4 + wave;
    const bool live_tile = tile_id < ntile;
    if (!live_tile) tile_id = ntile - 1;                   // (its stores are skipped)
    const int unit = (int)(tile_id / nblk), bi = (int)(tile_id - (int64_t)unit * nblk);
    const int b = unit / nh_kv, hk = unit - b * nh_kv;
    const int lane = threadIdx.x & 63;
    constexpr int PITCH = 68;
    __shared__ uint32_t stage[4][32 * PITCH];              // per wave: the fp16 tile, then (reused) the block's 512 code words
    uint32_t* stw = stage[wave];
    {
        const uint16_t* tb = k + b * k_sb + hk * k_sh + (int64_t)bi * 32 * k_st;
        u32x4 in[8];
#pragma unroll
        for (int j = 0; j < 8; j++) in[j] = __builtin_nontemporal_load((const u32x4*)(tb + (int64_t)(4 * j + (lane >> 4)) * k_st + 8 * (lane & 15)));
#pragma unroll
        for (int j = 0; j < 8; j++) *(u32x4*)(stw + (4 * j + (lane >> 4)) * PITCH + 4 * (lane & 15)) = in[j];
    }
    __builtin_amdgcn_wave_barrier();                       // (every region of `stage` belongs to one wave)
    uint32_t x[32];
#pragma unroll
    for (int t = 0; t < 32; t++) x[t] = stw[t * PITCH + lane];      // the lane's channel pair (2 l, 2 l + 1) down the 32 tokens
    uint32_t cq[32], scale2, mn2;
    pk16_pair_quantN<32, 4>(x, cq, scale2, mn2);
    __builtin_amdgcn_wave_barrier();                       // the tile has been read: its memory takes the code words
    // word (tile, n, kb, c) = the 8 channels 32 c + 8 kb + e of token n + 16 tile, element e at bits 4 (e >> 1) + 16 (e & 1): this lane
    // holds e = 2 i, 2 i + 1 (i = l & 3) in the halves of cq[t]; the four lanes of a quad complete a word
    const int i = lane & 3;
    const int c = lane >> 4, kb = (lane >> 2) & 3;
#pragma unroll
    for (int t = 0; t < 32; t++) cq[t] <<= 4 * i;
    uint32_t own[8];                                       // the words of tokens t = 8 i + jj
    quad_or_scatter32(cq, own);
    {
        uint32_t* dst = stw + (i >> 1) * 256 + (8 * (i & 1) + 16 * kb) * 4 + c;      // (t >> 4) * 256 + ((t & 15) + 16 kb) * 4 + c
#pragma unroll
        for (int jj = 0; jj < 8; jj++) dst[4 * jj] = own[jj];
    }
    __builtin_amdgcn_wave_barrier();
    if (!live_tile) return;
    const int64_t blk = blk0 + bi;
    uint32_t* sb = mf_sb(st, b, hk, blk >> 4);
    uint32_t* cw = sb + (blk & 15) * KIVI_MF4_BLOCK_WORDS;
    *(u32x4*)(cw + lane * 4) = *(const u32x4*)(stw + lane * 4);
    *(u32x4*)(cw + 256 + lane * 4) = *(const u32x4*)(stw + 256 + lane * 4);
    const int hidx = kt_sm_half((int)(blk & 15), 2 * lane);    // channel 2 l (even): the pair (2 l, 2 l + 1) is one word
    (sb + KIVI_MF4_SB_SCALE_WORD0)[hidx >> 1] = scale2;
    (sb + KIVI_MF4_SB_MN_WORD0)[hidx >> 1] = mn2;
    mf_range_mark(range + unit, scale2 & 0xFFFFu, scale2 >> 16);          // range marks of the unit (kt_pack_kernel)
}

// Lane (kb, c, ee) = 16 kb + 4 c + ee owns the token PAIR (8 kb + 2 ee, + 1) of channel group c (cf. vt_pack_kernel): even token in
// the low halves, odd token in the high halves of x[j], j = channel inside the group.  Tokens at or past T read as zeros.
__global__ __launch_bounds__(256) void vt_pack4_kernel(const uint16_t* v, int64_t v_sb, int64_t v_sh, int64_t v_st, MfStore st,
                                                       int* range, int64_t T, int nblk, int nh_kv, int64_t ntile) {
    const int wave = threadIdx.x >> 6;
    int64_t tile_id = (int64_t)blockIdx.x * 4 + wave;
    const bool live_tile = tile_id < ntile;
    if (!live_tile) tile_id = ntile - 1;
    const int unit = (int)(tile_id / nblk), bi = (int)(tile_id - (int64_t)unit * nblk);
    const int b = unit / nh_kv, hk = unit - b * nh_kv;
    const int lane = threadIdx.x & 63;
    const int kb = lane >> 4, c = (lane >> 2) & 3, ee = lane & 3;
    constexpr int PITCH = 68;                              // words per staged row (256 bytes + 16: rows start 4 banks apart)
    __shared__ uint32_t stage[4][32 * PITCH];              // per wave: the fp16 tile, then (reused) the block's 512 code words
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
    __builtin_amdgcn_wave_barrier();
    u32x4 ra[4], rb[4];                                   // 32 channels of the even / odd token
    {
        const uint32_t* pa = stw + (8 * kb + 2 * ee) * PITCH + 16 * c;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ra[q] = *(const u32x4*)(pa + 4 * q);
            rb[q] = *(const u32x4*)(pa + PITCH + 4 * q);
        }
    }
    uint32_t x[32];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const uint32_t a = ra[q][kk], bq = rb[q][kk];   // channels 8 q + 2 kk, + 1
            x[8 * q + 2 * kk] = __builtin_amdgcn_perm(bq, a, 0x05040100u);       // (a.lo, b.lo)
            x[8 * q + 2 * kk + 1] = __builtin_amdgcn_perm(bq, a, 0x07060302u);   // (a.hi, b.hi)
        }
    uint32_t cq[32], scale2, mn2;
    pk16_pair_quantN<32, 4>(x, cq, scale2, mn2);
    __builtin_amdgcn_wave_barrier();                       // the tile has been read: its memory takes the code words
    // word (tile, n, kb, c) = the 8 tokens 8 kb + e of channel 32 c + 16 tile + n, element e at bits 4 (e >> 1) + 16 (e & 1): this lane
    // holds e = 2 ee, 2 ee + 1 in the halves of cq[j], j = 16 tile + n; the four lanes of a quad (ee) complete a word
#pragma unroll
    for (int j = 0; j < 32; j++) cq[j] <<= 4 * ee;
    uint32_t own[8];                                       // the words j = 8 ee + jj
    quad_or_scatter32(cq, own);
    {
        uint32_t* dst = stw + (ee >> 1) * 256 + (8 * (ee & 1) + 16 * kb) * 4 + c;    // (j >> 4) * 256 + ((j & 15) + 16 kb) * 4 + c
#pragma unroll
        for (int jj = 0; jj < 8; jj++) dst[4 * jj] = own[jj];
    }
    __builtin_amdgcn_wave_barrier();
    if (!live_tile) return;
    uint32_t* sb = mf_sb(st, b, hk, bi >> 4);
    uint32_t* cw = sb + (bi & 15) * KIVI_MF4_BLOCK_WORDS;
    *(u32x4*)(cw + lane * 4) = *(const u32x4*)(stw + lane * 4);
    *(u32x4*)(cw + 256 + lane * 4) = *(const u32x4*)(stw + 256 + lane * 4);
    (sb + KIVI_MF4_SB_SCALE_WORD0 + (bi & 15) * 64)[lane] = scale2;   // vt_half(8 kb + 2 ee, c) / 2 == lane
    (sb + KIVI_MF4_SB_MN_WORD0 + (bi & 15) * 64)[lane] = mn2;
    mf_range_mark(range + unit, scale2 & 0xFFFFu, scale2 >> 16);
}

// KT4 <-> K_code_T (B, nh_kv, D, T/8), K_scale_T / K_mn_T (B, nh_kv, D, T/32): 4 reference words per (channel, block)
template <bool TO_REF>
__global__ __launch_bounds__(128) void kt_relayout4_kernel(MfStore st, uint32_t* code, int64_t code_sb, int64_t code_sh,
                                                           int64_t code_sr, uint16_t* scale, uint16_t* mn, int64_t sm_sb,
                                                           int64_t sm_sh, int64_t sm_sr, int nblk, int nh_kv, int* range) {
    __shared__ uint32_t lds[512];
    const int unit = blockIdx.x / nblk, blk = blockIdx.x - unit * nblk;
    const int b = unit / nh_kv, hk = unit - b * nh_kv;
    const int d = threadIdx.x;
    uint32_t* sb = mf_sb(st, b, hk, blk >> 4);
    uint32_t* cw = sb + (blk & 15) * KIVI_MF4_BLOCK_WORDS;
    uint16_t* ks = (uint16_t*)(sb + KIVI_MF4_SB_SCALE_WORD0);
    uint16_t* km = (uint16_t*)(sb + KIVI_MF4_SB_MN_WORD0);
    const int gsb = blk & 15;
    uint32_t* cref = code + b * code_sb + hk * code_sh + (int64_t)d * code_sr + (int64_t)blk * 4;
    const int64_t sidx = b * sm_sb + hk * sm_sh + (int64_t)d * sm_sr + blk;
    if constexpr (TO_REF) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) lds[d + 128 * rep] = cw[d + 128 * rep];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t w = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) w |= ((lds[kt4_word(8 * j + i, d)] >> kt4_bit(d)) & 15u) << (4 * i);
            cref[j] = w;
        }
        scale[sidx] = ks[kt_sm_half(gsb, d)];
        mn[sidx] = km[kt_sm_half(gsb, d)];
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) lds[4 * d + j] = cref[j];
        __syncthreads();
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
            const int wi = d + 128 * rep;
            const Mf4Word q = mf4_word_of(wi);
            const int tt = q.n + 16 * q.tile;
            uint32_t w = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int ch = 32 * q.c + 8 * q.kb + e;
                w |= ((lds[4 * ch + (tt >> 3)] >> (4 * (tt & 7))) & 15u) << kt4_bit(ch);
            }
            cw[wi] = w;
        }
        const uint16_t sc = scale[sidx];
        ks[kt_sm_half(gsb, d)] = sc;
        km[kt_sm_half(gsb, d)] = mn[sidx];
        mf_range_mark(range + unit, sc);
    }
}

// VT4 <-> V_code (B, nh_kv, T, D/8), V_scale / V_mn (B, nh_kv, T, D/32); tokens [0, T), the slots of the last block past T are zeros
template <bool TO_REF>
__global__ __launch_bounds__(256) void vt_relayout4_kernel(MfStore st, uint32_t* code, int64_t code_sb, int64_t code_sh,
                                                           int64_t code_sr, uint16_t* scale, uint16_t* mn, int64_t sm_sb,
                                                           int64_t sm_sh, int64_t sm_sr, int64_t T, int nblk, int nh_kv, int* range) {
    __shared__ uint32_t lds[512];
    const int unit = blockIdx.x / nblk, blk = blockIdx.x - unit * nblk;
    const int b = unit / nh_kv, hk = unit - b * nh_kv;
    const int tid = threadIdx.x;
    uint32_t* sb = mf_sb(st, b, hk, blk >> 4);
    uint32_t* cw = sb + (blk & 15) * KIVI_MF4_BLOCK_WORDS;
    uint16_t* vs = (uint16_t*)(sb + KIVI_MF4_SB_SCALE_WORD0) + (blk & 15) * 128;
    uint16_t* vm = (uint16_t*)(sb + KIVI_MF4_SB_MN_WORD0) + (blk & 15) * 128;
    if constexpr (TO_REF) {
        lds[tid] = cw[tid];
        lds[tid + 256] = cw[tid + 256];
        __syncthreads();
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            const int idx = tid + 256 * rep, tt = idx >> 4, rw = idx & 15;    // reference word rw of token tt: channels 8 rw .. + 7
            const int64_t t = (int64_t)blk * 32 + tt;
            if (t < T) {
                uint32_t w = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) w |= ((lds[vt4_word(tt, 8 * rw + j)] >> vt4_bit(tt)) & 15u) << (4 * j);
                code[b * code_sb + hk * code_sh + t * code_sr + rw] = w;
            }
        }
        if (tid < 128) {
            const int tt = tid >> 2, g = tid & 3;
            const int64_t t = (int64_t)blk * 32 + tt;
            if (t < T) {
                scale[b * sm_sb + hk * sm_sh + t * sm_sr + g] = vs[vt_half(tt, g)];
                mn[b * sm_sb + hk * sm_sh + t * sm_sr + g] = vm[vt_half(tt, g)];
            }
        }
    } else {
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            const int idx = tid + 256 * rep, tt = idx >> 4, rw = idx & 15;
            const int64_t t = (int64_t)blk * 32 + tt;
            lds[idx] = (t < T) ? code[b * code_sb + hk * code_sh + t * code_sr + rw] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            const int wi = tid + 256 * rep;
            const Mf4Word q = mf4_word_of(wi);
            const int ch = 32 * q.c + 16 * q.tile + q.n;
            uint32_t w = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int t2 = 8 * q.kb + e;
                w |= ((lds[16 * t2 + (ch >> 3)] >> (4 * (ch & 7))) & 15u) << vt4_bit(t2);
            }
            cw[wi] = w;
        }
        if (tid < 128) {
            const int tt = tid >> 2, g = tid & 3;
            const int64_t t = (int64_t)blk * 32 + tt;
            const uint16_t sc = (t < T) ? scale[b * sm_sb + hk * sm_sh + t * sm_sr + g] : (uint16_t)0;
            vs[vt_half(tt, g)] = sc;
            mf_range_mark(range + unit, sc);
            vm[vt_half(tt, g)] = (t < T) ? mn[b * sm_sb + hk * sm_sh + t * sm_sr + g] : (uint16_t)0;
        }
    }
}

// Largest byte extent a buffer descriptor over a unit's store may have: requests past the end of a wave's stream carry the
// per-lane offset MF_DEAD_OFF (kivi_mf_dev.h) and must fall OUTSIDE the descriptor's range (zeros, no memory access)
constexpr uint32_t MF_DESC_LIMIT = 0xFFFE0000u;

bool mf_store_ok(const void* base, int64_t sb_b, int64_t sb_h, int64_t sb_s, int bits = 2) {
    return base && (uintptr_t)base % 16 == 0 && sb_b % 4 == 0 && sb_h % 4 == 0 && sb_s % 4 == 0 &&
           sb_s >= (bits == 4 ? KIVI_MF4_SB_WORDS : KIVI_MF_SB_WORDS);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI

#define KIVI_MF_SHAPE_CHECK(who)                                                                                       \
    KIVI_REQUIRE((bits == 2 || bits == 4) && group_size == 32 && D == 128, KIVI_EUNSUPPORTED,                           \
                 who ": the MFMA cache layout covers 2- and 4-bit codes, group_size 32, head_dim 128 (got %d / %d / %d)", bits, \
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
    KIVI_REQUIRE(mf_store_ok(kt, kt_sb, kt_sh, kt_ss, bits), KIVI_EALIGN, "kivi_kt_pack: cache storage must be 16-byte aligned super-blocks");
    KIVI_REQUIRE(k && (uintptr_t)k % 16 == 0 && k_sb % 8 == 0 && k_sh % 8 == 0 && k_st % 8 == 0, KIVI_EALIGN,
                 "kivi_kt_pack: key rows must be 16-byte aligned");
    if (T == 0) return 0;
    const int nblk = (int)(T / 32);
    const MfStore st = {(uint32_t*)kt, kt_sb, kt_sh, kt_ss};
    const int64_t ntile = (int64_t)B * nh_kv * nblk;
    if (bits == 4) {
        hipLaunchKernelGGL(kt_pack4_kernel, dim3((unsigned)((ntile + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)k, k_sb, k_sh, k_st,
                           st, (int*)kt_range, token_offset / 32, nblk, nh_kv, ntile);
        return kivi_launch_status("kt_pack4");
    }
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
    KIVI_REQUIRE(mf_store_ok(vt, vt_sb, vt_sh, vt_ss, bits), KIVI_EALIGN, "kivi_vt_pack: cache storage must be 16-byte aligned super-blocks");
    KIVI_REQUIRE(v && (uintptr_t)v % 16 == 0 && v_sb % 8 == 0 && v_sh % 8 == 0 && v_st % 8 == 0, KIVI_EALIGN,
                 "kivi_vt_pack: value rows must be 16-byte aligned");
    if (T == 0) return 0;
    const int nblk = (int)((T + 31) / 32);
    const MfStore st = {(uint32_t*)vt, vt_sb, vt_sh, vt_ss};
    const int64_t ntile = (int64_t)B * nh_kv * nblk;
    if (bits == 4) {
        hipLaunchKernelGGL(vt_pack4_kernel, dim3((unsigned)((ntile + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)v, v_sb, v_sh, v_st,
                           st, (int*)vt_range, T, nblk, nh_kv, ntile);
        return kivi_launch_status("vt_pack4");
    }
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
    KIVI_REQUIRE(mf_store_ok(kt, kt_sb, kt_sh, kt_ss, bits) && code && scale && mn, KIVI_EALIGN, "kivi_kt_relayout: bad buffers");
    KIVI_REQUIRE(to_ref || (kt_range && (uintptr_t)kt_range % 4 == 0), KIVI_EINVAL, "kivi_kt_relayout: writing a store needs its range flags");
    if (T == 0) return 0;
    const int nblk = (int)(T / 32);
    const MfStore st = {(uint32_t*)kt, kt_sb, kt_sh, kt_ss};
    const dim3 grid((unsigned)((int64_t)B * nh_kv * nblk));
    if (bits == 4) {
        if (to_ref)
            hipLaunchKernelGGL(kt_relayout4_kernel<true>, grid, dim3(128), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                               code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, nblk, nh_kv, (int*)kt_range);
        else
            hipLaunchKernelGGL(kt_relayout4_kernel<false>, grid, dim3(128), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                               code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, nblk, nh_kv, (int*)kt_range);
        return kivi_launch_status("kt_relayout4");
    }
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
    KIVI_REQUIRE(mf_store_ok(vt, vt_sb, vt_sh, vt_ss, bits) && code && scale && mn, KIVI_EALIGN, "kivi_vt_relayout: bad buffers");
    KIVI_REQUIRE(to_ref || (vt_range && (uintptr_t)vt_range % 4 == 0), KIVI_EINVAL, "kivi_vt_relayout: writing a store needs its range flags");
    if (T == 0) return 0;
    const int nblk = (int)((T + 31) / 32);
    const MfStore st = {(uint32_t*)vt, vt_sb, vt_sh, vt_ss};
    const dim3 grid((unsigned)((int64_t)B * nh_kv * nblk));
    if (bits == 4) {
        if (to_ref)
            hipLaunchKernelGGL(vt_relayout4_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                               code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, T, nblk, nh_kv, (int*)vt_range);
        else
            hipLaunchKernelGGL(vt_relayout4_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                               code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, T, nblk, nh_kv, (int*)vt_range);
        return kivi_launch_status("vt_relayout4");
    }
    if (to_ref)
        hipLaunchKernelGGL(vt_relayout_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                           code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, T, nblk, nh_kv, (int*)vt_range);
    else
        hipLaunchKernelGGL(vt_relayout_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                           code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, T, nblk, nh_kv, (int*)vt_range);
    return kivi_launch_status("vt_relayout");
}

// the kernels (kivi_mf.hip); the argument blocks cross the translation-unit boundary as void*
int kivi_mf_run_k(void* k_args, int units, int bits, hipStream_t s);
int kivi_mf_run_v(const void* v_args, int prob, int bits, hipStream_t s);
int kivi_mf_run_row_sp(const void* p, int64_t p_sb, int64_t p_sh, int B, int nh, int nh_kv, int64_t T, const int* range, int* sp,
                       hipStream_t s);
int kivi_mf_run_row(void* k_args, const void* v_args, int units, int64_t n_rows, int dump, int bits, int S, int res_cap, int slice_kernel, hipStream_t s);

// Launch plan of a step whose LONGEST row has n_rows keys (nsbk super-blocks of packed keys + up to res_cap + 1 fp16 ones):
//   0      two launches (mf_k_kernel -> score rows + statistics in memory -> mf_v_kernel)
//   S >= 1 one launch, every unit's row cut into S slices of whole super-blocks, one block each (nh == nh_kv: mf_row_kernel, S = 1;
//          nh / nh_kv in {4, 8}: mf_row4_kernel -- S > 1 when the R score rows of a whole row do not fit the LDS, or to fill the chip
//          when there are few units; the slices exchange their softmax statistics inside the launch)
// A pure function of the step's geometry class (kivi_mf_step_key) and the call's constants, so eager and replayed steps agree.
// KIVI_GQA_SLICES(n) in the flags forces n slices (tests, tuning).
// Longest row a block can hold.  The plan is made for the longest row of a geometry class, nsbk * 512 packed keys + residual_length
// (<= 128) fp16 ones, so the caps are "whole super-blocks + a full residual": 16 super-blocks for a multi-head row (one fp16 row of
// 8320 scores: 16.3 KiB, four blocks per CU), 18 for nh / nh_kv = 4 (four rows: 73 KiB, two blocks per CU in 160 KiB with ~4 KiB of
// static LDS each), 9 super-blocks less the residual for nh / nh_kv = 8 (eight rows + ~6 KiB static: 4608 keys is what two blocks fit).
constexpr int64_t MF_ROW1_CAP = 8192 + 128, MF_ROW4_CAP = 9216 + 128, MF_ROW8_CAP = 4608;
static int mf_plan(int R, int units, int64_t n_rows, int nsbk, int res_cap, int flags, int bits = 2) {
    static const char* norow = KIVI_TUNE_ENV("KIVI_MF_NO_ROW");     // tuning aid: keep the two-launch form
    if ((flags & KIVI_GQA_FORCE_SPLIT) || (norow && atoi(norow))) return 0;
    if (R == 1) {
        const int f1 = (flags >> 8) & 0xFF;                         // KIVI_GQA_SLICES(n): the sliced form of multi-head rows (tests, tuning)
        if (f1 == 1 && !(flags & KIVI_GQA_FORCE_ROW)) return n_rows <= MF_ROW1_CAP ? 1 : 0;
        if (f1 > 1 && !(flags & KIVI_GQA_FORCE_ROW)) {
            const int spb = (nsbk + f1 - 1) / f1;
            const bool ok = f1 <= nsbk && f1 <= 64 && (int64_t)(spb > 2 ? spb : 2) * KIVI_MF_SB_TOKENS + res_cap + 1 <= 8192 && units <= KIVI_GQA_MAX_SLICED_UNITS;
            return ok ? f1 : 0;
        }
        // Multi-head rows beyond 16 super-blocks (the reference's LongChat-7B-32K runs, docs/long_bench.md:5-26): TWO launches.  Round 6
        // put the sliced one-launch form (mf_row4_kernel<R = 1>: the fewest slices whose score row fits a block, doubled while the grid
        // stays within the 4 blocks per CU) into this plan and measured it against the two launches on one box (profiles/
        // r06_long_rows.log, ms per 32-layer step, sliced / two launches): B = 8 x 32k 6.05 / 5.08, B = 16 x 32k 10.48 / 9.48, B = 8 x 16k
        // 2.96 / 2.87, B = 16 x 16k 5.52 / 5.10, one 32k row 2.30 / 1.11 -- a multi-head row is ONE head, so the slices' statistics
        // exchange and the in-stream softmax are pure overhead (as for the half-row slices of the headline shape, round 5), while the
        // two launches already stream at 0.66-0.71 of the roofline each.  KIVI_GQA_SLICES(n) keeps the sliced form reachable (tests, A/B).
        if (n_rows > MF_ROW1_CAP) return 0;
        // too few units: the split two-launch form fills the chip better -- unless the rows are short enough for the eight waves
        // of a row block to take one super-block each (<= 4096 packed keys): then one launch beats two whatever the batch
        // (32-160 rows: 0.64-0.68 ms per 32-layer step against 0.68-0.86; at 8000 keys 1.02 against 0.79, profiles/r03_other_shapes.log)
        return (units >= 192 || nsbk <= 8 || (flags & KIVI_GQA_FORCE_ROW)) ? 1 : 0;
    }
    if (R != 4 && R != 8) return 0;
    const int64_t cap = R == 4 ? MF_ROW4_CAP : MF_ROW8_CAP;         // keys whose R score rows fit the LDS of a block (mf_row4_kernel)
    auto blk_rows = [&](int S) -> int64_t {                         // the longest row of a block when a row is cut into S slices
        if (S <= 1) return n_rows;
        const int spb = (nsbk + S - 1) / S;
        return (int64_t)(spb > 2 ? spb : 2) * KIVI_MF_SB_TOKENS + res_cap + 1;
    };
    const int forced = (flags & KIVI_GQA_FORCE_ROW) ? 1 : ((flags >> 8) & 0xFF);          // FORCE_ROW: a block per row
    if (forced) return (forced <= (nsbk > 1 ? nsbk : 1) && forced <= 64 && blk_rows(forced) <= cap && (forced == 1 || units <= KIVI_GQA_MAX_SLICED_UNITS)) ? forced : 0;
    const int nsb1 = nsbk > 1 ? nsbk : 1;                          // (Tq = 0 before the first K flush: one, empty, slice)
    int S = 1;
    while (S <= nsb1 && S <= 64 && blk_rows(S) > cap) S++;
    if (S > nsb1 || S > 64) return 0;
    // more, shorter slices while a slice keeps >= 4 super-blocks (one per wave of its block in the K walk): rows that must be cut
    // anyway until the grid fills the 2 blocks per CU that are resident at once; rows that fit only while there are fewer blocks than
    // CUs -- a block that holds a whole row runs the faster phase-softmax flow and pays no exchange (32 / 8 heads, 8k keys, ms per
    // 32-layer step: 32 units 1.84 unsliced, 1.01 in 4 slices, 1.08 in two launches; 256 units 1.97 unsliced, 2.04 in 2 slices, 2.05
    // in two launches -- profiles/r05_forms.log)
    auto can_double = [&](int S_) { return (nsbk + 2 * S_ - 1) / (2 * S_) >= 4 && 2 * S_ <= 64 && blk_rows(2 * S_) <= cap; };
    if (S == 1) { while ((int64_t)units * S < 256 && can_double(S)) S *= 2; }
    else { while ((int64_t)units * S * 2 <= 512 && can_double(S)) S *= 2; }
    if (S > 1 && units > KIVI_GQA_MAX_SLICED_UNITS) S = blk_rows(1) <= cap ? 1 : 0;
    if (S == 1) {
        // (R = 8, 4000 keys: 128 units 1.65 ms per 32-layer step in one launch against 1.39 in two, 512 units 2.42 against 3.65;
        // R = 4 at 4 bits, one launch vs two: 128 units x 8k keys 1.97 vs 1.73, 256 units x 2k 1.02 vs 1.17, 512 units x 2k 1.51 vs 2.05;
        // profiles/r04_other_shapes.log, r04_mf4_config4.log)
        const int min_units = R == 4 ? (bits == 4 ? 192 : 128) : 192;
        return units >= min_units ? 1 : 0;
    }
    return S;
}

static_assert(sizeof(MfStep) == sizeof(kivi_mf_step) && offsetof(MfStep, Tv) == offsetof(kivi_mf_step, Tv) &&
                  offsetof(MfStep, k_res_len) == offsetof(kivi_mf_step, k_res_len) && offsetof(MfStep, v_flush) == offsetof(kivi_mf_step, v_flush),
              "the kernels read a kivi_mf_step through MfStep");

// Geometry class of a decode step (include/kivi_hip.h): launches captured for one step may be replayed for every later step
// with the same key -- the super-block counts of both stores and whether the step flushes a value.  The launch plan (mf_plan) is a
// function of the class and of constants of the call (shape, bits, flags), so it needs no bit of its own.
extern "C" int64_t kivi_mf_step_key(const kivi_mf_step* st, int B, int nh, int nh_kv, int residual_length, int flags) {
    if (!st || nh_kv <= 0 || nh <= 0 || nh % nh_kv) return -1;
    (void)B; (void)residual_length; (void)flags;
    const int64_t nsbk = (st->Tq + KIVI_MF_SB_TOKENS - 1) / KIVI_MF_SB_TOKENS, nsbv = (st->Tv + KIVI_MF_SB_TOKENS - 1) / KIVI_MF_SB_TOKENS;
    return (nsbk << 40) | (nsbv << 16) | ((int64_t)(st->v_flush != 0) << 1);
}

// The plan kivi_gqa_decode follows for a step (include/kivi_hip.h): 0 = two launches, S >= 1 = one launch with S slices per row
// (ABI version 3: eager steps are planned for the longest row of their geometry class too -- `dyn` and `k_res_len` no longer enter --, so
// an eager and a replayed step of the same position always take the same form and agree bit for bit)
extern "C" int kivi_mf_launch_plan(int B, int nh, int nh_kv, int64_t Tq, int k_res_len, int residual_length, int flags, int bits, int dyn) {
    if (B <= 0 || nh_kv <= 0 || nh <= 0 || nh % nh_kv || Tq < 0 || k_res_len < 0 || residual_length <= 0) return -1;
    (void)dyn;
    const int nsbk = (int)((Tq + KIVI_MF_SB_TOKENS - 1) / KIVI_MF_SB_TOKENS);
    return mf_plan(nh / nh_kv, B * nh_kv, (int64_t)nsbk * KIVI_MF_SB_TOKENS + residual_length, nsbk, residual_length, flags, bits);
}

extern "C" int kivi_gqa_scores(const void* q, int64_t q_sb, int64_t q_sh, const void* kt, int64_t kt_sb, int64_t kt_sh,
                               int64_t kt_ss, const void* kt_range, void* out, int64_t out_sb, int64_t out_sh, int B, int nh,
                               int nh_kv, int D, int64_t T, int group_size, int bits, kivi_stream_t stream) {
    KIVI_MF_SHAPE_CHECK("kivi_gqa_scores");
    KIVI_REQUIRE(nh > 0 && nh % nh_kv == 0 && (nh / nh_kv == 1 || nh / nh_kv == 4 || nh / nh_kv == 8), KIVI_EUNSUPPORTED,
                 "kivi_gqa_scores: nh / nh_kv must be 1, 4 or 8 (got %d / %d)", nh, nh_kv);
    KIVI_REQUIRE(bits == 2 || nh / nh_kv == 4 || nh == nh_kv, KIVI_EUNSUPPORTED, "kivi_gqa_scores: 4-bit codes on the matrix pipe need nh / nh_kv in {1, 4} (got %d / %d)", nh, nh_kv);
    KIVI_REQUIRE(T >= 0 && T % 32 == 0, KIVI_EINVAL, "kivi_gqa_scores: T=%lld must be a multiple of 32", (long long)T);
    KIVI_REQUIRE(mf_store_ok(kt, kt_sb, kt_sh, kt_ss, bits), KIVI_EALIGN, "kivi_gqa_scores: cache storage must be 16-byte aligned super-blocks");
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
    a.res_blocks = 0; a.kres = nullptr; a.knew = nullptr; a.res_len = 0;
    a.kres_sb = a.kres_sh = a.kres_st = a.knew_sb = a.knew_sh = 0;
    a.range = (const int*)kt_range;
    a.dyn = nullptr;
    a.dump = 0; a.xcount = nullptr; a.ticket = nullptr; a.err_ws = nullptr; a.err_host = nullptr;
    return kivi_mf_run_k(&a, B * nh_kv, bits, (hipStream_t)stream);
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
    KIVI_REQUIRE(nh > 0 && nh % nh_kv == 0 && (nh / nh_kv == 1 || nh / nh_kv == 4 || nh / nh_kv == 8), KIVI_EUNSUPPORTED,
                 "kivi_gqa_output: nh / nh_kv must be 1, 4 or 8 (got %d / %d)", nh, nh_kv);
    KIVI_REQUIRE(bits == 2 || nh / nh_kv == 4 || nh == nh_kv, KIVI_EUNSUPPORTED, "kivi_gqa_output: 4-bit codes on the matrix pipe need nh / nh_kv in {1, 4} (got %d / %d)", nh, nh_kv);
    KIVI_REQUIRE(T >= 0, KIVI_EINVAL, "kivi_gqa_output: negative length");
    KIVI_REQUIRE(mf_store_ok(vt, vt_sb, vt_sh, vt_ss, bits), KIVI_EALIGN, "kivi_gqa_output: cache storage must be 16-byte aligned super-blocks");
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
    return kivi_mf_run_v(&v, 1, bits, s);
}

extern "C" int kivi_gqa_decode(const kivi_gqa_decode_args* p, kivi_stream_t stream) {
    KIVI_REQUIRE(p != nullptr, KIVI_EINVAL, "kivi_gqa_decode: null arguments");
    {   // the sticky device-side error of an EARLIER sliced launch (include/kivi_hip.h, kivi_device_error): reported once, here, before
        // anything of this step is enqueued; the counters of that launch are back at zero, so the caller may simply call again
        int eu = -1;
        if (kivi_take_device_error(&eu)) {
            kivi_set_error("kivi_gqa_decode: a block of an earlier sliced decode launch gave up waiting for a partner block of (batch row, kv head) "
                           "unit %d: that step's output holds NaN for the unit; nothing was enqueued for this call", eu);
            return KIVI_ETIMEOUT;
        }
    }
    const int B = p->B, nh = p->nh, nh_kv = p->nh_kv, D = p->D, group_size = p->group_size, bits = p->bits;
    const int64_t T = p->Tq > p->Tv ? p->Tq : p->Tv;
    KIVI_MF_SHAPE_CHECK("kivi_gqa_decode");
    KIVI_REQUIRE(nh > 0 && nh % nh_kv == 0 && (nh / nh_kv == 1 || nh / nh_kv == 4 || nh / nh_kv == 8), KIVI_EUNSUPPORTED,
                 "kivi_gqa_decode: nh / nh_kv must be 1, 4 or 8 (got %d / %d)", nh, nh_kv);
    KIVI_REQUIRE(bits == 2 || nh / nh_kv == 4 || nh == nh_kv, KIVI_EUNSUPPORTED, "kivi_gqa_decode: 4-bit codes on the matrix pipe need nh / nh_kv in {1, 4} (got %d / %d)", nh, nh_kv);
    const int R = nh / nh_kv;
    const int units = B * nh_kv;
    KIVI_REQUIRE(p->Tq >= 0 && p->Tq % 32 == 0 && p->Tv >= 0 && p->k_res_len >= 0 && p->k_res_len <= 128 && p->v_res_len >= 0 &&
                     p->v_res_len <= 128 && p->Tq + p->k_res_len == p->Tv + p->v_res_len,
                 KIVI_EINVAL, "kivi_gqa_decode: inconsistent lengths (Tq=%lld k_res=%d Tv=%lld v_res=%d)", (long long)p->Tq,
                 p->k_res_len, (long long)p->Tv, p->v_res_len);
    const int64_t n = p->Tq + p->k_res_len + 1;
    // device-resident lengths (dyn_step): the launch geometry is sized for EVERY step with the same super-block counts
    // (kivi_mf_step_key), whose longest row has ceil(Tq / 512) * 512 + residual_length keys
    const bool dyn = p->dyn_step != nullptr;
    KIVI_REQUIRE(!dyn || (uintptr_t)p->dyn_step % 8 == 0, KIVI_EALIGN, "kivi_gqa_decode: dyn_step must be 8-byte aligned");
    const int64_t n_rows = dyn ? (p->Tq + KIVI_MF_SB_TOKENS - 1) / KIVI_MF_SB_TOKENS * KIVI_MF_SB_TOKENS + p->residual_length : n;
    // what the step WRITES: the K append (row k_res_len of kres), the V append (row v_win_start + v_res_len of vres), the
    // slot of the token leaving the window (token Tv of the VT store)
    KIVI_REQUIRE(p->residual_length > 0 && p->residual_length <= 128 && p->k_res_len < p->residual_length &&
                     p->v_res_len <= p->residual_length,
                 KIVI_EINVAL, "kivi_gqa_decode: residual of %d keys / window of %d values do not fit residual_length %d",
                 p->k_res_len, p->v_res_len, p->residual_length);
    const bool win_ring = (p->flags & KIVI_GQA_WINDOW_RING) != 0;
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
    KIVI_REQUIRE(mf_store_ok(p->kt, p->kt_sb, p->kt_sh, p->kt_ss, bits) && mf_store_ok(p->vt, p->vt_sb, p->vt_sh, p->vt_ss, bits), KIVI_EALIGN,
                 "kivi_gqa_decode: cache storage must be 16-byte aligned super-blocks");
    KIVI_REQUIRE(p->q && (uintptr_t)p->q % 16 == 0 && p->q_sb % 8 == 0 && p->q_sh % 8 == 0, KIVI_EALIGN,
                 "kivi_gqa_decode: q rows must be 16-byte aligned");
    KIVI_REQUIRE(p->kres && p->knew && (uintptr_t)p->kres % 16 == 0 && (uintptr_t)p->knew % 16 == 0 && p->kres_sb % 8 == 0 &&
                     p->kres_sh % 8 == 0 && p->kres_st % 8 == 0 && p->knew_sb % 8 == 0 && p->knew_sh % 8 == 0,
                 KIVI_EALIGN, "kivi_gqa_decode: key rows must be 16-byte aligned");
    KIVI_REQUIRE(p->vres && p->vnew && (uintptr_t)p->vres % 16 == 0 && (uintptr_t)p->vnew % 16 == 0 && p->vres_sb % 8 == 0 &&
                     p->vres_sh % 8 == 0 && p->vres_st % 8 == 0 && p->vnew_sb % 8 == 0 && p->vnew_sh % 8 == 0,
                 KIVI_EALIGN, "kivi_gqa_decode: value rows must be 16-byte aligned");
    KIVI_REQUIRE(p->scores && (uintptr_t)p->scores % 16 == 0 && p->s_sb % 8 == 0 && p->s_sh % 8 == 0 && p->s_sh >= ((n_rows + 7) & ~(int64_t)7),
                 KIVI_EALIGN, "kivi_gqa_decode: score rows must be 16-byte aligned and hold %lld scores", (long long)n_rows);
    KIVI_REQUIRE((int64_t)(R - 1) * p->s_sh * 2 + n_rows * 2 + 16 < ((int64_t)1 << 32), KIVI_EINVAL, "kivi_gqa_decode: score rows too long");
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
    // launch plan: one launch (rows, or slices of rows, in the LDS) or two -- decided for the longest row of the step's geometry class
    // (ceil(Tq / 512) * 512 + residual_length keys) whether or not the lengths are device-resident, so that an eager step and a
    // replayed one of the same position take the same form (round 5 planned eager steps for their own row: in the bands where only
    // the class bound exceeds a block -- nh == nh_kv: Tq in (7680, 8192], nh / nh_kv = 4: (8704, 9216] -- the two then differed in
    // rounding).  The LDS of an eager launch is still sized for the step's own row (n_rows).
    const int64_t n_class = (int64_t)nsbk * KIVI_MF_SB_TOKENS + p->residual_length;
    const int plan = mf_plan(R, units, n_class, nsbk, p->residual_length, p->flags, bits);
    const int nslot = plan > 1 ? plan : (S + (win_blocks ? 1 : 0));
    const int64_t need = (int64_t)KIVI_GQA_WS_COUNTERS * 4 + (plan == 1 ? 0 : (int64_t)units * nslot * 2 * R * 128 * 4);
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
    k.kres = (uint16_t*)p->kres; k.kres_sb = p->kres_sb; k.kres_sh = p->kres_sh; k.kres_st = p->kres_st;
    k.knew = (const uint16_t*)p->knew; k.knew_sb = p->knew_sb; k.knew_sh = p->knew_sh; k.res_len = p->k_res_len;
    k.range = (const int*)p->kt_range;
    k.dyn = (const MfStep*)p->dyn_step;
    // sliced one-launch form: the second half of the counter area holds the statistics-exchange counters, its last word the ticket
    k.dump = (p->flags & KIVI_GQA_DUMP_SCORES) != 0;
    k.xcount = (int*)p->workspace + KIVI_GQA_WS_COUNTERS / 2;
    k.ticket = (int*)p->workspace + KIVI_GQA_WS_COUNTERS - 1;
    k.err_ws = (int*)p->workspace + KIVI_GQA_WS_COUNTERS - KIVI_GQA_TICKETS - 1;
    k.err_host = plan > 1 ? kivi_device_error_word((hipStream_t)stream) : nullptr;
    static const char* skipk = KIVI_TUNE_ENV("KIVI_GQA_SKIP_K");       // diagnostic (tools/mf_stage_error.py): the caller filled scores / stats
    static const char* timev = KIVI_TUNE_ENV("KIVI_GQA_TIME_V");       // tuning aid: a pending event pair brackets the sV launch instead
    KiviLaunchEvents held = {nullptr, nullptr};
    if (timev) held = kivi_take_launch_events();
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
    v.dyn = (const MfStep*)p->dyn_step;
    // rows (or slices of rows) that fit the LDS: the whole step of a (batch row, kv head) in one launch (nh == nh_kv: 4 blocks of
    // 4 waves per CU; nh / nh_kv in {4, 8}: the R score rows of a unit / a slice in one block, 2 blocks per CU)
    if (plan >= 1) {
        if (plan > 1) { v.S = plan; v.nslot = plan; v.win_blocks = 0; }
        return kivi_mf_run_row(&k, &v, units, n_rows, (p->flags & KIVI_GQA_DUMP_SCORES) != 0, bits, plan, p->residual_length,
                               ((p->flags >> 8) & 0xFF) != 0, s);
    }
    int rc = skipk ? 0 : kivi_mf_run_k(&k, units, bits, s);
    if (rc) return rc;
    if (timev) kivi_set_launch_events(held.start, held.stop);
    return kivi_mf_run_v(&v, 0, bits, s);
}

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
#include "kivi_mfma_layout.h"
#include "kivi_quant.h"
#include "kivi_row_softmax.h"

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct MfStore {              // one cache side (K or V) in the super-block layout
    uint32_t* base;
    int64_t sb_b, sb_h, sb_s; // word strides: batch row, kv head, super-block
};

__device__ __forceinline__ uint32_t* mf_sb(const MfStore& s, int b, int hk, int64_t sb) {
    return s.base + b * s.sb_b + hk * s.sb_h + sb * s.sb_s;
}

// ------------------------------------------------------------------------------------------------ pack / relayout

// Per-channel K quantise + pack of whole 32-token blocks straight into the KT layout (prompt pass, models/
// llama_kivi.py:436, and the flush of the fp16 residual every R tokens, :343-356).  Reference arithmetic through the
// shared quantiser (kivi_quant.h = new_pack.py:236-241 op for op); group_size == 32 == the block.
// One wave per block; lane l owns channels 2l, 2l+1 over the 32 tokens (min / max without any cross-lane step).
__global__ __launch_bounds__(64) void kt_pack_kernel(const uint16_t* k, int64_t k_sb, int64_t k_sh, int64_t k_st, MfStore st,
                                                     int64_t blk0, int nblk, int nh_kv) {
    const int unit = blockIdx.x / nblk, bi = blockIdx.x - unit * nblk;
    const int b = unit / nh_kv, hk = unit - b * nh_kv;
    const int lane = threadIdx.x;
    const uint16_t* src = k + b * k_sb + hk * k_sh + (int64_t)bi * 32 * k_st + 2 * lane;
    uint32_t x[32];
#pragma unroll
    for (int t = 0; t < 32; t++) x[t] = *(const uint32_t*)(src + (int64_t)t * k_st);
    uint32_t mn0 = 0xFFFFu, mx0 = 0u, mn1 = 0xFFFFu, mx1 = 0u;
#pragma unroll
    for (int t = 0; t < 32; t++) {
        const uint32_t k0 = h_key(x[t] & 0xFFFFu), k1 = h_key(x[t] >> 16);
        mn0 = k0 < mn0 ? k0 : mn0; mx0 = k0 > mx0 ? k0 : mx0;
        mn1 = k1 < mn1 ? k1 : mn1; mx1 = k1 > mx1 ? k1 : mx1;
    }
    const GroupQ g0 = make_group(mn0, mx0, 3), g1 = make_group(mn1, mx1, 3);
    const int i = lane & 3;
    uint32_t pw[16];
#pragma unroll
    for (int n = 0; n < 16; n++) {
        const uint32_t ce = quant_one<2>((uint16_t)(x[n] & 0xFFFFu), g0), co = quant_one<2>((uint16_t)(x[n] >> 16), g1);
        const uint32_t ce2 = quant_one<2>((uint16_t)(x[n + 16] & 0xFFFFu), g0), co2 = quant_one<2>((uint16_t)(x[n + 16] >> 16), g1);
        uint32_t w = (ce << (2 * i)) | (co << (2 * i + 16)) | (ce2 << (2 * i + 8)) | (co2 << (2 * i + 24));
        w |= (uint32_t)__shfl_xor((int)w, 1);      // the 4 lanes of a quad hold the 4 channel pairs of one word
        w |= (uint32_t)__shfl_xor((int)w, 2);
        pw[n] = w;
    }
    const int c = lane >> 4, kb = (lane >> 2) & 3;
    const int64_t blk = blk0 + bi;
    uint32_t* sb = mf_sb(st, b, hk, blk >> 4);
    uint32_t* cw = sb + (blk & 15) * KIVI_MF_BLOCK_WORDS;
#pragma unroll
    for (int n = 0; n < 16; n++)
        if ((n >> 2) == i) cw[(n + 16 * kb) * 4 + c] = pw[n];
    const int hidx = kb * 32 + c * 8 + 2 * i;    // kt_half of channel 2l (even): the pair (2l, 2l+1) is one word
    (sb + KIVI_MF_SB_SCALE_WORD0 + (blk & 15) * 64)[hidx >> 1] = (uint32_t)g0.scale | ((uint32_t)g1.scale << 16);
    (sb + KIVI_MF_SB_MN_WORD0 + (blk & 15) * 64)[hidx >> 1] = (uint32_t)g0.mn | ((uint32_t)g1.mn << 16);
}

// KT <-> reference layout K_code_T (B, nh_kv, D, T/16), K_scale_T / K_mn_T (B, nh_kv, D, T/32) (llama_kivi.py:454-455).
// Pure bit moves; one 128-thread block per 32-token block (thread = channel).
template <bool TO_REF>
__global__ __launch_bounds__(128) void kt_relayout_kernel(MfStore st, uint32_t* code, int64_t code_sb, int64_t code_sh,
                                                          int64_t code_sr, uint16_t* scale, uint16_t* mn, int64_t sm_sb,
                                                          int64_t sm_sh, int64_t sm_sr, int nblk, int nh_kv) {
    __shared__ uint32_t lds[256];
    const int unit = blockIdx.x / nblk, blk = blockIdx.x - unit * nblk;
    const int b = unit / nh_kv, hk = unit - b * nh_kv;
    const int d = threadIdx.x;
    uint32_t* sb = mf_sb(st, b, hk, blk >> 4);
    uint32_t* cw = sb + (blk & 15) * KIVI_MF_BLOCK_WORDS;
    uint16_t* ks = (uint16_t*)(sb + KIVI_MF_SB_SCALE_WORD0) + (blk & 15) * 128;
    uint16_t* km = (uint16_t*)(sb + KIVI_MF_SB_MN_WORD0) + (blk & 15) * 128;
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
        scale[sidx] = ks[kt_half(d)];
        mn[sidx] = km[kt_half(d)];
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
        ks[kt_half(d)] = scale[sidx];
        km[kt_half(d)] = mn[sidx];
    }
}

// VT <-> reference layout V_code (B, nh_kv, T, D/16), V_scale / V_mn (B, nh_kv, T, D/32); tokens [0, T), the slots of
// the last block past T are written as zeros (the kernels rely on never-written slots being zero).
template <bool TO_REF>
__global__ __launch_bounds__(256) void vt_relayout_kernel(MfStore st, uint32_t* code, int64_t code_sb, int64_t code_sh,
                                                          int64_t code_sr, uint16_t* scale, uint16_t* mn, int64_t sm_sb,
                                                          int64_t sm_sh, int64_t sm_sr, int64_t T, int nblk, int nh_kv) {
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
            vs[vt_half(tt, wi)] = (t < T) ? scale[b * sm_sb + hk * sm_sh + t * sm_sr + wi] : (uint16_t)0;
            vm[vt_half(tt, wi)] = (t < T) ? mn[b * sm_sb + hk * sm_sh + t * sm_sr + wi] : (uint16_t)0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ shared MFMA pieces

__device__ __forceinline__ h8 as_h8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_bit_cast(h8, (u32x4){a, b, c, d});
}
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, a) * __builtin_bit_cast(h2, b));
}
// a * b - c, one rounding (v_pk_fma_f16): with c = fp16(a * b) the exact remainder of the product
__device__ __forceinline__ uint32_t pk_fms(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b),
                                                                  -__builtin_bit_cast(h2, c)));
}
// 2^(2 i - 6) in both halves: brings a zero-point operand to the scale of the A rows (which carry 2^(6 - 2 i))
__device__ __forceinline__ constexpr uint32_t zfac(int i) { return i == 0 ? 0x24002400u : i == 1 ? 0x2C002C00u : i == 2 ? 0x34003400u : 0x3C003C00u; }
// 2^(6 - 2 i) in both halves
__device__ __forceinline__ constexpr uint32_t afac(int i) { return i == 0 ? 0x54005400u : i == 1 ? 0x4C004C00u : i == 2 ? 0x44004400u : 0x3C003C00u; }

// hi / lo rows of the A operand without a branch: `xf` is x in the lanes of a "lo" row and 0 in the lanes of a "hi" row,
// so  fma(x, s, -fp16(xf * s))  is the rounded product in hi rows and its exact remainder in lo rows (two packed ops).
// Without the split (HILO = false) `xf` is x in hi rows and 0 in lo rows and the element is one packed multiply.
template <bool HILO>
__device__ __forceinline__ uint32_t a_elem(uint32_t x, uint32_t xf, uint32_t s) {
    if constexpr (HILO) return pk_fms(x, s, pk_mul(xf, s));
    else return pk_mul(xf, s);
}

// one 32-channel (K) / 32-token (V) chunk: two MFMAs, the masked code words are the B operands
__device__ __forceinline__ void mfma_pair(const uint32_t* A, uint32_t w, f4& acc0, f4& acc1) {
    const uint32_t ws = w >> 8;
    const h8 a = as_h8(A[0], A[1], A[2], A[3]);
    const h8 b0 = as_h8(w & 0x00030003u, w & 0x000C000Cu, w & 0x00300030u, w & 0x00C000C0u);
    const h8 b1 = as_h8(ws & 0x00030003u, ws & 0x000C000Cu, ws & 0x00300030u, ws & 0x00C000C0u);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, acc1, 0, 0, 0);
}

// rows hi + lo of two accumulator registers (tile 0 in x, tile 1 in y) in one swap + add:
//   R = 4: lanes 0-15 <- tile 0 rows j + (4 + j), lanes 16-31 <- tile 1 (v_permlane16_swap: odd rows of x <-> even rows of y)
//   R = 8: lanes 0-31 <- tile 0 rows (0..7) + (8..15), lanes 32-63 <- tile 1 (v_permlane32_swap)
// (inline asm: on ROCm 7.2 the __builtin_amdgcn_permlane16_swap / 32_swap builtins return the FIRST result in both slots
// -- hipcc emits v_add v, v, v after the swap; checked with hipcc -S)
template <int R>
__device__ __forceinline__ float fold_rows(float x, float y) {
    if constexpr (R == 4) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}

// ------------------------------------------------------------------------------------------------ qK^T

struct GqaKArgs {
    const uint16_t* q;
    int64_t q_sb, q_sh;
    MfStore kt;
    uint16_t* out;
    int64_t out_sb, out_sh;
    int nh_kv, ratio;
    int64_t Tq;                 // packed tokens (multiple of 32)
    int nsb, sb_blocks;         // super-blocks of a row, thread blocks per (b, kv head)
};

// W waves per thread block, one super-block each; nothing is shared between the waves of a block.
template <int R, int W, bool HILO>
__global__ __launch_bounds__(64 * W) void gqa_k_kernel(const GqaKArgs a) {
    constexpr int S = KIVI_MF_SHIFT;
    extern __shared__ uint32_t lds_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* lds_s = lds_all + wave * (1024 + R * 256);          // scale of the super-block: 16 groups x 64 words
    uint16_t* lds_o = (uint16_t*)(lds_s + 1024);                  // R x 512 fp16 scores
    const int unit = (int)blockIdx.x / a.sb_blocks;
    const int sb = ((int)blockIdx.x - unit * a.sb_blocks) * W + wave;
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
    // (n, kb) takes group n), the first two code blocks
    u32x4 sreg[4], zreg[4];
#pragma unroll
    for (int j = 0; j < 4; j++) sreg[j] = buf_load<u32x4, true>(rk, (uint32_t)(KIVI_MF_SB_SCALE_WORD0 * 4 + (j * 64 + lane) * 16), 0);
#pragma unroll
    for (int c = 0; c < 4; c++) zreg[c] = buf_load<u32x4, true>(rk, (uint32_t)(KIVI_MF_SB_MN_WORD0 * 4 + n * 256 + kb * 64 + c * 16), 0);
    u32x4 w0 = buf_load<u32x4, true>(rk, (uint32_t)(lane * 16), 0);
    u32x4 w1 = buf_load<u32x4, true>(rk, (uint32_t)(1024 + lane * 16), 0);

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
            const float f0 = __builtin_ldexpf(h2f_bits(qv[c][2 * i]), sq + S - 2 * i);
            const float f1 = __builtin_ldexpf(h2f_bits(qv[c][2 * i + 1]), sq + S - 2 * i);
            qq[c][i] = (uint32_t)f2h_bits(f0) | ((uint32_t)f2h_bits(f1) << 16);
            qf[c][i] = (lo_row == HILO) ? qq[c][i] : 0u;           // see a_elem
        }
    // per output register j of a lane: which head, its 2^(24 - S - Sq) and 2^-Sq
    float cmul[4], zmul[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int rj = (R == 4) ? j : 4 * ((lane >> 4) & 1) + j;  // head of output register j after fold_rows
        const int sqj = __shfl(sq, rj);                           // lane rj (kb = 0, n = rj) holds head rj's exponent
        cmul[j] = __builtin_ldexpf(1.0f, 24 - S - sqj);
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
        for (int c = 0; c < 4; c++) s[c] = *(const u32x4*)(lds_s + g * 64 + kb * 16 + c * 4);
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

    // ring of code blocks: two groups in flight ahead of the one being multiplied
#pragma unroll
    for (int g = 0; g < 16; g += 2) {
        if (g >= ng) break;
        const u32x4 w2 = buf_load<u32x4, true>(rk, (uint32_t)((g + 2) * 1024 + lane * 16), 0);
        group(g, w0);
        const u32x4 w3 = buf_load<u32x4, true>(rk, (uint32_t)((g + 3) * 1024 + lane * 16), 0);
        if (g + 1 < ng) group(g + 1, w1);
        w0 = w2;
        w1 = w3;
    }
    __builtin_amdgcn_wave_barrier();
    // 512 tokens x R heads of fp16 scores: one 16-byte store per lane and head
#pragma unroll
    for (int rr = 0; rr < R; rr++) {
        if (lane * 8 < ng * 32) {
            const u16x8 v = *(const u16x8*)(lds_o + rr * 512 + lane * 8);
            *(u16x8*)(a.out + b * a.out_sb + (int64_t)(h0 + rr) * a.out_sh + (int64_t)sb * KIVI_MF_SB_TOKENS + lane * 8) = v;
        }
    }
}

template <int R, int W, bool HILO>
void launch_gqa_k(const GqaKArgs& a, int units, hipStream_t s) {
    const size_t lds = (size_t)W * (1024 + R * 256) * 4;
    KIVI_LAUNCH_LDS((gqa_k_kernel<R, W, HILO>), dim3((unsigned)(units * a.sb_blocks)), dim3(64 * W), lds, s, a);
}

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
                            int64_t kt_ss, int64_t token_offset, int B, int nh_kv, int64_t T, int D, int group_size, int bits,
                            kivi_stream_t stream) {
    KIVI_MF_SHAPE_CHECK("kivi_kt_pack");
    KIVI_REQUIRE(T >= 0 && T % 32 == 0 && token_offset >= 0 && token_offset % 32 == 0, KIVI_EINVAL,
                 "kivi_kt_pack: T=%lld and token_offset=%lld must be multiples of the 32-token block", (long long)T,
                 (long long)token_offset);
    KIVI_REQUIRE(mf_store_ok(kt, kt_sb, kt_sh, kt_ss), KIVI_EALIGN, "kivi_kt_pack: cache storage must be 16-byte aligned super-blocks");
    KIVI_REQUIRE(k && (uintptr_t)k % 4 == 0 && k_sb % 2 == 0 && k_sh % 2 == 0 && k_st % 2 == 0, KIVI_EALIGN,
                 "kivi_kt_pack: key rows must be 4-byte aligned");
    if (T == 0) return 0;
    const int nblk = (int)(T / 32);
    const MfStore st = {(uint32_t*)kt, kt_sb, kt_sh, kt_ss};
    hipLaunchKernelGGL(kt_pack_kernel, dim3((unsigned)((int64_t)B * nh_kv * nblk)), dim3(64), 0, (hipStream_t)stream,
                       (const uint16_t*)k, k_sb, k_sh, k_st, st, token_offset / 32, nblk, nh_kv);
    return kivi_launch_status("kt_pack");
}

extern "C" int kivi_kt_relayout(int to_ref, void* kt, int64_t kt_sb, int64_t kt_sh, int64_t kt_ss, void* code,
                                int64_t code_sb, int64_t code_sh, int64_t code_sr, void* scale, void* mn, int64_t sm_sb,
                                int64_t sm_sh, int64_t sm_sr, int B, int nh_kv, int64_t T, int D, int group_size, int bits,
                                kivi_stream_t stream) {
    KIVI_MF_SHAPE_CHECK("kivi_kt_relayout");
    KIVI_REQUIRE(T >= 0 && T % 32 == 0, KIVI_EINVAL, "kivi_kt_relayout: T=%lld must be a multiple of 32", (long long)T);
    KIVI_REQUIRE(mf_store_ok(kt, kt_sb, kt_sh, kt_ss) && code && scale && mn, KIVI_EALIGN, "kivi_kt_relayout: bad buffers");
    if (T == 0) return 0;
    const int nblk = (int)(T / 32);
    const MfStore st = {(uint32_t*)kt, kt_sb, kt_sh, kt_ss};
    const dim3 grid((unsigned)((int64_t)B * nh_kv * nblk));
    if (to_ref)
        hipLaunchKernelGGL(kt_relayout_kernel<true>, grid, dim3(128), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                           code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, nblk, nh_kv);
    else
        hipLaunchKernelGGL(kt_relayout_kernel<false>, grid, dim3(128), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                           code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, nblk, nh_kv);
    return kivi_launch_status("kt_relayout");
}

extern "C" int kivi_vt_relayout(int to_ref, void* vt, int64_t vt_sb, int64_t vt_sh, int64_t vt_ss, void* code,
                                int64_t code_sb, int64_t code_sh, int64_t code_sr, void* scale, void* mn, int64_t sm_sb,
                                int64_t sm_sh, int64_t sm_sr, int B, int nh_kv, int64_t T, int D, int group_size, int bits,
                                kivi_stream_t stream) {
    KIVI_MF_SHAPE_CHECK("kivi_vt_relayout");
    KIVI_REQUIRE(T >= 0, KIVI_EINVAL, "kivi_vt_relayout: negative length");
    KIVI_REQUIRE(mf_store_ok(vt, vt_sb, vt_sh, vt_ss) && code && scale && mn, KIVI_EALIGN, "kivi_vt_relayout: bad buffers");
    if (T == 0) return 0;
    const int nblk = (int)((T + 31) / 32);
    const MfStore st = {(uint32_t*)vt, vt_sb, vt_sh, vt_ss};
    const dim3 grid((unsigned)((int64_t)B * nh_kv * nblk));
    if (to_ref)
        hipLaunchKernelGGL(vt_relayout_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                           code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, T, nblk, nh_kv);
    else
        hipLaunchKernelGGL(vt_relayout_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, st, (uint32_t*)code, code_sb,
                           code_sh, code_sr, (uint16_t*)scale, (uint16_t*)mn, sm_sb, sm_sh, sm_sr, T, nblk, nh_kv);
    return kivi_launch_status("vt_relayout");
}

extern "C" int kivi_gqa_scores(const void* q, int64_t q_sb, int64_t q_sh, const void* kt, int64_t kt_sb, int64_t kt_sh,
                               int64_t kt_ss, void* out, int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv, int D,
                               int64_t T, int group_size, int bits, kivi_stream_t stream) {
    KIVI_MF_SHAPE_CHECK("kivi_gqa_scores");
    KIVI_REQUIRE(nh > 0 && nh % nh_kv == 0 && (nh / nh_kv == 4 || nh / nh_kv == 8), KIVI_EUNSUPPORTED,
                 "kivi_gqa_scores: nh / nh_kv must be 4 or 8 (got %d / %d)", nh, nh_kv);
    KIVI_REQUIRE(T >= 0 && T % 32 == 0, KIVI_EINVAL, "kivi_gqa_scores: T=%lld must be a multiple of 32", (long long)T);
    KIVI_REQUIRE(mf_store_ok(kt, kt_sb, kt_sh, kt_ss), KIVI_EALIGN, "kivi_gqa_scores: cache storage must be 16-byte aligned super-blocks");
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
    const int units = B * nh_kv;
    static const char* nohilo = getenv("KIVI_GQA_NO_HILO");      // tuning aid: fp16-rounded q * scale (no remainder rows)
    static const char* fw = getenv("KIVI_GQA_K_WAVES");          // tuning aid: waves per block (1 or 4)
    int W = ((int64_t)units * a.nsb >= 2048) ? 4 : 1;            // few super-blocks: one wave per block spreads them over the CUs
    if (fw) W = atoi(fw) == 1 ? 1 : 4;
    a.sb_blocks = (a.nsb + W - 1) / W;
    hipStream_t s = (hipStream_t)stream;
#define KIVI_GK(RR, WW, HL) launch_gqa_k<RR, WW, HL>(a, units, s)
    if (a.ratio == 4) {
        if (nohilo) { if (W == 4) KIVI_GK(4, 4, false); else KIVI_GK(4, 1, false); }
        else { if (W == 4) KIVI_GK(4, 4, true); else KIVI_GK(4, 1, true); }
    } else {
        if (nohilo) { if (W == 4) KIVI_GK(8, 4, false); else KIVI_GK(8, 1, false); }
        else { if (W == 4) KIVI_GK(8, 4, true); else KIVI_GK(8, 1, true); }
    }
#undef KIVI_GK
    return kivi_launch_status("gqa_k");
}

// MFMA-friendly internal cache layouts for grouped-query decode ("KT" for K, "VT" for V), gfx950.
//
// Why: the reference packs 16 codes of ONE channel (K: 16 tokens, quant/new_pack.py:146-154 along the token axis) or of
// one token (V: 16 channels) into a word, i.e. along the OUTPUT axis of the fused GEMV (quant/csrc/gemv_cuda.cu:348-427,
// "outer dim").  v_mfma_f32_16x16x32_f16 wants every lane to hold 8 consecutive elements of the REDUCTION axis of one
// output column.  With nh / nh_kv = R query heads per kv head the VALU form costs one mask + R FMAs per code and is
// VALU bound at R >= 4; on the matrix pipe the R heads are free.  So for grouped queries the
// cache keeps the SAME codes / scales / zero points (bit for bit: the 9-tuple of models/llama_kivi.py:454-455 is
// reproduced exactly by the *_to_ref kernels) in a layout whose words ARE B-operand registers after one mask:
//
//   block  = 32 consecutive tokens (= one K quantisation group at group_size 32) of one (batch row, kv head),
//            1024 B of codes = one 16-byte load per lane of a wave, lane = n + 16 * kb
//   super-block (SB) = 16 blocks = 512 tokens, stored contiguously:
//            [ codes 16 x 256 words | scale 16 x 128 halves | mn 16 x 128 halves ]  = 6144 words = 24 KiB
//
//   K block, token tt (0..31), channel d (0..127):      c = d >> 5, kb = (d >> 3) & 3, e = d & 7
//        word (n + 16 kb) * 4 + c   with n = tt & 15, tile = tt >> 4
//        bits mf_pos(tile, e >> 1) + 16 * (e & 1)               (lo half: even channel, hi half: odd channel)
//        scale / mn of (channel d, group g = block index inside the super-block): half kt_sm_half(g, d) of the region =
//        (g >> 3) * 1024 + c * 256 + kb * 64 + (g & 7) * 8 + e  -- for each 32-channel chunk c the 16 bytes (kb, g) of the
//        8 groups of a half super-block are contiguous, i.e. the A-operand load of a wave whose MFMA rows are GROUPS (lane
//        (row = group, kb) takes 16 bytes per chunk) reads 512 dense bytes per instruction.  (Round 2 kept the 128 halves of a
//        group together: the same load then touched 32 cache lines for 16 bytes each, 4x the L2 requests of the whole code
//        stream -- measured 49 us vs 35 us per BASELINE configs[1] launch, profiles/r03_kt_scale_layout.log.)
//   V block, token tt, channel d:                        c = d >> 5 (= channel group), tile = (d >> 4) & 1, n = d & 15
//        word (n + 16 kb) * 4 + c   with kb = tt >> 3, e = tt & 7
//        bits mf_pos(tile, e >> 1) + 16 * (e & 1)               (lo half: even token, hi half: odd token)
//        scale / mn of (token tt, channel group c): half kb * 32 + c * 8 + e
//
// Field positions inside a 16-bit half, mf_pos(tile, i) for the pair i = 0..3 (k = 2 i, 2 i + 1) of the 8 reduction-axis
// elements a lane feeds to one MFMA:   tile 0: 8, 4, 6, 2     tile 1: 12, 0, 10, 14.
// With the four views  w,  w << 4,  w >> 4,  byteswap16(w)  every field lands on bits 9:8 (i = 0, 1) or 7:6 (i = 2, 3), so one
// AND per register makes the B operand: an fp16 SUBNORMAL code * 2^-16 or code * 2^-18 whose significant bits sit at the TOP
// of the mantissa.  That placement matters: the matrix pipe keeps subnormal operands but aligns products by their exponent
// FIELDS, so a subnormal with z leading zero bits loses z bits of the ~26 the adder keeps below the largest term -- codes
// left in the low mantissa bits (code * 4^i * 2^-24 read in place) cost 1e-5 relative error per 32-term dot on softmax-
// shaped operands (tools/mfma_dot_probe.hip), 2e-3 of an sV output after the code and zero-point sums cancel.  The A
// operand carries q * scale (K) or p * scale (V) times 2^(4 + 2 (i >> 1)), split into an fp16 hi and lo row so the product
// is exact; a product is then a * code * 2^-12.
// (Round 5 tried the cheaper placement 8, 6, 4, 2 | 0, 14, 12, 10 -- tile 0 read in place, tile 1 in the byte-swapped halves, 9 instead
// of 11 instructions per word, one A exponent per register: -2.8 % on the row kernels, but a field at bits 3:2 costs 6 of the ~24 bits
// the adder keeps below the largest term, and the sV sums, which chain thousands of instructions through C, came out 1.0-1.7e-3 off on
// 8k-32k rows.  Not kept: profiles/r05_inplace_fields.log.)
#pragma once
#include <stdint.h>

#define KIVI_MF_BLOCK_TOKENS 32
#define KIVI_MF_SB_BLOCKS 16
#define KIVI_MF_SB_TOKENS 512
#define KIVI_MF_BLOCK_WORDS 256        // code words of one block
#define KIVI_MF_SB_CODE_WORDS 4096
#define KIVI_MF_SB_SCALE_WORD0 4096    // scale halves start here (as words)
#define KIVI_MF_SB_MN_WORD0 5120
#define KIVI_MF_SB_WORDS 6144
#define KIVI_MF_SHIFT 4                // A operands carry 2^(4 + 2 (i >> 1))
#define KIVI_MF_PROD_SHIFT 12          // accumulated products = a * code * 2^-12
// Range of the fp16 A operands (q * scale, p * scale).  With q normalised to [1, 2) (times up to 2^6) and the probabilities
// of a row to <= 2^6, a group scale >= 512 would overflow the fp16 hi part where the reference's fp32 scale * code + zero
// (quant/csrc/gemv_cuda.cu:407-413) stays finite -- and group scales in the fp16 subnormals (values ~1e-7) would push the hi part
// into the subnormals and the lo part (the exact remainder) below the grid, where the reference still carries 24 bits.  Every
// kernel that WRITES scales into a store (kivi_kt_pack, kivi_vt_pack, the relayouts, the V flush of the decode step) records three
// sticky per-(batch row, kv head) marks in the unit's range word (one BYTE each, written with byte stores: concurrent writers
// never lose each other's mark):
//   byte 0  a scale whose fp16 bits are >= KIVI_MF_BIG_SCALE_BITS (256.0; NaN / inf included) was written
//   byte 1  a scale >= KIVI_MF_SMALL_SCALE_BITS (2^-8) was written
//   byte 2  (ABI version 3) "the writers of this unit keep byte 1": set by every library writer with every scale it writes
// The consumers place the A operand by mf_range_shift(word): LOWER for a unit with byte 0 set -- qK^T through the placement of q''
// (2^KIVI_MF_BIG_SHIFT: 128 * 65504 * 2^-10 < 2^13, every finite fp16 scale is safe), sV by 2^KIVI_MF_BIG_SHIFT_V = 2^7 split between
// the probabilities and the scales so that neither is rounded (mf_sp / mf_ksh, kivi_mf_dev.h: the lossless part into p'', at most 3 bits
// into the scales of a peaked row; found and refined with tools/fuzz_decode.py in round 6's last sessions) --, 2^KIVI_MF_SMALL_SHIFT
// HIGHER (q'' and p'' alike) for a unit that is KNOWN to hold only
// scales below 2^-8 -- byte 2 set AND byte 1 clear (round 5; q'' / p'' <= 2^15, the A operand < 2^7: a scale of 2^-24 still gives a
// hi part with all its bits) --, and as before otherwise: units whose scales straddle neither bound compute bit for bit what they
// did before the marks existed.  A ZERO word means the default placement (ABI version 2 read it as "all scales < 2^-8" and placed
// 2^8 higher: a store whose words were not written by the library's own writers -- a caller-side packer, a store copied without
// its words -- then overflowed fp16 on ordinary data; version 3 needs the explicit byte 2 for the higher placement).
#define KIVI_MF_BIG_SCALE_BITS 0x5C00u
#define KIVI_MF_BIG_SHIFT 10
#define KIVI_MF_BIG_SHIFT_V 7           // sV of such a unit: p'' * scale 2^7 lower in total (p'' <= 2^6: 2^6 * 65504 * 2^-7 < 2^15), split by mf_sp / mf_ksh
#define KIVI_MF_SMALL_SCALE_BITS 0x1C00u
#define KIVI_MF_SMALL_SHIFT 8

// ---- 4-bit codes ("KT4" / "VT4"; round 4, nh / nh_kv = 4): the same super-block with TWO code tiles per 32-token block.  A word
// holds the 8 reduction-axis elements one lane feeds to ONE matrix instruction (the 2-bit word holds both tiles' 2 x 8):
//   block = 2048 B of codes = two 16-byte loads per lane: bytes [0, 1024) tile 0, [1024, 2048) tile 1; lane = n + 16 kb
//   K block, token tt, channel d:  tile = tt >> 4, n = tt & 15, c = d >> 5, kb = (d >> 3) & 3, e = d & 7
//   V block, token tt, channel d:  tile = (d >> 4) & 1, n = d & 15, c = d >> 5, kb = tt >> 3, e = tt & 7
//        word tile * 256 + (n + 16 kb) * 4 + c, bits 4 (e >> 1) + 16 (e & 1)
//   super-block = [ codes 16 x 512 words | scale 16 x 128 halves | mn 16 x 128 halves ] = 10240 words = 40 KiB; scale / mn as above.
// B operand: register i = shift_i(w) & 0x03C003C0 with shift = << 6, << 2, >> 2, >> 6: every code on mantissa bits 9:6 of an fp16
// subnormal = code * 2^-18, ONE exponent for the four registers: the A operand carries 2^6 throughout (aexp), a product is
// a * code * 2^-12 as for 2 bits; the centre of a code is -7.5.
#define KIVI_MF4_BLOCK_WORDS 512
#define KIVI_MF4_SB_SCALE_WORD0 8192
#define KIVI_MF4_SB_MN_WORD0 9216
#define KIVI_MF4_SB_WORDS 10240

#ifdef __HIPCC__
// placement of q'' / p'' for a unit from its range word: -KIVI_MF_BIG_SHIFT, 0 or +KIVI_MF_SMALL_SHIFT (see above)
__device__ __forceinline__ int mf_range_shift(int word) {
    return (word & 0xFF) ? -KIVI_MF_BIG_SHIFT : (((word & 0xFFFF00) == 0x010000) ? KIVI_MF_SMALL_SHIFT : 0);
}
// The marks a writer leaves for (up to two) scales with these fp16 bits: sticky, byte stores of the value 1.  The word is READ first
// and a byte is stored only when its mark is missing: nearly every scale of ordinary data is >= 2^-8, and an unconditional store of
// byte 1 by every lane of every packing wave serialised on the unit's one address (measured: kivi_kt_pack 10x slower).  A stale
// read costs a redundant store, never a lost mark.
__device__ __forceinline__ void mf_range_mark(int* word, uint32_t scale_bits, uint32_t scale_bits2 = 0u) {
    const uint32_t top = scale_bits > scale_bits2 ? scale_bits : scale_bits2;
    const int cur = *reinterpret_cast<const volatile int*>(word);
    if ((cur & 0xFF0000) == 0) reinterpret_cast<volatile unsigned char*>(word)[2] = 1;
    if (top < KIVI_MF_SMALL_SCALE_BITS) return;
    if ((cur & 0xFF00) == 0) reinterpret_cast<volatile unsigned char*>(word)[1] = 1;
    if (top >= KIVI_MF_BIG_SCALE_BITS && (cur & 0xFF) == 0) reinterpret_cast<volatile unsigned char*>(word)[0] = 1;
}
template <int BITS> struct MfL;        // per-width constants of the super-block
template <> struct MfL<2> {
    static constexpr int BLOCK_WORDS = KIVI_MF_BLOCK_WORDS, SCALE_WORD0 = KIVI_MF_SB_SCALE_WORD0, MN_WORD0 = KIVI_MF_SB_MN_WORD0,
                         SB_WORDS = KIVI_MF_SB_WORDS;
};
template <> struct MfL<4> {
    static constexpr int BLOCK_WORDS = KIVI_MF4_BLOCK_WORDS, SCALE_WORD0 = KIVI_MF4_SB_SCALE_WORD0, MN_WORD0 = KIVI_MF4_SB_MN_WORD0,
                         SB_WORDS = KIVI_MF4_SB_WORDS;
};
__device__ __forceinline__ int kt4_word(int tt, int d) { return (tt >> 4) * 256 + ((tt & 15) + 16 * ((d >> 3) & 3)) * 4 + (d >> 5); }
__device__ __forceinline__ int kt4_bit(int d) { return 4 * ((d & 7) >> 1) + 16 * (d & 1); }
__device__ __forceinline__ int vt4_word(int tt, int d) { return ((d >> 4) & 1) * 256 + ((d & 15) + 16 * (tt >> 3)) * 4 + (d >> 5); }
__device__ __forceinline__ int vt4_bit(int tt) { return 4 * ((tt & 7) >> 1) + 16 * (tt & 1); }
__device__ __forceinline__ int kt_word(int tt, int d) { return ((tt & 15) + 16 * ((d >> 3) & 3)) * 4 + (d >> 5); }
__device__ __forceinline__ int mf_pos(int tile, int i) { return ((tile ? 0xEA0C : 0x2648) >> (4 * i)) & 15; }
__device__ __forceinline__ int kt_bit(int tt, int d) { return mf_pos(tt >> 4, (d & 7) >> 1) + 16 * (d & 1); }
__device__ __forceinline__ int kt_sm_half(int g, int d) {
    return (g >> 3) * 1024 + (d >> 5) * 256 + ((d >> 3) & 3) * 64 + (g & 7) * 8 + (d & 7);
}
// word (= 2 halves) offset of the 16 bytes (channels 32 c + 8 kb .. + 7) of group g inside the scale / mn region
__device__ __forceinline__ int kt_sm_word4(int g, int kb, int c) { return (g >> 3) * 512 + c * 128 + kb * 32 + (g & 7) * 4; }
__device__ __forceinline__ int vt_word(int tt, int d) { return ((d & 15) + 16 * (tt >> 3)) * 4 + (d >> 5); }
__device__ __forceinline__ int vt_bit(int tt, int d) { return mf_pos((d >> 4) & 1, (tt & 7) >> 1) + 16 * (tt & 1); }
__device__ __forceinline__ int vt_half(int tt, int c) { return (tt >> 3) * 32 + c * 8 + (tt & 7); }
#endif

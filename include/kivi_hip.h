/*
 * kivi_hip.h -- C ABI of libkivi_hip.so: the MI355X (gfx950) implementation of
 * KIVI's quant/ hot path.  Plain pointers and sizes only: every pointer is a
 * DEVICE pointer owned by the caller, every kernel is enqueued on the
 * hipStream_t passed as `stream` (NULL = the default stream) and nothing here
 * synchronises or allocates.  Return value: 0 on success, a positive hipError_t
 * if a launch failed, a negative KIVI_E* code if the arguments are rejected
 * (kivi_last_error() then holds a message; thread-local).
 *
 * Each entry point names the reference interface it replaces (paths relative
 * to the jy-yuan/KIVI tree).  INTEGRATION.md shows the binding a maintainer of
 * the reference would add.
 *
 * Layout vocabulary (reference: models/llama_kivi.py:454-455, "hook state"):
 *   fpi = 32 / bits codes per int32 word, element i of a word at bit bits*i
 *   K_code_T (B, nh_kv, D, Tq/fpi) int32   K_scale_T, K_mn_T (B, nh_kv, D, Tq/g) fp16
 *   V_code   (B, nh_kv, Tv, D/fpi) int32   V_scale,  V_mn    (B, nh_kv, Tv, D/g) fp16
 * Strides are in ELEMENTS of the tensor they describe (int32 words / halves).
 *
 * SURFACE.  The reference's boundary for this path is two functions (pybind.cpp:5-8) + the Python pack module; what a
 * caller is meant to bind here is correspondingly small:
 *   SUPPORTED   pack / unpack:  kivi_quant_pack_lastdim, kivi_quant_pack_k_tmajor, kivi_unpack_dequant_lastdim,
 *                               kivi_pack_codes_lastdim, kivi_unpack_codes_lastdim
 *               fused GEMVs:    kivi_gemv_k, kivi_gemv_v (hook-state tensors), kivi_gemv_outer_dim, kivi_gemv_awq (the
 *                               reference extension's own argument layouts)
 *               a layer step:   kivi_decode_layer (hook-state cache: any 2- / 4-bit shape), kivi_mf_decode_layer and its
 *                               hipGraph form kivi_mf_decode_layer_dyn + kivi_mf_step_* (matrix-pipe cache: g = 32, D = 128,
 *                               2-bit with nh / nh_kv in {1, 4, 8} or 4-bit with nh / nh_kv in {1, 4}), with the packers of that
 *                               cache: kivi_kt_pack, kivi_vt_pack, kivi_kt_relayout, kivi_vt_relayout
 *   BUILDING BLOCKS (what the layer steps are composed of; exported for tests, tools and callers that keep their own cache
 *               bookkeeping -- same contracts, but no stability promise beyond the ABI version):  kivi_gemv_k_paged,
 *               kivi_decode_scores, kivi_softmax_scaled, kivi_decode_output, kivi_decode_softmax_output, kivi_decode_attend,
 *               kivi_gqa_scores, kivi_gqa_output, kivi_gqa_decode, kivi_mf_launch_plan
 *   INSTRUMENTATION (bench.py, tools/, the parity sweeps; not part of the drop-in):  kivi_gemv_{k,v}_variant*, kivi_event_*,
 *               kivi_set_launch_events, kivi_last_timed_kernel, kivi_debug_set_stamps
 * Environment: the product library reads no environment variable; tuning knobs, losing / result-changing kernel variants and the
 * stamping instantiations exist only in -DKIVI_TUNING builds (tools/build_variant.sh).
 */
#ifndef KIVI_HIP_H
#define KIVI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KIVI_ABI_VERSION 3   /* 3: range words carry an explicit byte 2 (a zero word = default placement); sliced launches always take
                                  ticket ids, the workspace's counter area holds a device error word; KIVI_ETIMEOUT, kivi_device_error */

#define KIVI_EINVAL (-1)       /* unsupported bits / group size / shape */
#define KIVI_EALIGN (-2)       /* pointer or stride alignment the kernels rely on is violated */
#define KIVI_EUNSUPPORTED (-3) /* valid in the reference, not implemented here */
#define KIVI_ETIMEOUT (-4)     /* a block of an EARLIER sliced launch gave up waiting for a partner block (see kivi_device_error) */

typedef void* kivi_stream_t; /* hipStream_t */

int kivi_abi_version(void);
const char* kivi_last_error(void);
/* Sticky, asynchronous device-side error of this process (like a stream error of the runtime): 0, or KIVI_ETIMEOUT when a block of
 * a sliced one-launch decode step (kivi_gqa_decode / kivi_mf_decode_layer*, rows cut into S > 1 slices) gave up after ~1 s of
 * waiting for a partner block of its unit -- that unit's output of that step is NaN.  The blocks report through one word of
 * host-visible memory the library owns (the only allocation it ever makes: 64 bytes of pinned host memory, on first use), so the
 * host can look without synchronising: the NEXT kivi_gqa_decode / kivi_mf_decode_layer* call returns KIVI_ETIMEOUT once (with a
 * message naming the unit) and clears the error -- the arrival counters of the timed-out launch have been put back to zero by its
 * last block, the call after that runs normally.  kivi_device_error() returns and clears the same state explicitly (after a
 * stream synchronisation it is exact).  The same code is also left in the device error word of the launch's workspace (word
 * 16375 of the counter area) for callers that keep everything on the device. */
int kivi_device_error(void);

/* ---------------------------------------------------------------- pack --- */

/*
 * Fused group-wise asymmetric quantise + pack along the LAST dim.
 * Replaces triton_quantize_and_pack_along_last_dim (quant/new_pack.py:217-252:
 * two Triton kernels + five elementwise torch kernels + an int32 temporary) and
 * its pure-torch twin quant_and_pack_vcache (quant/new_pack.py:30-48).
 *   x      (rows, T) fp16, contiguous          T % group_size == 0, T % fpi == 0
 *   code   (rows, T/fpi) int32                 scale, mn (rows, T/group_size) fp16
 * bits in {2,4,8}.  Results are bit-identical to the reference op sequence
 * (per-op fp16 rounding, round-half-even); a constant group (scale 0 -> 0/0)
 * yields code 0 as the reference's CUDA float->int conversion does.
 */
int kivi_quant_pack_lastdim(const void* x, void* code, void* scale, void* mn, int64_t rows, int64_t T,
                            int group_size, int bits, kivi_stream_t stream);

/*
 * Per-channel K quantise + pack straight from the un-transposed K tensor:
 * reads k[b, h, t, :] = k + b*k_sb + h*k_sh + t*k_st (D contiguous halves) and
 * writes the hook-state layout (groups of `group_size` TOKENS per channel d):
 *   code [b, h, d, code_off + t/fpi],  scale/mn [b, h, d, sm_off + t/group_size]
 * Replaces `key_states.transpose(2, 3).contiguous()` + the pack call at
 * models/llama_kivi.py:345 and :436 (and quant_and_pack_kcache,
 * quant/new_pack.py:8-27, whose codes are the transpose of these).
 * The *_off arguments let a pre-allocated cache be appended in place.
 */
int kivi_quant_pack_k_tmajor(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_st, void* code, int64_t code_sb,
                             int64_t code_sh, int64_t code_sr, int64_t code_off, void* scale, void* mn,
                             int64_t sm_sb, int64_t sm_sh, int64_t sm_sr, int64_t sm_off, int B, int nh, int64_t T,
                             int D, int group_size, int bits, kivi_stream_t stream);

/*
 * Unpack + dequantise along the last dim: out = fp16(fp16(fp16(q) * scale) + mn).
 * Replaces unpack_and_dequant_vcache (quant/new_pack.py:69-83); with the K^T
 * layout it also serves unpack_and_dequant_kcache (:51-66).
 */
int kivi_unpack_dequant_lastdim(const void* code, const void* scale, const void* mn, void* out, int64_t rows,
                                int64_t T, int group_size, int bits, kivi_stream_t stream);

/* Raw code pack along the last dim: pack_tensor (quant/new_pack.py:86-107) / _pack_along_last_dim (:132-154).
 * data (rows, T) int32 -> code (rows, T/fpi) int32, OR of data << bits*i (no masking, like the reference). */
int kivi_pack_codes_lastdim(const void* data_i32, void* code, int64_t rows, int64_t T, int bits,
                            kivi_stream_t stream);

/* Raw code unpack (quant/new_pack.py:110-129, unpack_tensor with pack_dim = last): int16 out. */
int kivi_unpack_codes_lastdim(const void* code, void* out_i16, int64_t rows, int64_t T, int bits,
                              kivi_stream_t stream);

/* ---------------------------------------------------------- fused GEMV --- */

/*
 * qK^T over the packed per-channel K cache, hook-state layout, no transposes:
 *   out[b, h, t] = fp16( sum_d q[b,h,d] * (scale[b,hk,d,t/g] * code[b,hk,d,t] + mn[b,hk,d,t/g]) )
 * with hk = h / (nh / nh_kv), fp32 arithmetic, one final rounding.
 * Replaces cuda_bmm_fA_qB_outer (quant/matmul.py:178-219, including its three
 * `.transpose(1,2).contiguous()` copies) -> kivi_gemv.gemv_forward_cuda_outer_dim
 * (quant/csrc/gemv_cuda.cu:511-557) -> bgemv2/4_kernel_outer_dim (:265-427) at the
 * call site models/llama_kivi.py:324.  bits in {2,4}; q_len must be 1.
 *   q    (B, nh, D)  at q + b*q_sb + h*q_sh          out (B, nh, T) at out + b*out_sb + h*out_sh
 *   code (B, nh_kv, D, >=T/fpi) strides code_sb/sh/sr  scale, mn (B, nh_kv, D, >=T/g) strides sm_sb/sh/sr
 */
int kivi_gemv_k(const void* q, int64_t q_sb, int64_t q_sh, const void* code, int64_t code_sb, int64_t code_sh,
                int64_t code_sr, const void* scale, const void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr,
                void* out, int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv, int D, int64_t T,
                int group_size, int bits, kivi_stream_t stream);

/*
 * Same product over PAGED per-channel K storage (what an in-place, appendable cache needs: with the plain
 * (B, nh_kv, D, capacity) layout every channel row is strided by the capacity and HBM efficiency collapses,
 * see DESIGN.md).  A page holds `page_tokens` tokens of all D channels as its own contiguous block:
 *   code  (B, nh_kv, P, D, page_tokens/fpi)   strides code_sb, code_sh, code_sp (page), code_sr (channel row)
 *   scale (B, nh_kv, P, D, page_tokens/g)     strides sm_sb,   sm_sh,   sm_sp,          sm_sr
 * Token t lives in page t / page_tokens.  `variant` = -1 picks the default kernel.  No reference twin: the
 * reference re-allocates the whole packed K with torch.cat instead (models/llama_kivi.py:350-352).
 */
int kivi_gemv_k_paged(int variant, int64_t page_tokens, int64_t code_sp, int64_t sm_sp, const void* q, int64_t q_sb,
                      int64_t q_sh, const void* code, int64_t code_sb, int64_t code_sh, int64_t code_sr,
                      const void* scale, const void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr, void* out,
                      int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv, int D, int64_t T, int group_size,
                      int bits, kivi_stream_t stream);

/*
 * sV over the packed per-token V cache:
 *   out[b, h, d] = fp16( sum_t a[b,h,t] * (scale[b,hk,t,d/g] * code[b,hk,t,d] + mn[b,hk,t,d/g]) )
 * Replaces the same reference chain at the call site models/llama_kivi.py:382
 * (fA there is the non-contiguous slice attn_weights[..., :-value_full_length];
 * a_sb/a_sh carry its strides so no copy is needed).
 *   a    (B, nh, Tv) at a + b*a_sb + h*a_sh          out (B, nh, D)
 *   code (B, nh_kv, Tv, D/fpi) strides code_sb/sh/sr  scale, mn (B, nh_kv, Tv, D/g) strides sm_sb/sh/sr
 */
int kivi_gemv_v(const void* a, int64_t a_sb, int64_t a_sh, const void* code, int64_t code_sb, int64_t code_sh,
                int64_t code_sr, const void* scale, const void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr,
                void* out, int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv, int64_t Tv, int D,
                int group_size, int bits, kivi_stream_t stream);

/*
 * ABI twin of the reference's native entry point on ITS kernel-input layout
 * (torch::Tensor gemv_forward_cuda_outer_dim, quant/csrc/gemv_cuda.h:13-21,
 * gemv_cuda.cu:511-557): in (BS, 1, IC) fp16, kernel (BS_kv, OC/fpi, IC) int32,
 * scale/zeros (BS_kv, OC/g, IC) fp16, out (BS, 1, OC) fp16, BS_kv = BS*nh_kv/nh.
 * Kept for callers that already hold transposed tensors (quant/gemv.py:117,154).
 */
int kivi_gemv_outer_dim(const void* in, const void* kernel, const void* scale, const void* zeros, void* out,
                        int64_t BS, int64_t IC, int64_t OC, int bit, int group_size, int nh, int nh_kv,
                        kivi_stream_t stream);

/*
 * ABI twin of the reference's second native entry point, the legacy AWQ-style inner-dim 4-bit GEMV
 * (torch::Tensor gemv_forward_cuda, quant/csrc/gemv_cuda.h:4-10, gemv_cuda.cu:60-246): in (B, IC) fp16,
 * kernel (OC, IC/8) int32 packed along IC, scale / zeros (OC, sz_pitch >= IC/g) fp16, out (B, OC) fp16;
 * bit must be 4 and group_size 64 or 128 like the reference.  Off the KV-cache path (surface parity only).
 */
int kivi_gemv_awq(const void* in, const void* kernel, const void* scale, const void* zeros, void* out, int64_t B,
                  int64_t IC, int64_t OC, int bit, int group_size, int64_t sz_pitch, kivi_stream_t stream);

/* --------------------------------------------------- fused decode step --- */

/*
 * The three launches of one decode step of the attention hook (models/llama_kivi.py:314-399) for one layer.
 * Together they replace ~20 kernels of the reference sequence with identical roundings:
 *
 * kivi_decode_scores: out[b,h,:T] = fused qK^T over the packed K pages (as kivi_gemv_k_paged) and
 *   out[b,h,T:T+res_len+1] = q . k for the fp16 residual keys plus the new one (:333-337); also appends `knew`
 *   to the residual buffer at index res_len (the torch.cat of :334).  `out` is the pre-scale score row (:339 cat).
 * kivi_softmax_scaled: probs = softmax_fp32(fp16(scores * inv_scale) [+ mask, clamped at fp16 min]) -> fp16
 *   (:339 division, :364-372 mask, :375 softmax).  rows = B*nh, row pitches in halves, mask (B,1,1,n) or NULL.
 * kivi_decode_output: out[b,h,:] = fp16( fp16(fused sV over the packed V) + fp16(probs[..., Tv:] @ V_window) )
 *   (:382-384; window = rows [win_start, win_start+res_len) of `vres` plus `vnew`), appends `vnew` to the window
 *   (:377) and, if `flush`, quantises the oldest window row into cache row Tv (:386-399), bit-identical to
 *   kivi_quant_pack_lastdim.  The caller advances its lengths afterwards.
 * Return KIVI_EUNSUPPORTED when no tuned kernel covers the shape (callers then compose the unfused entry points).
 */
int kivi_decode_scores(int64_t page_tokens, int64_t code_sp, int64_t sm_sp, const void* q, int64_t q_sb, int64_t q_sh,
                       const void* code, int64_t code_sb, int64_t code_sh, int64_t code_sr, const void* scale,
                       const void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr, void* kres, int64_t kres_sb,
                       int64_t kres_sh, int64_t kres_st, const void* knew, int64_t knew_sb, int64_t knew_sh, int res_len,
                       void* out, int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv, int D, int64_t T,
                       int group_size, int bits, kivi_stream_t stream);
int kivi_softmax_scaled(const void* scores, void* probs, int64_t rows, int64_t n, int64_t s_pitch, int64_t p_pitch,
                        float inv_scale, const void* mask, int64_t mask_sb, int nh, kivi_stream_t stream);
int kivi_decode_output(const void* probs, int64_t a_sb, int64_t a_sh, void* code, int64_t code_sb, int64_t code_sh,
                       int64_t code_sr, void* scale, void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr, void* vres,
                       int64_t vres_sb, int64_t vres_sh, int64_t vres_st, int win_start, int res_len, const void* vnew,
                       int64_t vnew_sb, int64_t vnew_sh, int flush, void* out, int64_t out_sb, int64_t out_sh, int B,
                       int nh, int nh_kv, int64_t Tv, int D, int group_size, int bits, kivi_stream_t stream);

/* kivi_softmax_scaled + kivi_decode_output in ONE call: `scores` are the pre-softmax rows written by
 * kivi_decode_scores (row length Tv + res_len + 1).  MHA rows of <= 8192 keys: one launch, every block turns its
 * row into fp16 probabilities in LDS (bit-identical to kivi_softmax_scaled) and goes on with the sV product.
 * Grouped queries (nh > nh_kv) or longer rows: a row-softmax launch first OVERWRITES `scores` with the probabilities
 * (same arithmetic), then the sV launch reads them; `scores` is scratch in that case despite the const. */
int kivi_decode_softmax_output(const void* scores, int64_t a_sb, int64_t a_sh, float inv_scale, const void* mask,
                               int64_t mask_sb, void* code, int64_t code_sb, int64_t code_sh, int64_t code_sr,
                               void* scale, void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr, void* vres,
                               int64_t vres_sb, int64_t vres_sh, int64_t vres_st, int win_start, int res_len,
                               const void* vnew, int64_t vnew_sb, int64_t vnew_sh, int flush, void* out, int64_t out_sb,
                               int64_t out_sh, int B, int nh, int nh_kv, int64_t Tv, int D, int group_size, int bits,
                               kivi_stream_t stream);

/* Everything of the decode step that follows the packed qK^T GEMV, in ONE launch: residual scores
 * q . [fp16 K residual | new key] (+ append of the new key), scale + mask + softmax, packed sV + fp16 V window
 * (+ append of the new value, + quantisation of the oldest window token).  A layer's decode step is then
 * kivi_gemv_k_paged (scores[..., :Tq]) followed by this call.  `scores` rows have length Tq + k_res_len + 1 ==
 * Tv + v_res_len + 1.  KIVI_EUNSUPPORTED when a shape / alignment requirement is not met (use the finer entry points). */
typedef struct kivi_decode_attend_args {
    const void* q; int64_t q_sb, q_sh;                           /* (B, nh, D) queries */
    void* kres; int64_t kres_sb, kres_sh, kres_st;               /* (B, nh_kv, R, D) fp16 K residual buffer */
    const void* knew; int64_t knew_sb, knew_sh; int k_res_len;   /* (B, nh_kv, D) new key; keys already in kres */
    void* scores; int64_t s_sb, s_sh;                            /* (B, nh, >= n) pre-softmax score rows; scratch:
        on return the rows may hold the probabilities instead */
    float inv_scale; const void* mask; int64_t mask_sb;          /* 1/sqrt(D); additive (B,1,1,n) fp16 mask or NULL */
    void* v_code; int64_t vc_sb, vc_sh, vc_sr;                   /* packed V (B, nh_kv, >=Tv+1, D/fpi) */
    void* v_scale; void* v_mn; int64_t vs_sb, vs_sh, vs_sr;
    void* vres; int64_t vres_sb, vres_sh, vres_st; int v_win_start, v_res_len;   /* fp16 V window buffer */
    const void* vnew; int64_t vnew_sb, vnew_sh; int v_flush;     /* new value; quantise the oldest window token */
    void* out; int64_t out_sb, out_sh;                           /* (B, nh, D) attention output */
    int B, nh, nh_kv, D, group_size, v_bits; int64_t Tq, Tv;
    void* workspace; int64_t workspace_bytes;                    /* optional (may be NULL): zero-initialised device
        scratch, >= 65536 + 4096 + B*nh*512 + 4 * B*nh*D*65 bytes.  With it, rows are split over several blocks when
        B*nh_kv is too small to fill the GPU (long context, small batch) or the probabilities of a block's rows would
        not fit a small LDS budget (grouped queries, long rows); counters in it are left at zero after every call. */
    /* optional (k_code may be NULL): the packed-K side of the same step, as kivi_gemv_k_paged takes it.  With it the
       call covers the WHOLE step -- the library runs the packed qK^T itself (do not call kivi_gemv_k_paged first) and,
       for the MHA decode shape (rows <= 8192 keys, nothing split), fuses it into the same launch: the block that
       owns a (b, head) row computes the row's packed scores into LDS and goes straight on (no score round trip
       through HBM, one launch per layer). */
    const void* k_code; int64_t kc_sb, kc_sh, kc_sp, kc_sr;
    const void* k_scale; const void* k_mn; int64_t ks_sb, ks_sh, ks_sp, ks_sr;
    int64_t k_page_tokens; int k_bits;
} kivi_decode_attend_args;
int kivi_decode_attend(const kivi_decode_attend_args* args, kivi_stream_t stream);

/* ------------------------------------ grouped queries on the matrix pipe --- */

/*
 * MFMA-friendly cache layout (group_size 32, head_dim 128; bits = 2: nh / nh_kv in {1, 4, 8}; bits = 4: nh / nh_kv = 4 (round 4) or 1 (round 6: multi-head KIVI-4, e.g. LongChat-7B-32K),
 * the reference's published Mistral-7B + KIVI-4 shape): round 2 introduced it for grouped-query models, round 3 uses it for
 * multi-head models too (the matrix pipe takes the per-code multiply-adds off the vector ALU, which is what bounds the
 * hook-layout kernels).  Every entry point below takes `bits` and refuses (KIVI_EUNSUPPORTED) what is outside these sets.
 * Same codes, scales and zero points as the hook-state tensors above -- kivi_kt_relayout / kivi_vt_relayout convert
 * both ways bit for bit -- stored so that one masked code word IS a B-operand register of v_mfma_f32_16x16x32_f16
 * (kivi_amd/csrc/kivi_mfma_layout.h): per (batch row, kv head) a sequence of super-blocks of 512 tokens,
 *   [ codes 16 x 256 words | scale 16 x 128 halves | mn 16 x 128 halves ] = 6144 int32 words each (bits = 2), or
 *   [ codes 16 x 512 words | scale 16 x 128 halves | mn 16 x 128 halves ] = 10240 words (bits = 4: sb_s >= 10240),
 * addressed as base + b*sb_b + hk*sb_h + (t / 512)*sb_s (strides in words).  Never-written slots must be ZERO.
 * RANGE WORDS ("range flags"): every store comes with `range`, B * nh_kv int32 (index b * nh_kv + hk), zeroed by the caller
 *   together with the store.  The matrix pipe takes q * scale (qK^T) and p * scale (sV) as fp16 hi / lo pairs; with the default
 *   placement of q and p a group scale >= 512 would overflow them where the reference's fp32 `scale * code + zero`
 *   (quant/csrc/gemv_cuda.cu:407-413) stays finite, and group scales in the fp16 subnormals would leave the hi part without its
 *   low bits.  So every entry point that WRITES scales (kivi_kt_pack, kivi_vt_pack, the relayouts towards the layout, the V flush
 *   inside kivi_gqa_decode) marks BYTE 0 of the unit's word when it writes a scale >= 256 (inf / NaN included), BYTE 1 when it
 *   writes a scale >= 2^-8 and BYTE 2 with every scale it writes ("the writers of this unit keep byte 1"; byte stores of the value
 *   1: concurrent writers never lose a mark), and the consumers take the operand lower for a unit with byte 0 set (finite for
 *   every finite fp16 scale: qK^T places q 2^10 lower; sV needs 2^7 and splits it -- the part that rounds no probability goes
 *   into their placement, at most 3 bits into the unit's scales, exact for scales >= 2^-11), q / p 2^8 higher for a unit with byte 2 set
 *   and bytes 0, 1 clear (all its scales are KNOWN to be < 2^-8),
 *   unchanged otherwise -- in particular for a ZERO word: a store filled by a caller's own packer, or copied without its words, gets
 *   the default placement (ABI version 2 placed it higher and overflowed on ordinary data).  Sticky (never cleared by the
 *   library); a caller that copies a store copies its words; a caller that writes scales itself marks the bytes the same way.
 * The reference has no counterpart: it expands codes / scale / mn nh / nh_kv times (models/mistral_kivi.py:58-67,
 * :381-385, :441-445) or lets the CUDA kernel map heads (quant/csrc/gemv_cuda.cu:361-365).
 *
 * kivi_kt_pack: per-channel K quantise + pack of whole 32-token blocks (T % 32 == 0) from un-transposed keys
 *   k[b, h, t, :] = k + b*k_sb + h*k_sh + t*k_st (16-byte aligned rows) straight into the layout at token_offset (prompt pass
 *   models/llama_kivi.py:436, residual flush :343-356); bit-identical to kivi_quant_pack_k_tmajor + relayout.
 * kivi_vt_pack: per-token V quantise + pack of tokens [0, T) of a prompt (any T; 16-byte aligned rows) into the VT
 *   layout at token 0 -- triton_quantize_and_pack_along_last_dim (new_pack.py:217-252) as llama_kivi.py:441-448 applies
 *   it to value_states[:, :, :-R], without the intermediate hook-state tensors.
 * kivi_kt_relayout / kivi_vt_relayout: to_ref != 0 writes tokens [0, T) of the hook-state tensors
 *   (K_code_T (B,nh_kv,D,T/fpi) / V_code (B,nh_kv,T,D/fpi), fpi = 32 / bits, + scale, mn) from the layout, to_ref == 0 the reverse.
 * kivi_gqa_scores: out[b, h, :T] = packed qK^T (the arithmetic of kivi_gemv_k: fp32 accumulate, one fp16 rounding; the
 *   q * scale products enter the matrix pipe as exact hi + lo fp16 pairs).  = cuda_bmm_fA_qB_outer at llama_kivi.py:324.
 * kivi_gqa_output (nh / nh_kv in {1, 4}): out[b, h, :128] = packed sV for given fp16 attention weights
 *   probs[b, h, :T] = probs + b*p_sb + h*p_sh (rows 16-byte aligned, pitch >= T rounded up to 8) -- cuda_bmm_fA_qB_outer
 *   at llama_kivi.py:382 on the VT layout: fp32 accumulate of exact products, one fp16 rounding.  `workspace`: 64 KiB of
 *   arrival counters (zeroed once by the caller) + 4 * B * nh bytes (rounded up to 256) + B * nh_kv * slices * 2 *
 *   (nh / nh_kv) * 128 floats, slices <= max(1, ceil(T / 512)).
 */
int kivi_kt_pack(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_st, void* kt, int64_t kt_sb, int64_t kt_sh,
                 int64_t kt_ss, void* kt_range, int64_t token_offset, int B, int nh_kv, int64_t T, int D, int group_size,
                 int bits, kivi_stream_t stream);
int kivi_vt_pack(const void* v, int64_t v_sb, int64_t v_sh, int64_t v_st, void* vt, int64_t vt_sb, int64_t vt_sh,
                 int64_t vt_ss, void* vt_range, int B, int nh_kv, int64_t T, int D, int group_size, int bits,
                 kivi_stream_t stream);
/* (kt_range / vt_range may be null when to_ref != 0: reading a store does not touch its flags) */
int kivi_kt_relayout(int to_ref, void* kt, int64_t kt_sb, int64_t kt_sh, int64_t kt_ss, void* kt_range, void* code,
                     int64_t code_sb, int64_t code_sh, int64_t code_sr, void* scale, void* mn, int64_t sm_sb, int64_t sm_sh,
                     int64_t sm_sr, int B, int nh_kv, int64_t T, int D, int group_size, int bits, kivi_stream_t stream);
int kivi_vt_relayout(int to_ref, void* vt, int64_t vt_sb, int64_t vt_sh, int64_t vt_ss, void* vt_range, void* code,
                     int64_t code_sb, int64_t code_sh, int64_t code_sr, void* scale, void* mn, int64_t sm_sb, int64_t sm_sh,
                     int64_t sm_sr, int B, int nh_kv, int64_t T, int D, int group_size, int bits, kivi_stream_t stream);
int kivi_gqa_scores(const void* q, int64_t q_sb, int64_t q_sh, const void* kt, int64_t kt_sb, int64_t kt_sh, int64_t kt_ss,
                    const void* kt_range, void* out, int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv, int D, int64_t T,
                    int group_size, int bits, kivi_stream_t stream);
int kivi_gqa_output(const void* probs, int64_t p_sb, int64_t p_sh, const void* vt, int64_t vt_sb, int64_t vt_sh, int64_t vt_ss,
                    const void* vt_range, void* out, int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv, int D, int64_t T,
                    int group_size, int bits, void* workspace, int64_t workspace_bytes, kivi_stream_t stream);

/*
 * kivi_gqa_decode: the whole decode step of one layer over the KT / VT layouts.  ONE launch (packed qK^T -> LDS scores [softmax
 * statistics per 512-token segment inside the K walk] -> residual scores -> window -> packed sV with the probabilities made on the
 * fly) when the score rows of a (batch row, kv head) unit fit the LDS and the units fill the chip:
 *   nh == nh_kv      rows of <= 16 super-blocks + a full residual (8320 keys) and (>= 192 units, or <= 4096 packed keys at any batch)
 *                                                                                                      -> mf_row_kernel (2- and 4-bit codes)
 *   nh / nh_kv == 4  rows of <= 18 super-blocks + a full residual (9344 keys) and >= 128 units (4-bit codes: >= 192) -> mf_row4_kernel
 *                    (2-bit rows of <= ~6.3k keys with >= 3 blocks per CU: the instantiation that fits three blocks per CU)
 *   nh / nh_kv == 8  rows of <= 4608 keys and >= 192 units                                            -> mf_row4_kernel<R = 8>
 * The plan is made for the LONGEST row of the step's geometry class (ceil(Tq / 512) * 512 + residual_length keys: kivi_mf_step_key),
 * whether the lengths are passed by value or device-resident, so an eager step and a replayed one of the same position take the same form.
 * For nh / nh_kv in {4, 8} LONGER rows (and rows of few units, to fill the chip) are cut into S slices of whole super-blocks, one
 * block per slice, still in ONE launch: the slices of a unit exchange their (max, sum exp) through `stats` (arrival counters in the
 * second half of the workspace's counter area), form the same probabilities a single block would, and their partial outputs meet
 * in `workspace` (S slots per unit).  Blocks of such a launch wait for each other, so their ids are ALWAYS handed out in start
 * order by ticket counters (the last eight words of the counter area, one per blockIdx % 8: a single counter serialises a launch's
 * atomics on one address; ABI version 2 skipped the ticket when the grid fitted the chip, which an ordinary launch cannot
 * guarantee: other streams, a CU mask): a waiting block's partners have started, or belong to the units at the dispatch front
 * and start as soon as any older block finishes.  The wait is bounded (~1 s); a block that gives up poisons its
 * unit's output with NaN, records KIVI_ETIMEOUT (kivi_device_error; word 16375 of the counter area) and the launch's last blocks
 * still put every counter back to zero.  At most 8183 units take a sliced form.
 * Otherwise two launches (incl. multi-head rows beyond 16 super-blocks: the sliced one-launch form of such rows is reachable through
 * KIVI_GQA_SLICES(n) only -- measured slower than the two launches):
 *   1. packed qK^T on the matrix pipe + fp16 residual scores + K append (llama_kivi.py:323-337); the epilogue applies
 *      1/sqrt(D) and the mask (:339, :364-372), writes the scaled scores to `scores` and (max, sum exp) of every
 *      512-token segment to `stats`;
 *   2. packed sV on the matrix pipe with the softmax applied on the fly (p = fp16(exp(x - max) / sum), :375) + fp16
 *      window (:377-384) + V append + quantisation of the token leaving the window into the VT layout (:386-399);
 *      rows are cut into slices, partial sums meet in `workspace`.
 * Lengths: Tq packed keys (multiple of 32) + k_res_len residual keys == Tv packed values + v_res_len window values.
 * `stats`: >= B * nh * (ceil(Tq / 512) + 4) * 2 floats.  `workspace`: 64 KiB of arrival counters (zeroed once by the
 * caller) followed by B * nh_kv * 2 * (slices + 1) * (nh / nh_kv) * 128 floats (1 <= slices <= max(1, ceil(Tv / 512));
 * + 1: the slot of the unit's window block).
 * Bounds the step writes against: residual_length (k_res_len < residual_length: the K append goes to row k_res_len of
 * kres), v_window_rows (v_win_start + v_res_len + 1 <= v_window_rows: the V append), vt_superblocks (Tv + 1 <= 512 *
 * vt_superblocks when v_flush), kt_superblocks (Tq <= 512 * kt_superblocks).
 * flags: KIVI_GQA_FORCE_SPLIT = the two-launch form even where the one-launch form applies, KIVI_GQA_FORCE_ROW = the
 * one-launch form for any number of units (rows that fit the LDS as above), KIVI_GQA_SLICES(n) = n slices per row (nh / nh_kv in
 * {4, 8}) -- tests and tuning.  KIVI_GQA_DUMP_SCORES (tests): the one-launch form also writes the fp16 rows its softmax statistics
 * are taken from (scaled, mask added: what the two-launch form leaves in `scores`) to `scores` -- nh / nh_kv in {4, 8}: the product
 * kernel, through a run-time pointer; nh == nh_kv: separate instantiations.
 * kt_range / vt_range: the range flags of the two stores (see above); the V flush may set vt_range.
 */
#define KIVI_GQA_FORCE_SPLIT 1
#define KIVI_GQA_FORCE_ROW 2
#define KIVI_GQA_DUMP_SCORES 8
#define KIVI_GQA_SLICES(n) (((n) & 0xFF) << 8)
/* the fp16 value window is a RING of v_window_rows rows (row of window token t = (v_win_start + t) mod v_window_rows):
 * residual_length + 1 rows suffice and nothing is ever compacted; nh / nh_kv in {1, 4} */
#define KIVI_GQA_WINDOW_RING 4
/*
 * Device-resident step lengths (hipGraph capture).  A decode step's launches depend on six lengths that change every step;
 * with `dyn_step` = a DEVICE pointer to a kivi_mf_step the kernels read them from there, and the launch geometry (grids, LDS,
 * slices) is sized for the step's whole geometry class: every step with the same kivi_mf_step_key -- the same number of
 * 512-token super-blocks of packed keys and values and the same one- / two-launch decision -- can REPLAY the captured
 * launches after the caller has updated the six numbers in device memory.  The host-side lengths passed with the call are the
 * ones of the step being captured (validated as usual); keeping later steps inside the caller's buffers is the caller's job
 * (kivi_mf_step_advance is the same bookkeeping kivi_mf_decode_layer does).  Buffers named by the arguments (q, new key /
 * value, mask with a fixed row pitch, out, scratch) must stay where they are across replays.
 */
typedef struct {
    int64_t Tq, Tv;                                    /* packed keys (multiple of 32), packed values */
    int32_t k_res_len, v_res_len, v_win_start, v_flush;
} kivi_mf_step;
typedef struct {
    int B, nh, nh_kv, D, group_size, bits;
    float inv_scale;
    const void* q; int64_t q_sb, q_sh;
    const void* mask; int64_t mask_sb;
    void* kt; int64_t kt_sb, kt_sh, kt_ss; int64_t Tq;
    void* kres; int64_t kres_sb, kres_sh, kres_st; const void* knew; int64_t knew_sb, knew_sh; int k_res_len;
    void* vt; int64_t vt_sb, vt_sh, vt_ss; int64_t Tv;
    void* vres; int64_t vres_sb, vres_sh, vres_st; int v_win_start, v_res_len;
    const void* vnew; int64_t vnew_sb, vnew_sh; int v_flush;
    void* scores; int64_t s_sb, s_sh;
    void* stats; int64_t stats_bytes;
    void* workspace; int64_t workspace_bytes;
    void* out; int64_t out_sb, out_sh;
    int residual_length; int64_t v_window_rows, kt_superblocks, vt_superblocks;
    int flags;
    void* kt_range; void* vt_range;                    /* B * nh_kv int32 each */
    const void* dyn_step;                              /* device kivi_mf_step or null (lengths from the fields above) */
} kivi_gqa_decode_args;
int kivi_gqa_decode(const kivi_gqa_decode_args* args, kivi_stream_t stream);
/* the launch plan kivi_gqa_decode follows for a step of this geometry: 0 = two launches, S >= 1 = one launch with every row cut
 * into S slices (1: a block per row); -1: bad arguments.  dyn != 0: the plan of the step's whole geometry class (dyn_step). */
int kivi_mf_launch_plan(int B, int nh, int nh_kv, int64_t Tq, int k_res_len, int residual_length, int flags, int bits, int dyn);

/* ------------------------------------------------------ layer step --- */

/*
 * One decode step of one layer in ONE host call: kivi_gemv_k_paged + kivi_decode_attend + the cache bookkeeping of
 * the hook (models/llama_kivi.py:314-399): lengths, the K flush when the residual reaches R tokens (:343-356, packed
 * in place at the end of the prefix), the V flush of the token leaving the window (:386-399), window compaction.
 * Stateless: the descriptor names the caller's buffers (all device memory, layouts as in kivi_gemv_k_paged /
 * kivi_decode_attend; `scores` = (B, nh, s_pitch) fp16 scratch rows), `state` = {k_quant_len, k_res_len, v_quant_len,
 * v_win_start, v_res_len, kv_seq_len} is read and advanced.  KIVI_EUNSUPPORTED = no tuned kernel for the shape, state
 * untouched (except a completed window compaction): compose the step from the entry points above.
 * Failure atomicity: every argument the K flush could reject is validated before the first launch, so an argument
 * error never leaves a half-done step; `state` is written after each phase that has been enqueued (window compaction,
 * attend launch, K flush).  If the K flush LAUNCH itself fails after the attend launch was enqueued, the state reads
 * k_res_len == residual_length ("flush pending") and the next call performs that flush first.
 */
typedef struct {
    int B, nh_kv, D, k_bits, v_bits, group_size, residual_length;
    float inv_scale;                                   /* 1 / sqrt(D) */
    int64_t cap, page_tokens, v_window_rows, s_pitch;
    void* k_code; int64_t kc_sb, kc_sh, kc_sp, kc_sr;  /* (B, nh_kv, P, D, page_tokens/fpi) int32 */
    void* k_scale; void* k_mn; int64_t ks_sb, ks_sh, ks_sp, ks_sr;
    void* k_res; int64_t kr_sb, kr_sh, kr_st;          /* (B, nh_kv, R, D) fp16 */
    void* v_code; int64_t vc_sb, vc_sh, vc_sr;         /* (B, nh_kv, cap, D/fpi) int32 */
    void* v_scale; void* v_mn; int64_t vs_sb, vs_sh, vs_sr;
    void* v_res; int64_t vr_sb, vr_sh, vr_st;          /* (B, nh_kv, v_window_rows, D) fp16, contiguous */
    void* scores; int64_t s_sb, s_sh;
    void* workspace; int64_t workspace_bytes;          /* as in kivi_decode_attend_args */
} kivi_layer_desc;
int kivi_decode_layer(const kivi_layer_desc* layer, int64_t* state, const void* q, int64_t q_sb, int64_t q_sh, int nh,
                      const void* knew, int64_t kn_sb, int64_t kn_sh, const void* vnew, int64_t vn_sb, int64_t vn_sh,
                      const void* mask, int64_t mask_sb, void* out, int64_t out_sb, int64_t out_sh, kivi_stream_t stream);

/*
 * The same for a cache in the KT / VT layouts (kivi_gqa_decode + bookkeeping + the K flush through kivi_kt_pack every
 * residual_length steps + window compaction), same state array and the same atomicity contract.
 */
typedef struct {
    int B, nh_kv, D, bits, group_size, residual_length;
    float inv_scale;
    int64_t cap, v_window_rows, s_pitch;               /* cap: tokens the stores can hold (512 * super-blocks) */
    void* kt; int64_t kt_sb, kt_sh, kt_ss;
    void* vt; int64_t vt_sb, vt_sh, vt_ss;
    void* k_res; int64_t kr_sb, kr_sh, kr_st;          /* (B, nh_kv, R, D) fp16 */
    void* v_res; int64_t vr_sb, vr_sh, vr_st;          /* (B, nh_kv, v_window_rows, D) fp16, contiguous */
    void* scores; int64_t s_sb, s_sh;                  /* (B, nh, s_pitch) fp16 scratch rows */
    void* stats; int64_t stats_bytes;
    void* workspace; int64_t workspace_bytes;
    int flags;                                         /* KIVI_GQA_* */
    void* kt_range; void* vt_range;                    /* range flags of the two stores: B * nh_kv int32 each */
} kivi_mf_layer_desc;
int kivi_mf_decode_layer(const kivi_mf_layer_desc* layer, int64_t* state, const void* q, int64_t q_sb, int64_t q_sh, int nh,
                         const void* knew, int64_t kn_sb, int64_t kn_sh, const void* vnew, int64_t vn_sb, int64_t vn_sh,
                         const void* mask, int64_t mask_sb, void* out, int64_t out_sb, int64_t out_sh, kivi_stream_t stream);

/*
 * The attend phase of the same step with device-resident lengths, for hipGraph capture (see kivi_mf_step above): launches
 * kivi_gqa_decode with dyn_step = `dev_step` and NOTHING else -- no bookkeeping, no K flush.  `host_step` = the lengths of the
 * step being enqueued (what *dev_step will hold when the launches run).  The caller, per step: writes the lengths to
 * *dev_step, replays (or calls this), kivi_mf_step_advance(host_step, ...), and when that returns 1 packs the full K residual
 * with kivi_kt_pack at token offset Tq and sets Tq += residual_length, k_res_len = 0.  A captured step may be replayed while
 * kivi_mf_step_key of the current lengths equals the key at capture time (and the cache buffers have not been reallocated).
 */
int kivi_mf_decode_layer_dyn(const kivi_mf_layer_desc* layer, const kivi_mf_step* host_step, const void* dev_step, const void* q,
                             int64_t q_sb, int64_t q_sh, int nh, const void* knew, int64_t kn_sb, int64_t kn_sh, const void* vnew,
                             int64_t vn_sb, int64_t vn_sh, const void* mask, int64_t mask_sb, void* out, int64_t out_sb,
                             int64_t out_sh, kivi_stream_t stream);
/* geometry class of a step (-1: bad arguments): the super-block counts of both stores and whether the step flushes a value; the
 * launch plan (one launch / S slices / two launches) is a function of the class and of constants of the call (shape, bits, flags) */
int64_t kivi_mf_step_key(const kivi_mf_step* step, int B, int nh, int nh_kv, int residual_length, int flags);
/* lengths after the attend phase of one step (llama_kivi.py:333-336, :377, :386-399); returns 1 when the K residual is full
 * (the K flush of :343-356 is due), 0 otherwise, < 0 on inconsistent lengths.  window_rows: rows of the ring window buffer. */
int kivi_mf_step_advance(kivi_mf_step* step, int residual_length, int64_t window_rows);
/* *dev_step <- *host_step by a one-thread kernel on `stream` (the values travel as kernel arguments: the host struct may be
 * reused as soon as the call returns, nothing synchronises) */
int kivi_mf_step_upload(const kivi_mf_step* host_step, void* dev_step, kivi_stream_t stream);

/* ------------------------------------------------- tuning / bench hooks --- */

/* Kernel variants of kivi_gemv_k (same arguments + variant id; -1 = the default heuristic).
 * Used by bench.py and the parity tests to check every variant; not part of the drop-in surface. */
int kivi_gemv_k_num_variants(void);
const char* kivi_gemv_k_variant_name(int variant);
int kivi_gemv_k_variant(int variant, const void* q, int64_t q_sb, int64_t q_sh, const void* code, int64_t code_sb,
                        int64_t code_sh, int64_t code_sr, const void* scale, const void* mn, int64_t sm_sb,
                        int64_t sm_sh, int64_t sm_sr, void* out, int64_t out_sb, int64_t out_sh, int B, int nh,
                        int nh_kv, int D, int64_t T, int group_size, int bits, kivi_stream_t stream);

int kivi_gemv_v_num_variants(void);
const char* kivi_gemv_v_variant_name(int variant);
int kivi_gemv_v_variant(int variant, const void* a, int64_t a_sb, int64_t a_sh, const void* code, int64_t code_sb,
                        int64_t code_sh, int64_t code_sr, const void* scale, const void* mn, int64_t sm_sb,
                        int64_t sm_sh, int64_t sm_sr, void* out, int64_t out_sb, int64_t out_sh, int B, int nh,
                        int nh_kv, int64_t Tv, int D, int group_size, int bits, kivi_stream_t stream);

/* Per-dispatch timing: the NEXT fused-GEMV launch issued by this thread stamps `start` / `stop` (hipEvent_t
 * from kivi_event_create) with the dispatch's own begin / end (hipExtLaunchKernelGGL), i.e. what a profiler
 * reports as the kernel duration.  kivi_event_elapsed_us synchronises on `stop`. */
void* kivi_event_create(void);
void kivi_event_destroy(void* event);
void kivi_set_launch_events(void* start, void* stop);
float kivi_event_elapsed_us(void* start, void* stop);
/* source text of the kernel instantiation the last consumed event pair bracketed ("" if none yet) */
const char* kivi_last_timed_kernel(void);
/* Diagnostic, -DKIVI_TUNING builds only (the product library accepts the call and ignores it): device buffer of (blocks x 4 waves
 * x 16) uint64 that the stamping instantiations of the row kernels fill with per-wave phase time stamps (shader clock; slot 1 =
 * 100 MHz realtime at entry); null switches it off (default). */
void kivi_debug_set_stamps(void* device_buffer);

#ifdef __cplusplus
}
#endif
#endif /* KIVI_HIP_H */

// qK^T over the packed per-channel K cache (hook-state layout), gfx950.
//
// Replaces cuda_bmm_fA_qB_outer + bgemv{2,4}_kernel_outer_dim of the reference
// (quant/matmul.py:178-219, quant/csrc/gemv_cuda.cu:265-427) for the call at
// models/llama_kivi.py:324, reading K_code_T (B,nh_kv,D,T/fpi) directly.
//
// Mapping (wave64).  The dot axis d is the LOOP axis; the packed axis t is
// spread over lanes, so a lane owns WPL consecutive words (= WPL*fpi tokens)
// of every channel row and there is NO cross-lane reduction:
//   one wave-instruction reads 64*WPL*4 contiguous bytes of one code row,
//   plus 64*NGL halves of the matching scale and mn rows.
// Per code the VALU work is one mask (shared by two codes) + one v_fma_mix_f32;
// scale is folded into q once per (row, group) and the zero-point term
// sum_d q[d]*mn[d,G] is hoisted out of the per-token work (one fma per group).
// DSPLIT waves of a block split the channel range and combine through LDS.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "kivi_common.h"
#include "kivi_gemv_k_dev.h"

namespace {

template <int BITS, int G, int WPL, int DSPLIT, int R, int U, int MODE, bool NT>
__global__ __launch_bounds__(256) void gemv_k_kernel(const GemvKArgs a) {
    if ((int)blockIdx.x < a.res_blocks) {   // block-uniform role switch
        k_residual_role<R>(a, (int)blockIdx.x);
        return;
    }
    k_tile_body<BITS, G, WPL, DSPLIT, R, U, MODE, NT>(a, (int)blockIdx.x - a.res_blocks, nullptr);
}

// Shape-agnostic fallback: one thread per output word, any group size that is a
// multiple of fpi, any D.  Same arithmetic, BFE unpack.
template <int BITS>
__global__ __launch_bounds__(256) void gemv_k_generic(const GemvKArgs a, int G) {
    constexpr int FPI = 32 / BITS;
    const int wblocks = (int)((a.Tw + 255) / 256);
    const int bh = blockIdx.x / wblocks;
    const int64_t w = (int64_t)(blockIdx.x - bh * wblocks) * 256 + threadIdx.x;
    const int b = bh / a.nh, h = bh - b * a.nh;
    const int hk = h / a.ratio;
    if (w >= a.Tw) return;
    const int64_t page = a.page_words ? w / a.page_words : 0;
    const int64_t win = a.page_words ? w - page * a.page_words : w;
    const int64_t g = (win * FPI) / G;
    const uint32_t* cp = a.code + b * a.code_sb + hk * a.code_sh + page * a.code_sp + win;
    const uint16_t* sp = a.scale + b * a.sm_sb + hk * a.sm_sh + page * a.sm_sp + g;
    const uint16_t* mp = a.mn + b * a.sm_sb + hk * a.sm_sh + page * a.sm_sp + g;
    const uint16_t* qp = a.q + b * a.q_sb + (int64_t)h * a.q_sh;
    float acc[FPI];
#pragma unroll
    for (int p = 0; p < FPI; p++) acc[p] = 0.f;
    float z = 0.f;
    for (int d = 0; d < a.D; d++) {
        const float qd = h2f_bits(qp[d]);
        const float qs = qd * h2f_bits(sp[(int64_t)d * a.sm_sr]);
        z = __builtin_fmaf(qd, h2f_bits(mp[(int64_t)d * a.sm_sr]), z);
        accum_word<BITS, KIVI_UNPACK_BFE>(cp[(int64_t)d * a.code_sr], qs, acc);
    }
    uint16_t* op = a.out + b * a.out_sb + (int64_t)h * a.out_sh + w * FPI;
#pragma unroll
    for (int p = 0; p < FPI; p++) op[p] = f2h_bits(acc[p] + z);
}

// ------------------------------------------------------------------ host side

typedef void (*KLaunch)(const GemvKArgs&, dim3, hipStream_t);

template <int BITS, int G, int WPL, int DSPLIT, int R, int U, int MODE, bool NT>
void launch_k(const GemvKArgs& a, dim3 grid, hipStream_t s) {
    static const char* xl = KIVI_TUNE_ENV("KIVI_K_EXTRA_LDS");   // diagnostic: cap the blocks per CU by reserving extra LDS
    KIVI_LAUNCH_LDS((gemv_k_kernel<BITS, G, WPL, DSPLIT, R, U, MODE, NT>), grid, dim3(256), xl ? (size_t)atoi(xl) : 0, s, a);
}

struct KVariant {
    const char* name;
    int bits, G, wpl, dsplit, R, U, mode, nt;
    KLaunch fn;
};

#define KV(BITS, G, WPL, DS, R, U, MODE, NT)                                                        \
    {"k_b" #BITS "_g" #G "_w" #WPL "_ds" #DS "_r" #R "_u" #U "_m" #MODE "_nt" #NT, BITS, G, WPL, DS, R, U, \
     MODE, NT, launch_k<BITS, G, WPL, DS, R, U, MODE, (NT != 0)>}

const KVariant k_variants[] = {
    // ---- 2-bit, g=32, MHA: the north-star shape.  Table order = dispatch preference (measured, profiles/).
    KV(2, 32, 2, 4, 1, 4, 2, 1),
    KV(2, 32, 4, 2, 1, 4, 2, 0),
    KV(2, 32, 4, 2, 1, 4, 2, 1),
    KV(2, 32, 4, 2, 1, 8, 2, 0),
    KV(2, 32, 4, 2, 1, 8, 2, 1),
    KV(2, 32, 4, 1, 1, 4, 2, 0),
    KV(2, 32, 4, 1, 1, 4, 2, 1),
    KV(2, 32, 4, 1, 1, 8, 2, 0),
    KV(2, 32, 4, 1, 1, 8, 2, 1),
    KV(2, 32, 4, 4, 1, 4, 2, 0),
    KV(2, 32, 4, 4, 1, 8, 2, 1),
    KV(2, 32, 2, 4, 1, 4, 2, 0),
    KV(2, 32, 2, 4, 1, 8, 2, 0),
    KV(2, 32, 2, 4, 1, 8, 2, 1),
    KV(2, 32, 2, 2, 1, 8, 2, 0),
    KV(2, 32, 2, 2, 1, 8, 2, 1),
    KV(2, 32, 2, 1, 1, 8, 2, 0),
    KV(2, 32, 2, 1, 1, 8, 2, 1),
    KV(2, 32, 2, 1, 1, 16, 2, 1),
    KV(2, 32, 2, 4, 1, 2, 2, 1),
    KV(2, 32, 4, 4, 1, 2, 2, 1),
    KV(2, 32, 4, 2, 1, 2, 2, 1),
    KV(2, 32, 2, 2, 1, 4, 2, 1),
    // the plain bit-field unpack (v_bfe + v_cvt + v_fma) on two geometries: an independent arithmetic the parity tests cross-check
    // the FMA-mix kernels with
    KV(2, 32, 4, 2, 1, 4, 0, 0),
    KV(2, 32, 2, 4, 1, 4, 0, 0),
#ifdef KIVI_TUNING
    // -DKIVI_TUNING builds only (tools/build_variant.sh): unpack-strategy A/B forms the dispatch never picks ...
    KV(2, 32, 2, 4, 1, 4, 4, 1),
    KV(2, 32, 2, 4, 1, 2, 4, 1),
    KV(2, 32, 4, 2, 1, 4, 4, 1),
    KV(2, 32, 4, 2, 1, 4, 1, 0),
    KV(2, 32, 2, 4, 1, 4, 1, 0),
    // ... and the memory-side ceiling of each geometry: the unpack skipped, WRONG RESULTS (diagnostic)
    KV(2, 32, 2, 4, 1, 4, 3, 1),
    KV(2, 32, 2, 2, 1, 8, 3, 1),
    KV(2, 32, 2, 1, 1, 8, 3, 1),
    KV(2, 32, 4, 4, 1, 4, 3, 1),
    KV(2, 32, 4, 2, 1, 4, 3, 1),
    KV(2, 32, 4, 1, 1, 8, 3, 1),
#endif
    // ---- 2-bit, other group sizes
    KV(2, 64, 2, 4, 1, 4, 2, 1),
    KV(2, 64, 4, 2, 1, 4, 2, 0),
    KV(2, 128, 2, 4, 1, 4, 2, 1),
    KV(2, 128, 4, 2, 1, 4, 2, 0),
    // ---- 4-bit (fpi = 8: four words per lane give the same 32 tokens per lane as 2-bit w2)
    KV(4, 32, 4, 4, 1, 4, 2, 1),
    KV(4, 32, 2, 4, 1, 4, 2, 1),
    KV(4, 32, 4, 2, 1, 4, 2, 1),
    KV(4, 64, 4, 4, 1, 4, 2, 1),
    KV(4, 64, 2, 4, 1, 4, 2, 1),
    KV(4, 128, 4, 4, 1, 4, 2, 1),
    KV(4, 128, 2, 4, 1, 4, 2, 1),
    KV(4, 32, 4, 2, 1, 4, 0, 0),
    // ---- GQA: R query heads share every unpacked code (R*TPL accumulators per lane).  The unpack is R FMAs per
    // code, so the cheaper-per-FMA fp32-subnormal form (mask + R x v_fmac_f32) beats the FMA-mix form here.
    KV(2, 32, 1, 2, 4, 4, 4, 1),
    KV(2, 32, 1, 1, 4, 4, 4, 1),
    KV(2, 32, 2, 2, 2, 4, 4, 1),
    KV(2, 32, 1, 1, 8, 4, 4, 1),
    KV(2, 32, 2, 1, 4, 2, 4, 1),
    KV(4, 32, 2, 2, 4, 4, 4, 1),
    KV(2, 64, 1, 2, 4, 4, 4, 1),
    KV(2, 128, 1, 2, 4, 4, 4, 1),
    KV(2, 32, 1, 1, 4, 4, 2, 1),
    KV(2, 32, 1, 2, 4, 4, 2, 1),
    KV(2, 32, 2, 2, 2, 4, 2, 1),
    KV(2, 32, 2, 4, 2, 4, 2, 1),
    KV(2, 32, 1, 1, 8, 4, 2, 1),
    KV(2, 32, 1, 1, 2, 8, 2, 1),
    KV(4, 32, 2, 2, 4, 4, 2, 1),
    KV(4, 32, 2, 2, 2, 4, 2, 1),
    KV(2, 64, 1, 2, 4, 4, 2, 1),
    KV(2, 128, 1, 2, 4, 4, 2, 1),
    KV(2, 64, 2, 2, 2, 4, 2, 1),
};
constexpr int k_nvariants = sizeof(k_variants) / sizeof(k_variants[0]);

int ngl_of(const KVariant& v, int bits, int G) {
    const int tpl = v.wpl * (32 / bits);
    return tpl >= G ? tpl / G : 1;
}

bool k_variant_fits(const KVariant& v, const GemvKArgs& a, int bits, int G) {
    if (v.bits != bits || v.G != G) return false;
    if (a.ratio % v.R != 0 && !(v.R == 1)) return false;
    if (a.D > KQ_MAXD) return false;
    // one tile's D rows must fit a 32-bit buffer descriptor
    if ((int64_t)a.D * a.code_sr * 4 >= ((int64_t)1 << 31) || (int64_t)a.D * a.sm_sr * 2 >= ((int64_t)1 << 31)) return false;
    if (a.page_words) {   // pages are whole tiles, aligned like rows
        if (a.page_words % (64 * v.wpl)) return false;
        if ((a.code_sp % v.wpl) || (a.sm_sp % ngl_of(v, bits, G))) return false;
    }
    // q is read as aligned fp16 pairs with scalar loads; every wave's channel range starts on an even channel
    if ((a.D % (2 * v.dsplit)) || (a.q_sh % 2) || (a.q_sb % 2) || ((uintptr_t)a.q % 4)) return false;
    const int fpi = 32 / bits;
    const int tpl = v.wpl * fpi;
    const int ngl = tpl >= G ? tpl / G : 1;
    // vector-load alignment: every row start must be a multiple of the vector width
    if (a.Tw % v.wpl) return false;
    if ((a.code_sr % v.wpl) || (a.code_sh % v.wpl) || (a.code_sb % v.wpl)) return false;
    if (((uintptr_t)a.code) % (4 * v.wpl)) return false;
    if ((a.sm_sr % ngl) || (a.sm_sh % ngl) || (a.sm_sb % ngl)) return false;
    if (((uintptr_t)a.scale) % (2 * ngl) || ((uintptr_t)a.mn) % (2 * ngl)) return false;
    // 16-byte output stores
    if ((a.out_sh % 8) || (a.out_sb % 8) || ((uintptr_t)a.out % 16) || (a.T % 8)) return false;
    return true;
}

int k_run(int variant, GemvKArgs a, int B, int nh_kv, int G, int bits, hipStream_t s) {
    (void)nh_kv;
    const int fpi = 32 / bits;
    if (variant >= 0) {
        KIVI_REQUIRE(variant < k_nvariants, KIVI_EINVAL, "kivi_gemv_k_variant: no variant %d", variant);
        const KVariant& v = k_variants[variant];
        KIVI_REQUIRE(k_variant_fits(v, a, bits, G), KIVI_EINVAL,
                     "kivi_gemv_k_variant: %s does not fit this problem (bits=%d g=%d ratio=%d Tw=%lld)", v.name, bits,
                     G, a.ratio, (long long)a.Tw);
        const int tiles = (int)((a.Tw + 64 * v.wpl - 1) / (64 * v.wpl));
        const int tpb = 4 / v.dsplit;
        a.units_per_b = a.nh / v.R;
        a.tile_blocks = (tiles + tpb - 1) / tpb;
        int64_t blocks = (int64_t)B * a.units_per_b * a.tile_blocks;
        a.res_blocks = 0;
        if (a.main_blocks >= 0) {   // fused decode step: one residual block per (b, head unit), scheduled first
            a.main_blocks = (int)blocks;
            a.res_blocks = B * a.units_per_b;
            blocks += a.res_blocks;
        }
        v.fn(a, dim3((unsigned)blocks), s);
        return kivi_launch_status(v.name);
    }
    // heuristic: first fitting variant with the largest usable R; table order = preference
    int best = -1;
    static const char* forced = KIVI_TUNE_ENV("KIVI_GEMV_K_VARIANT");   // tuning aid: force a variant by name if it fits
    if (forced)
        for (int i = 0; i < k_nvariants; i++)
            if (!strcmp(forced, k_variants[i].name) && k_variant_fits(k_variants[i], a, bits, G))
                return k_run(i, a, B, nh_kv, G, bits, s);
    for (int i = 0; i < k_nvariants; i++) {
        const KVariant& v = k_variants[i];
        // production unpacks: FMA-mix, and the fp32-subnormal form for GQA units; the others are A/B references
        if (!(v.mode == KIVI_UNPACK_MIX || (v.mode == KIVI_UNPACK_DEN32 && v.R > 1))) continue;
        if (!k_variant_fits(v, a, bits, G)) continue;
        if (a.ratio % v.R) continue;
        if (best < 0 || v.R > k_variants[best].R) best = i;
    }
    if (best >= 0) return k_run(best, a, B, nh_kv, G, bits, s);
    KIVI_REQUIRE(a.main_blocks < 0, KIVI_EUNSUPPORTED,
                 "kivi_decode_scores: no tuned kernel for this shape (bits=%d g=%d D=%d); use the unfused path", bits, G,
                 a.D);
    // generic fallback
    dim3 grid((unsigned)(((a.Tw + 255) / 256) * (int64_t)B * a.nh));
    if (bits == 2) hipLaunchKernelGGL(gemv_k_generic<2>, grid, dim3(256), 0, s, a, G);
    else hipLaunchKernelGGL(gemv_k_generic<4>, grid, dim3(256), 0, s, a, G);
    (void)fpi;
    return kivi_launch_status("gemv_k_generic");
}

}  // namespace

extern "C" int kivi_gemv_k_num_variants(void) { return k_nvariants; }
extern "C" const char* kivi_gemv_k_variant_name(int v) {
    return (v >= 0 && v < k_nvariants) ? k_variants[v].name : "";
}

static int k_entry(int variant, int64_t page_tokens, int64_t code_sp, int64_t sm_sp, const void* q, int64_t q_sb,
                   int64_t q_sh, const void* code, int64_t code_sb, int64_t code_sh, int64_t code_sr, const void* scale,
                   const void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr, void* out, int64_t out_sb,
                   int64_t out_sh, int B, int nh, int nh_kv, int D, int64_t T, int group_size, int bits,
                   kivi_stream_t stream) {
    GemvKArgs a;
    int rc = k_check_and_fill(a, q, q_sb, q_sh, code, code_sb, code_sh, code_sr, scale, mn, sm_sb, sm_sh, sm_sr, out,
                              out_sb, out_sh, B, nh, nh_kv, D, T, group_size, bits);
    if (rc) return rc;
    if (page_tokens) {
        const int fpi = 32 / bits;
        KIVI_REQUIRE(page_tokens > 0 && page_tokens % group_size == 0 && page_tokens % fpi == 0, KIVI_EINVAL,
                     "kivi_gemv_k_paged: page_tokens=%lld must be a multiple of group_size=%d", (long long)page_tokens,
                     group_size);
        a.page_words = page_tokens / fpi;
        a.page_groups = page_tokens / group_size;
        a.code_sp = code_sp;
        a.sm_sp = sm_sp;
    }
    if (T == 0) return 0;
    return k_run(variant, a, B, nh_kv, group_size, bits, (hipStream_t)stream);
}

extern "C" int kivi_gemv_k_variant(int variant, const void* q, int64_t q_sb, int64_t q_sh, const void* code,
                                   int64_t code_sb, int64_t code_sh, int64_t code_sr, const void* scale,
                                   const void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr, void* out,
                                   int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv, int D, int64_t T,
                                   int group_size, int bits, kivi_stream_t stream) {
    return k_entry(variant, 0, 0, 0, q, q_sb, q_sh, code, code_sb, code_sh, code_sr, scale, mn, sm_sb, sm_sh, sm_sr, out,
                   out_sb, out_sh, B, nh, nh_kv, D, T, group_size, bits, stream);
}

extern "C" int kivi_gemv_k(const void* q, int64_t q_sb, int64_t q_sh, const void* code, int64_t code_sb,
                           int64_t code_sh, int64_t code_sr, const void* scale, const void* mn, int64_t sm_sb,
                           int64_t sm_sh, int64_t sm_sr, void* out, int64_t out_sb, int64_t out_sh, int B, int nh,
                           int nh_kv, int D, int64_t T, int group_size, int bits, kivi_stream_t stream) {
    return k_entry(-1, 0, 0, 0, q, q_sb, q_sh, code, code_sb, code_sh, code_sr, scale, mn, sm_sb, sm_sh, sm_sr, out,
                   out_sb, out_sh, B, nh, nh_kv, D, T, group_size, bits, stream);
}

extern "C" int kivi_gemv_k_paged(int variant, int64_t page_tokens, int64_t code_sp, int64_t sm_sp, const void* q,
                                 int64_t q_sb, int64_t q_sh, const void* code, int64_t code_sb, int64_t code_sh,
                                 int64_t code_sr, const void* scale, const void* mn, int64_t sm_sb, int64_t sm_sh,
                                 int64_t sm_sr, void* out, int64_t out_sb, int64_t out_sh, int B, int nh, int nh_kv,
                                 int D, int64_t T, int group_size, int bits, kivi_stream_t stream) {
    KIVI_REQUIRE(page_tokens > 0, KIVI_EINVAL, "kivi_gemv_k_paged: page_tokens must be positive");
    return k_entry(variant, page_tokens, code_sp, sm_sp, q, q_sb, q_sh, code, code_sb, code_sh, code_sr, scale, mn, sm_sb,
                   sm_sh, sm_sr, out, out_sb, out_sh, B, nh, nh_kv, D, T, group_size, bits, stream);
}

extern "C" int kivi_decode_scores(int64_t page_tokens, int64_t code_sp, int64_t sm_sp, const void* q, int64_t q_sb,
                                  int64_t q_sh, const void* code, int64_t code_sb, int64_t code_sh, int64_t code_sr,
                                  const void* scale, const void* mn, int64_t sm_sb, int64_t sm_sh, int64_t sm_sr,
                                  void* kres, int64_t kres_sb, int64_t kres_sh, int64_t kres_st, const void* knew,
                                  int64_t knew_sb, int64_t knew_sh, int res_len, void* out, int64_t out_sb,
                                  int64_t out_sh, int B, int nh, int nh_kv, int D, int64_t T, int group_size, int bits,
                                  kivi_stream_t stream) {
    GemvKArgs a;
    int rc = k_check_and_fill(a, q, q_sb, q_sh, code, code_sb, code_sh, code_sr, scale, mn, sm_sb, sm_sh, sm_sr, out,
                              out_sb, out_sh, B, nh, nh_kv, D, T, group_size, bits);
    if (rc) return rc;
    const int fpi = 32 / bits;
    KIVI_REQUIRE(page_tokens > 0 && page_tokens % group_size == 0 && page_tokens % fpi == 0, KIVI_EINVAL,
                 "kivi_decode_scores: page_tokens=%lld must be a positive multiple of group_size=%d", (long long)page_tokens,
                 group_size);
    KIVI_REQUIRE(res_len >= 0 && kres && knew, KIVI_EINVAL, "kivi_decode_scores: residual buffers missing");
    KIVI_REQUIRE(D <= 256, KIVI_EUNSUPPORTED, "kivi_decode_scores: head_dim %d > 256", D);
    KIVI_REQUIRE(D % 2 == 0 && kres_sb % 2 == 0 && kres_sh % 2 == 0 && kres_st % 2 == 0 && knew_sb % 2 == 0 &&
                     knew_sh % 2 == 0 && (uintptr_t)kres % 4 == 0 && (uintptr_t)knew % 4 == 0,
                 KIVI_EALIGN, "kivi_decode_scores: residual rows must be 4-byte aligned");
    a.page_words = page_tokens / fpi;
    a.page_groups = page_tokens / group_size;
    a.code_sp = code_sp;
    a.sm_sp = sm_sp;
    a.main_blocks = 0;
    a.kres = (const uint16_t*)kres; a.kres_sb = kres_sb; a.kres_sh = kres_sh; a.kres_st = kres_st;
    a.knew = (const uint16_t*)knew; a.knew_sb = knew_sb; a.knew_sh = knew_sh;
    a.res_len = res_len;
    return k_run(-1, a, B, nh_kv, group_size, bits, (hipStream_t)stream);
}

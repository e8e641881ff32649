// One decode step of one layer in ONE host call: cache bookkeeping of the attention hook
// (models/llama_kivi.py:314-399 -- residual / window lengths, the K flush every R tokens, the V flush of the token
// leaving the window) around the two fused launches.  kivi_amd/attention.py does the same in Python with ~40 us of
// interpreter + ctypes work per layer; at small batch the GPU finishes a step faster than that, so the host side of
// the step lives here.  Stateless: the caller owns the buffers (descriptor) and the six lengths (state array).
#include "kivi_common.h"

extern "C" int kivi_decode_layer(const kivi_layer_desc* L, int64_t* st, const void* q, int64_t q_sb, int64_t q_sh, int nh,
                                 const void* knew, int64_t kn_sb, int64_t kn_sh, const void* vnew, int64_t vn_sb,
                                 int64_t vn_sh, const void* mask, int64_t mask_sb, void* out, int64_t out_sb,
                                 int64_t out_sh, kivi_stream_t stream) {
    KIVI_REQUIRE(L && st && q && knew && vnew && out, KIVI_EINVAL, "kivi_decode_layer: null argument");
    int64_t Tq = st[0], kres = st[1], Tv = st[2], wstart = st[3], vres = st[4], kv = st[5];
    const int R = L->residual_length;
    KIVI_REQUIRE(R > 0 && Tq >= 0 && kres >= 0 && kres <= R && Tv >= 0 && wstart >= 0 && vres >= 0 && vres <= R &&
                     kv == Tq + kres && kv == Tv + vres,
                 KIVI_EINVAL, "kivi_decode_layer: inconsistent lengths (Tq=%lld kres=%lld Tv=%lld vres=%lld kv=%lld)",
                 (long long)Tq, (long long)kres, (long long)Tv, (long long)vres, (long long)kv);
    // Everything the K flush (kivi_quant_pack_k_tmajor, below) can reject is checked HERE, before anything is launched:
    // a step is either refused with `state` untouched or committed phase by phase (state is written after every phase
    // that has been enqueued, so the caller's lengths always describe the buffers).
    KIVI_REQUIRE((L->k_bits == 2 || L->k_bits == 4) && (L->v_bits == 2 || L->v_bits == 4), KIVI_EINVAL,
                 "kivi_decode_layer: k_bits / v_bits must be 2 or 4 (matmul.py:215), got %d / %d", L->k_bits, L->v_bits);
    KIVI_REQUIRE(L->group_size > 0 && L->group_size % (32 / L->k_bits) == 0 && L->group_size % (32 / L->v_bits) == 0,
                 KIVI_EINVAL, "kivi_decode_layer: group_size %d must be a positive multiple of the codes per word", L->group_size);
    KIVI_REQUIRE(L->B > 0 && L->nh_kv > 0 && L->D > 0 && nh > 0 && nh % L->nh_kv == 0, KIVI_EINVAL,
                 "kivi_decode_layer: bad shape (B=%d nh=%d nh_kv=%d D=%d)", L->B, nh, L->nh_kv, L->D);
    KIVI_REQUIRE((int64_t)(R / L->group_size) * L->B * L->nh_kv < ((int64_t)1 << 31), KIVI_EINVAL,
                 "kivi_decode_layer: K flush grid too large");
    KIVI_REQUIRE(L->k_code && L->k_scale && L->k_mn && L->k_res && L->v_code && L->v_scale && L->v_mn && L->v_res && L->scores,
                 KIVI_EINVAL, "kivi_decode_layer: null cache buffer in the descriptor");
    KIVI_REQUIRE(kv + 1 <= L->cap && Tv + 1 <= L->cap, KIVI_EINVAL, "kivi_decode_layer: cache capacity %lld exceeded",
                 (long long)L->cap);
    KIVI_REQUIRE(R % L->group_size == 0 && L->page_tokens % R == 0, KIVI_EINVAL,
                 "kivi_decode_layer: residual_length must be a multiple of group_size and divide the page");
    KIVI_REQUIRE(kv + 1 <= L->s_pitch, KIVI_EINVAL, "kivi_decode_layer: score rows too short");
    hipStream_t s = (hipStream_t)stream;

    // K flush (llama_kivi.py:343-356): quantise the R residual tokens in place, at token offset Tq of the packed prefix
    auto flush_k = [&]() -> int {
        const int64_t page = Tq / L->page_tokens, off = Tq - page * L->page_tokens;
        const int kfpi = 32 / L->k_bits;
        const int rc = kivi_quant_pack_k_tmajor(
            L->k_res, L->kr_sb, L->kr_sh, L->kr_st, (char*)L->k_code + (size_t)page * L->kc_sp * 4, L->kc_sb, L->kc_sh,
            L->kc_sr, off / kfpi, (char*)L->k_scale + (size_t)page * L->ks_sp * 2, (char*)L->k_mn + (size_t)page * L->ks_sp * 2,
            L->ks_sb, L->ks_sh, L->ks_sr, off / L->group_size, L->B, L->nh_kv, R, L->D, L->group_size, L->k_bits, stream);
        if (rc) return rc;
        Tq += R;
        kres = 0;
        st[0] = Tq; st[1] = 0;
        return 0;
    };
    if (kres == R) {   // a previous call committed its attend phase but its K flush launch failed: finish that first
        const int rc = flush_k();
        if (rc) return rc;
    }

    if (wstart + vres + 1 > L->v_window_rows) {
        // live rows to the front of the window buffer (every ~R steps): [wstart, wstart+vres) and [0, vres) never
        // overlap because wstart + vres == window rows >= 2R + 1 and vres <= R
        KIVI_REQUIRE(L->vr_sb == (int64_t)L->nh_kv * L->vr_sh && wstart >= vres, KIVI_EUNSUPPORTED,
                     "kivi_decode_layer: window buffer layout not compactable in place");
        const size_t pitch = (size_t)L->vr_sh * 2, width = (size_t)vres * L->vr_st * 2;
        if (vres) {
            const hipError_t e = hipMemcpy2DAsync(L->v_res, pitch, (const char*)L->v_res + (size_t)wstart * L->vr_st * 2,
                                                  pitch, width, (size_t)L->B * L->nh_kv, hipMemcpyDeviceToDevice, s);
            KIVI_REQUIRE(e == hipSuccess, (int)e, "kivi_decode_layer: window compaction: %s", hipGetErrorString(e));
        }
        wstart = 0;
        st[3] = 0;   // a compaction is complete on its own: commit it even if a later launch is refused
    }
    int rc;
    const int flush = vres + 1 > R;
    kivi_decode_attend_args a;
    a.q = q; a.q_sb = q_sb; a.q_sh = q_sh;
    a.kres = L->k_res; a.kres_sb = L->kr_sb; a.kres_sh = L->kr_sh; a.kres_st = L->kr_st;
    a.knew = knew; a.knew_sb = kn_sb; a.knew_sh = kn_sh; a.k_res_len = (int)kres;
    a.scores = L->scores; a.s_sb = L->s_sb; a.s_sh = L->s_sh;
    a.inv_scale = L->inv_scale; a.mask = mask; a.mask_sb = mask_sb;
    a.v_code = L->v_code; a.vc_sb = L->vc_sb; a.vc_sh = L->vc_sh; a.vc_sr = L->vc_sr;
    a.v_scale = L->v_scale; a.v_mn = L->v_mn; a.vs_sb = L->vs_sb; a.vs_sh = L->vs_sh; a.vs_sr = L->vs_sr;
    a.vres = L->v_res; a.vres_sb = L->vr_sb; a.vres_sh = L->vr_sh; a.vres_st = L->vr_st;
    a.v_win_start = (int)wstart; a.v_res_len = (int)vres;
    a.vnew = vnew; a.vnew_sb = vn_sb; a.vnew_sh = vn_sh; a.v_flush = flush;
    a.out = out; a.out_sb = out_sb; a.out_sh = out_sh;
    a.B = L->B; a.nh = nh; a.nh_kv = L->nh_kv; a.D = L->D; a.group_size = L->group_size; a.v_bits = L->v_bits;
    a.Tq = Tq; a.Tv = Tv;
    a.workspace = L->workspace; a.workspace_bytes = L->workspace_bytes;
    // the packed-K side goes in the same call: the library fuses the qK^T of a row into the launch when it can and
    // runs it as its own launch first otherwise
    a.k_code = Tq ? L->k_code : nullptr; a.kc_sb = L->kc_sb; a.kc_sh = L->kc_sh; a.kc_sp = L->kc_sp; a.kc_sr = L->kc_sr;
    a.k_scale = L->k_scale; a.k_mn = L->k_mn; a.ks_sb = L->ks_sb; a.ks_sh = L->ks_sh; a.ks_sp = L->ks_sp; a.ks_sr = L->ks_sr;
    a.k_page_tokens = L->page_tokens; a.k_bits = L->k_bits;
    rc = kivi_decode_attend(&a, stream);
    if (rc) return rc;            // nothing of the step has been committed: the caller may compose it instead

    // phase committed: the attend launch appended the new key (llama_kivi.py:333-336) and value (:377) and, when the
    // window was full, quantised the token leaving it (:386-399)
    kres += 1;
    vres += 1;
    if (flush) {
        Tv += 1;
        wstart += 1;
        vres -= 1;
    }
    st[1] = kres; st[2] = Tv; st[3] = wstart; st[4] = vres; st[5] = kv + 1;
    if (kres == R) {              // :343-356; on a launch failure the state says "R residual tokens, flush pending"
        rc = flush_k();
        if (rc) return rc;
    }
    return 0;
}

// The same for a cache in the KT / VT layouts (kivi_mfma_layout.h): kivi_gqa_decode (one launch for multi-head rows that fit
// the LDS, two launches otherwise) + lengths + the K flush through kivi_kt_pack every residual_length steps
// (llama_kivi.py:343-356) + window compaction.  Same state array, same atomicity contract as kivi_decode_layer.
extern "C" int kivi_mf_decode_layer(const kivi_mf_layer_desc* L, int64_t* st, const void* q, int64_t q_sb, int64_t q_sh, int nh,
                                    const void* knew, int64_t kn_sb, int64_t kn_sh, const void* vnew, int64_t vn_sb,
                                    int64_t vn_sh, const void* mask, int64_t mask_sb, void* out, int64_t out_sb,
                                    int64_t out_sh, kivi_stream_t stream) {
    KIVI_REQUIRE(L && st && q && knew && vnew && out, KIVI_EINVAL, "kivi_mf_decode_layer: null argument");
    int64_t Tq = st[0], kres = st[1], Tv = st[2], wstart = st[3], vres = st[4], kv = st[5];
    const int R = L->residual_length;
    KIVI_REQUIRE(R > 0 && R % 32 == 0 && R <= 128 && Tq >= 0 && Tq % 32 == 0 && kres >= 0 && kres <= R && Tv >= 0 && wstart >= 0 &&
                     vres >= 0 && vres <= R && kv == Tq + kres && kv == Tv + vres,
                 KIVI_EINVAL, "kivi_mf_decode_layer: inconsistent lengths (Tq=%lld kres=%lld Tv=%lld vres=%lld kv=%lld R=%d)",
                 (long long)Tq, (long long)kres, (long long)Tv, (long long)vres, (long long)kv, R);
    // everything the K flush (kivi_kt_pack, below) can reject is checked before anything is launched
    KIVI_REQUIRE((L->bits == 2 || L->bits == 4) && L->group_size == 32 && L->D == 128, KIVI_EUNSUPPORTED,
                 "kivi_mf_decode_layer: the MFMA cache layout covers 2- and 4-bit codes, group_size 32, head_dim 128 (got %d / %d / %d)",
                 L->bits, L->group_size, L->D);
    KIVI_REQUIRE(L->bits == 2 || (L->nh_kv > 0 && (nh == 4 * L->nh_kv || nh == L->nh_kv)), KIVI_EUNSUPPORTED,
                 "kivi_mf_decode_layer: 4-bit codes on the matrix pipe need nh / nh_kv in {1, 4} (got %d / %d)", nh, L->nh_kv);
    KIVI_REQUIRE(L->B > 0 && L->nh_kv > 0 && nh > 0 && nh % L->nh_kv == 0, KIVI_EINVAL, "kivi_mf_decode_layer: bad shape (B=%d nh=%d nh_kv=%d)",
                 L->B, nh, L->nh_kv);
    KIVI_REQUIRE(L->kt && L->vt && L->k_res && L->v_res && L->scores && L->stats && L->workspace && L->kt_range && L->vt_range, KIVI_EINVAL,
                 "kivi_mf_decode_layer: null cache buffer in the descriptor");
    KIVI_REQUIRE((uintptr_t)L->kt_range % 4 == 0 && (uintptr_t)L->vt_range % 4 == 0, KIVI_EALIGN, "kivi_mf_decode_layer: range flags alignment");
    KIVI_REQUIRE(L->cap % 512 == 0 && kv + 1 <= L->cap, KIVI_EINVAL, "kivi_mf_decode_layer: cache capacity %lld exceeded", (long long)L->cap);
    KIVI_REQUIRE((uintptr_t)L->kt % 16 == 0 && L->kt_sb % 4 == 0 && L->kt_sh % 4 == 0 && L->kt_ss % 4 == 0 && L->kt_ss >= (L->bits == 4 ? 10240 : 6144) &&
                     (uintptr_t)L->k_res % 4 == 0 && L->kr_sb % 2 == 0 && L->kr_sh % 2 == 0 && L->kr_st % 2 == 0,
                 KIVI_EALIGN, "kivi_mf_decode_layer: K store / residual alignment");
    KIVI_REQUIRE((int64_t)L->B * L->nh_kv * (R / 32) < ((int64_t)1 << 31), KIVI_EINVAL, "kivi_mf_decode_layer: K flush grid too large");
    KIVI_REQUIRE(kv + 1 <= L->s_pitch, KIVI_EINVAL, "kivi_mf_decode_layer: score rows too short");
    hipStream_t s = (hipStream_t)stream;

    auto flush_k = [&]() -> int {
        const int rc = kivi_kt_pack(L->k_res, L->kr_sb, L->kr_sh, L->kr_st, L->kt, L->kt_sb, L->kt_sh, L->kt_ss, L->kt_range, Tq, L->B,
                                    L->nh_kv, R, L->D, L->group_size, L->bits, stream);
        if (rc) return rc;
        Tq += R;
        kres = 0;
        st[0] = Tq; st[1] = 0;
        return 0;
    };
    if (kres == R) {   // a previous call committed its attend phase but its K flush launch failed: finish that first
        const int rc = flush_k();
        if (rc) return rc;
    }
    const bool ring = (L->flags & KIVI_GQA_WINDOW_RING) != 0;     // ring window: no compaction, R + 1 rows suffice
    KIVI_REQUIRE(!ring || (wstart < L->v_window_rows && vres + 1 <= L->v_window_rows), KIVI_EINVAL,
                 "kivi_mf_decode_layer: ring window of %lld rows cannot hold %lld + 1 values", (long long)L->v_window_rows, (long long)vres);
    if (!ring && wstart + vres + 1 > L->v_window_rows) {
        KIVI_REQUIRE(L->vr_sb == (int64_t)L->nh_kv * L->vr_sh && wstart >= vres, KIVI_EUNSUPPORTED,
                     "kivi_mf_decode_layer: window buffer layout not compactable in place");
        const size_t pitch = (size_t)L->vr_sh * 2, width = (size_t)vres * L->vr_st * 2;
        if (vres) {
            const hipError_t e = hipMemcpy2DAsync(L->v_res, pitch, (const char*)L->v_res + (size_t)wstart * L->vr_st * 2,
                                                  pitch, width, (size_t)L->B * L->nh_kv, hipMemcpyDeviceToDevice, s);
            KIVI_REQUIRE(e == hipSuccess, (int)e, "kivi_mf_decode_layer: window compaction: %s", hipGetErrorString(e));
        }
        wstart = 0;
        st[3] = 0;
    }
    const int flush = vres + 1 > R;
    kivi_gqa_decode_args a;
    a.B = L->B; a.nh = nh; a.nh_kv = L->nh_kv; a.D = L->D; a.group_size = L->group_size; a.bits = L->bits;
    a.inv_scale = L->inv_scale;
    a.q = q; a.q_sb = q_sb; a.q_sh = q_sh;
    a.mask = mask; a.mask_sb = mask_sb;
    a.kt = L->kt; a.kt_sb = L->kt_sb; a.kt_sh = L->kt_sh; a.kt_ss = L->kt_ss; a.Tq = Tq;
    a.kres = L->k_res; a.kres_sb = L->kr_sb; a.kres_sh = L->kr_sh; a.kres_st = L->kr_st;
    a.knew = knew; a.knew_sb = kn_sb; a.knew_sh = kn_sh; a.k_res_len = (int)kres;
    a.vt = L->vt; a.vt_sb = L->vt_sb; a.vt_sh = L->vt_sh; a.vt_ss = L->vt_ss; a.Tv = Tv;
    a.vres = L->v_res; a.vres_sb = L->vr_sb; a.vres_sh = L->vr_sh; a.vres_st = L->vr_st;
    a.v_win_start = (int)wstart; a.v_res_len = (int)vres;
    a.vnew = vnew; a.vnew_sb = vn_sb; a.vnew_sh = vn_sh; a.v_flush = flush;
    a.scores = L->scores; a.s_sb = L->s_sb; a.s_sh = L->s_sh;
    a.stats = L->stats; a.stats_bytes = L->stats_bytes;
    a.workspace = L->workspace; a.workspace_bytes = L->workspace_bytes;
    a.out = out; a.out_sb = out_sb; a.out_sh = out_sh;
    a.residual_length = R; a.v_window_rows = L->v_window_rows;
    a.kt_superblocks = L->cap / 512; a.vt_superblocks = L->cap / 512;
    a.flags = L->flags;
    a.kt_range = L->kt_range; a.vt_range = L->vt_range;
    a.dyn_step = nullptr;
    int rc = kivi_gqa_decode(&a, stream);
    if (rc) return rc;            // nothing of the step has been committed
    kres += 1;
    vres += 1;
    if (flush) {
        Tv += 1;
        wstart += 1;
        if (ring && wstart >= L->v_window_rows) wstart -= L->v_window_rows;
        vres -= 1;
    }
    st[1] = kres; st[2] = Tv; st[3] = wstart; st[4] = vres; st[5] = kv + 1;
    if (kres == R) {              // :343-356; on a launch failure the state says "R residual tokens, flush pending"
        rc = flush_k();
        if (rc) return rc;
    }
    return 0;
}


// The attend phase with device-resident lengths (hipGraph capture): include/kivi_hip.h, kivi_mf_step.
extern "C" int kivi_mf_decode_layer_dyn(const kivi_mf_layer_desc* L, const kivi_mf_step* hs, const void* dev_step, const void* q,
                                        int64_t q_sb, int64_t q_sh, int nh, const void* knew, int64_t kn_sb, int64_t kn_sh,
                                        const void* vnew, int64_t vn_sb, int64_t vn_sh, const void* mask, int64_t mask_sb, void* out,
                                        int64_t out_sb, int64_t out_sh, kivi_stream_t stream) {
    KIVI_REQUIRE(L && hs && dev_step && q && knew && vnew && out, KIVI_EINVAL, "kivi_mf_decode_layer_dyn: null argument");
    const int R = L->residual_length;
    const int64_t kv = hs->Tq + hs->k_res_len;
    KIVI_REQUIRE(R > 0 && R % 32 == 0 && R <= 128 && hs->Tq >= 0 && hs->Tq % 32 == 0 && hs->k_res_len >= 0 && hs->k_res_len < R && hs->Tv >= 0 &&
                     hs->v_win_start >= 0 && hs->v_res_len >= 0 && hs->v_res_len <= R && kv == hs->Tv + hs->v_res_len &&
                     (hs->v_flush != 0) == (hs->v_res_len + 1 > R),
                 KIVI_EINVAL, "kivi_mf_decode_layer_dyn: inconsistent lengths (Tq=%lld kres=%d Tv=%lld vres=%d flush=%d R=%d)",
                 (long long)hs->Tq, hs->k_res_len, (long long)hs->Tv, hs->v_res_len, hs->v_flush, R);
    KIVI_REQUIRE((L->bits == 2 || L->bits == 4) && L->group_size == 32 && L->D == 128, KIVI_EUNSUPPORTED,
                 "kivi_mf_decode_layer_dyn: the MFMA cache layout covers 2- and 4-bit codes, group_size 32, head_dim 128 (got %d / %d / %d)",
                 L->bits, L->group_size, L->D);
    KIVI_REQUIRE(L->B > 0 && L->nh_kv > 0 && nh > 0 && nh % L->nh_kv == 0, KIVI_EINVAL, "kivi_mf_decode_layer_dyn: bad shape (B=%d nh=%d nh_kv=%d)",
                 L->B, nh, L->nh_kv);
    KIVI_REQUIRE(L->kt && L->vt && L->k_res && L->v_res && L->scores && L->stats && L->workspace && L->kt_range && L->vt_range, KIVI_EINVAL,
                 "kivi_mf_decode_layer_dyn: null cache buffer in the descriptor");
    KIVI_REQUIRE((L->flags & KIVI_GQA_WINDOW_RING) != 0, KIVI_EUNSUPPORTED, "kivi_mf_decode_layer_dyn: needs the ring window (nothing to compact between replays)");
    // the class of steps a capture of this one may be replayed for must fit the buffers: its longest row and the stores
    const int64_t nsbk = (hs->Tq + 511) / 512;
    KIVI_REQUIRE(L->cap % 512 == 0 && kv + 1 <= L->cap && nsbk * 512 + R <= L->s_pitch, KIVI_EINVAL,
                 "kivi_mf_decode_layer_dyn: capacity %lld / score pitch %lld too small for the step's geometry class", (long long)L->cap,
                 (long long)L->s_pitch);
    kivi_gqa_decode_args a;
    a.B = L->B; a.nh = nh; a.nh_kv = L->nh_kv; a.D = L->D; a.group_size = L->group_size; a.bits = L->bits;
    a.inv_scale = L->inv_scale;
    a.q = q; a.q_sb = q_sb; a.q_sh = q_sh;
    a.mask = mask; a.mask_sb = mask_sb;
    a.kt = L->kt; a.kt_sb = L->kt_sb; a.kt_sh = L->kt_sh; a.kt_ss = L->kt_ss; a.Tq = hs->Tq;
    a.kres = L->k_res; a.kres_sb = L->kr_sb; a.kres_sh = L->kr_sh; a.kres_st = L->kr_st;
    a.knew = knew; a.knew_sb = kn_sb; a.knew_sh = kn_sh; a.k_res_len = hs->k_res_len;
    a.vt = L->vt; a.vt_sb = L->vt_sb; a.vt_sh = L->vt_sh; a.vt_ss = L->vt_ss; a.Tv = hs->Tv;
    a.vres = L->v_res; a.vres_sb = L->vr_sb; a.vres_sh = L->vr_sh; a.vres_st = L->vr_st;
    a.v_win_start = hs->v_win_start; a.v_res_len = hs->v_res_len;
    a.vnew = vnew; a.vnew_sb = vn_sb; a.vnew_sh = vn_sh; a.v_flush = hs->v_flush;
    a.scores = L->scores; a.s_sb = L->s_sb; a.s_sh = L->s_sh;
    a.stats = L->stats; a.stats_bytes = L->stats_bytes;
    a.workspace = L->workspace; a.workspace_bytes = L->workspace_bytes;
    a.out = out; a.out_sb = out_sb; a.out_sh = out_sh;
    a.residual_length = R; a.v_window_rows = L->v_window_rows;
    a.kt_superblocks = L->cap / 512; a.vt_superblocks = L->cap / 512;
    a.flags = L->flags;
    a.kt_range = L->kt_range; a.vt_range = L->vt_range;
    a.dyn_step = dev_step;
    return kivi_gqa_decode(&a, stream);
}

extern "C" int kivi_mf_step_advance(kivi_mf_step* s, int R, int64_t window_rows) {
    if (!s || R <= 0 || s->k_res_len < 0 || s->k_res_len >= R || s->v_res_len < 0 || s->v_res_len > R ||
        s->Tq + s->k_res_len != s->Tv + s->v_res_len || window_rows < R + 1 || s->v_win_start < 0 || s->v_win_start >= window_rows)
        return KIVI_EINVAL;
    s->k_res_len += 1;                                 // the K append (llama_kivi.py:333-336)
    if (s->v_res_len + 1 > R) {                        // the window was full: its oldest token was quantised (:386-399)
        s->Tv += 1;
        s->v_win_start = (int32_t)((s->v_win_start + 1) % window_rows);
    } else {
        s->v_res_len += 1;                             // the V append (:377)
    }
    s->v_flush = s->v_res_len + 1 > R;                 // what the NEXT step does
    return s->k_res_len == R;
}

namespace {
__global__ void mf_step_store_kernel(kivi_mf_step v, kivi_mf_step* dst) { *dst = v; }
}  // namespace

extern "C" int kivi_mf_step_upload(const kivi_mf_step* hs, void* dev_step, kivi_stream_t stream) {
    KIVI_REQUIRE(hs && dev_step && (uintptr_t)dev_step % 8 == 0, KIVI_EINVAL, "kivi_mf_step_upload: null / misaligned argument");
    hipLaunchKernelGGL(mf_step_store_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, *hs, (kivi_mf_step*)dev_step);
    return kivi_launch_status("mf_step_store");
}

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
#include "kivi_gemv_v_dev.h"
#include "kivi_quant.h"
#include "kivi_row_softmax.h"

namespace {

template <int BITS, int G, int DW, int WPL, int R, int U, int MODE, bool NT, bool SPLIT>
__global__ __launch_bounds__(256) void gemv_v_kernel(const GemvVArgs a) {
    v_row_body<BITS, G, DW, WPL, R, U, MODE, NT, SPLIT>(a);
}

// The whole decode step of one (b, head) row in ONE block (MHA, rows <= 8192 keys): the packed qK^T of the row tile
// by tile (k_tile_body, the scores go to the LDS row instead of memory), then everything v_row_body does with them
// (residual scores, softmax, window, packed sV).  Against the two-launch form this drops the 2 x 8 MB score round
// trip through HBM, one launch ramp/drain and the cold start of the second kernel.
// NW = 8 (round 2): the same row on eight waves, two blocks per CU.  Co-resident blocks are served oldest first (DESIGN.md
// section 3.2b), so the last blocks of a launch stream alone, each wave bound by its own issue rate: with twice the waves
// per row that tail is half as long.
template <int BITS, int G, int DW, int KWPL, int KDS, int KU, int VWPL, int VU, bool PRE = true, bool DBG = false, int EARLY = 0,
          int PRIO = 0, int NW = 4, int VDEPTH = 2>
__global__ __launch_bounds__(NW * 64, (DBG || EARLY || PRIO || NW == 8 || VDEPTH == 3) ? 4 : 1) void decode_row_kernel(const GemvKArgs ak, const GemvVArgs av) {
    extern __shared__ uint16_t pl_row[];
    const int unit = (int)blockIdx.x;
    if constexpr (PRIO != 0) {   // experiment: issue priority by hardware wave slot (the SIMD arbiter is oldest-first otherwise)
        const int slot = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 3;
        const int pr = PRIO == 1 ? 3 - slot : slot;
        if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
        else if (pr == 3) __builtin_amdgcn_s_setprio(3);
    }
    kivi_stamp<DBG>(av.dbg, 0);
    if (DBG && (threadIdx.x & 63) == 0)
        av.dbg[((size_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 16 + 1] = __builtin_amdgcn_s_memrealtime();   // 100 MHz, chip-wide
    RowPre<DW * (32 / BITS), NW> pre;
    if constexpr (PRE) row_prefetch(av, pre);   // in flight during the whole qK^T phase (26 VGPRs, where they fit)
    kivi_stamp<DBG>(av.dbg, 2);
    for (int tb = 0; tb < ak.tile_blocks; tb++) {
        k_tile_body<BITS, G, KWPL, KDS, 1, KU, KIVI_UNPACK_MIX, true, DBG, NW>(ak, unit * ak.tile_blocks + tb, pl_row);
        if (tb == 0) kivi_stamp<DBG>(av.dbg, 4);
        __syncthreads();   // the exchange buffer is reused by the next tile; the scores must be visible below
    }
    kivi_stamp<DBG>(av.dbg, 5);
    v_row_body<BITS, G, DW, VWPL, 1, VU, KIVI_UNPACK_MIX, true, false, PRE, DBG, true, EARLY, NW, VDEPTH>(av, &pre);
    kivi_stamp<DBG>(av.dbg, 11);
    if (DBG && (threadIdx.x & 63) == 0) {
        unsigned long long* rec = av.dbg + ((size_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 16;
        rec[12] = __builtin_amdgcn_s_memrealtime();
        rec[13] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_ID: wave slot, SIMD, CU, SE
        rec[14] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // XCC_ID
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
    // table order = dispatch preference (measured, profiles/); -DKIVI_TUNING builds add the losing / diagnostic shapes
    // ---- 2-bit, D=128 (DW=8), g=32, MHA
    VV(2, 32, 8, 4, 1, 1, 2, 1),
    VV(2, 32, 8, 2, 1, 4, 2, 1),
    VV(2, 32, 8, 4, 1, 2, 2, 0),
#ifdef KIVI_TUNING
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
#endif
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
#ifdef KIVI_TUNING
    VV(4, 32, 16, 4, 1, 4, 0, 0),
#endif
    // ---- GQA (R heads share the unpack; 2 words per lane keeps R*EPL accumulators in registers)
    VV(2, 32, 8, 2, 4, 2, 4, 1),
    VV(2, 32, 8, 1, 8, 4, 4, 1),
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
#ifdef KIVI_TUNING
    VV(2, 32, 8, 1, 4, 2, 4, 1),
    VV(2, 32, 8, 1, 4, 1, 4, 1),
    VV(2, 32, 8, 2, 4, 1, 4, 1),
    VV(2, 32, 8, 1, 4, 4, 4, 1),
    VV(2, 32, 8, 1, 4, 8, 4, 1),
#endif
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
            static const char* forced_split = KIVI_TUNE_ENV("KIVI_V_SPLIT");   // tuning aid
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
            static const char* nofuse = KIVI_TUNE_ENV("KIVI_NO_ROW_FUSION");   // tuning aid
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
            static const char* fp = KIVI_TUNE_ENV("KIVI_SOFTMAX_P");   // tuning aid: blocks per row of the row softmax
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
            ak.dbg = a.dbg;
            ak.res_blocks = 0;
            a.scores_lds = 1;
            const dim3 grid((unsigned)units);
            size_t lds = (size_t)a.n_pad * sizeof(uint16_t);
            if (bits == 4) {   // 4-bit: 8 codes per word, 4 words per lane = the same 2048-token tile; sV over 16 words per row
                ak.tile_blocks = (int)(((ak.Tw + 255) / 256 + 1) / 2);
                KIVI_LAUNCH_LDS((decode_row_kernel<4, 32, 16, 4, 2, 4, 4, 2, false, false, 0, 0, 4, 3>), grid, dim3(256), lds, s, ak, a);
                return kivi_launch_status("decode_row");
            }
            const int tiles = (int)((ak.Tw + 127) / 128);
            ak.tile_blocks = (tiles + 1) / 2;                    // DSPLIT = 2 (four waves) / two tiles per pass (eight waves)
            // Eight waves per row (2 tiles of 2048 tokens per pass, D over 4 waves): with fewer than ~1.75 four-wave blocks per
            // CU the chip is under-occupied and the row's own waves are what hides latency -- 256 rows: 37.8 -> 32.5 us,
            // 512 rows: equal, 768 rows: 65.9 vs 72.8 us (profiles/r02_row_nw8_small_batch.log)
            bool few_rows = units < 448;
#ifdef KIVI_TUNING
            static const char* xl = KIVI_TUNE_ENV("KIVI_ROW_EXTRA_LDS");   // diagnostic: fewer co-resident blocks per CU
            static const char* rx = KIVI_TUNE_ENV("KIVI_ROW_X");           // experimental instantiations: d2 / d3 / nw8ds4
            lds += xl ? (size_t)atoi(xl) : 0;
            if (rx && (!strcmp(rx, "d3") || !strcmp(rx, "d2"))) few_rows = false;
            if (rx && !strcmp(rx, "nw8ds4")) few_rows = true;
            if (G == 32 && !few_rows && a.dbg) {
                KIVI_LAUNCH_LDS((decode_row_kernel<2, 32, 8, 2, 2, 4, 4, 1, true, true, 0, 0, 4, 3>), grid, dim3(256), lds, s, ak, a);
                return kivi_launch_status("decode_row");
            }
            if (G == 32 && !few_rows && rx && !strcmp(rx, "d2")) {    // the two-deep V ring (round 1 / first half of round 2)
                KIVI_LAUNCH_LDS((decode_row_kernel<2, 32, 8, 2, 2, 4, 4, 1>), grid, dim3(256), lds, s, ak, a);
                return kivi_launch_status("decode_row");
            }
#endif
            if (G == 32 && few_rows) KIVI_LAUNCH_LDS((decode_row_kernel<2, 32, 8, 2, 4, 4, 4, 1, true, false, 0, 0, 8>), grid, dim3(512), lds, s, ak, a);
            else if (G == 64) KIVI_LAUNCH_LDS((decode_row_kernel<2, 64, 8, 2, 2, 4, 4, 1, true, false, 0, 0, 4, 3>), grid, dim3(256), lds, s, ak, a);
            else if (G == 128) KIVI_LAUNCH_LDS((decode_row_kernel<2, 128, 8, 2, 2, 4, 4, 1, true, false, 0, 0, 4, 3>), grid, dim3(256), lds, s, ak, a);
            else KIVI_LAUNCH_LDS((decode_row_kernel<2, 32, 8, 2, 2, 4, 4, 1, true, false, 0, 0, 4, 3>), grid, dim3(256), lds, s, ak, a);   // three-deep V ring
            return kivi_launch_status("decode_row");
        }
        v.fn(a, dim3((unsigned)(units * S)), s);
        return kivi_launch_status(v.name);
    }
    int best = -1;
    static const char* forced = KIVI_TUNE_ENV("KIVI_GEMV_V_VARIANT");   // tuning aid: force a variant by name if it fits
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
    a.dbg = kivi_debug_stamps();
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

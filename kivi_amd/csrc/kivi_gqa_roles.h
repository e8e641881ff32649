// Roles shared by the matrix-pipe decode kernels (kivi_mf.hip, nh / nh_kv in {1, 4, 8}): argument blocks, the fp16 K-residual role of the qK^T launch, the softmax
// constants of a row from its segment statistics, the hand-off of partial sums between the blocks of a unit.
#pragma once
#include "kivi_common.h"
#include "kivi_gqa_dev.h"
#include "kivi_quant.h"

namespace {

constexpr int KIVI_GQA_WS_COUNTERS = 16384;   // arrival counters at the head of the caller's workspace: [0, 8192) one per unit (partial sums),
                                              // [8192, 16375) one per unit (statistics exchange of sliced launches), 16375 the device error word,
                                              // [16376, 16384) the eight ticket counters of sliced launches
constexpr int KIVI_GQA_TICKETS = 8;           // ticket counters: a block takes its id from counter blockIdx % 8 (see mf_row4_kernel)
constexpr int KIVI_GQA_MAX_SLICED_UNITS = KIVI_GQA_WS_COUNTERS / 2 - KIVI_GQA_TICKETS - 1;

// The six lengths of a decode step in DEVICE memory (= kivi_mf_step of include/kivi_hip.h): when an argument block carries a
// pointer to one, the kernels take the lengths from it instead of from their by-value arguments, so that a captured launch
// (hipGraph) can be replayed step after step while the launch geometry -- sized by the super-block counts -- stays valid.
struct MfStep {
    long long Tq, Tv;
    int k_res_len, v_res_len, v_win_start, v_flush;
};

struct GqaKArgs {
    const uint16_t* q;
    int64_t q_sb, q_sh;
    MfStore kt;
    uint16_t* out;              // score rows: raw fp16 scores (stats == null) or scaled + masked scores (decode step)
    int64_t out_sb, out_sh;
    int nh_kv, ratio, nh;
    int64_t Tq;                 // packed tokens (multiple of 32)
    int nsb, sb_blocks;         // super-blocks of a row, thread blocks per (b, kv head)
    // decode step (kivi_gqa_decode): the epilogue applies 1/sqrt(D) + mask exactly as the reference feeds its softmax
    // (llama_kivi.py:339, :364-372) and leaves (max, sum exp(x - max)) of every segment of the row in `stats`
    float* stats;               // [B][nh][nseg][2] or null
    int nseg;                   // nsb + KIVI_GQA_RES_SEGS
    float inv_scale;
    const uint16_t* mask;       // (B, 1, 1, n) additive fp16 mask or null
    int64_t mask_sb;
    // residual role (the FIRST res_blocks blocks of the grid): q . [fp16 K residual | new key] (:333-337) + the K append
    int res_blocks;             // units * KIVI_GQA_RES_SEGS or 0
    uint16_t* kres;
    int64_t kres_sb, kres_sh, kres_st;
    const uint16_t* knew;
    int64_t knew_sb, knew_sh;
    int res_len;                // keys already in the residual; the new one becomes index res_len
    const int* range;           // [B * nh_kv] range words of the K store (kivi_mfma_layout.h: marks for scales >= 256 / >= 2^-8)
    const MfStep* dyn;          // device-resident lengths (or null): Tq, res_len are read from it
    // one-launch form (mf_row4_kernel): dump != 0 (KIVI_GQA_DUMP_SCORES, tests): the fp16 rows the softmax statistics are taken
    // from (scaled, mask added) also go to `out`; rows cut into S > 1 slices: `stats` is the exchange buffer [unit][slice][R][2]
    // of the slices' (max, sum exp), `xcount` [units] their arrival counters (zero between launches), `ticket` (null for S = 1) the
    // LAST of KIVI_GQA_TICKETS counters (ticket[-c], c = blockIdx % 8) that hand out the block ids in the order the blocks START
    // (always, for S > 1: blocks that wait for each other must not depend on the whole grid being resident at once -- other streams,
    // a CU mask), `err_ws` the device
    // error word of the workspace and `err_host` (or null) the process's host-visible one: a block that gives up waiting records
    // KIVI_ETIMEOUT in both (kivi_device_error)
    int dump;                   // bit 0: dump; bit 1 (-DKIVI_TUNING builds only): fault injection, slice 0 of unit 0 never arrives
    int* xcount;
    int* ticket;
    int* err_ws;
    int* err_host;
    __device__ __forceinline__ void take_dyn() {
        if (dyn) { Tq = dyn->Tq; res_len = dyn->k_res_len; }
    }
};

// Residual role of the decode step: block (unit, j) scores keys [j c, (j + 1) c) of the residual (c = ceil(L / 4), L =
// res_len + 1 incl. the new key) for the R query heads of the unit, 8 lanes per (head, key) with 16-byte loads, fp32
// accumulate, one rounding (the reference's fp16 torch.matmul, llama_kivi.py:337), writes the scaled scores and the
// statistics of its segment, and appends the new key (:333-336).  Short, latency-bound blocks: first in the grid.
// Tq / res_len: the step's lengths (from the arguments or, device-resident, from a.dyn: the callers resolve that).
template <int R>
__device__ __forceinline__ void gqa_k_residual(const GqaKArgs& a, int bid, long long Tq, int res_len) {
    constexpr int CH = 36;                                         // keys per segment: L <= 129 -> c <= 33
    __shared__ float xs[R][CH];
    const int unit = bid / KIVI_GQA_RES_SEGS, j = bid - unit * KIVI_GQA_RES_SEGS;
    const int b = unit / a.nh_kv, hk = unit - b * a.nh_kv;
    const int h0 = hk * a.ratio;
    const int L = res_len + 1;
    const int c = (L + KIVI_GQA_RES_SEGS - 1) / KIVI_GQA_RES_SEGS;
    const int t0 = j * c;
    const int nt = (t0 + c <= L ? c : L - t0) > 0 ? (t0 + c <= L ? c : L - t0) : 0;
    const uint16_t* knew = a.knew + b * a.knew_sb + hk * a.knew_sh;
    uint16_t* kres = a.kres + b * a.kres_sb + hk * a.kres_sh;
    const uint16_t* mrow = a.mask ? a.mask + b * a.mask_sb : nullptr;
    const int nthr = (int)blockDim.x;
    for (int idx = threadIdx.x; idx < R * nt * 8; idx += nthr) {
        const int sub = idx & 7, rt = idx >> 3;
        const int r = rt / nt, t = t0 + (rt - r * nt);
        const uint16_t* krow = ((t < res_len) ? kres + (int64_t)t * a.kres_st : knew) + sub * 16;
        const uint16_t* qrow = a.q + b * a.q_sb + (int64_t)(h0 + r) * a.q_sh + sub * 16;
        const u16x8 k0 = *(const u16x8*)krow, k1 = *(const u16x8*)(krow + 8);
        const u16x8 q0 = *(const u16x8*)qrow, q1 = *(const u16x8*)(qrow + 8);
        float sc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) sc = __builtin_fmaf(h2f_bits(q0[e]), h2f_bits(k0[e]), sc);
#pragma unroll
        for (int e = 0; e < 8; e++) sc = __builtin_fmaf(h2f_bits(q1[e]), h2f_bits(k1[e]), sc);
        if (t == res_len && r == 0) {                               // append the new key
            *(u16x8*)(kres + (int64_t)t * a.kres_st + sub * 16) = k0;
            *(u16x8*)(kres + (int64_t)t * a.kres_st + sub * 16 + 8) = k1;
        }
        sc += __shfl_xor(sc, 1);
        sc += __shfl_xor(sc, 2);
        sc += __shfl_xor(sc, 4);
        if (sub == 0) {
            const uint16_t x = kivi_scaled_score(f2h_bits(sc), a.inv_scale, mrow != nullptr, mrow ? mrow[Tq + t] : 0);
            a.out[b * a.out_sb + (int64_t)(h0 + r) * a.out_sh + Tq + t] = x;
            xs[r][t - t0] = h2f_bits(x);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nthr >> 6;
    for (int r = wave; r < R; r += nw) {
        const float v = lane < nt ? xs[r][lane] : -__builtin_inff();
        const float m = wave_max(v);
        const float l = wave_sum(lane < nt ? kivi_exp(v - m) : 0.f);
        if (lane == 0) {
            float* st = a.stats + (((int64_t)b * a.nh + h0 + r) * a.nseg + a.nsb + j) * 2;
            st[0] = m;
            st[1] = l;
        }
    }
}

struct GqaVArgs {
    const uint16_t* x;          // scaled + masked scores of the row (written by the qK^T launch)
    int64_t x_sb, x_sh;
    const float* stats;         // [B][nh][nseg][2]
    int nseg;
    MfStore vt;
    int nh_kv, ratio, nh;
    int64_t Tv;                 // packed tokens
    int nsb;                    // super-blocks holding them
    int S, spb;                 // stream blocks per (b, kv head), super-blocks per stream block
    int units;                  // B * nh_kv
    int win_blocks;             // 0: every stream block takes a share of the fp16 window; else (= units): one window block per
                                // unit at the TAIL of the grid does the window, the V append and the flush
    int nslot;                  // partial-sum slots per unit: S (+ 1 for the window block)
    uint16_t* vres;             // (B, nh_kv, W, D) fp16 window buffer
    int64_t vres_sb, vres_sh, vres_st;
    int win_start, res_len;     // live rows [win_start, win_start + res_len); the new token goes right after
    const uint16_t* vnew;
    int64_t vnew_sb, vnew_sh;
    int flush;                  // quantise the oldest window row into the layout at token Tv (llama_kivi.py:386-399)
    uint16_t* out;
    int64_t out_sb, out_sh;
    float* ws;                  // [units][nslot][2][R * 128] fp32 partial sums of every block: quantised part, window part
    int* counters;              // [units] arrival counters, zero between launches
    unsigned long long* dbg;    // phase time stamps or null
    int win_rows;               // > 0: the window buffer is a RING of that many rows (row of token t = (win_start + t) % win_rows); 0: linear
    const int* sp_rows;         // kivi_gqa_output: [B][nh] exponent Sp of every probability row (mf_row_sp_kernel)
    int* range;                 // [B * nh_kv] range words of the V store: read by every block, marked by the V flush
    const MfStep* dyn;          // device-resident lengths (or null): Tv, res_len, win_start, flush are read from it
    __device__ __forceinline__ void take_dyn() {
        if (dyn) { Tv = dyn->Tv; res_len = dyn->v_res_len; win_start = dyn->v_win_start; flush = dyn->v_flush; }
    }
};

// softmax constants of the R rows of a unit from the segment statistics: M = max, 1 / sum exp(x - M).  Every lane of the
// calling wave ends up with the same values.
template <int R>
__device__ __forceinline__ void gqa_row_consts(const GqaVArgs& a, int b, int h0, float* M, float* invS) {
    const int lane = threadIdx.x & 63;
    typedef float f2 __attribute__((ext_vector_type(2)));
    if (a.nseg <= 128) {
        // all 2 R loads of the wave are requested before the first reduction: one memory round trip, not 2 R of them
        f2 v0[R], v1[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const f2* st = reinterpret_cast<const f2*>(a.stats + ((int64_t)b * a.nh + h0 + r) * a.nseg * 2);
            v0[r] = lane < a.nseg ? st[lane] : f2{-__builtin_inff(), 0.f};
            v1[r] = lane + 64 < a.nseg ? st[lane + 64] : f2{-__builtin_inff(), 0.f};
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float m = wave_max(__builtin_fmaxf(v0[r][0], v1[r][0]));
            const float l = wave_sum(v0[r][1] * kivi_exp(v0[r][0] - m) + v1[r][1] * kivi_exp(v1[r][0] - m));
            M[r] = m;
            invS[r] = 1.0f / l;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const float* st = a.stats + ((int64_t)b * a.nh + h0 + r) * a.nseg * 2;
        float m = -__builtin_inff();
        for (int i = lane; i < a.nseg; i += 64) m = __builtin_fmaxf(m, st[2 * i]);
        m = wave_max(m);
        float l = 0.f;
        for (int i = lane; i < a.nseg; i += 64) l += st[2 * i + 1] * kivi_exp(st[2 * i] - m);
        l = wave_sum(l);
        M[r] = m;
        invS[r] = 1.0f / l;
    }
}

// Combine of a unit's partial sums by the block that arrives last (hand-off as in gemv_v_kernel<SPLIT>: write-through
// payload, drained, one relaxed arrival counter; cdna_hip_programming.md G16).
// also_reset: a second per-unit counter the last block puts back to zero (the statistics exchange of the sliced one-launch form:
// a block arrives here only after it has left that exchange, so nobody still reads the counter).
template <int R, int NTH = 256>
__device__ __forceinline__ void gqa_arrive_and_combine(const GqaVArgs& a, int unit, int slot, const float* part_lds, int b, int h0,
                                                       int* also_reset = nullptr) {
    __shared__ int last_flag;
    constexpr int RD = R * 128;
    // part_lds = [quantised part | window part] of this block; workspace [unit][slot][2][RD]
    uint32_t* dst = reinterpret_cast<uint32_t*>(a.ws + ((size_t)unit * a.nslot + slot) * 2 * RD);
    for (int i = threadIdx.x; i < 2 * RD; i += NTH)
        __hip_atomic_store(dst + i, __builtin_bit_cast(uint32_t, part_lds[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(a.counters + unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old == a.nslot - 1);
        if (last) {                                                 // next launch
            __hip_atomic_store(a.counters + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (also_reset) __hip_atomic_store(also_reset, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    const uint32_t* p0 = reinterpret_cast<const uint32_t*>(a.ws + (size_t)unit * a.nslot * 2 * RD);
    for (int i = threadIdx.x; i < RD; i += NTH) {
        const int r = i >> 7, d = i & 127;
        float q = 0.f, w = 0.f;
        for (int s0 = 0; s0 < a.nslot; s0 += 4) {   // 8 independent loads in flight, added in slot order
            uint32_t v8[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v8[2 * k] = (s0 + k < a.nslot) ? __hip_atomic_load(p0 + (size_t)(s0 + k) * 2 * RD + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                v8[2 * k + 1] = (s0 + k < a.nslot) ? __hip_atomic_load(p0 + (size_t)(s0 + k) * 2 * RD + RD + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                q += __builtin_bit_cast(float, v8[2 * k]);
                w += __builtin_bit_cast(float, v8[2 * k + 1]);
            }
        }
        // fp16(quantised part) + fp16(window part), rounded: the reference's `attn_output += matmul(...)` (llama_kivi.py:382-384);
        // only the window part exists before anything is quantised (:380)
        const uint16_t o = (a.Tv > 0) ? f2h_bits(h2f_bits(f2h_bits(q)) + h2f_bits(f2h_bits(w))) : f2h_bits(w);
        a.out[b * a.out_sb + (int64_t)(h0 + r) * a.out_sh + d] = o;
    }
}

// Window role of the sV launch: out_w[r][2 lane, 2 lane + 1] += probs[r, Tv + t] * V_window[t] for the window tokens
// t in [w0, w1) (llama_kivi.py:384; the last token is the new value, appended here, :377), and -- `flusher` -- the
// quantisation of the token leaving the window into its VT slot (:386-399).  NTH threads; `pw`: R x PW halves of LDS
// that hold the fp16 probabilities of tokens [w0, w1) at index t - w0 and ZEROS from w1 - w0 up to the next multiple of 8.  A lane
// owns two channels; wave w takes the GROUPS of eight consecutive tokens w, w + NW, ..., so the probabilities of eight tokens of a
// head are ONE 16-byte LDS read and a group is walked branch-free (round 4 walked tokens w, w + NW, ... with two branches and R
// two-byte LDS reads per token, each behind its own wait: 6.7 us per block at residual_length 128, profiles/r05_row4_flows.log).
// Two halves, so that the loads fly while the caller does something else (the row kernels request before their softmax):
// request() issues the loads of the first WPRE tokens of every wave (and of the token that leaves the window, and of the code
// word it will be merged into), finish() loads what is left in one batch and consumes everything once pw holds the probabilities.
template <int R, int NTH, int PW, int WPRE, int BITS = 2>
struct GqaWindow {
    static constexpr int NW = NTH / 64;
    static constexpr int TW = (((129 + NW - 1) / NW) + 7) / 8 * 8;       // tokens per wave: 40 (4 waves), 24 (8 waves)
    static constexpr int NPRE = WPRE < TW ? WPRE : TW;
    static_assert(PW >= NW * TW && PW % 8 == 0, "a probability row holds NW * TW halves, rows 16-byte aligned");
    typedef MfL<BITS> LY;
    uint32_t vv[NPRE];
    uint32_t wold;
    uint16_t xflush;
    int ngr;                   // groups of eight tokens in this window

    __device__ __forceinline__ static uint16_t* wrow(const GqaVArgs& a, uint16_t* vbuf, int t) {
        // row of window token t: a ring of win_rows rows (no compaction, residual_length + 1 rows suffice) or the linear buffer
        int r = a.win_start + t;
        if (a.win_rows) r = r >= a.win_rows ? r - a.win_rows : r;       // t <= residual_length < win_rows: one wrap at most
        return vbuf + (int64_t)r * a.vres_st;
    }
    // the code word of the token that leaves the window (token Tv) for channel d (2 bits: shared by the channels d, d ^ 16)
    __device__ __forceinline__ static uint32_t* flush_word(const GqaVArgs& a, int b, int hk, int d) {
        const int tt = (int)(a.Tv & 31), blk = (int)((a.Tv >> 5) & 15);
        const int kbq = tt >> 3, c = d >> 5, nn = d & 15;
        if constexpr (BITS == 4) return mf_sb(a.vt, b, hk, a.Tv >> 9) + blk * LY::BLOCK_WORDS + vt4_word(tt, d);
        return mf_sb(a.vt, b, hk, a.Tv >> 9) + blk * LY::BLOCK_WORDS + (nn + 16 * kbq) * 4 + c;
    }
    // The two channels (2 lane, 2 lane + 1) of the eight window tokens of group gidx (tokens w0 + 8 gidx + e; past the window: of the
    // new value's row, the walk masks those).  gidx is WAVE-UNIFORM (the callers pass the wave index through readfirstlane): the row
    // addresses are scalar arithmetic -- one multiply per group, then a stride and the ring's wrap per token -- and the loads are
    // unconditional; as conditional loads of per-lane pointers every token cost two branches and ~30 vector instructions (40 tokens
    // per wave: ~4 us in front of the barrier, profiles/r05_row4_flows.log).  E0 .. E1: the elements of the group to load.
    template <int E0, int E1>
    __device__ __forceinline__ static void gload(const GqaVArgs& a, uint16_t* vbuf, const uint16_t* vnew, int gidx, int w0, uint32_t* dst) {
        const int lane = threadIdx.x & 63;
        int r = a.win_start + w0 + 8 * gidx + E0;
        if (a.win_rows && r >= a.win_rows) r -= a.win_rows;        // (t <= residual_length < win_rows: one wrap at most)
        const uint16_t* p = vbuf + (int64_t)r * a.vres_st;
        if (!a.win_rows || r + (E1 - E0) <= a.win_rows) {          // the group's rows do not wrap (all but one group of a ring): a stride per token
#pragma unroll
            for (int e = E0; e < E1; e++) {
                const int t = w0 + 8 * gidx + e;
                const uint16_t* vrow = (t < a.res_len) ? p : vnew; // (t >= res_len: the new value, or past the window)
                dst[e - E0] = *(const uint32_t*)(vrow + 2 * lane); // (no select on the value here: it would wait for the load at once)
                p += a.vres_st;
            }
        } else {
#pragma unroll
            for (int e = E0; e < E1; e++) {
                const int t = w0 + 8 * gidx + e;
                const uint16_t* vrow = (t < a.res_len) ? p : vnew;
                dst[e - E0] = *(const uint32_t*)(vrow + 2 * lane);
                p += a.vres_st;
                r++;
                if (r == a.win_rows) { r = 0; p = vbuf; }
            }
        }
    }

    // w0, w1: the window's tokens [w0, w1), w1 = res_len + 1 (the new value is its last token).  Wave w takes the groups of eight
    // tokens w, w + NW, ...
    __device__ __forceinline__ void request(const GqaVArgs& a, int b, int hk, int w0, int w1, bool flusher) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        uint16_t* vbuf = a.vres + b * a.vres_sb + hk * a.vres_sh;
        const uint16_t* vnew = a.vnew + b * a.vnew_sb + hk * a.vnew_sh;
        xflush = 0; wold = 0;
        if (flusher && threadIdx.x < 128) {
            xflush = wrow(a, vbuf, 0)[threadIdx.x];
            if (BITS == 4 || ((threadIdx.x >> 4) & 1) == 0) wold = *flush_word(a, b, hk, threadIdx.x);
        }
        ngr = (w1 - w0 + 7) >> 3;                                  // groups of the window (<= 17)
#pragma unroll
        for (int u = 0; u < NPRE; u++) vv[u] = 0u;
#pragma unroll
        for (int k = 0; k < (NPRE + 7) / 8; k++) {                 // the first NPRE token slots of the wave (whole and one partial group)
            if (wave + k * NW >= ngr) break;                       // (wave-uniform: nothing of the window there)
            if (8 * k + 8 <= NPRE) gload<0, 8>(a, vbuf, vnew, wave + k * NW, w0, vv + 8 * k);
            else gload<0, NPRE % 8 ? NPRE % 8 : 8>(a, vbuf, vnew, wave + k * NW, w0, vv + 8 * k);
        }
    }

    __device__ __forceinline__ void finish(const GqaVArgs& a, int b, int hk, int w0, int w1, bool flusher,
                                           const uint16_t (*pw)[PW], float (*ow)[2]) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        uint16_t* vbuf = a.vres + b * a.vres_sb + hk * a.vres_sh;
        const uint16_t* vnew = a.vnew + b * a.vnew_sb + hk * a.vnew_sh;
#pragma unroll
        for (int rr = 0; rr < R; rr++) ow[rr][0] = ow[rr][1] = 0.f;
        // the token slots request() did not prefetch are loaded a group ahead of their use
        constexpr int NG = TW / 8;                                 // groups per wave at most
        const int gnew = (a.res_len >= w0 && a.res_len < w1) ? ((a.res_len - w0) >> 3) : -1, enew = (a.res_len - w0) & 7;   // group / slot of the new value
        uint32_t nx[8];
#pragma unroll
        for (int e = 0; e < 8; e++) nx[e] = 0u;
        if constexpr (NPRE < 8) gload<NPRE, 8>(a, vbuf, vnew, wave, w0, nx + NPRE);
#pragma unroll
        for (int k = 0; k < NG; k++) {
            const int gidx = wave + k * NW;
            if (gidx >= ngr) break;                                // (wave-uniform: a short window ends early)
            uint32_t cur[8];
#pragma unroll
            for (int e = 0; e < 8; e++) cur[e] = (8 * k + e < NPRE) ? vv[8 * k + e < NPRE ? 8 * k + e : 0] : nx[e];
            if (k + 1 < NG) {                                      // what the next group still needs from memory
                constexpr int dummy = 0; (void)dummy;
                const int e0 = (NPRE > 8 * (k + 1)) ? ((NPRE - 8 * (k + 1)) < 8 ? (NPRE - 8 * (k + 1)) : 8) : 0;   // (a constant after unrolling)
                if (e0 == 0) gload<0, 8>(a, vbuf, vnew, gidx + NW, w0, nx);
                else if (e0 < 8) {                                 // a group the prefetch covers partly (NPRE % 8 != 0)
                    uint32_t tmp[8];
                    gload<0, 8>(a, vbuf, vnew, gidx + NW, w0, tmp);
#pragma unroll
                    for (int e = 0; e < 8; e++) nx[e] = tmp[e];
                }
            }
            if (gidx == gnew) {                                    // V append (:377): the new value becomes window row res_len -- from the
                uint32_t vn_ = cur[0];                             // registers it was loaded into (a fresh load would put a memory round
#pragma unroll                                                     // trip into the latency-bound middle of the step)
                for (int e = 1; e < 8; e++) vn_ = (enew == e) ? cur[e] : vn_;
                *(uint32_t*)(wrow(a, vbuf, a.res_len) + 2 * lane) = vn_;
            }
            constexpr int RH = R > 4 ? 4 : R;                      // heads per pass (R = 8: two passes: 16 instead of 32 registers of probabilities)
#pragma unroll
            for (int r0 = 0; r0 < R; r0 += RH) {
                u32x4 pv[RH];                                      // the probabilities of 8 tokens of these heads (zeros past the window)
#pragma unroll
                for (int rr = 0; rr < RH; rr++) pv[rr] = *(const u32x4*)(&pw[r0 + rr][8 * gidx]);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const uint32_t v = (w0 + 8 * gidx + e < w1) ? cur[e] : 0u;     // (token slots past the window: zeros, whatever was read)
                    const float v0 = h2f_bits((uint16_t)(v & 0xFFFFu)), v1 = h2f_bits((uint16_t)(v >> 16));
#pragma unroll
                    for (int rr = 0; rr < RH; rr++) {
                        const uint32_t pp = pv[rr][e >> 1];
                        const float p = h2f_bits((uint16_t)((e & 1) ? (pp >> 16) : (pp & 0xFFFFu)));
                        ow[r0 + rr][0] = __builtin_fmaf(p, v0, ow[r0 + rr][0]);
                        ow[r0 + rr][1] = __builtin_fmaf(p, v1, ow[r0 + rr][1]);
                    }
                }
            }
            // (with loads inside the walk: nothing moves across a group's end -- hipcc otherwise hoists every load of the walk to its
            // top and the kernel pays for TW more registers)
            if constexpr (NPRE < TW) __builtin_amdgcn_sched_barrier(0);
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
            const GroupQ gq = make_group(kmin, kmax, (1 << BITS) - 1);
            const uint32_t code = quant_one<BITS>(xflush, gq);
            const int tt = (int)(a.Tv & 31), blk = (int)((a.Tv >> 5) & 15);
            const int e = tt & 7, kbq = tt >> 3;
            const int c = d >> 5, tile = (d >> 4) & 1;
            const int sh = 16 * (e & 1);
            uint32_t* sbp;
            if constexpr (BITS == 4) {
                // every channel has its own word (8 tokens of the channel): the token's field is cleared first (stale codes, below)
                const int pos = vt4_bit(tt);
                sbp = mf_sb(a.vt, b, hk, a.Tv >> 9);
                *flush_word(a, b, hk, d) = (wold & ~(15u << pos)) | (code << pos);
            } else {
                uint32_t val = code << (mf_pos(tile, e >> 1) + sh);
                val |= (uint32_t)__shfl_xor((int)val, 16);
                sbp = mf_sb(a.vt, b, hk, a.Tv >> 9);
                if (tile == 0) {
                    // the two fields of this token (channel tiles 0 / 1) are cleared first: a slot may hold stale codes of an earlier,
                    // longer sequence that used the same storage
                    const uint32_t clr = (3u << (mf_pos(0, e >> 1) + sh)) | (3u << (mf_pos(1, e >> 1) + sh));
                    *flush_word(a, b, hk, d) = (wold & ~clr) | val;
                }
            }
            if ((d & 31) == 0) {
                const int hidx = blk * 128 + kbq * 32 + c * 8 + e;
                ((uint16_t*)(sbp + LY::SCALE_WORD0))[hidx] = gq.scale;
                ((uint16_t*)(sbp + LY::MN_WORD0))[hidx] = gq.mn;
                // range marks of the unit (kivi_mfma_layout.h): the token becomes part of the packed prefix with the NEXT step
                mf_range_mark(a.range + b * a.nh_kv + hk, gq.scale);
            }
        }
    }
};

// The window role as the two-launch form runs it (mf_v_kernel: window blocks at the tail of the grid, or shares inside the stream
// blocks; register-lean -- the sV stream sets that kernel's occupancy): wave w takes tokens w0 + w, w0 + w + NW, ...
// Window role of the sV launch: out_w[r][2 lane, 2 lane + 1] += probs[r, Tv + t] * V_window[t] for the window tokens
// t in [w0, w1) (llama_kivi.py:384; the last token is the new value, appended here, :377), and -- `flusher` -- the
// quantisation of the token leaving the window into its VT slot (:386-399).  NTH threads; `pw`: R x >= 136 halves of LDS
// that already hold the fp16 probabilities of tokens [w0, w1) (index t - w0).  A lane owns two channels, wave w takes
// tokens w0 + w, w0 + w + NW, ...; all loads of a batch in flight.
// Two halves, so that the loads fly while the caller does something else (the row kernels request before their softmax):
// request() issues the loads of the first WPRE tokens of every wave (and of the token that leaves the window, and of the code
// word it will be merged into), finish() consumes them once pw holds the probabilities and walks what is left in batches.
template <int R, int NTH, int PW, int WPRE, int BITS = 2>
struct GqaWindowStrided {
    static constexpr int NW = NTH / 64, WB = 12;
    typedef MfL<BITS> LY;
    uint32_t vv[WPRE];
    uint32_t wold;
    uint16_t xflush;

    __device__ __forceinline__ static uint16_t* wrow(const GqaVArgs& a, uint16_t* vbuf, int t) {
        // row of window token t: a ring of win_rows rows (no compaction, residual_length + 1 rows suffice) or the linear buffer
        int r = a.win_start + t;
        if (a.win_rows) r = r >= a.win_rows ? r - a.win_rows : r;       // t <= residual_length < win_rows: one wrap at most
        return vbuf + (int64_t)r * a.vres_st;
    }
    // the code word of the token that leaves the window (token Tv) for channel d (2 bits: shared by the channels d, d ^ 16)
    __device__ __forceinline__ static uint32_t* flush_word(const GqaVArgs& a, int b, int hk, int d) {
        const int tt = (int)(a.Tv & 31), blk = (int)((a.Tv >> 5) & 15);
        const int kbq = tt >> 3, c = d >> 5, nn = d & 15;
        if constexpr (BITS == 4) return mf_sb(a.vt, b, hk, a.Tv >> 9) + blk * LY::BLOCK_WORDS + vt4_word(tt, d);
        return mf_sb(a.vt, b, hk, a.Tv >> 9) + blk * LY::BLOCK_WORDS + (nn + 16 * kbq) * 4 + c;
    }

    __device__ __forceinline__ void request(const GqaVArgs& a, int b, int hk, int w0, int w1, bool flusher) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint16_t* vbuf = a.vres + b * a.vres_sb + hk * a.vres_sh;
        const uint16_t* vnew = a.vnew + b * a.vnew_sb + hk * a.vnew_sh;
        xflush = 0; wold = 0;
        if (flusher && threadIdx.x < 128) {
            xflush = wrow(a, vbuf, 0)[threadIdx.x];
            if (BITS == 4 || ((threadIdx.x >> 4) & 1) == 0) wold = *flush_word(a, b, hk, threadIdx.x);
        }
#pragma unroll
        for (int u = 0; u < WPRE; u++) {
            const int t = w0 + wave + NW * u;
            const uint16_t* vrow = (t < a.res_len) ? wrow(a, vbuf, t) : vnew;
            vv[u] = (t < w1) ? *(const uint32_t*)(vrow + 2 * lane) : 0u;
        }
    }

    __device__ __forceinline__ void finish(const GqaVArgs& a, int b, int hk, int w0, int w1, bool flusher,
                                           const uint16_t (*pw)[PW], float (*ow)[2]) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint16_t* vbuf = a.vres + b * a.vres_sb + hk * a.vres_sh;
        const uint16_t* vnew = a.vnew + b * a.vnew_sb + hk * a.vnew_sh;
#pragma unroll
        for (int rr = 0; rr < R; rr++) ow[rr][0] = ow[rr][1] = 0.f;
        const int nwt = w1 > w0 ? w1 - w0 : 0;
        auto use = [&](int t, uint32_t v) {
            if (t < w1) {
                const float v0 = h2f_bits((uint16_t)(v & 0xFFFFu)), v1 = h2f_bits((uint16_t)(v >> 16));
#pragma unroll
                for (int rr = 0; rr < R; rr++) {
                    const float p = h2f_bits(pw[rr][t - w0]);
                    ow[rr][0] = __builtin_fmaf(p, v0, ow[rr][0]);
                    ow[rr][1] = __builtin_fmaf(p, v1, ow[rr][1]);
                }
                if (t == a.res_len) *(uint32_t*)(wrow(a, vbuf, t) + 2 * lane) = v;   // V append
            }
        };
#pragma unroll
        for (int u = 0; u < WPRE; u++) use(w0 + wave + NW * u, vv[u]);
        for (int tb = wave + NW * WPRE; tb < nwt; tb += NW * WB) {
            uint32_t vb[WB];
#pragma unroll
            for (int u = 0; u < WB; u++) {
                const int t = w0 + tb + NW * u;
                const uint16_t* vrow = (t < a.res_len) ? wrow(a, vbuf, t) : vnew;
                vb[u] = (t < w1) ? *(const uint32_t*)(vrow + 2 * lane) : 0u;
            }
#pragma unroll
            for (int u = 0; u < WB; u++) use(w0 + tb + NW * u, vb[u]);
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
            const GroupQ gq = make_group(kmin, kmax, (1 << BITS) - 1);
            const uint32_t code = quant_one<BITS>(xflush, gq);
            const int tt = (int)(a.Tv & 31), blk = (int)((a.Tv >> 5) & 15);
            const int e = tt & 7, kbq = tt >> 3;
            const int c = d >> 5, tile = (d >> 4) & 1;
            const int sh = 16 * (e & 1);
            uint32_t* sbp;
            if constexpr (BITS == 4) {
                // every channel has its own word (8 tokens of the channel): the token's field is cleared first (stale codes, below)
                const int pos = vt4_bit(tt);
                sbp = mf_sb(a.vt, b, hk, a.Tv >> 9);
                *flush_word(a, b, hk, d) = (wold & ~(15u << pos)) | (code << pos);
            } else {
                uint32_t val = code << (mf_pos(tile, e >> 1) + sh);
                val |= (uint32_t)__shfl_xor((int)val, 16);
                sbp = mf_sb(a.vt, b, hk, a.Tv >> 9);
                if (tile == 0) {
                    // the two fields of this token (channel tiles 0 / 1) are cleared first: a slot may hold stale codes of an earlier,
                    // longer sequence that used the same storage
                    const uint32_t clr = (3u << (mf_pos(0, e >> 1) + sh)) | (3u << (mf_pos(1, e >> 1) + sh));
                    *flush_word(a, b, hk, d) = (wold & ~clr) | val;
                }
            }
            if ((d & 31) == 0) {
                const int hidx = blk * 128 + kbq * 32 + c * 8 + e;
                ((uint16_t*)(sbp + LY::SCALE_WORD0))[hidx] = gq.scale;
                ((uint16_t*)(sbp + LY::MN_WORD0))[hidx] = gq.mn;
                // range marks of the unit (kivi_mfma_layout.h): the token becomes part of the packed prefix with the NEXT step
                mf_range_mark(a.range + b * a.nh_kv + hk, gq.scale);
            }
        }
    }
};

template <int R, int NTH, int PW, int BITS = 2>
__device__ __forceinline__ void gqa_window_part(const GqaVArgs& a, int b, int hk, int w0, int w1, bool flusher,
                                                const uint16_t (*pw)[PW], float (*ow)[2]) {
    GqaWindowStrided<R, NTH, PW, 12, BITS> w;
    w.request(a, b, hk, w0, w1, flusher);
    w.finish(a, b, hk, w0, w1, flusher, pw, ow);
}

}  // namespace

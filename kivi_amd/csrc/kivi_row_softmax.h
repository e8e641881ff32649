// Row softmax of the decode step as its own launch(es) (grouped queries, long rows, rows split over blocks).
#pragma once
#include "kivi_common.h"

namespace {

// Stand-alone row softmax of the decode step, used when the block-prologue softmax of gemv_v_kernel does not pay
// (grouped queries: R rows per block; rows longer than the register-resident form; rows split over blocks): one
// block per (b, h) score row computes the residual scores q . [fp16 K residual | new key] (+ the K append) when a
// query is given, then scale + mask + softmax with exactly the element order and reduction tree of
// kivi_softmax_scaled, and overwrites the score row with the fp16 probabilities (llama_kivi.py:339, :364-375).
struct RowSoftmaxArgs {
    uint16_t* scores;
    int64_t s_sb, s_sh;
    int n, Tq;
    float inv_scale;
    const uint16_t* mask;
    int64_t mask_sb;
    const uint16_t* q;                 // null: the score rows are complete (no residual part to compute)
    int64_t q_sb, q_sh;
    uint16_t* kres;
    int64_t k_sb, k_sh, k_st;
    const uint16_t* knew;
    int64_t kn_sb, kn_sh;
    int rk_len, ratio, nh, D;
    // long rows / few rows: P blocks per row, each owning `chunk` scores (the last one the rest, incl. the residual
    // part); launch 1 leaves (max, sum exp) of every chunk in `partial`, launch 2 combines them and normalises.
    int P, chunk;
    float* partial;                    // [rows][P][2]
};

// MODE 0: one block per row does everything.  MODE 1: chunk statistics.  MODE 2: combine + normalise the chunk.
template <int MODE>
__global__ __launch_bounds__(256) void row_softmax_kernel(const RowSoftmaxArgs p) {
    typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
    __shared__ uint16_t rs_lds[136];
    const int row = (MODE == 0) ? (int)blockIdx.x : (int)blockIdx.x / p.P;
    const int c = (MODE == 0) ? 0 : (int)blockIdx.x - row * p.P;
    const int P = (MODE == 0) ? 1 : p.P;
    const int b = row / p.nh, h = row - b * p.nh;
    const int hk = h / p.ratio;
    uint16_t* srow = p.scores + b * p.s_sb + (int64_t)h * p.s_sh;
    const uint16_t* mrow = p.mask ? p.mask + b * p.mask_sb : nullptr;
    const int n = p.n;
    const int lo = c * p.chunk;                               // multiple of 1024: the 8-byte loads stay aligned
    const int hi = (c == P - 1) ? n : lo + p.chunk;
    const bool res_here = (MODE != 2) && p.q != nullptr && c == P - 1;   // the last chunk contains [Tq, n)
    if (res_here) {
        const int L = p.rk_len + 1;
        const uint16_t* knew = p.knew + b * p.kn_sb + hk * p.kn_sh;
        uint16_t* kres = p.kres + b * p.k_sb + hk * p.k_sh;
        const int cpl = p.D / 8;
        for (int idx = threadIdx.x; idx < L * 8; idx += 256) {
            const int sub = idx & 7, t = idx >> 3;
            const uint16_t* krow = ((t < p.rk_len) ? kres + (int64_t)t * p.k_st : knew) + sub * cpl;
            const uint16_t* qrow = p.q + b * p.q_sb + (int64_t)h * p.q_sh + sub * cpl;
            const bool append = (t == p.rk_len) && (h % p.ratio) == 0;
            float sc = 0.f;
            for (int d = 0; d < cpl; d += 8) {
                const u16x8 kv = *(const u16x8*)(krow + d);
                const u16x8 qv = *(const u16x8*)(qrow + d);
#pragma unroll
                for (int e = 0; e < 8; e++) sc = __builtin_fmaf(h2f_bits(qv[e]), h2f_bits(kv[e]), sc);
                if (append) *(u16x8*)(kres + (int64_t)t * p.k_st + sub * cpl + d) = kv;
            }
            sc += __shfl_xor(sc, 1);
            sc += __shfl_xor(sc, 2);
            sc += __shfl_xor(sc, 4);
            if (sub == 0) {
                const uint16_t hs = f2h_bits(sc);
                rs_lds[t] = hs;
                srow[p.Tq + t] = hs;
            }
        }
        __syncthreads();
    }
    auto sval = [&](int j) {
        const uint16_t raw1 = (res_here && j >= p.Tq) ? rs_lds[j - p.Tq] : srow[j];
        return h2f_bits(kivi_scaled_score(raw1, p.inv_scale, mrow != nullptr, mrow ? mrow[j] : 0));
    };
    const int nvec = (res_here ? p.Tq : n) & ~3;   // scores below this index come straight from memory, 4 at a time
    float mx = -__builtin_inff();
    float sum = 0.f;
    if constexpr (MODE != 2) {
        // ONE pass over the chunk: every thread keeps a running (max, sum of exp(x - max)) of its scores and rescales
        // the sum when the max grows; the 256 pairs are then merged the same way (wave shuffles, 4 values of LDS).
        for (int j0 = lo + threadIdx.x * 4; j0 < hi; j0 += 1024) {
            float x[4];
            if (j0 + 4 <= nvec && j0 + 4 <= hi) {
                const u16x4 v4 = *(const u16x4*)(srow + j0);
#pragma unroll
                for (int e = 0; e < 4; e++)
                    x[e] = h2f_bits(kivi_scaled_score(v4[e], p.inv_scale, mrow != nullptr, mrow ? mrow[j0 + e] : 0));
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) x[e] = (j0 + e < hi) ? sval(j0 + e) : -__builtin_inff();
            }
            const float m4 = __builtin_fmaxf(__builtin_fmaxf(x[0], x[1]), __builtin_fmaxf(x[2], x[3]));
            const float mn = __builtin_fmaxf(mx, m4);   // finite: every score is a finite fp16 (masked ones sit at -65504)
            sum = sum * kivi_exp(mx - mn) + ((kivi_exp(x[0] - mn) + kivi_exp(x[1] - mn)) + (kivi_exp(x[2] - mn) + kivi_exp(x[3] - mn)));
            mx = mn;
        }
        auto merge = [](float& m, float& l, float m2, float l2) {
            const float mn = __builtin_fmaxf(m, m2);
            const float a = (m == mn) ? 1.0f : kivi_exp(m - mn);      // also covers -inf - -inf (an empty side)
            const float b = (m2 == mn) ? 1.0f : kivi_exp(m2 - mn);
            l = l * a + l2 * b;
            m = mn;
        };
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) merge(mx, sum, __shfl_xor(mx, k), __shfl_xor(sum, k));
        __shared__ float ml_lds[8];
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            ml_lds[threadIdx.x >> 6] = mx;
            ml_lds[4 + (threadIdx.x >> 6)] = sum;
        }
        __syncthreads();
        mx = ml_lds[0];
        sum = ml_lds[4];
#pragma unroll
        for (int w = 1; w < 4; w++) merge(mx, sum, ml_lds[w], ml_lds[4 + w]);
    }
    if constexpr (MODE == 1) {
        if (threadIdx.x == 0) {
            p.partial[2 * ((int64_t)row * P + c)] = mx;
            p.partial[2 * ((int64_t)row * P + c) + 1] = sum;
        }
        return;
    }
    if constexpr (MODE == 2) {   // every thread combines the P chunk statistics the same way (chunk order)
        const float* pp = p.partial + 2 * (int64_t)row * P;
        for (int i = 0; i < P; i++) mx = __builtin_fmaxf(mx, pp[2 * i]);
        for (int i = 0; i < P; i++) sum += pp[2 * i + 1] * kivi_exp(pp[2 * i] - mx);
    }
    // every thread rewrites exactly the elements it read (the reductions above are barriers), so in place is safe
    const float inv = 1.0f / sum;
    for (int j0 = lo + threadIdx.x * 4; j0 < hi; j0 += 1024) {
        if (j0 + 4 <= nvec && j0 + 4 <= hi) {
            const u16x4 v4 = *(const u16x4*)(srow + j0);
            u16x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = f2h_bits(kivi_exp(h2f_bits(kivi_scaled_score(v4[e], p.inv_scale, mrow != nullptr,
                                                                         mrow ? mrow[j0 + e] : 0)) - mx) * inv);
            *(u16x4*)(srow + j0) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (j0 + e < hi) srow[j0 + e] = f2h_bits(kivi_exp(sval(j0 + e) - mx) * inv);
        }
    }
}


}  // namespace

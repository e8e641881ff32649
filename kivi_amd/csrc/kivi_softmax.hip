// Scale + (mask) + softmax over one row of attention scores, one launch.
//
// Replaces three torch kernels of the reference decode branch (models/llama_kivi.py):
//   attn_weights = cat([...]) / math.sqrt(head_dim)                     :339   fp16 result
//   attn_weights = max(attn_weights + attention_mask, finfo.min)        :364-372 (optional)
//   softmax(attn_weights, dim=-1, dtype=float32).to(fp16)               :375
// with the same roundings: the scaled score is rounded to fp16 (torch's CUDA/HIP division by a Python scalar
// multiplies by the fp32 reciprocal -- proven equal to true division for every finite half and the divisors
// used here in oracle/pin_reference.py), the mask add is rounded to fp16, the softmax runs in fp32 and the
// probabilities are rounded once.  One 256-thread block per (b, h) row; the row lives in registers.
#include "kivi_common.h"

namespace {

typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));

// CHUNKS x 1024 elements per row are held in registers (4 halves per thread per chunk).
template <int CHUNKS>
__global__ __launch_bounds__(256) void softmax_scaled_kernel(const uint16_t* __restrict__ scores, uint16_t* __restrict__ probs,
                                                             int64_t n, int64_t s_pitch, int64_t p_pitch, float inv_scale,
                                                             const uint16_t* __restrict__ mask, int64_t mask_sb, int nh) {
    __shared__ float lds[4];
    const int64_t row = blockIdx.x;
    const uint16_t* srow = scores + row * s_pitch;
    uint16_t* prow = probs + row * p_pitch;
    const uint16_t* mrow = mask ? mask + (row / nh) * mask_sb : nullptr;
    float x[CHUNKS][4];
    float mx = -__builtin_inff();
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        const int64_t j0 = (int64_t)c * 1024 + threadIdx.x * 4;
        u16x4 raw = {0, 0, 0, 0};
        if (j0 + 4 <= n) raw = *(const u16x4*)(srow + j0);
        else
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (j0 + e < n) raw[e] = srow[j0 + e];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float v = -__builtin_inff();
            if (j0 + e < n) {
                v = h2f_bits(kivi_scaled_score(raw[e], inv_scale, mrow != nullptr, mrow ? mrow[j0 + e] : 0));
            }
            x[c][e] = v;
            mx = __builtin_fmaxf(mx, v);
        }
    }
    mx = kivi_block_reduce(mx, true, lds);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            x[c][e] = kivi_exp(x[c][e] - mx);   // exp(-inf) = 0 for the padding lanes
            sum += x[c][e];
        }
    sum = kivi_block_reduce(sum, false, lds);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        const int64_t j0 = (int64_t)c * 1024 + threadIdx.x * 4;
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = f2h_bits(x[c][e] * inv);
        if (j0 + 4 <= n) *(u16x4*)(prow + j0) = o;
        else
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (j0 + e < n) prow[j0 + e] = o[e];
    }
}

// Any row length: three passes over the (L2-resident) row.
__global__ __launch_bounds__(256) void softmax_scaled_generic(const uint16_t* __restrict__ scores, uint16_t* __restrict__ probs,
                                                              int64_t n, int64_t s_pitch, int64_t p_pitch, float inv_scale,
                                                              const uint16_t* __restrict__ mask, int64_t mask_sb, int nh) {
    __shared__ float lds[4];
    const int64_t row = blockIdx.x;
    const uint16_t* srow = scores + row * s_pitch;
    uint16_t* prow = probs + row * p_pitch;
    const uint16_t* mrow = mask ? mask + (row / nh) * mask_sb : nullptr;
    auto val = [&](int64_t j) {
        return h2f_bits(kivi_scaled_score(srow[j], inv_scale, mrow != nullptr, mrow ? mrow[j] : 0));
    };
    float mx = -__builtin_inff();
    for (int64_t j = threadIdx.x; j < n; j += 256) mx = __builtin_fmaxf(mx, val(j));
    mx = kivi_block_reduce(mx, true, lds);
    float sum = 0.f;
    for (int64_t j = threadIdx.x; j < n; j += 256) sum += kivi_exp(val(j) - mx);
    sum = kivi_block_reduce(sum, false, lds);
    const float inv = 1.0f / sum;
    for (int64_t j = threadIdx.x; j < n; j += 256) prow[j] = f2h_bits(kivi_exp(val(j) - mx) * inv);
}

}  // namespace

extern "C" int kivi_softmax_scaled(const void* scores, void* probs, int64_t rows, int64_t n, int64_t s_pitch,
                                   int64_t p_pitch, float inv_scale, const void* mask, int64_t mask_sb, int nh,
                                   kivi_stream_t stream) {
    KIVI_REQUIRE(rows >= 0 && n >= 0 && nh > 0, KIVI_EINVAL, "kivi_softmax_scaled: bad shape");
    KIVI_REQUIRE(scores != probs || s_pitch == p_pitch, KIVI_EINVAL, "kivi_softmax_scaled: in-place needs equal pitches");
    if (rows == 0 || n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (s_pitch % 4 == 0) && (p_pitch % 4 == 0) && ((uintptr_t)scores % 8 == 0) && ((uintptr_t)probs % 8 == 0);
    dim3 grid((unsigned)rows);
#define KIVI_SM_CASE(C)                                                                                              \
    if (vec && n <= (C) * 1024) {                                                                                    \
        hipLaunchKernelGGL(softmax_scaled_kernel<C>, grid, dim3(256), 0, s, (const uint16_t*)scores, (uint16_t*)probs, \
                           n, s_pitch, p_pitch, inv_scale, (const uint16_t*)mask, mask_sb, nh);                      \
        return kivi_launch_status("softmax_scaled");                                                                \
    }
    KIVI_SM_CASE(1) KIVI_SM_CASE(2) KIVI_SM_CASE(4) KIVI_SM_CASE(5) KIVI_SM_CASE(8) KIVI_SM_CASE(16)
#undef KIVI_SM_CASE
    KIVI_REQUIRE(scores != probs, KIVI_EINVAL, "kivi_softmax_scaled: rows longer than 16384 cannot be in place");
    hipLaunchKernelGGL(softmax_scaled_generic, grid, dim3(256), 0, s, (const uint16_t*)scores, (uint16_t*)probs, n,
                       s_pitch, p_pitch, inv_scale, (const uint16_t*)mask, mask_sb, nh);
    return kivi_launch_status("softmax_scaled_generic");
}

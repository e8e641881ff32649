// ABI twin of the reference's native entry point on ITS kernel-input layout
// (gemv_forward_cuda_outer_dim, quant/csrc/gemv_cuda.h:13-21, gemv_cuda.cu:511-557):
//   in (BS, 1, IC) fp16, kernel (BS_kv, OC/fpi, IC) int32, scale/zeros (BS_kv, OC/g, IC) fp16,
//   out (BS, 1, OC) fp16.
// The fast paths (kivi_gemv_k / kivi_gemv_v) read the hook-state layout and never need
// these transposed tensors; this kernel exists for callers that already hold them
// (quant/gemv.py:117,154).  One wave per (batch row, packed output row): lanes stride over
// IC with coalesced dword loads, fpi fp32 accumulators per lane, butterfly reduction.
#include "kivi_common.h"

namespace {

template <int BITS>
__global__ __launch_bounds__(256) void gemv_outer_dim_kernel(const uint16_t* __restrict__ in,
                                                             const uint32_t* __restrict__ kernel,
                                                             const uint16_t* __restrict__ scale,
                                                             const uint16_t* __restrict__ zeros,
                                                             uint16_t* __restrict__ out, int64_t IC, int64_t OC,
                                                             int g, int ratio, int nrow_blocks) {
    constexpr int FPI = 32 / BITS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t bidx = blockIdx.x / nrow_blocks;            // batch_idx
    const int64_t row = (int64_t)(blockIdx.x % nrow_blocks) * 4 + wave;  // packed_oc_idx
    const int64_t nrow = (OC + FPI - 1) / FPI;
    if (row >= nrow) return;
    const int64_t bk = bidx / ratio;                           // gemv_cuda.cu:361-365
    const int64_t grp = (row * FPI) / g;                       // :357
    const uint32_t* wp = kernel + (bk * nrow + row) * IC;
    const uint16_t* sp = scale + (bk * (OC / g) + grp) * IC;
    const uint16_t* zp = zeros + (bk * (OC / g) + grp) * IC;
    const uint16_t* ip = in + bidx * IC;
    float acc[FPI];
#pragma unroll
    for (int p = 0; p < FPI; p++) acc[p] = 0.f;
    float z = 0.f;
    for (int64_t ic = lane; ic < IC; ic += 64) {
        const float x = h2f_bits(ip[ic]);
        const float xs = x * h2f_bits(sp[ic]) * qs_factor<KIVI_UNPACK_MIX>();
        z = __builtin_fmaf(x, h2f_bits(zp[ic]), z);
        accum_word<BITS, KIVI_UNPACK_MIX>(wp[ic], xs, acc);
    }
#pragma unroll
    for (int p = 0; p < FPI; p++) acc[p] *= post_scale<BITS, KIVI_UNPACK_MIX>(p);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) z += __shfl_xor(z, m);
    // halve the accumulator set while there is more than one value, then plain xor-adds
    int n = FPI, off = 0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        if (n > 1) {
            const int half = n / 2;
            const bool upper = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < FPI / 2; i++) {
                if (i < half) {
                    const float send = upper ? acc[i] : acc[i + half];
                    const float keep = upper ? acc[i + half] : acc[i];
                    acc[i] = keep + __shfl_xor(send, m);
                }
            }
            off += upper ? half : 0;
            n = half;
        } else {
            acc[0] += __shfl_xor(acc[0], m);
        }
    }
    // lanes 0..FPI-1 hold distinct channels `off`
    if (lane < FPI) {
        const int64_t oc = row * FPI + off;
        if (oc < OC) out[bidx * OC + oc] = f2h_bits(acc[0] + z);
    }
}

// Legacy AWQ-style INNER-dim grouped 4-bit GEMV (gemv_kernel_g64 / gemv_kernel_g128, gemv_cuda.cu:60-184):
//   out[b, oc] = fp16( sum_ic (scale[oc, ic/g] * code[oc, ic] + zero[oc, ic/g]) * in[b, ic] )
// weight (OC, IC/8) int32 packed along IC, scale / zeros (OC, >= IC/g) fp16 with row pitch `sz_pitch`.
// Not on the KV-cache path (only the reference's disabled scripts call it, quant/gemv.py:188,225); kept for surface
// parity, written for clarity: one wave per (b, oc), 32 codes per lane per pass.
__global__ __launch_bounds__(256) void gemv_awq_kernel(const uint16_t* __restrict__ in, const uint32_t* __restrict__ weight,
                                                       const uint16_t* __restrict__ scale,
                                                       const uint16_t* __restrict__ zeros, uint16_t* __restrict__ out,
                                                       int64_t IC, int64_t OC, int g, int64_t sz_pitch) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t oc = (int64_t)blockIdx.x * 4 + wave;
    const int64_t b = blockIdx.y;
    if (oc >= OC) return;
    const int64_t ww = IC / 8;
    const uint32_t* wrow = weight + oc * ww;
    const uint16_t* xrow = in + b * IC;
    float psum = 0.f;
    for (int64_t w0 = (int64_t)lane * 4; w0 < ww; w0 += 256) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t w = w0 + j;
            if (w < ww) {
                uint32_t word = wrow[w];
                const int64_t gi = (w * 8) / g;
                const float sc = h2f_bits(scale[oc * sz_pitch + gi]), zp = h2f_bits(zeros[oc * sz_pitch + gi]);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float dq = __builtin_fmaf(sc, (float)(word & 0xFu), zp);   // gemv_cuda.cu:101
                    psum = __builtin_fmaf(dq, h2f_bits(xrow[w * 8 + i]), psum);      // :103
                    word >>= 4;
                }
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) psum += __shfl_xor(psum, m);
    if (lane == 0) out[b * OC + oc] = f2h_bits(psum);
}

}  // namespace

extern "C" int kivi_gemv_awq(const void* in, const void* kernel, const void* scale, const void* zeros, void* out, int64_t B,
                             int64_t IC, int64_t OC, int bit, int group_size, int64_t sz_pitch, kivi_stream_t stream) {
    KIVI_REQUIRE(bit == 4, KIVI_EINVAL, "kivi_gemv_awq: the reference kernels are 4-bit only (PACK_FACTOR 8), got %d", bit);
    KIVI_REQUIRE(group_size == 64 || group_size == 128, KIVI_EINVAL,
                 "kivi_gemv_awq: group_size must be 64 or 128 (gemv_cuda.cu:227-244), got %d", group_size);
    KIVI_REQUIRE(IC > 0 && IC % group_size == 0 && OC >= 0 && B >= 0 && sz_pitch >= IC / group_size, KIVI_EINVAL,
                 "kivi_gemv_awq: IC=%lld must be a multiple of group_size=%d", (long long)IC, group_size);
    KIVI_REQUIRE(B < 65536, KIVI_EINVAL, "kivi_gemv_awq: batch too large");
    if (B == 0 || OC == 0) return 0;
    dim3 grid((unsigned)((OC + 3) / 4), (unsigned)B);
    hipLaunchKernelGGL(gemv_awq_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in,
                       (const uint32_t*)kernel, (const uint16_t*)scale, (const uint16_t*)zeros, (uint16_t*)out, IC, OC,
                       group_size, sz_pitch);
    return kivi_launch_status("gemv_awq");
}

extern "C" int kivi_gemv_outer_dim(const void* in, const void* kernel, const void* scale, const void* zeros, void* out,
                                   int64_t BS, int64_t IC, int64_t OC, int bit, int group_size, int nh, int nh_kv,
                                   kivi_stream_t stream) {
    KIVI_REQUIRE(bit == 2 || bit == 4, KIVI_EINVAL, "kivi_gemv_outer_dim: bit must be 2 or 4 (matmul.py:215), got %d", bit);
    KIVI_REQUIRE(nh_kv > 0 && nh > 0 && nh % nh_kv == 0, KIVI_EINVAL,
                 "kivi_gemv_outer_dim: nh %% nh_kv != 0 (matmul.py:216): nh=%d nh_kv=%d", nh, nh_kv);
    const int fpi = 32 / bit;
    KIVI_REQUIRE(group_size > 0 && group_size % fpi == 0 && OC % group_size == 0, KIVI_EINVAL,
                 "kivi_gemv_outer_dim: OC=%lld must be a multiple of group_size=%d", (long long)OC, group_size);
    KIVI_REQUIRE(BS >= 0 && IC >= 0 && OC >= 0, KIVI_EINVAL, "kivi_gemv_outer_dim: negative size");
    if (BS == 0 || OC == 0) return 0;
    const int64_t nrow = (OC + fpi - 1) / fpi;
    const int nrb = (int)((nrow + 3) / 4);
    KIVI_REQUIRE(BS * nrb < ((int64_t)1 << 31), KIVI_EINVAL, "kivi_gemv_outer_dim: grid too large");
    dim3 grid((unsigned)(BS * nrb));
    hipStream_t s = (hipStream_t)stream;
    if (bit == 2)
        hipLaunchKernelGGL(gemv_outer_dim_kernel<2>, grid, dim3(256), 0, s, (const uint16_t*)in, (const uint32_t*)kernel,
                           (const uint16_t*)scale, (const uint16_t*)zeros, (uint16_t*)out, IC, OC, group_size,
                           nh / nh_kv, nrb);
    else
        hipLaunchKernelGGL(gemv_outer_dim_kernel<4>, grid, dim3(256), 0, s, (const uint16_t*)in, (const uint32_t*)kernel,
                           (const uint16_t*)scale, (const uint16_t*)zeros, (uint16_t*)out, IC, OC, group_size,
                           nh / nh_kv, nrb);
    return kivi_launch_status("gemv_outer_dim");
}

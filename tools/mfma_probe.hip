// Diagnostic for the grouped-query GEMV design: v_mfma_f32_4x4x1_16b_f32 on gfx950 --
//  (1) operand / result layout, (2) fp32-subnormal B operands (the masked 2-bit codes), (3) issue rate alone and
//  next to the VALU ANDs of the unpack.   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int i = 0; i < 4; i++) d[l * 4 + i] = c[i];
}

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, uint32_t seed) {
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t w = seed * (threadIdx.x + 1);
    float qs = 1.0f + threadIdx.x * 1e-6f;
    float facc[8];
    for (int i = 0; i < 8; i++) facc[i] = (float)i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if constexpr (KIND == 0) {            // MFMA only, 8 independent accumulators
                acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(qs, __builtin_bit_cast(float, w), acc[i], 0, 0, 0);
            } else if constexpr (KIND == 1) {     // one AND (fp32-subnormal mask) + MFMA per code
                const uint32_t m = w & (3u << (2 * i));
                acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(qs, __builtin_bit_cast(float, m), acc[i], 0, 0, 0);
            } else if constexpr (KIND == 2) {     // the VALU form it would replace: AND + 4 fmac
                const float m = __builtin_bit_cast(float, w & (3u << (2 * i)));
#pragma unroll
                for (int r = 0; r < 4; r++) acc[i][r] = __builtin_fmaf(m, qs + r, acc[i][r]);
            } else if constexpr (KIND == 3) {     // MFMA with normal (non-denormal) B
                acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(qs, qs, acc[i], 0, 0, 0);
            }
        }
        w = w * 1664525u + 1013904223u;
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + facc[i];
    if (s == 123.456f) out[0] = s;
}

template <int KIND>
void run(const char* name, float* out) {
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    // waves per SIMD = blocks*4 / (256 CUs * 4 SIMDs) = 8; "codes" per wave = iters * 8
    const double per = best * 1e-3 / (8.0 * iters * 8);
    printf("%-44s %8.3f ms   %6.2f ns per (wave, code) -> %5.1f cycles @2.4GHz\n", name, best, per * 1e9, per * 2.4e9);
}

int main() {
    float *a, *b, *d;
    CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&d, 1024));
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; l++) { ha[l] = (float)(l + 1); hb[l] = (float)(100 * (l + 1)); }
    CK(hipMemcpy(a, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, a, b, d);
    CK(hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost));
    // hypothesis: lane l = 4*blk + j holds D[i][j] = A[blk][i] * B[blk][j] in register i, A[blk][i] from lane 4*blk+i
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 4; i++) {
            const float want = ha[(l & ~3) + i] * hb[l];
            if (hd[l * 4 + i] != want) bad++;
        }
    printf("layout D[i][j] (reg i of lane 4b+j) = A(lane 4b+i) * B(lane 4b+j): %s (%d mismatches); lane5: %g %g %g %g\n",
           bad ? "NO" : "yes", bad, hd[20], hd[21], hd[22], hd[23]);
    // subnormal B: code 3 in field 5 of an fp32 word
    for (int l = 0; l < 64; l++) { ha[l] = 0x1p100f; uint32_t m = 3u << 10; hb[l] = *(float*)&m; }
    CK(hipMemcpy(a, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, a, b, d);
    CK(hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost));
    printf("subnormal B (3<<10 as fp32) * 2^100 = %g, expected %g -> %s\n", hd[0], 3.0 * 1024 * 0x1p-149 * 0x1p100,
           hd[0] == (float)(3.0 * 1024 * 0x1p-149 * 0x1p100) ? "exact" : "FLUSHED/WRONG");
    float* out; CK(hipMalloc(&out, 64));
    run<0>("mfma_f32_4x4x1 only", out);
    run<3>("mfma_f32_4x4x1 only, normal operands", out);
    run<1>("v_and + mfma_f32_4x4x1 per code", out);
    run<2>("v_and + 4 v_fmac per code (VALU form)", out);
    return 0;
}

// Probe for DESIGN.md section 8, item 1: gfx950's scaled matrix instruction with FP4 (E2M1) B operands.
//   (1) decode: every element of B = the nibble v (0 .. 15), every element of A = fp8 e4m3 1.0, scales 2^0 -> D = 128 * value(v).
//       A 2-bit code zero-extended to a nibble (v = 0 .. 3) must come out as v / 2 exactly.
//   (2) rate: cycles per v_mfma_scale_f32_16x16x128_f8f6f4 (A fp8, B fp4) against v_mfma_f32_16x16x32_f16, four independent chains.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_fp4_probe.hip -o tools/mfma_fp4_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int i8v __attribute__((ext_vector_type(8)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void decode(float* out) {
    for (int v = 0; v < 16; v++) {
        const int nib = v * 0x11111111;
        i8v a;
        for (int i = 0; i < 8; i++) a[i] = 0x38383838;      // e4m3 1.0
        i8v b = {nib, nib, nib, nib, 0, 0, 0, 0};            // fp4: the first four registers carry the 32 elements of a lane
        f4 c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 4, 0, 127, 0, 127);
        out[v * 64 + threadIdx.x] = c[0];        // (all lanes store: a store under `lane == 0` lets hipcc sink the matrix instruction into the branch)
    }
}

template <bool FP4>
__global__ void rate(long long* cyc, float* sink, int iters) {
    i8v a, b;
    for (int i = 0; i < 8; i++) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x11111111 * (threadIdx.x & 3); }
    h8 ah, bh;
    for (int i = 0; i < 8; i++) { ah[i] = (_Float16)(1.0f + threadIdx.x); bh[i] = (_Float16)0.5f; }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if constexpr (FP4) {
            c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 0, 4, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 0, 4, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 0, 4, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 0, 4, 0, 127, 0, 127);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c3, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    float *out, *sink; long long* cyc;
    (void)hipMalloc(&out, 16 * 64 * 4); (void)hipMalloc(&sink, 256); (void)hipMalloc(&cyc, 8);
    decode<<<1, 64>>>(out);
    float hall[16 * 64], h[16]; (void)hipMemcpy(hall, out, sizeof hall, hipMemcpyDeviceToHost);
    for (int v = 0; v < 16; v++) h[v] = hall[v * 64];
    printf("fp4 nibble -> value (D / 128, A = fp8 1.0):\n");
    for (int v = 0; v < 16; v++) printf("  %2d (%d%d%d%d) -> %g%s\n", v, (v >> 3) & 1, (v >> 2) & 1, (v >> 1) & 1, v & 1, h[v] / 128.0f, v < 4 ? (h[v] / 128.0f == v * 0.5f ? "   = code / 2" : "   MISMATCH") : "");
    const int iters = 4096;
    long long c;
    for (int rep = 0; rep < 2; rep++) {
        rate<true><<<1, 64>>>(cyc, sink, iters); (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("v_mfma_scale_f32_16x16x128_f8f6f4 (A fp8, B fp4): %.1f shader-clock ticks per instruction (one wave, 4 chains)\n", (double)c / (4.0 * iters));
        rate<false><<<1, 64>>>(cyc, sink, iters); (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("v_mfma_f32_16x16x32_f16:                           %.1f ticks per instruction\n", (double)c / (4.0 * iters));
    }
    return 0;
}

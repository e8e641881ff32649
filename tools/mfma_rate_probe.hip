// Issue rate of v_mfma_f32_16x16x32_f16 on gfx950 as the matrix-pipe decode kernels use it (kivi_mf_dev.h): does a
// SUBNORMAL fp16 B operand (a masked 2-bit code) run at the rate of a normal one?  chains of 2 accumulators vs 8
// independent ones; bare vs interleaved with the 11 view / mask VALU instructions per code word.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o tools/mfma_rate_probe.bin && tools/mfma_rate_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC, bool VALU>
__global__ __launch_bounds__(256) void k(const uint32_t* in, float* out, int iters) {
    const int l = threadIdx.x;
    u32x4 w = *(const u32x4*)(in + (l & 63) * 4);
    u32x4 aw = *(const u32x4*)(in + 256 + (l & 63) * 4);
    const h8 a = __builtin_bit_cast(h8, aw);
    f4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            uint32_t x = w[c];
            u32x4 b0, b1;
            if constexpr (VALU) {
                const uint32_t x1 = x << 4, x2 = x >> 4, x3 = __builtin_amdgcn_perm(x, x, 0x02030001u);
                b0 = u32x4{x & 0x03000300u, x1 & 0x03000300u, x & 0x00C000C0u, x1 & 0x00C000C0u};
                b1 = u32x4{x2 & 0x03000300u, x3 & 0x03000300u, x2 & 0x00C000C0u, x3 & 0x00C000C0u};
            } else {
                b0 = u32x4{x, x, x, x};
                b1 = b0;
            }
            if constexpr (MODE == 1) {        // normal fp16 operands: set an exponent
                b0 |= 0x3C003C00u;
                b1 |= 0x3C003C00u;
            }
            const h8 hb0 = __builtin_bit_cast(h8, b0), hb1 = __builtin_bit_cast(h8, b1);
            acc[(2 * c) % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, hb0, acc[(2 * c) % NACC], 0, 0, 0);
            acc[(2 * c + 1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, hb1, acc[(2 * c + 1) % NACC], 0, 0, 0);
            acc[(2 * c) % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, hb0, acc[(2 * c) % NACC], 0, 0, 0);
            acc[(2 * c + 1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, hb1, acc[(2 * c + 1) % NACC], 0, 0, 0);
        }
        w = w + (uint32_t)(__builtin_bit_cast(uint32_t, acc[0][0]) & 1u);   // keep the loop honest
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[l] = s;
}

template <int MODE, int NACC, bool VALU>
static void run(const char* name, const uint32_t* in, float* out, int blocks_per_cu) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NACC, VALU><<<256 * blocks_per_cu, 256>>>(in, out, 10);
    hipEventRecord(e0);
    k<MODE, NACC, VALU><<<256 * blocks_per_cu, 256>>>(in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)blocks_per_cu * iters * 16;   // 4 waves of a block = one per SIMD
    printf("%-44s waves/SIMD %d: %7.2f ns per MFMA per SIMD = %5.1f cycles at 2.4 GHz\n", name, blocks_per_cu, ms * 1e6 / mfma_per_simd,
           ms * 1e6 / mfma_per_simd * 2.4);
}

int main() {
    uint32_t h[512];
    for (int i = 0; i < 256; i++) h[i] = 0x9E3779B9u * (i + 1);               // code words
    for (int i = 256; i < 512; i++) h[i] = 0x3C003800u + ((i * 37) & 0x3FF);  // A operand: normal halves
    uint32_t* d; float* o;
    hipMalloc(&d, sizeof h); hipMalloc(&o, 1024);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int bpc = 1; bpc <= 4; bpc += 3) {
        run<0, 2, true>("subnormal B, 2 chained acc, + views/masks", d, o, bpc);
        run<1, 2, true>("normal B,    2 chained acc, + views/masks", d, o, bpc);
        run<0, 8, true>("subnormal B, 8 independent acc, + views/masks", d, o, bpc);
        run<0, 2, false>("raw words B, 2 chained acc, no VALU", d, o, bpc);
        run<1, 2, false>("normal B,    2 chained acc, no VALU", d, o, bpc);
        run<0, 8, false>("raw words B, 8 independent acc, no VALU", d, o, bpc);
    }
    return 0;
}

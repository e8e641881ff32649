// Diagnostic: issue rate of the VALU instructions the unpack loop can be built from (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NACC = 16;

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, uint32_t seed) {
    float acc[NACC];
    f32x2 acc2[NACC / 2];
    uint32_t w = seed + threadIdx.x;
    float qs = 1.0f + threadIdx.x * 1e-6f;
    f32x2 qq = {qs, qs};
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (float)i;
#pragma unroll
    for (int i = 0; i < NACC / 2; i++) acc2[i] = f32x2{(float)i, 1.f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if constexpr (KIND == 0) {            // v_fma_f32
                acc[i] = __builtin_fmaf(acc[i], qs, 0.5f);
            } else if constexpr (KIND == 1) {     // v_fma_mix_f32 (fp16 lo operand)
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(w), "v"(qs));
            } else if constexpr (KIND == 2) {     // v_and_b32
                uint32_t t;
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(t) : "v"(w), "v"(__builtin_bit_cast(uint32_t, acc[i])));
                acc[i] = __builtin_bit_cast(float, t);
            } else if constexpr (KIND == 3) {     // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i % (NACC / 2)]) : "v"(qq), "v"(qq));
            } else if constexpr (KIND == 4) {     // v_cvt_scalef32_pk_f32_fp4
                asm volatile("v_cvt_scalef32_pk_f32_fp4 %0, %1, 1.0" : "=v"(acc2[i % (NACC / 2)]) : "v"(w));
            } else if constexpr (KIND == 5) {     // v_cvt_f32_ubyte1
                asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(acc[i]) : "v"(w));
            } else if constexpr (KIND == 6) {     // v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc2[i % (NACC / 2)]) : "v"(qq));
            } else if constexpr (KIND == 7) {     // v_dot2c_f32_f16
                asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc[i]) : "v"(w), "v"(w));
            } else if constexpr (KIND == 8) {     // v_lshrrev_b32
                uint32_t t;
                asm volatile("v_lshrrev_b32 %0, 6, %1" : "=v"(t) : "v"(__builtin_bit_cast(uint32_t, acc[i])));
                acc[i] = __builtin_bit_cast(float, t);
            } else if constexpr (KIND == 9) {     // v_cvt_pk_f32_fp8
                asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(acc2[i % (NACC / 2)]) : "v"(w));
            } else if constexpr (KIND == 10) {    // v_pk_fma_f16
                uint32_t t = __builtin_bit_cast(uint32_t, acc[i]);
                asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(t) : "v"(w));
                acc[i] = __builtin_bit_cast(float, t);
            } else if constexpr (KIND == 11) {    // v_and_or_b32
                uint32_t t;
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(t) : "v"(w), "v"(seed), "v"(__builtin_bit_cast(uint32_t, acc[i])));
                acc[i] = __builtin_bit_cast(float, t);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
#pragma unroll
    for (int i = 0; i < NACC / 2; i++) s += acc2[i].x + acc2[i].y;
    if (s == 123.456f) out[0] = s;
}

template <int KIND>
void run(const char* name, float* out, double ghz_guess) {
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
    double wave_instr = (double)blocks * 4 * iters * NACC;          // wave-instructions issued
    double per_simd_per_s = wave_instr / 1024.0 / (best * 1e-3);     // 1024 SIMDs
    printf("%-32s %8.3f ms   %6.2f G wave-instr/s/SIMD -> %.2f cycles per wave64 instr @%.2f GHz   %.2e lane-ops/s chip\n",
           name, best, per_simd_per_s / 1e9, ghz_guess * 1e9 / per_simd_per_s, ghz_guess, wave_instr * 64 / (best * 1e-3));
}

int main() {
    float* out; CK(hipMalloc(&out, 64));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    double ghz = p.clockRate / 1e6;
    printf("device %s, clockRate %.2f GHz, CUs %d\n", p.name, ghz, p.multiProcessorCount);
    run<0>("v_fma_f32", out, ghz);
    run<1>("v_fma_mix_f32", out, ghz);
    run<2>("v_and_b32", out, ghz);
    run<11>("v_and_or_b32", out, ghz);
    run<8>("v_lshrrev_b32", out, ghz);
    run<3>("v_pk_fma_f32", out, ghz);
    run<6>("v_pk_mul_f32", out, ghz);
    run<4>("v_cvt_scalef32_pk_f32_fp4", out, ghz);
    run<9>("v_cvt_pk_f32_fp8", out, ghz);
    run<5>("v_cvt_f32_ubyte1", out, ghz);
    run<7>("v_dot2c_f32_f16", out, ghz);
    run<10>("v_pk_fma_f16", out, ghz);
    return 0;
}

// Numerics of v_mfma_f32_16x16x32_f16 on gfx950 beyond "subnormals are kept": how exact are the products and the
// accumulation when B holds fp16 subnormals (masked 2-bit codes) and A holds full-mantissa halves (hi) or tiny remainders
// (lo)?  Every case prints the worst error in units of 2^-24 of the largest |product| (1.0 = one fp32 ulp-ish).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_prec_probe.bin mfma_prec_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void mfma_once(const uint16_t* A, const uint16_t* B, const float* C, float* D) {
    const int l = threadIdx.x;
    h8 a, b;
    for (int e = 0; e < 8; e++) {
        a[e] = __builtin_bit_cast(_Float16, A[(l & 15) * 32 + 8 * (l >> 4) + e]);
        b[e] = __builtin_bit_cast(_Float16, B[(8 * (l >> 4) + e) * 16 + (l & 15)]);
    }
    f4 c;
    for (int j = 0; j < 4; j++) c[j] = C[(4 * (l >> 4) + j) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; j++) D[(4 * (l >> 4) + j) * 16 + (l & 15)] = c[j];
}

// chain of `steps` accumulating MFMAs with fresh operands each step (operands [step][16][32] / [step][32][16])
__global__ void mfma_chain(const uint16_t* A, const uint16_t* B, float* D, int steps) {
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; s++) {
        h8 a, b;
        for (int e = 0; e < 8; e++) {
            a[e] = __builtin_bit_cast(_Float16, A[s * 512 + (l & 15) * 32 + 8 * (l >> 4) + e]);
            b[e] = __builtin_bit_cast(_Float16, B[s * 512 + (8 * (l >> 4) + e) * 16 + (l & 15)]);
        }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    for (int j = 0; j < 4; j++) D[(4 * (l >> 4) + j) * 16 + (l & 15)] = c[j];
}

static float h2f(uint16_t u) { _Float16 h = __builtin_bit_cast(_Float16, u); return (float)h; }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }

int main() {
    uint16_t hA[512], hB[512];
    float hC[256], hD[256];
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    srand(3);
    auto run = [&](const char* what) {
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        mfma_once<<<1, 64>>>(dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double worst = 0, worst_rel_sum = 0; int exact = 0;
        for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) {
            long double ref = hC[m * 16 + n]; double big = fabs(hC[m * 16 + n]);
            for (int k = 0; k < 32; k++) {
                const double p = (double)h2f(hA[m * 32 + k]) * (double)h2f(hB[k * 16 + n]);
                ref += p; big = fmax(big, fabs(p));
            }
            const double err = fabs((double)ref - (double)hD[m * 16 + n]);
            if ((float)ref == hD[m * 16 + n]) exact++;
            if (big > 0) worst = fmax(worst, err / (big * 5.9604644775390625e-08));
            if (ref != 0) worst_rel_sum = fmax(worst_rel_sum, err / fabs((double)ref));
        }
        printf("%-72s worst err %.3f x 2^-24 |max term|, rel to sum %.2e, correctly rounded %d / 256\n", what, worst, worst_rel_sum, exact);
    };
    auto rnd_full = [&]() { return (uint16_t)(0x3800 + (rand() % 0x0C00)) | (uint16_t)((rand() & 1) << 15); };   // |x| in [0.5, 4), random mantissa
    for (int i = 0; i < 256; i++) hC[i] = 0.f;
    // T1: ONE product per output
    for (int i = 0; i < 512; i++) { hA[i] = 0; hB[i] = 0; }
    for (int m = 0; m < 16; m++) hA[m * 32 + (m % 32)] = rnd_full();
    for (int k = 0; k < 32; k++) for (int n = 0; n < 16; n++) hB[k * 16 + n] = (uint16_t)(3 << (2 * (k & 3)));
    run("T1 one product: full-mantissa A x subnormal B");
    // T2: 32 products, B subnormal codes
    for (int i = 0; i < 512; i++) hA[i] = rnd_full();
    for (int k = 0; k < 32; k++) for (int n = 0; n < 16; n++) hB[k * 16 + n] = (uint16_t)((rand() & 3) << (2 * (k & 3)));
    run("T2 32 products: full-mantissa A x subnormal code B (mixed 4^i)");
    // T2b: A pre-scaled by 2^(6-2i) so that all products have equal magnitude (the kernels' case)
    for (int m = 0; m < 16; m++) for (int k = 0; k < 32; k++) hA[m * 32 + k] = f2h(h2f(rnd_full()) * (float)(1 << (6 - 2 * (k & 3))));
    run("T2b same, A carries 2^(6-2i)");
    // T3: the same values with NORMAL B (codes as integers), A as T2b without prescale
    for (int i = 0; i < 512; i++) hA[i] = rnd_full();
    for (int k = 0; k < 32; k++) for (int n = 0; n < 16; n++) hB[k * 16 + n] = f2h((float)(rand() & 3));
    run("T3 32 products: full-mantissa A x normal integer B");
    // T4: accumulate rounding: C = 1, products sum to an odd multiple of 2^-25 (exactly half an ulp of C) and 3 * 2^-25
    for (int i = 0; i < 512; i++) { hA[i] = 0; hB[i] = 0; }
    for (int m = 0; m < 16; m++) { hA[m * 32] = f2h(1.0f); }
    for (int n = 0; n < 16; n++) hB[0 * 16 + n] = (uint16_t)(n + 1);            // (n + 1) * 2^-24
    for (int i = 0; i < 256; i++) hC[i] = 1.0f;
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    mfma_once<<<1, 64>>>(dA, dB, dC, dD); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    printf("T4 C = 1 + n * 2^-24 (ulp(1) = 2^-23): ");
    for (int n = 0; n < 8; n++) printf(" n=%d -> 1+%g*2^-23", n + 1, (hD[n] - 1.0) * 8388608.0);
    printf("\n");
    // T4b: C = -S, products sum to S + small (cancellation like the sV zero-point term)
    // T5: lo-row case: A subnormal / tiny x B subnormal
    for (int i = 0; i < 256; i++) hC[i] = 0.f;
    for (int i = 0; i < 512; i++) hA[i] = (uint16_t)(rand() & 0x3ff) | (uint16_t)((rand() & 1) << 15);      // subnormal A
    for (int k = 0; k < 32; k++) for (int n = 0; n < 16; n++) hB[k * 16 + n] = (uint16_t)((rand() & 3) << (2 * (k & 3)));
    run("T5 32 products: subnormal A x subnormal B");
    // T7: accumulation chain, all-positive products (the sV code term): error relative to the final sum
    {
        const int steps = 32;
        static uint16_t cA[32 * 512], cB[32 * 512];
        for (int i = 0; i < steps * 512; i++) cA[i] = (uint16_t)(0x2800 + (rand() % 0x1000));     // p' in [0.03, 0.5): positive
        for (int st = 0; st < steps; st++) for (int k = 0; k < 32; k++) for (int n = 0; n < 16; n++) cB[st * 512 + k * 16 + n] = (uint16_t)((rand() & 3) << (2 * (k & 3)));
        for (int st = 0; st < steps; st++) for (int m = 0; m < 16; m++) for (int k = 0; k < 32; k++)
            cA[st * 512 + m * 32 + k] = f2h(h2f(cA[st * 512 + m * 32 + k]) * (float)(1 << (6 - 2 * (k & 3))));
        uint16_t *dcA, *dcB; hipMalloc(&dcA, sizeof cA); hipMalloc(&dcB, sizeof cB);
        hipMemcpy(dcA, cA, sizeof cA, hipMemcpyHostToDevice); hipMemcpy(dcB, cB, sizeof cB, hipMemcpyHostToDevice);
        mfma_chain<<<1, 64>>>(dcA, dcB, dD, steps); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) {
            long double ref = 0;
            for (int st = 0; st < steps; st++) for (int k = 0; k < 32; k++) ref += (long double)h2f(cA[st * 512 + m * 32 + k]) * (long double)h2f(cB[st * 512 + k * 16 + n]);
            worst = fmax(worst, fabs((double)ref - (double)hD[m * 16 + n]) / fabs((double)ref));
        }
        printf("T7 chain of %d accumulating MFMAs, positive terms: worst relative error of the sum %.3e (2^%.1f)\n", steps, worst, log2(worst));
    }
    // T6: hi + lo rows emulation: exact 22-bit products of two random halves x and s, split into hi = RN(x s), lo = x s - hi
    {
        double worst = 0;
        for (int trial = 0; trial < 8; trial++) {
            double exact[16][16]; double big[16][16];
            uint16_t xs[16][32], ss[16][32];
            // rows 0..7: hi of (x, s) pairs, rows 8..15: lo of the same pairs
            for (int m = 0; m < 8; m++) for (int k = 0; k < 32; k++) {
                const float x = h2f(rnd_full()) * (float)(1 << (6 - 2 * (k & 3))) * (trial < 4 ? 1.0f : 0.01f), s = h2f(rnd_full());
                xs[m][k] = f2h(x); ss[m][k] = f2h(s);
                const float xr = h2f(xs[m][k]), sr = h2f(ss[m][k]);
                const uint16_t hi = f2h(xr * sr);
                const float lo = fmaf(xr, sr, -h2f(hi));
                hA[m * 32 + k] = hi; hA[(m + 8) * 32 + k] = f2h(lo);
            }
            for (int k = 0; k < 32; k++) for (int n = 0; n < 16; n++) hB[k * 16 + n] = (uint16_t)((rand() & 3) << (2 * (k & 3)));
            hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
            mfma_once<<<1, 64>>>(dA, dB, dC, dD); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
            for (int m = 0; m < 8; m++) for (int n = 0; n < 16; n++) {
                double ref = 0, bg = 0;
                for (int k = 0; k < 32; k++) {
                    const double p = (double)h2f(xs[m][k]) * (double)h2f(ss[m][k]) * (double)h2f(hB[k * 16 + n]);
                    ref += p; bg = fmax(bg, fabs(p));
                }
                const double got = (double)hD[m * 16 + n] + (double)hD[(m + 8) * 16 + n];
                exact[m][n] = ref; big[m][n] = bg;
                if (bg > 0) worst = fmax(worst, fabs(got - ref) / (bg * 5.9604644775390625e-08));
                if (bg > 0 && fabs(got - ref) / (bg * 5.9604644775390625e-08) > 50) {
                    if (trial < 4) {
                        long double eh = 0, el = 0;
                        for (int k = 0; k < 32; k++) { eh += (long double)h2f(hA[m * 32 + k]) * (long double)h2f(hB[k * 16 + n]); el += (long double)h2f(hA[(m + 8) * 32 + k]) * (long double)h2f(hB[k * 16 + n]); }
                        printf("   row sums: exact hi %.12e  gpu hi %.12e | exact lo %.12e  gpu lo %.12e\n", (double)eh, hD[m * 16 + n], (double)el, hD[(m + 8) * 16 + n]);
                        static int once = 0;
                        if (!once++) for (int k = 0; k < 32; k++) printf("      k=%2d hi 0x%04x %12g lo 0x%04x %14g  B 0x%04x\n", k, hA[m * 32 + k], h2f(hA[m * 32 + k]), hA[(m + 8) * 32 + k], h2f(hA[(m + 8) * 32 + k]), hB[k * 16 + n]);
                    }
                    printf("   BIG trial %d m %d n %d: got %.12e ref %.12e D_hi %.12e D_lo %.12e bg %.3e\n", trial, m, n, got, ref, hD[m * 16 + n], hD[(m + 8) * 16 + n], bg);
                    for (int k = 0; k < 32; k++) {
                        const double pe = (double)h2f(xs[m][k]) * (double)h2f(ss[m][k]);
                        const double hl = (double)h2f(hA[m * 32 + k]) + (double)h2f(hA[(m + 8) * 32 + k]);
                        if (pe != hl) printf("      k=%d x=%g (0x%04x) s=%g P=%.12g hi=%g lo=%.12g (0x%04x)\n", k, h2f(xs[m][k]), xs[m][k], h2f(ss[m][k]), pe, h2f(hA[m * 32 + k]), h2f(hA[(m + 8) * 32 + k]), hA[(m + 8) * 32 + k]);
                    }
                }
                if (trial == 0 && m == 0 && n < 1) {
                    double rh = 0, rl = 0;
                    for (int k = 0; k < 32; k++) { rh += (double)h2f(hA[m * 32 + k]) * (double)h2f(hB[k * 16 + n]); rl += (double)h2f(hA[(m + 8) * 32 + k]) * (double)h2f(hB[k * 16 + n]); }
                    printf("   n=%d: D_hi %.10e (exact %.10e)  D_lo %.10e (exact %.10e)  sum %.10e  ref %.10e  max term %.3e\n", n, hD[m * 16 + n], rh, hD[(m + 8) * 16 + n], rl, got, ref, bg);
                }
            }
            if (trial == 3 || trial == 7) { printf("T6 hi + lo rows vs exact 22-bit products x codes (%s A): worst err %.3f x 2^-24 |max term|\n", trial == 3 ? "O(1..100)" : "O(0.01..1)", worst); worst = 0; }
        }
    }
    return 0;
}

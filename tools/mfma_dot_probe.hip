// Error of ONE v_mfma_f32_16x16x32_f16 dot (C = 0) on sV-like operands: A[t] = p'_t * 2^(6 - 2 i) with log-normal p'
// (softmax of N(0, sigma) scores, max normalised into [1, 2)), B[t] = code << 2 i as fp16 subnormal bits.  Exact products,
// so the only error is the pipe's alignment / truncation.  Prints mean (bias) and rms of (D - exact) / exact.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t* A, const uint16_t* B, float* D) {
    const int l = threadIdx.x;
    h8 a, b;
    for (int e = 0; e < 8; e++) {
        a[e] = __builtin_bit_cast(_Float16, A[(l & 15) * 32 + 8 * (l >> 4) + e]);
        b[e] = __builtin_bit_cast(_Float16, B[(8 * (l >> 4) + e) * 16 + (l & 15)]);
    }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; j++) D[(4 * (l >> 4) + j) * 16 + (l & 15)] = c[j];
}
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
static float h2f(uint16_t u) { _Float16 h = __builtin_bit_cast(_Float16, u); return (float)h; }
static double gauss() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }
int main() {
    uint16_t hA[512], hB[512]; float hD[256];
    uint16_t *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    srand(7);
    for (int mode = 0; mode < 2; mode++)
    for (double sigma = 0; sigma <= 4.01; sigma += 1.0) {
        double sum = 0, sum2 = 0, worst = 0; int cnt = 0;
        for (int trial = 0; trial < 64; trial++) {
            for (int m = 0; m < 16; m++) {
                double p[32], mx = 0;
                for (int t = 0; t < 32; t++) { p[t] = exp(sigma * gauss()); mx = fmax(mx, p[t]); }
                for (int t = 0; t < 32; t++) {
                    const float pv = h2f(f2h((float)(p[t] / mx * 1.5)));                         // fp16 p' <= 1.5
                    hA[m * 32 + t] = f2h(pv * (mode == 0 ? (float)(1 << (6 - 2 * (t & 3))) : 1.0f));   // mode 1: the Z operand (no prefactor)
                }
            }
            for (int t = 0; t < 32; t++) for (int n = 0; n < 16; n++)
                hB[t * 16 + n] = mode == 0 ? (uint16_t)((rand() & 3) << (2 * (t & 3))) : f2h(-1.5f - 0.25f * (rand() & 3));
            hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
            k<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
            for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) {
                long double ref = 0;
                for (int t = 0; t < 32; t++) ref += (long double)h2f(hA[m * 32 + t]) * (long double)h2f(hB[t * 16 + n]);
                if (ref == 0) continue;
                const double rel = ((double)hD[m * 16 + n] - (double)ref) / fabs((double)ref);
                sum += rel; sum2 += rel * rel; worst = fmax(worst, fabs(rel)); cnt++;
            }
        }
        printf("%s sigma %.0f: relative error of the dot: mean %+.3e  rms %.3e  max %.3e   (2^-24 = 5.96e-08)\n", mode == 0 ? "codes " : "zero-pt", sigma, sum / cnt, sqrt(sum2 / cnt), worst);
    }
    return 0;
}

// How wide is the adder of v_mfma_f32_16x16x32_f16 (gfx950)?  One dot: term 0 = 1.0 * 1.0, terms 1..31 = a * b with a * b =
// (1 + 2^-10) * 2^-e (a = 1 + 2^-10, b = 2^-e for e <= 24, split across both operands beyond); C = 0 or C = 1.
// Prints (D - big) / (31 * small) -- 1.0 = every small term fully accumulated, 0.0 = all lost -- and the same with negative
// small terms (floor vs round-toward-zero shows in the sign of the error).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t* A, const uint16_t* B, float c0, float* D) {
    const int l = threadIdx.x;
    h8 a, b;
    for (int e = 0; e < 8; e++) {
        a[e] = __builtin_bit_cast(_Float16, A[(l & 15) * 32 + 8 * (l >> 4) + e]);
        b[e] = __builtin_bit_cast(_Float16, B[(8 * (l >> 4) + e) * 16 + (l & 15)]);
    }
    f4 c = {c0, c0, c0, c0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; j++) D[(4 * (l >> 4) + j) * 16 + (l & 15)] = c[j];
}
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
int main() {
    uint16_t hA[512], hB[512]; float hD[256];
    uint16_t *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    for (int sign = 1; sign >= -1; sign -= 2)
        for (int withc = 0; withc < 2; withc++) {
            printf("small terms %s, C = %d:\n", sign > 0 ? "positive" : "negative", withc);
            for (int e = 8; e <= 34; e++) {
                // row m, column n all identical
                for (int m = 0; m < 16; m++) for (int kk = 0; kk < 32; kk++) {
                    float av = kk == 0 ? (withc ? 0.f : 1.f) : (1.0f + 0.0009765625f) * (e > 14 ? ldexpf(1.f, -(e - 14)) : 1.f) * sign;
                    hA[m * 32 + kk] = f2h(av);
                }
                for (int kk = 0; kk < 32; kk++) for (int n = 0; n < 16; n++) hB[kk * 16 + n] = f2h(kk == 0 ? 1.f : ldexpf(1.f, -(e > 14 ? 14 : e)));
                hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
                k<<<1, 64>>>(dA, dB, withc ? 1.f : 0.f, dD);
                hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
                const double small = (1.0 + 0.0009765625) * ldexp(1.0, -e) * sign;
                printf("  small = (1+2^-10) 2^-%2d: (D - 1) / (31 small) = %.6f   D - 1 = %.9e (exact %.9e, RN fp32 %.9e)\n", e, (hD[0] - 1.0) / (31 * small), hD[0] - 1.0, 31 * small,
                       (double)(float)(1.0 + 31 * small) - 1.0);
            }
        }
    return 0;
}

// v_mfma_f32_16x16x32_f16 on gfx950 for the grouped-query GEMV redesign:
//  (1) operand / result lane maps (A row = lane & 15, B column = lane & 15, k = 8 * (lane >> 4) + e for both,
//      D[row = 4 * (lane >> 4) + reg][col = lane & 15]) checked against a CPU product with asymmetric operands;
//  (2) fp16 SUBNORMAL B operands (a masked 2-bit code read in place: value code * 4^i * 2^-24): kept or flushed?
//  (3) the biased alternative (code | 0x6400 = 1024 + code * 4^i);
//  (4) hi / lo split of the A operand (q * scale as two fp16 rows) = exact 22-bit products;
//  (5) issue rate of the planned inner loop of one 32-token block: 8 MFMA + 36 mask ops + 32 packed-half ops per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_probe.bin mfma_f16_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void mfma_once(const uint16_t* A /*[16][32]*/, const uint16_t* B /*[32][16]*/, float* D /*[16][16]*/) {
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

// (5) compute-only model of one 32-token block of the qK^T kernel: A build (16 pk_mul + 16 pk_fma), B masks
// (4 words: 8 and + 1 shift each), 8 MFMAs into two accumulators; data stays in registers, `iters` blocks per wave.
__global__ __launch_bounds__(256) void block_model(const uint32_t* src, float* out, int iters, int with_valu, int with_mfma) {
    const int l = threadIdx.x & 63;
    u4 w = *(const u4*)(src + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4);
    uint32_t q[16], qf[16], sc[16];
    for (int i = 0; i < 16; i++) {
        q[i] = src[(i * 7 + l) & 1023] | 0x3c003c00u;
        qf[i] = (l & 4) ? q[i] : 0u;
        sc[i] = src[(i * 13 + l) & 1023] | 0x38003800u;
    }
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        uint32_t A[16];
        if (with_valu) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                h2 t, r;
                asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(t) : "v"(qf[i]), "v"(sc[i]));
                asm volatile("v_pk_fma_f16 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(q[i]), "v"(sc[i]), "v"(t));
                A[i] = __builtin_bit_cast(uint32_t, r);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) A[i] = q[i];
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t wc = w[c] + it;   // keeps the masks from being hoisted
            const uint32_t ws = wc >> 8;
            uint32_t b0[4], b1[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                b0[i] = wc & (0x00030003u << (2 * i));
                b1[i] = ws & (0x00030003u << (2 * i));
            }
            if (with_mfma) {
                h8 av = __builtin_bit_cast(h8, (u4){A[4 * c], A[4 * c + 1], A[4 * c + 2], A[4 * c + 3]});
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(h8, (u4){b0[0], b0[1], b0[2], b0[3]}), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(h8, (u4){b1[0], b1[1], b1[2], b1[3]}), acc1, 0, 0, 0);
            } else {
                acc0[0] += __builtin_bit_cast(float, b0[0] ^ b0[1] ^ b0[2] ^ b0[3] ^ A[4 * c]);
                acc1[0] += __builtin_bit_cast(float, b1[0] ^ b1[1] ^ b1[2] ^ b1[3] ^ A[4 * c + 1]);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; u = __builtin_bit_cast(uint16_t, h); return u; }
static float h2f(uint16_t u) { _Float16 h = __builtin_bit_cast(_Float16, u); return (float)h; }

int main() {
    uint16_t hA[16 * 32], hB[32 * 16];
    float hD[256];
    uint16_t *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    srand(1);
    auto run = [&](const char* what, double post) {
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        mfma_once<<<1, 64>>>(dA, dB, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) {
            double ref = 0;
            for (int k = 0; k < 32; k++) ref += (double)h2f(hA[m * 32 + k]) * (double)h2f(hB[k * 16 + n]);
            worst = fmax(worst, fabs(ref * post - hD[m * 16 + n] * post)); scale = fmax(scale, fabs(ref * post));
        }
        printf("%-58s max |err| %.3e  (max |ref| %.3e)  rel %.2e\n", what, worst, scale, worst / scale);
    };
    // (1) layout: asymmetric integer-valued operands, exact in fp32
    for (int m = 0; m < 16; m++) for (int k = 0; k < 32; k++) hA[m * 32 + k] = f2h((float)((m * 3 + k * 5) % 11 - 4));
    for (int k = 0; k < 32; k++) for (int n = 0; n < 16; n++) hB[k * 16 + n] = f2h((float)((k * 7 + n * 2) % 5));
    run("(1) lane maps, integer operands", 1.0);
    // (2) subnormal B: bits = code << (2 i), i = k & 3  -> value code * 4^i * 2^-24; A random
    for (int m = 0; m < 16; m++) for (int k = 0; k < 32; k++) hA[m * 32 + k] = f2h(((rand() % 2001) - 1000) / 250.0f);
    for (int k = 0; k < 32; k++) for (int n = 0; n < 16; n++) hB[k * 16 + n] = (uint16_t)((rand() & 3) << (2 * (k & 3)));
    run("(2) fp16-subnormal B operand (x 2^24)", 16777216.0);
    for (int k = 0; k < 32; k++) for (int n = 0; n < 16; n++) hB[k * 16 + n] = (uint16_t)((rand() & 3) << (2 * (k & 3))) | 0x6400;
    run("(3) biased B operand 0x6400 | code<<2i", 1.0);
    // subnormal A too (tiny q*scale)
    for (int m = 0; m < 16; m++) for (int k = 0; k < 32; k++) hA[m * 32 + k] = (uint16_t)(rand() & 0x3ff) | ((rand() & 1) << 15);
    for (int k = 0; k < 32; k++) for (int n = 0; n < 16; n++) hB[k * 16 + n] = f2h((float)(rand() % 4));
    run("(2b) fp16-subnormal A operand (x 2^24)", 16777216.0);

    // (5) issue-rate model
    const int nb = 1024, iters = 4096;
    uint32_t* dsrc; float* dout;
    hipMalloc(&dsrc, (size_t)nb * 256 * 16 + 4096); hipMalloc(&dout, (size_t)nb * 256 * 4);
    hipMemset(dsrc, 0x5a, (size_t)nb * 256 * 16 + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; mode++) {
        const int wv = mode & 1, wm = (mode >> 1) & 1;
        block_model<<<nb, 256>>>(dsrc, dout, 16, wv, wm);
        hipEventRecord(e0);
        block_model<<<nb, 256>>>(dsrc, dout, iters, wv, wm);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // one iteration = one 32-token block of one kv head = 1536 bytes of cache
        const double blocks = (double)nb * 4 * iters;
        printf("(5) A-build %d  mfma %d : %.1f ns per block-iteration per wave-slot, compute ceiling %.1f TB/s of cache bytes (%d blocks x 4 waves)\n",
               wv, wm, ms * 1e6 / iters / (nb * 4.0 / (256 * 4 * 4.0) > 1 ? nb * 4.0 / (256 * 4) / 4.0 : 1.0) / 4.0, blocks * 1536 / (ms * 1e-3) / 1e12, nb);
    }
    return 0;
}

// LDS read cost when many lanes read the SAME address (the scale operands of the matrix-pipe kernels: the 16 lanes of a
// k-block group all need the same 64 bytes).  16 waves per CU issue `iters` reads each; cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(uint32_t* out, int iters, unsigned long long* cyc) {
    __shared__ uint32_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, kb = lane >> 4, n = lane & 15;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        const int g = it & 15;
        if constexpr (MODE == 0) {          // b128, 16 lanes share an address (4 distinct per wave)
            u4 v = *(const u4*)(lds + g * 64 + kb * 16 + (it & 3) * 4);
            acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        } else if constexpr (MODE == 1) {   // b128, every lane its own address (conflict-free, 1 KiB contiguous)
            u4 v = *(const u4*)(lds + ((g * 256 + lane * 4) & 8191));
            acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        } else if constexpr (MODE == 2) {   // 2 x b64 shared
            u2 a = *(const u2*)(lds + g * 64 + kb * 16 + (it & 3) * 4), b = *(const u2*)(lds + g * 64 + kb * 16 + (it & 3) * 4 + 2);
            acc += a[0] ^ a[1] ^ b[0] ^ b[1];
        } else if constexpr (MODE == 3) {   // 4 x b32 shared
            const uint32_t* p = lds + g * 64 + kb * 16 + (it & 3) * 4;
            uint32_t a, b, c, d;
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"((uint32_t)(uintptr_t)p) : "memory");
            acc += a ^ b ^ c ^ d;
        } else if constexpr (MODE == 4) {   // b128, 4 lanes share an address (16 distinct per wave: rows = (head, channel group))
            u4 v = *(const u4*)(lds + g * 64 + kb * 16 + (n >> 2) * 4);
            acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        } else {                            // b128 all 64 lanes the same address
            u4 v = *(const u4*)(lds + g * 64);
            acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    uint32_t* out; unsigned long long* cyc; const int nb = 1024, iters = 4096;
    hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
    unsigned long long h[1024];
    const char* names[] = {"b128, 16 lanes per address", "b128, distinct addresses", "2 x b64, 16 lanes per address", "4 x b32, 16 lanes per address", "b128, 4 lanes per address", "b128, 64 lanes one address"};
#define RUN(M) { probe<M><<<nb, 256>>>(out, 64, cyc); probe<M><<<nb, 256>>>(out, iters, cyc); hipDeviceSynchronize(); hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost); \
    double s = 0; for (int i = 0; i < nb; i++) s += h[i]; printf("%-34s %7.1f cycles per wave-read of 16 bytes/lane at 16 waves/CU  (= %.1f LDS cycles per CU)\n", names[M], s / nb / iters, s / nb / iters / 16.0 * 1.0); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}

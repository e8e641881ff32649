// Diagnostic: how often does the eight-counter ticket of kivi_mf.hip (ids c, c + 8, ... in the start order of the blocks of class
// c = blockIdx % 8) hand a block exactly its own blockIdx?  (walking the keys of block id blockIdx while the atomic
// is in flight pays only if that is nearly always.)  Blocks of 256 threads with the LDS of a slice block (two per CU), each busy for
// ~50 us after its ticket, grids of one and of several residency waves.
//   hipcc --offload-arch=gfx950 -O3 tools/ticket_order_probe.hip -o tools/ticket_order_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void ticket_kernel(int* counters, int* ids, unsigned long long* t_atomic, int busy_ticks) {
    extern __shared__ uint32_t lds[];
    __shared__ int bid_lds;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const int c = (int)blockIdx.x & 7;
        const int nc = ((int)gridDim.x - c + 7) / 8;
        const int t = __hip_atomic_fetch_add(counters + c * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == nc - 1) __hip_atomic_store(counters + c * 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bid_lds = c + 8 * t;
        ids[blockIdx.x] = c + 8 * t;
        t_atomic[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
    }
    __syncthreads();
    lds[threadIdx.x] = (uint32_t)bid_lds;
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t1) < busy_ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[threadIdx.x ^ 1] == 0xFFFFFFFFu) ids[0] = -1;           // keeps the LDS allocation alive
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    int *counters, *ids;
    unsigned long long* ta;
    const int max_grid = 8192;
    CK(hipMalloc(&counters, 8 * 32 * sizeof(int)));
    CK(hipMemset(counters, 0, 8 * 32 * sizeof(int)));
    CK(hipMalloc(&ids, max_grid * sizeof(int)));
    CK(hipMalloc(&ta, max_grid * sizeof(unsigned long long)));
    CK(hipFuncSetAttribute((const void*)ticket_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    const int lds_bytes[] = {73 * 1024, 36 * 1024};                 // two / four blocks per CU
    const int grids[] = {128, 256, 512, 1024, 2048, 4096};
    std::vector<int> h(max_grid);
    std::vector<unsigned long long> ht(max_grid);
    for (int lb : lds_bytes)
        for (int g : grids) {
            long mism = 0, total = 0, maxdist = 0;
            double tsum = 0, tmax_sum = 0, t99_sum = 0;
            int launches_with_any = 0;
            for (int r = 0; r < reps; r++) {
                ticket_kernel<<<g, 256, lb>>>(counters, ids, ta, 5000);   // 100 MHz clock: 5000 ticks = 50 us
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h.data(), ids, g * sizeof(int), hipMemcpyDeviceToHost));
                CK(hipMemcpy(ht.data(), ta, g * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                long m = 0;
                for (int i = 0; i < g; i++) {
                    if (h[i] != i) { m++; long d = labs((long)h[i] - i) / 8; if (d > maxdist) maxdist = d; }
                    tsum += (double)ht[i];
                }
                std::vector<unsigned long long> srt(ht.begin(), ht.begin() + g);
                std::sort(srt.begin(), srt.end());
                tmax_sum += (double)srt[g - 1];
                t99_sum += (double)srt[(size_t)(0.99 * (g - 1))];
                mism += m;
                total += g;
                launches_with_any += m > 0;
            }
            printf("lds %2d KiB grid %5d: %6ld of %7ld blocks got another id (%.2f %%), launches with any %d / %d, furthest %ld tickets, atomic round trip %.2f us avg, p99 %.2f, slowest of a launch %.2f\n",
                   lb / 1024, g, mism, total, 100.0 * mism / total, launches_with_any, reps, maxdist, tsum / total / 100.0, t99_sum / reps / 100.0, tmax_sum / reps / 100.0);
        }
    return 0;
}

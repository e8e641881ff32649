// Diagnostic: streaming-READ bandwidth ceiling of this MI355X for a few access shapes
// (the roofline denominators quoted in DESIGN.md come from the microarch guide; this measures
// what a plain read-only kernel reaches on the same box, beside our GEMV).
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_read_bw.hip -o tools/hbm_read_bw.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// Each block streams a contiguous chunk; UNR loads in flight per lane.
template <typename V, int UNR, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const V* __restrict__ p, size_t n_per_block, uint32_t* out) {
    const V* base = p + (size_t)blockIdx.x * n_per_block;
    uint32_t acc = 0;
    for (size_t i = threadIdx.x; i < n_per_block; i += 256 * UNR) {
        V v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            size_t j = i + (size_t)u * 256;
            if (j < n_per_block) v[u] = NT ? __builtin_nontemporal_load(base + j) : base[j];
            else v[u] = V{};
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if constexpr (sizeof(V) == 16) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
            else acc ^= v[u][0] ^ v[u][1];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;  // never true in practice; keeps the loads alive
}

template <typename V, int UNR, bool NT>
double run(const char* name, const void* buf, size_t bytes, int blocks, uint32_t* out, int iters) {
    size_t n = bytes / sizeof(V);
    size_t per = n / blocks;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int it = 0; it < iters + 2; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((read_kernel<V, UNR, NT>), dim3(blocks), dim3(256), 0, 0, (const V*)buf, per, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    double med = ts[ts.size() / 2];
    printf("%-44s blocks=%6d  median %8.3f ms  %6.2f TB/s  (min %.3f ms -> %.2f TB/s)\n", name, blocks, med,
           per * blocks * sizeof(V) / med / 1e9, ts[0], per * blocks * sizeof(V) / ts[0] / 1e9);
    return med;
}

int main(int argc, char** argv) {
    size_t gib = argc > 1 ? atoi(argv[1]) : 4;
    size_t bytes = gib << 30;
    void* buf; uint32_t* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    printf("streaming read of %zu GiB\n", gib);
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 64}) {
        run<u32x4, 4, false>("dwordx4 x4 in flight", buf, bytes, blocks, out, 7);
        run<u32x4, 4, true>("dwordx4 x4 in flight, nt", buf, bytes, blocks, out, 7);
        run<u32x4, 8, true>("dwordx4 x8 in flight, nt", buf, bytes, blocks, out, 7);
        run<u32x2, 8, true>("dwordx2 x8 in flight, nt", buf, bytes, blocks, out, 7);
    }
    // a 210 MB launch like one K-GEMV (cold: rotate through the 4 GiB buffer)
    {
        size_t chunk = 210u << 20;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<float> ts;
        for (int it = 0; it < 18; it++) {
            const char* p = (const char*)buf + (size_t)(it % (bytes / chunk)) * chunk;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((read_kernel<u32x4, 4, true>), dim3(2048), dim3(256), 0, 0, (const u32x4*)p, chunk / 16 / 2048, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("210 MiB cold launches (2048 blocks, dwordx4 nt): median %.1f us -> %.2f TB/s, min %.1f us\n", ts[9] * 1e3,
               chunk / ts[9] / 1e9, ts[0] * 1e3);
    }
    return 0;
}

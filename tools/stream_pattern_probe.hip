// How should a wave re-request its ring?  Pure streaming reads in the geometry of the matrix-pipe kernels: every wave of a
// 256-thread block walks its own segment (SEG KiB, the segments of consecutive waves STRIDE KiB apart) in 1 KiB
// wave-loads (16 bytes per lane, nt), RING loads in flight, re-requested either one slot at a time right after its use
// (BURST = 1) or BURST slots together (contiguous BURST KiB).  Prints TB/s for a ~200 MB launch over a rotating 4 GiB buffer.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_pattern_probe.hip -o tools/stream_pattern_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int RING, int BURST>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ p, size_t stride_vec, int pieces, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u32x4* base = p + wave * stride_vec + lane;
    u32x4 r[RING];
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < RING; i++) r[i] = __builtin_nontemporal_load(base + (size_t)(i < pieces ? i : pieces - 1) * 64);
    for (int g0 = 0; g0 < pieces; g0 += RING) {
#pragma unroll
        for (int b0 = 0; b0 < RING; b0 += BURST) {
#pragma unroll
            for (int j = 0; j < BURST; j++) acc ^= r[b0 + j][0] ^ r[b0 + j][1] ^ r[b0 + j][2] ^ r[b0 + j][3];
#pragma unroll
            for (int j = 0; j < BURST; j++) {
                const int gn = g0 + b0 + j + RING;
                r[b0 + j] = __builtin_nontemporal_load(base + (size_t)(gn < pieces ? gn : pieces - 1) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int RING, int BURST>
static void run(const void* buf, size_t bytes, int waves, int seg_kib, int stride_kib, uint32_t* out) {
    const size_t per_launch = (size_t)waves * stride_kib * 1024;
    const int nrot = (int)(bytes / per_launch);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int it = 0; it < 14; it++) {
        const char* b = (const char*)buf + (size_t)(it % nrot) * per_launch;
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<RING, BURST>), dim3(waves / 4), dim3(256), 0, 0, (const u32x4*)b, (size_t)stride_kib * 64, seg_kib, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double mb = (double)waves * seg_kib * 1024 / 1e6;
    printf("waves %5d seg %3d KiB stride %4d KiB  ring %2d burst %d:  %7.2f us  %5.2f TB/s  (%.0f MB)\n", waves, seg_kib, stride_kib, RING, BURST,
           ts[ts.size() / 2] * 1e3, mb / ts[ts.size() / 2] / 1e3, mb);
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    void* buf; uint32_t* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int waves : {4096, 5120, 8192}) {
        for (int seg : {24, 48}) {
            const int stride = seg;               // dense segments
            run<4, 1>(buf, bytes, waves, seg, stride, out);
            run<4, 2>(buf, bytes, waves, seg, stride, out);
            run<4, 4>(buf, bytes, waves, seg, stride, out);
            run<8, 1>(buf, bytes, waves, seg, stride, out);
            run<8, 4>(buf, bytes, waves, seg, stride, out);
            run<8, 8>(buf, bytes, waves, seg, stride, out);
            run<2, 1>(buf, bytes, waves, seg, stride, out);
            run<2, 2>(buf, bytes, waves, seg, stride, out);
        }
    }
    // the super-blocks of one unit 768 KiB apart (the KT store of the bench shape): waves of a block on consecutive super-blocks
    run<4, 1>(buf, bytes, 4096, 24, 768, out);
    run<4, 4>(buf, bytes, 4096, 24, 768, out);
    return 0;
}

// ABI twin of the reference's native entry point on ITS kernel-input layout
// (gemv_forward_cuda_outer_dim, quant/csrc/gemv_cuda.h:13-21, gemv_cuda.cu:511-557):
//   in (BS, 1, IC) fp16, kernel (BS_kv, OC/fpi, IC) int32, scale/zeros (BS_kv, OC/g, IC) fp16,
//   out (BS, 1, OC) fp16.
// The fast paths (kivi_gemv_k / kivi_gemv_v) read the hook-state layout and never need
// these transposed tensors; these kernels exist for callers that already hold them -- an
// UNMODIFIED quant/matmul.py:198-219 lands here (quant/gemv.py:117,154 too).
//   gemv_outer_dim_wide_kernel (round 6; IC % 4 == 0, group_size 32 / 64): a wave per quantisation GROUP of output
//       columns -- 16-byte loads along IC for the code rows, scale / zero point / input loaded ONCE per group (the
//       reference re-reads them for every packed row, gemv_cuda.cu:370-387), several groups' loads in flight per
//       wave, the fpi sums of a packed row reduced by a halving butterfly; long IC (the sV shape) is split over the
//       four waves of a block.  The same fp32 arithmetic as gemv_cuda.cu:401-426 up to summation order.
//   gemv_outer_dim_kernel: the general fallback (any IC / group size): one wave per (batch row, packed output row),
//       dword loads, written for clarity.
#include "kivi_common.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int BITS>
__global__ __launch_bounds__(256) void gemv_outer_dim_kernel(const uint16_t* __restrict__ in,
                                                             const uint32_t* __restrict__ kernel,
                                                             const uint16_t* __restrict__ scale,
                                                             const uint16_t* __restrict__ zeros,
                                                             uint16_t* __restrict__ out, int64_t IC, int64_t OC,
                                                             int g, int ratio, int nrow_blocks) {
    constexpr int FPI = 32 / BITS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t bidx = blockIdx.x / nrow_blocks;            // batch_idx
    const int64_t row = (int64_t)(blockIdx.x % nrow_blocks) * 4 + wave;  // packed_oc_idx
    const int64_t nrow = (OC + FPI - 1) / FPI;
    if (row >= nrow) return;
    const int64_t bk = bidx / ratio;                           // gemv_cuda.cu:361-365
    const int64_t grp = (row * FPI) / g;                       // :357
    const uint32_t* wp = kernel + (bk * nrow + row) * IC;
    const uint16_t* sp = scale + (bk * (OC / g) + grp) * IC;
    const uint16_t* zp = zeros + (bk * (OC / g) + grp) * IC;
    const uint16_t* ip = in + bidx * IC;
    float acc[FPI];
#pragma unroll
    for (int p = 0; p < FPI; p++) acc[p] = 0.f;
    float z = 0.f;
    for (int64_t ic = lane; ic < IC; ic += 64) {
        const float x = h2f_bits(ip[ic]);
        const float xs = x * h2f_bits(sp[ic]) * qs_factor<KIVI_UNPACK_MIX>();
        z = __builtin_fmaf(x, h2f_bits(zp[ic]), z);
        accum_word<BITS, KIVI_UNPACK_MIX>(wp[ic], xs, acc);
    }
#pragma unroll
    for (int p = 0; p < FPI; p++) acc[p] *= post_scale<BITS, KIVI_UNPACK_MIX>(p);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) z += __shfl_xor(z, m);
    // halve the accumulator set while there is more than one value, then plain xor-adds
    int n = FPI, off = 0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        if (n > 1) {
            const int half = n / 2;
            const bool upper = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < FPI / 2; i++) {
                if (i < half) {
                    const float send = upper ? acc[i] : acc[i + half];
                    const float keep = upper ? acc[i + half] : acc[i];
                    acc[i] = keep + __shfl_xor(send, m);
                }
            }
            off += upper ? half : 0;
            n = half;
        } else {
            acc[0] += __shfl_xor(acc[0], m);
        }
    }
    // lanes 0..FPI-1 hold distinct channels `off`
    if (lane < FPI) {
        const int64_t oc = row * FPI + off;
        if (oc < OC) out[bidx * OC + oc] = f2h_bits(acc[0] + z);
    }
}

// ---- the tuned form.  LPR lanes per packed row (each lane 4 consecutive ic = one 16-byte code load): 32 when IC <= 128 (two rows
// side by side in a wave), else 64.  RPL packed rows per lane = (g / fpi) / (64 / LPR).
// NG > 1 (rows of at most LPR * 4 ic: ONE pass): NG consecutive groups per wave, all their loads issued before the first is used (a
// wave then has NG x 1.5 KiB in flight instead of one round trip per group), the groups finished one after the other.
// SPLIT: the four waves of a block take a quarter of IC each and meet in LDS (long IC: few groups, long rows).

// sums over the LPR lanes of a packed row: FPI values per lane -> lane sl holds column `off` (sl < FPI; returned through `off`)
template <int V> struct od_ic { static constexpr int value = V; };
template <int BITS> struct OdRow { float a[32 / BITS]; };

// value of lane (l ^ M): a DPP operand modifier where the pattern exists on gfx950 (lanes 1, 2: quad permutations; 8: a rotation by half a
// 16-lane row), the LDS crossbar (ds_bpermute) otherwise -- 15 of the 20 exchanges of a 2-bit group's reduction are DPP
template <int M>
__device__ __forceinline__ float od_xor(float v) {
    if constexpr (M == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    else if constexpr (M == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    else if constexpr (M == 8) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
    else return __shfl_xor(v, M);
}
template <int LPR>
__device__ __forceinline__ float od_row_sum(float z) {          // sum over the LPR lanes of a packed row (every lane gets it)
    z += od_xor<1>(z);
    z += od_xor<2>(z);
    z += od_xor<4>(z);
    z += od_xor<8>(z);
    z += od_xor<16>(z);
    if constexpr (LPR == 64) z += od_xor<32>(z);
    return z;
}
template <int BITS, int LPR>
__device__ __forceinline__ float outer_dim_row_reduce(float (&a)[32 / BITS], int lane, int& off) {
    constexpr int FPI = 32 / BITS;
#pragma unroll
    for (int p = 0; p < FPI; p++) a[p] *= post_scale<BITS, KIVI_UNPACK_MIX>(p);
    off = 0;
    // step K exchanges across lane bit K: while more than one value is left (FPI >> K > 1) the set is halved -- the lane with the bit set
    // keeps the upper half --, then plain sums.  Every index into a[] is a compile-time constant (mf_ic steps, `if constexpr`).
    auto step = [&](auto kc) {
        constexpr int K = decltype(kc)::value, M = 1 << K;
        if constexpr (M < LPR) {
            if constexpr ((FPI >> K) > 1) {
                constexpr int half = FPI >> (K + 1);
                const bool upper = (lane & M) != 0;
#pragma unroll
                for (int i = 0; i < half; i++) {
                    // (the two values are pinned in registers first: hipcc otherwise folds the selects into ONE dynamically indexed
                    // read of a[] -- and the whole accumulator array moves to scratch memory)
                    float lo = a[i], hi = a[i + half];
                    asm volatile("" : "+v"(lo), "+v"(hi));
                    const float send = upper ? lo : hi;
                    const float keep = upper ? hi : lo;
                    a[i] = keep + od_xor<M>(send);
                }
                off += upper ? half : 0;
            } else {
                a[0] += od_xor<M>(a[0]);
            }
        }
    };
    step(od_ic<0>{}); step(od_ic<1>{}); step(od_ic<2>{}); step(od_ic<3>{}); step(od_ic<4>{}); step(od_ic<5>{});
    return a[0];
}

__device__ __forceinline__ uint16_t half_of(const u32x2& v, int j) {       // element j (0..3) of four packed halves
    const uint32_t w = (j >> 1) ? v[1] : v[0];
    return (uint16_t)((j & 1) ? (w >> 16) : (w & 0xFFFFu));
}

template <int BITS, int LPR, int RPL, int NG, bool SPLIT>
__global__ __launch_bounds__(256) void gemv_outer_dim_wide_kernel(const uint16_t* __restrict__ in, const uint32_t* __restrict__ kernel,
                                                                  const uint16_t* __restrict__ scale, const uint16_t* __restrict__ zeros,
                                                                  uint16_t* __restrict__ out, int64_t IC, int64_t OC, int g, int ratio,
                                                                  int64_t ntask) {
    constexpr int FPI = 32 / BITS;
    constexpr int RPP = 64 / LPR;                               // packed rows side by side in one wave
    constexpr int RPG = RPL * RPP;                              // packed rows per group = g / FPI
    static_assert(!(SPLIT && NG > 1), "split rows: one group per block");
    __shared__ float part[SPLIT ? 4 : 1][RPG * FPI + 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane / LPR, sl = lane % LPR;                // which of the RPP rows, position along IC
    const int64_t ngrp = OC / g;
    const int64_t nrow = OC / FPI;
    // tasks = (batch row, group); SPLIT: one task per block, else NG consecutive tasks per wave
    const int64_t task0 = SPLIT ? (int64_t)blockIdx.x : ((int64_t)blockIdx.x * 4 + wave) * NG;
    if (task0 >= ntask) return;
    if constexpr (NG > 1) {
        // ---- one pass (IC <= LPR * 4): requests of all NG groups, then group after group
        const int64_t ic0 = sl * 4;
        const bool live = ic0 < IC;
        u32x4 wv[NG][RPL];
        u32x2 sv[NG], zv[NG], xv[NG];
#pragma unroll
        for (int t = 0; t < NG; t++) {
            const int64_t task = task0 + t < ntask ? task0 + t : ntask - 1;
            const int64_t bidx = task / ngrp, grp = task - bidx * ngrp, bk = bidx / ratio;      // gemv_cuda.cu:361-365
            const bool on = live && task0 + t < ntask;
            sv[t] = on ? *(const u32x2*)(scale + (bk * ngrp + grp) * IC + ic0) : u32x2{0, 0};
            zv[t] = on ? *(const u32x2*)(zeros + (bk * ngrp + grp) * IC + ic0) : u32x2{0, 0};
            xv[t] = on ? *(const u32x2*)(in + bidx * IC + ic0) : u32x2{0, 0};
#pragma unroll
            for (int r = 0; r < RPL; r++)
                wv[t][r] = on ? __builtin_nontemporal_load((const u32x4*)(kernel + (bk * nrow + grp * RPG + r * RPP + sub) * IC + ic0)) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int t = 0; t < NG; t++) {
            float acc[RPL][FPI], z = 0.f;
#pragma unroll
            for (int r = 0; r < RPL; r++)
#pragma unroll
                for (int p = 0; p < FPI; p++) acc[r][p] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float x = h2f_bits(half_of(xv[t], j));
                const float xs = x * h2f_bits(half_of(sv[t], j)) * qs_factor<KIVI_UNPACK_MIX>();
                z = __builtin_fmaf(x, h2f_bits(half_of(zv[t], j)), z);
#pragma unroll
                for (int r = 0; r < RPL; r++) accum_word<BITS, KIVI_UNPACK_MIX>(wv[t][r][j], xs, acc[r]);
            }
            z = od_row_sum<LPR>(z);
            const int64_t task = task0 + t;
            const int64_t bidx = task / ngrp, grp = task - bidx * ngrp;
#pragma unroll
            for (int r = 0; r < RPL; r++) {
                int off;
                const float v = outer_dim_row_reduce<BITS, LPR>(acc[r], lane, off);
                if (task < ntask && sl < FPI) out[bidx * OC + grp * g + (r * RPP + sub) * FPI + off] = f2h_bits(v + z);
            }
        }
    } else {
        // ---- one group, any IC: passes of LPR * 4 ic (SPLIT: this wave's quarter of the row)
        int64_t ic_lo = 0, ic_hi = IC;
        if constexpr (SPLIT) {
            const int64_t chunk = ((IC + 3) / 4 + 3) / 4 * 4;  // a multiple of 4 ic per wave
            ic_lo = wave * chunk;
            ic_hi = ic_lo + chunk < IC ? ic_lo + chunk : IC;
        }
        const int64_t bidx = task0 / ngrp, grp = task0 - bidx * ngrp, bk = bidx / ratio;          // gemv_cuda.cu:361-365
        const uint32_t* wp = kernel + (bk * nrow + grp * RPG + sub) * IC;                       // the lane's first packed row (:357)
        const uint16_t* sp = scale + (bk * ngrp + grp) * IC;
        const uint16_t* zp = zeros + (bk * ngrp + grp) * IC;
        const uint16_t* ip = in + bidx * IC;
        OdRow<BITS> acc[RPL];
        float z = 0.f;
#pragma unroll
        for (int r = 0; r < RPL; r++)
#pragma unroll
            for (int p = 0; p < FPI; p++) acc[r].a[p] = 0.f;
        for (int64_t icb = ic_lo; icb < ic_hi; icb += LPR * 4) {  // (uniform trip count; lanes past the end contribute zeros)
            const int64_t ic0 = icb + sl * 4;
            const bool on = ic0 < ic_hi;
            const u32x2 sv = on ? *(const u32x2*)(sp + ic0) : u32x2{0, 0};
            const u32x2 zv = on ? *(const u32x2*)(zp + ic0) : u32x2{0, 0};
            const u32x2 xv = on ? *(const u32x2*)(ip + ic0) : u32x2{0, 0};
            u32x4 wv[RPL];
#pragma unroll
            for (int r = 0; r < RPL; r++)
                wv[r] = on ? __builtin_nontemporal_load((const u32x4*)(wp + (int64_t)(r * RPP) * IC + ic0)) : u32x4{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float x = h2f_bits(half_of(xv, j));
                const float xs = x * h2f_bits(half_of(sv, j)) * qs_factor<KIVI_UNPACK_MIX>();
                z = __builtin_fmaf(x, h2f_bits(half_of(zv, j)), z);
#pragma unroll
                for (int r = 0; r < RPL; r++) accum_word<BITS, KIVI_UNPACK_MIX>(wv[r][j], xs, acc[r].a);
            }
        }
        z = od_row_sum<LPR>(z);
#pragma unroll
        for (int r = 0; r < RPL; r++) {
            int off;
            const float v = outer_dim_row_reduce<BITS, LPR>(acc[r].a, lane, off);
            if constexpr (SPLIT) {
                if (sl < FPI) part[wave][(r * RPP + sub) * FPI + off] = v;
            } else if (sl < FPI) {
                out[bidx * OC + grp * g + (r * RPP + sub) * FPI + off] = f2h_bits(v + z);
            }
        }
        if constexpr (SPLIT) {
            if (lane == 0) part[wave][RPG * FPI] = z;
            __syncthreads();
            const float zt = (part[0][RPG * FPI] + part[1][RPG * FPI]) + (part[2][RPG * FPI] + part[3][RPG * FPI]);
            for (int i = threadIdx.x; i < RPG * FPI; i += 256)
                out[bidx * OC + grp * g + i] = f2h_bits((part[0][i] + part[1][i]) + (part[2][i] + part[3][i]) + zt);
        }
    }
}

// ---- short rows (IC <= 256: the qK^T shape), round 6: the dot axis back INSIDE the lane.  With IC contiguous a lane of the wide kernel
// above holds 4 channels of one packed row, so every row needs a 16-value reduction across its 32 lanes (~60 of ~180 vector instructions
// per group).  Here a wave takes 64 consecutive packed rows and TRANSPOSES them through the LDS, 32 ic at a time: coalesced 16-byte
// global loads (8 lanes per row segment) -> LDS tile [row][36 words] -> lane l reads ITS row with 16-byte LDS reads (the pitch of 36
// words keeps both directions conflict-free) and accumulates its fpi outputs over all of IC exactly as kivi_gemv_k does on the
// hook-state layout (accum_word, 1.56 instructions per code; no cross-lane step at all).  The group's scale / zero point come from a
// small second tile (one fp16 pair per two ic, lanes of a group read the same word), the input row from an fp32 copy in the LDS
// (broadcast reads).  The next 32 ic are in flight in registers while the current ones are multiplied.  Outputs: fpi consecutive fp16
// per lane, consecutive lanes -> consecutive addresses.
// CH: ic per chunk (32: fewer, larger stages, 29 KB of LDS per two-wave block; 16: 17 KB, twice the resident waves)
// float(low | high half of a packed fp16 pair) * x + c in ONE instruction (v_fma_mix_f32: the half is converted on the fly, one rounding)
__device__ __forceinline__ float od_mix_lo(uint32_t hp, float x, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(x), "v"(c));
    return d;
}
__device__ __forceinline__ float od_mix_hi(uint32_t hp, float x, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(x), "v"(c));
    return d;
}

template <int BITS, int CH>
__global__ __launch_bounds__(128) void gemv_outer_dim_rows_kernel(const uint16_t* __restrict__ in, const uint32_t* __restrict__ kernel,
                                                                  const uint16_t* __restrict__ scale, const uint16_t* __restrict__ zeros,
                                                                  uint16_t* __restrict__ out, int IC, int64_t OC, int g, int ratio, int tiles_per_b, int64_t ntile) {
    constexpr int FPI = 32 / BITS;
    constexpr int P = CH + 4, PS = CH / 2 + 1;                 // code-tile pitch (words: = 4 mod 8, so that 16-byte accesses of 8 consecutive lanes never share a bank); scale / zero tile pitch (words = 2 halves)
    constexpr int LPR = CH / 4;                                // lanes per row segment of a chunk (16 bytes each)
    constexpr int NLD = 64 / (64 / LPR);                       // code loads per lane and chunk: 64 rows / (64 / LPR rows per instruction)
    __shared__ __attribute__((aligned(16))) uint32_t codes_lds[2][64 * P];
    __shared__ uint32_t sm_lds[2][2][32 * PS];                 // [wave][scale | zero][group][ic pair]
    __shared__ __attribute__((aligned(16))) float x_lds[2][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tile = (int64_t)blockIdx.x * 2 + wave;
    if (tile >= ntile) return;                                  // (whole wave; the waves of a block share nothing but the LDS allocation)
    const int64_t bidx = tile / tiles_per_b;
    const int t_in_b = (int)(tile - bidx * tiles_per_b);
    const int64_t nrow = OC / FPI, ngrp = OC / g;
    const int rpg = g / FPI;                                    // packed rows per group: 2, 4, 8 (a power of two)
    const int rsh = rpg == 2 ? 1 : (rpg == 4 ? 2 : (rpg == 8 ? 3 : 4));
    const int64_t row0 = (int64_t)t_in_b * 64;                  // first packed row of the tile
    if (row0 >= nrow) return;                                   // (whole wave)
    const int64_t bk = bidx / ratio;                            // gemv_cuda.cu:361-365
    const int nrows = nrow - row0 < 64 ? (int)(nrow - row0) : 64;
    const int ngt = (nrows + rpg - 1) >> rsh;                   // groups of the tile
    const uint32_t* wp = kernel + (bk * nrow + row0) * IC;
    const uint16_t* sp = scale + (bk * ngrp + (row0 >> rsh)) * IC;
    const uint16_t* zp = zeros + (bk * ngrp + (row0 >> rsh)) * IC;
    uint32_t* ct = codes_lds[wave];
    uint32_t* st = sm_lds[wave][0];
    uint32_t* zt = sm_lds[wave][1];
    float* xl = x_lds[wave];
    // the input row as fp32 (zeros past IC: partial chunks then contribute nothing)
    for (int i = lane; i < 256; i += 64) xl[i] = i < IC ? h2f_bits(in[bidx * IC + i]) : 0.f;
    const int nch = (IC + CH - 1) / CH;
    // what a lane fetches per chunk: NLD x 16 bytes of codes (rows i * (64 / LPR) + lane / LPR, words (lane % LPR) * 4 ..), 16 bytes of
    // scale and of zero points per 64 / (CH / 8) groups (halves (lane % (CH / 8)) * 8 ..)
    constexpr int LPG = CH / 8;                                // lanes per group segment of scale / zero (8 halves each)
    constexpr int NSM = (32 * LPG + 63) / 64;                  // scale loads per lane and chunk (32 groups at most)
    u32x4 cw[NLD], sw[NSM], zw[NSM];
    auto fetch = [&](int c) {
        const int ic0 = c * CH;
        const int w = ic0 + (lane % LPR) * 4;
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int r = i * (64 / LPR) + lane / LPR;
            cw[i] = (r < nrows && w < IC) ? __builtin_nontemporal_load((const u32x4*)(wp + (int64_t)r * IC + w)) : u32x4{0, 0, 0, 0};
        }
        const int h = ic0 + (lane % LPG) * 8;
#pragma unroll
        for (int i = 0; i < NSM; i++) {
            const int gq = i * (64 / LPG) + lane / LPG;
            const bool on = gq < ngt && h < IC;                 // (IC % 8 == 0 is required by the dispatch)
            sw[i] = on ? *(const u32x4*)(sp + (int64_t)gq * IC + h) : u32x4{0, 0, 0, 0};
            zw[i] = on ? *(const u32x4*)(zp + (int64_t)gq * IC + h) : u32x4{0, 0, 0, 0};
        }
    };
    float acc[FPI], z = 0.f;
#pragma unroll
    for (int p = 0; p < FPI; p++) acc[p] = 0.f;
    const int gl = lane >> rsh;                                 // this lane's group inside the tile
    fetch(0);
    for (int c = 0; c < nch; c++) {
        __builtin_amdgcn_wave_barrier();                        // the previous chunk's LDS reads are over (LDS operations of a wave complete in order)
#pragma unroll
        for (int i = 0; i < NLD; i++) *(u32x4*)(ct + (i * (64 / LPR) + lane / LPR) * P + (lane % LPR) * 4) = cw[i];
#pragma unroll
        for (int i = 0; i < NSM; i++) {
            const int gq = i * (64 / LPG) + lane / LPG;
            if (gq < 32) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    st[gq * PS + (lane % LPG) * 4 + k] = sw[i][k];
                    zt[gq * PS + (lane % LPG) * 4 + k] = zw[i][k];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (c + 1 < nch) fetch(c + 1);                          // in flight during this chunk's arithmetic
        const float* xc = xl + c * CH;
#pragma unroll
        for (int j4 = 0; j4 < CH / 4; j4++) {
            const u32x4 w4 = *(const u32x4*)(ct + lane * P + j4 * 4);
            const f32x4_t x4 = *(const f32x4_t*)(xc + j4 * 4);
            const uint32_t s01 = st[gl * PS + j4 * 2], s23 = st[gl * PS + j4 * 2 + 1];
            const uint32_t z01 = zt[gl * PS + j4 * 2], z23 = zt[gl * PS + j4 * 2 + 1];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t sp2 = j < 2 ? s01 : s23, zp2 = j < 2 ? z01 : z23;
                const float x = x4[j];
                // x * scale and z += x * zero straight from the packed halves (no shift / convert: 2 instead of 6 instructions per word)
                const float xs = (j & 1) ? od_mix_hi(sp2, x, 0.f) : od_mix_lo(sp2, x, 0.f);
                z = (j & 1) ? od_mix_hi(zp2, x, z) : od_mix_lo(zp2, x, z);
                accum_word<BITS, KIVI_UNPACK_MIX>(w4[j], xs * qs_factor<KIVI_UNPACK_MIX>(), acc);
            }
        }
    }
    if (lane < nrows) {
        uint16_t o[FPI];
#pragma unroll
        for (int p = 0; p < FPI; p++) o[p] = f2h_bits(acc[p] * post_scale<BITS, KIVI_UNPACK_MIX>(p) + z);
        uint16_t* dst = out + bidx * OC + (row0 + lane) * FPI;
#pragma unroll
        for (int p = 0; p < FPI; p += 8) {
            u32x4 v;
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = (uint32_t)o[p + 2 * k] | ((uint32_t)o[p + 2 * k + 1] << 16);
            *(u32x4*)(dst + p) = v;
        }
    }
}

// Legacy AWQ-style INNER-dim grouped 4-bit GEMV (gemv_kernel_g64 / gemv_kernel_g128, gemv_cuda.cu:60-184):
//   out[b, oc] = fp16( sum_ic (scale[oc, ic/g] * code[oc, ic] + zero[oc, ic/g]) * in[b, ic] )
// weight (OC, IC/8) int32 packed along IC, scale / zeros (OC, >= IC/g) fp16 with row pitch `sz_pitch`.
// Not on the KV-cache path (only the reference's disabled scripts call it, quant/gemv.py:188,225); kept for surface
// parity, written for clarity: one wave per (b, oc), 32 codes per lane per pass.
__global__ __launch_bounds__(256) void gemv_awq_kernel(const uint16_t* __restrict__ in, const uint32_t* __restrict__ weight,
                                                       const uint16_t* __restrict__ scale,
                                                       const uint16_t* __restrict__ zeros, uint16_t* __restrict__ out,
                                                       int64_t IC, int64_t OC, int g, int64_t sz_pitch) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t oc = (int64_t)blockIdx.x * 4 + wave;
    const int64_t b = blockIdx.y;
    if (oc >= OC) return;
    const int64_t ww = IC / 8;
    const uint32_t* wrow = weight + oc * ww;
    const uint16_t* xrow = in + b * IC;
    float psum = 0.f;
    for (int64_t w0 = (int64_t)lane * 4; w0 < ww; w0 += 256) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t w = w0 + j;
            if (w < ww) {
                uint32_t word = wrow[w];
                const int64_t gi = (w * 8) / g;
                const float sc = h2f_bits(scale[oc * sz_pitch + gi]), zp = h2f_bits(zeros[oc * sz_pitch + gi]);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float dq = __builtin_fmaf(sc, (float)(word & 0xFu), zp);   // gemv_cuda.cu:101
                    psum = __builtin_fmaf(dq, h2f_bits(xrow[w * 8 + i]), psum);      // :103
                    word >>= 4;
                }
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) psum += __shfl_xor(psum, m);
    if (lane == 0) out[b * OC + oc] = f2h_bits(psum);
}

}  // namespace

extern "C" int kivi_gemv_awq(const void* in, const void* kernel, const void* scale, const void* zeros, void* out, int64_t B,
                             int64_t IC, int64_t OC, int bit, int group_size, int64_t sz_pitch, kivi_stream_t stream) {
    KIVI_REQUIRE(bit == 4, KIVI_EINVAL, "kivi_gemv_awq: the reference kernels are 4-bit only (PACK_FACTOR 8), got %d", bit);
    KIVI_REQUIRE(group_size == 64 || group_size == 128, KIVI_EINVAL,
                 "kivi_gemv_awq: group_size must be 64 or 128 (gemv_cuda.cu:227-244), got %d", group_size);
    KIVI_REQUIRE(IC > 0 && IC % group_size == 0 && OC >= 0 && B >= 0 && sz_pitch >= IC / group_size, KIVI_EINVAL,
                 "kivi_gemv_awq: IC=%lld must be a multiple of group_size=%d", (long long)IC, group_size);
    KIVI_REQUIRE(B < 65536, KIVI_EINVAL, "kivi_gemv_awq: batch too large");
    if (B == 0 || OC == 0) return 0;
    dim3 grid((unsigned)((OC + 3) / 4), (unsigned)B);
    hipLaunchKernelGGL(gemv_awq_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in,
                       (const uint32_t*)kernel, (const uint16_t*)scale, (const uint16_t*)zeros, (uint16_t*)out, IC, OC,
                       group_size, sz_pitch);
    return kivi_launch_status("gemv_awq");
}

extern "C" int kivi_gemv_outer_dim(const void* in, const void* kernel, const void* scale, const void* zeros, void* out,
                                   int64_t BS, int64_t IC, int64_t OC, int bit, int group_size, int nh, int nh_kv,
                                   kivi_stream_t stream) {
    KIVI_REQUIRE(bit == 2 || bit == 4, KIVI_EINVAL, "kivi_gemv_outer_dim: bit must be 2 or 4 (matmul.py:215), got %d", bit);
    KIVI_REQUIRE(nh_kv > 0 && nh > 0 && nh % nh_kv == 0, KIVI_EINVAL,
                 "kivi_gemv_outer_dim: nh %% nh_kv != 0 (matmul.py:216): nh=%d nh_kv=%d", nh, nh_kv);
    const int fpi = 32 / bit;
    KIVI_REQUIRE(group_size > 0 && group_size % fpi == 0 && OC % group_size == 0, KIVI_EINVAL,
                 "kivi_gemv_outer_dim: OC=%lld must be a multiple of group_size=%d", (long long)OC, group_size);
    KIVI_REQUIRE(BS >= 0 && IC >= 0 && OC >= 0, KIVI_EINVAL, "kivi_gemv_outer_dim: negative size");
    if (BS == 0 || OC == 0) return 0;
    const int64_t nrow = (OC + fpi - 1) / fpi;
    const int nrb = (int)((nrow + 3) / 4);
    KIVI_REQUIRE(BS * nrb < ((int64_t)1 << 31), KIVI_EINVAL, "kivi_gemv_outer_dim: grid too large");
    dim3 grid((unsigned)(BS * nrb));
    hipStream_t s = (hipStream_t)stream;
    // the tuned form: rows of whole 16-byte chunks (IC % 4 == 0, 16-byte aligned bases), group_size 32 / 64
    const bool wide_ok = IC > 0 && IC % 4 == 0 && (group_size == 32 || group_size == 64) && (uintptr_t)kernel % 16 == 0 &&
                         (uintptr_t)scale % 8 == 0 && (uintptr_t)zeros % 8 == 0 && (uintptr_t)in % 8 == 0;
    static const char* old_only = KIVI_TUNE_ENV("KIVI_COMPAT_OLD");      // tuning builds, A/B: the general one-wave-per-packed-row kernel
    static const char* no_rows = KIVI_TUNE_ENV("KIVI_COMPAT_NO_ROWS");   // tuning builds, A/B: the wide kernel for short rows too
    // short rows (the qK^T shape): the transposing kernel -- lane = packed row, no cross-lane sums
    const bool rows_ok = wide_ok && IC <= 256 && IC % 8 == 0 && (uintptr_t)scale % 16 == 0 && (uintptr_t)zeros % 16 == 0 && (uintptr_t)out % 16 == 0 &&
                         (group_size / fpi) <= 16;
    if (rows_ok && !(old_only && atoi(old_only)) && !(no_rows && atoi(no_rows))) {
        const int64_t tiles_per_b = (nrow + 63) / 64;
        const int64_t ntile = BS * tiles_per_b;
        KIVI_REQUIRE((ntile + 1) / 2 < ((int64_t)1 << 31), KIVI_EINVAL, "kivi_gemv_outer_dim: grid too large");
        const dim3 grid2((unsigned)((ntile + 1) / 2));
        static const char* fch = KIVI_TUNE_ENV("KIVI_COMPAT_CH");       // tuning builds, A/B: ic per chunk (16 | 32)
        const int ch = fch ? atoi(fch) : 32;
#define KIVI_ROWS(B_, CH_)                                                                                                                       \
    hipLaunchKernelGGL((gemv_outer_dim_rows_kernel<B_, CH_>), grid2, dim3(128), 0, s, (const uint16_t*)in, (const uint32_t*)kernel, (const uint16_t*)scale, \
                       (const uint16_t*)zeros, (uint16_t*)out, (int)IC, OC, group_size, nh / nh_kv, (int)tiles_per_b, ntile)
        if (bit == 2) { if (ch == 16) KIVI_ROWS(2, 16); else KIVI_ROWS(2, 32); }
        else { if (ch == 16) KIVI_ROWS(4, 16); else KIVI_ROWS(4, 32); }
#undef KIVI_ROWS
        return kivi_launch_status("gemv_outer_dim_rows");
    }
    if (wide_ok && !(old_only && atoi(old_only))) {
        const int64_t ntask = BS * (OC / group_size);
        const bool split = IC > 512;                              // long rows (the sV shape): a block per group, IC over its four waves
#define KIVI_WIDE(B_, LPR_, RPL_, NG_, SP_)                                                                                         \
    hipLaunchKernelGGL((gemv_outer_dim_wide_kernel<B_, LPR_, RPL_, NG_, SP_>), dim3((unsigned)((SP_) ? ntask : (ntask + 4 * (NG_) - 1) / (4 * (NG_)))), \
                       dim3(256), 0, s, (const uint16_t*)in, (const uint32_t*)kernel, (const uint16_t*)scale, (const uint16_t*)zeros,       \
                       (uint16_t*)out, IC, OC, group_size, nh / nh_kv, ntask)
        const int rpg = group_size / fpi;                         // packed rows per group: 2 / 4 (2-bit g = 32 / 64), 4 / 8 (4-bit)
        const bool narrow = IC <= 128;                            // two rows side by side in a wave
        bool done = true;
        const bool one = IC <= 256;                               // 64 lanes x 4 ic: one pass
        if (bit == 2 && rpg == 2) { if (split) KIVI_WIDE(2, 64, 2, 1, true); else if (narrow) KIVI_WIDE(2, 32, 1, 4, false); else if (one) KIVI_WIDE(2, 64, 2, 2, false); else KIVI_WIDE(2, 64, 2, 1, false); }
        else if (bit == 2 && rpg == 4) { if (split) KIVI_WIDE(2, 64, 4, 1, true); else if (narrow) KIVI_WIDE(2, 32, 2, 2, false); else KIVI_WIDE(2, 64, 4, 1, false); }
        else if (bit == 4 && rpg == 4) { if (split) KIVI_WIDE(4, 64, 4, 1, true); else if (narrow) KIVI_WIDE(4, 32, 2, 4, false); else KIVI_WIDE(4, 64, 4, 1, false); }
        else if (bit == 4 && rpg == 8) { if (split) KIVI_WIDE(4, 64, 8, 1, true); else if (narrow) KIVI_WIDE(4, 32, 4, 1, false); else KIVI_WIDE(4, 64, 8, 1, false); }
        else done = false;
#undef KIVI_WIDE
        if (done) return kivi_launch_status("gemv_outer_dim_wide");
    }
    if (bit == 2)
        hipLaunchKernelGGL(gemv_outer_dim_kernel<2>, grid, dim3(256), 0, s, (const uint16_t*)in, (const uint32_t*)kernel,
                           (const uint16_t*)scale, (const uint16_t*)zeros, (uint16_t*)out, IC, OC, group_size,
                           nh / nh_kv, nrb);
    else
        hipLaunchKernelGGL(gemv_outer_dim_kernel<4>, grid, dim3(256), 0, s, (const uint16_t*)in, (const uint32_t*)kernel,
                           (const uint16_t*)scale, (const uint16_t*)zeros, (uint16_t*)out, IC, OC, group_size,
                           nh / nh_kv, nrb);
    return kivi_launch_status("gemv_outer_dim");
}

/*
 * kivi_oracle.c -- CPU restatement of the KIVI quant/ hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kivi_amd/ may import, link or call
 * this file; it is the checker for tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  It is a scalar, single-threaded restatement
 * of the reference algorithm with every fp16 rounding made explicit (software
 * binary16 <-> binary32 conversion, round-to-nearest-even), so it runs on any
 * host and does not depend on torch.
 *
 * Parity status: the pack / unpack / dequant functions are PINNED against the
 * reference's own pure-PyTorch functions (quant/new_pack.py imported from
 * /root/reference in the build container, see oracle/pin_reference.py and the
 * fixtures under tests/golden/).  The fused GEMV has no runnable reference here
 * (CUDA only); it follows quant/csrc/gemv_cuda.cu line by line, including the
 * reference's lane partition and shuffle-tree summation order, and its
 * indexing is pinned through exact-arithmetic cases against the reference's
 * unpack_and_dequant_* + torch.matmul (same script).
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ fp16 */

static inline float h2f(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* binary32 -> binary16, round to nearest even (what torch's c10::Half and
 * v_cvt_f16_f32 / __float2half_rn do). */
static inline uint16_t f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) { /* inf / nan */
        if (ax > 0x7F800000u) return (uint16_t)(sign | 0x7E00u | ((ax >> 13) & 0x3FFu));
        return (uint16_t)(sign | 0x7C00u);
    }
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u); /* >= 65520 -> inf */
    if (ax < 0x33000001u) return (uint16_t)sign;              /* <= 2^-25 -> 0 (tie to even) */
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7FFFFFu) | 0x800000u; /* 24-bit significand */
    int shift;                                /* bits to drop */
    uint32_t base;
    if (e < -14) { /* subnormal result */
        shift = 13 + (-14 - e);
        base = 0;
    } else {
        shift = 13;
        base = (uint32_t)(e + 15) << 10;
        m &= 0x7FFFFFu;
    }
    uint32_t keep = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    uint32_t r = base + keep;
    if (rem > half || (rem == half && (keep & 1u))) r++; /* may carry into exponent: correct */
    return (uint16_t)(sign | r);
}

/* Round-trip helpers exported for the tests (pin the conversion itself
 * against numpy/torch over all 65536 halves and a float sweep). */
float kivi_oracle_h2f(uint16_t h) { return h2f(h); }
uint16_t kivi_oracle_f2h(float f) { return f2h(f); }

/* fp16 "less than" used for the group minimum.  -0 orders below +0, which is
 * what the device min instruction (and the reference's CUDA/Triton tl.min)
 * does; torch's CPU reduction keeps the first of the two.  Only the sign bit
 * of a zero `mn` can differ; see DESIGN.md. */
static inline int h_lt(uint16_t a, uint16_t b) {
    float fa = h2f(a), fb = h2f(b);
    if (fa < fb) return 1;
    if (fa == fb && fa == 0.0f) return (a & 0x8000u) && !(b & 0x8000u);
    return 0;
}

/* ------------------------------------------------------ quantise one group */

#define KIVI_NAN_CUDA 0 /* float->int of NaN gives 0 (reference CUDA path) */
#define KIVI_NAN_CPU 1  /* x86 cvttss2si gives INT_MIN (reference run on CPU) */

/* new_pack.py:36-44 (== :16-24, :236-241).  `x` points at the first element
 * of the group, elements are `stride` halves apart.  Writes g int32 codes. */
static void quant_group(const uint16_t* x, int64_t stride, int g, int bits, int nan_mode,
                        int32_t* codes, uint16_t* scale_out, uint16_t* mn_out) {
    const int maxq = (1 << bits) - 1;
    uint16_t mn = x[0], mx = x[0];
    for (int i = 1; i < g; i++) {
        uint16_t v = x[i * stride];
        if (h_lt(v, mn)) mn = v;
        if (h_lt(mx, v)) mx = v;
    }
    /* scale = (mx - mn) / max_int : two fp16 roundings (:42) */
    uint16_t range = f2h(h2f(mx) - h2f(mn));
    uint16_t scale = f2h(h2f(range) / (float)maxq);
    float fs = h2f(scale), fmn = h2f(mn);
    for (int i = 0; i < g; i++) {
        uint16_t d = f2h(h2f(x[i * stride]) - fmn); /* data - mn   (:43) */
        uint16_t q = f2h(h2f(d) / fs);              /* data.div_(scale) (:44) */
        float fq = h2f(q);
        int32_t c;
        if (fq != fq) { /* NaN survives clamp_ and round_ (:45) */
            c = (nan_mode == KIVI_NAN_CPU) ? INT32_MIN : 0;
        } else {
            if (fq < 0.0f) fq = 0.0f;
            if (fq > (float)maxq) fq = (float)maxq;
            c = (int32_t)nearbyintf(fq); /* round_: half to even */
        }
        codes[i] = c;
    }
    *scale_out = scale;
    *mn_out = mn;
}

/* code << (bits*i) with torch's int32 wrap-around semantics (pack_tensor,
 * new_pack.py:100-105). */
static inline uint32_t shl_wrap(int32_t c, int sh) { return (uint32_t)c << sh; }

/*
 * triton_quantize_and_pack_along_last_dim (new_pack.py:217-252) ==
 * quant_and_pack_vcache (new_pack.py:30-48) with scale/mn squeezed.
 *   x      (rows, T) fp16 bits, contiguous
 *   code   (rows, T/fpi) int32,  scale/mn (rows, T/g) fp16 bits
 */
int kivi_oracle_quant_pack_lastdim(const uint16_t* x, int64_t rows, int64_t T, int g, int bits,
                                   int nan_mode, int32_t* code, uint16_t* scale, uint16_t* mn) {
    if (!(bits == 2 || bits == 4 || bits == 8)) return -1; /* :90 */
    const int fpi = 32 / bits;
    if (T % g || T % fpi) return -2; /* :222, :91 */
    const int64_t ng = T / g, nw = T / fpi;
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)T);
    if (!tmp) return -3;
    for (int64_t r = 0; r < rows; r++) {
        for (int64_t G = 0; G < ng; G++)
            quant_group(x + r * T + G * g, 1, g, bits, nan_mode, tmp + G * g, &scale[r * ng + G],
                        &mn[r * ng + G]);
        for (int64_t w = 0; w < nw; w++) {
            uint32_t word = 0;
            for (int i = 0; i < fpi; i++) word |= shl_wrap(tmp[w * fpi + i], bits * i);
            code[r * nw + w] = (int32_t)word;
        }
    }
    free(tmp);
    return 0;
}

/*
 * quant_and_pack_kcache (new_pack.py:8-27): groups and packing run along
 * dim 2 (T) of an un-transposed (BH, T, D) tensor.
 *   code (BH, T/fpi, D) int32, scale/mn (BH, T/g, D) fp16 bits
 */
int kivi_oracle_quant_pack_kcache(const uint16_t* k, int64_t BH, int64_t T, int64_t D, int g, int bits,
                                  int nan_mode, int32_t* code, uint16_t* scale, uint16_t* mn) {
    if (!(bits == 2 || bits == 4 || bits == 8)) return -1;
    const int fpi = 32 / bits;
    if (T % g || T % fpi) return -2;
    const int64_t ng = T / g, nw = T / fpi;
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)T);
    if (!tmp) return -3;
    for (int64_t bh = 0; bh < BH; bh++)
        for (int64_t d = 0; d < D; d++) {
            const uint16_t* col = k + bh * T * D + d;
            for (int64_t G = 0; G < ng; G++)
                quant_group(col + G * g * D, D, g, bits, nan_mode, tmp + G * g,
                            &scale[(bh * ng + G) * D + d], &mn[(bh * ng + G) * D + d]);
            for (int64_t w = 0; w < nw; w++) {
                uint32_t word = 0;
                for (int i = 0; i < fpi; i++) word |= shl_wrap(tmp[w * fpi + i], bits * i);
                code[(bh * nw + w) * D + d] = (int32_t)word;
            }
        }
    free(tmp);
    return 0;
}

/*
 * pack_tensor (new_pack.py:86-107) on a 3-D view (outer, n, inner): packs
 * along the middle axis.  pack_dim=3 of a 4-D tensor is inner=1; pack_dim=2
 * is inner=shape[3].
 */
int kivi_oracle_pack_tensor(const int32_t* data, int64_t outer, int64_t n, int64_t inner, int bits,
                            int32_t* code) {
    if (!(bits == 2 || bits == 4 || bits == 8)) return -1;
    const int fpi = 32 / bits;
    if (n % fpi) return -2;
    const int64_t nw = n / fpi;
    for (int64_t o = 0; o < outer; o++)
        for (int64_t w = 0; w < nw; w++)
            for (int64_t in = 0; in < inner; in++) {
                uint32_t word = 0;
                for (int i = 0; i < fpi; i++)
                    word |= shl_wrap(data[(o * n + w * fpi + i) * inner + in], bits * i);
                code[(o * nw + w) * inner + in] = (int32_t)word;
            }
    return 0;
}

/* unpack_tensor (new_pack.py:110-129): arithmetic >> on int32, then & mask,
 * result int16. */
int kivi_oracle_unpack_tensor(const int32_t* code, int64_t outer, int64_t nw, int64_t inner, int bits,
                              int16_t* out) {
    if (!(bits == 2 || bits == 4 || bits == 8)) return -1;
    const int fpi = 32 / bits;
    const int mask = 0xFF >> (8 - bits); /* :120 */
    for (int64_t o = 0; o < outer; o++)
        for (int64_t w = 0; w < nw; w++)
            for (int64_t in = 0; in < inner; in++) {
                int32_t word = code[(o * nw + w) * inner + in];
                for (int i = 0; i < fpi; i++)
                    out[(o * nw * fpi + w * fpi + i) * inner + in] = (int16_t)((word >> (bits * i)) & mask);
            }
    return 0;
}

/*
 * unpack_and_dequant_vcache (new_pack.py:69-83): x = fp16(fp16(fp16(q)*scale)+mn),
 * groups along the last (packed) dim.  code (rows, T/fpi), scale/mn (rows, T/g).
 */
int kivi_oracle_unpack_dequant_lastdim(const int32_t* code, const uint16_t* scale, const uint16_t* mn,
                                       int64_t rows, int64_t T, int g, int bits, uint16_t* out) {
    if (!(bits == 2 || bits == 4 || bits == 8)) return -1;
    const int fpi = 32 / bits;
    const int mask = 0xFF >> (8 - bits);
    const int64_t ng = T / g, nw = T / fpi;
    for (int64_t r = 0; r < rows; r++)
        for (int64_t t = 0; t < T; t++) {
            int32_t word = code[r * nw + t / fpi];
            int q = (word >> (bits * (int)(t % fpi))) & mask;
            uint16_t qh = f2h((float)q);
            uint16_t p = f2h(h2f(qh) * h2f(scale[r * ng + t / g]));
            out[r * T + t] = f2h(h2f(p) + h2f(mn[r * ng + t / g]));
        }
    return 0;
}

/*
 * unpack_and_dequant_kcache (new_pack.py:51-66): code (BH, T/fpi, D),
 * scale/mn (BH, T/g, D) -> (BH, T, D).
 */
int kivi_oracle_unpack_dequant_kcache(const int32_t* code, const uint16_t* scale, const uint16_t* mn,
                                      int64_t BH, int64_t T, int64_t D, int g, int bits, uint16_t* out) {
    if (!(bits == 2 || bits == 4 || bits == 8)) return -1;
    const int fpi = 32 / bits;
    const int mask = 0xFF >> (8 - bits);
    const int64_t ng = T / g, nw = T / fpi;
    for (int64_t bh = 0; bh < BH; bh++)
        for (int64_t t = 0; t < T; t++)
            for (int64_t d = 0; d < D; d++) {
                int32_t word = code[(bh * nw + t / fpi) * D + d];
                int q = (word >> (bits * (int)(t % fpi))) & mask;
                uint16_t p = f2h(h2f(f2h((float)q)) * h2f(scale[(bh * ng + t / g) * D + d]));
                out[(bh * T + t) * D + d] = f2h(h2f(p) + h2f(mn[(bh * ng + t / g) * D + d]));
            }
    return 0;
}

/* ------------------------------------------------------------ fused GEMV */

/*
 * Core of bgemv2_kernel_outer_dim / bgemv4_kernel_outer_dim
 * (gemv_cuda.cu:348-427 / :265-345) for ONE (batch_idx, packed row):
 * 32 lanes, TILE_DIM 128, lane l owns ic = 128k + 4l .. 4l+3 (:370-387),
 * per-lane sequential fp32 accumulation in (k, ic_0, ic_1) order (:401-417),
 * then the 16,8,4,2,1 shuffle-down tree (:25-39, :419-426) and one RN
 * conversion to fp16.  nvcc contracts a*b+c into fma by default (and the
 * reference builds with --use_fast_math, setup.py:26) so `use_fma`=1 is the
 * faithful mode; use_fma=0 is kept to measure the sensitivity.
 *
 * Addressing is by strides so the same routine serves the reference's
 * kernel-input layout and the hook-state layouts:
 *   word(ocw, ic)  = w[ocw * w_socw + ic * w_sic]
 *   scale(G, ic)   = s[G * s_sg + ic * s_sic]   (zeros likewise)
 */
static void gemv_row(const uint16_t* in, const uint32_t* w, int64_t w_socw, int64_t w_sic,
                     const uint16_t* s, const uint16_t* z, int64_t s_sg, int64_t s_sic, int64_t IC,
                     int64_t OC, int64_t ocw, int g, int bits, int use_fma, uint16_t* out) {
    const int pf = 32 / bits;
    const uint32_t mask = 0xFFu >> (8 - bits);
    const int64_t oc0 = ocw * pf;
    const int64_t G = oc0 / g; /* :357 */
    float psum[32][16];
    memset(psum, 0, sizeof psum);
    const int64_t ntile = (IC + 127) / 128;
    for (int64_t k = 0; k < ntile; k++)
        for (int lane = 0; lane < 32; lane++)
            for (int i0 = 0; i0 < 4; i0++) {
                int64_t ic = k * 128 + lane * 4 + i0;
                if (ic >= IC) continue; /* zero-initialised qw/inp (:371-386) contribute +0 */
                uint32_t word = w[ocw * w_socw + ic * w_sic];
                float cin = h2f(in[ic]);
                float cs = h2f(s[G * s_sg + ic * s_sic]);
                float cz = h2f(z[G * s_sg + ic * s_sic]);
                for (int i1 = 0; i1 < pf; i1++) {
                    if (oc0 + i1 >= OC) break; /* :409 */
                    float wf = (float)(word & mask);
                    word >>= bits;
                    if (use_fma) {
                        float dq = fmaf(cs, wf, cz);
                        psum[lane][i1] = fmaf(dq, cin, psum[lane][i1]);
                    } else {
                        float dq = cs * wf + cz;
                        psum[lane][i1] = psum[lane][i1] + dq * cin;
                    }
                }
            }
    for (int i1 = 0; i1 < pf; i1++) {
        if (oc0 + i1 >= OC) break;
        float v[32];
        for (int l = 0; l < 32; l++) v[l] = psum[l][i1];
        for (int off = 16; off >= 1; off >>= 1) /* shfl_down tree; lane 0 is what is stored */
            for (int l = 0; l + off < 32; l++) v[l] += v[l + off];
        out[oc0 + i1] = f2h(v[0]);
    }
}

/*
 * gemv_forward_cuda_outer_dim (gemv_cuda.cu:511-557), reference kernel-input
 * layout: in (BS, 1, IC), kernel (BS_kv, OC/pf, IC) u32, scale/zeros
 * (BS_kv, OC/g, IC), out (BS, 1, OC).  GQA: _batch_idx = batch_idx / (nh/nh_kv)
 * (:361-365).
 */
int kivi_oracle_gemv_outer_dim(const uint16_t* in, const int32_t* kernel, const uint16_t* scale,
                               const uint16_t* zeros, uint16_t* out, int64_t BS, int64_t IC, int64_t OC,
                               int bits, int g, int nh, int nh_kv, int use_fma) {
    if (!(bits == 2 || bits == 4)) return -1; /* matmul.py:215 */
    if (nh_kv <= 0 || nh % nh_kv) return -2;  /* matmul.py:216 */
    const int pf = 32 / bits;
    const int ratio = nh / nh_kv;
    const int64_t nrow = (OC + pf - 1) / pf;
    for (int64_t b = 0; b < BS; b++) {
        const int64_t bk = b / ratio;
        for (int64_t r = 0; r < nrow; r++)
            gemv_row(in + b * IC, (const uint32_t*)kernel + bk * nrow * IC, IC, 1, scale + bk * (OC / g) * IC,
                     zeros + bk * (OC / g) * IC, IC, 1, IC, OC, r, g, bits, use_fma, out + b * OC);
    }
    return 0;
}

/*
 * The same arithmetic on the hook-state layouts (llama_kivi.py:454-455) that
 * cuda_bmm_fA_qB_outer (matmul.py:178-219) receives before it transposes:
 *   fA (BS, 1, K) with row stride fa_stride (halves),
 *   qB (BS_kv, K, N/pf) int32, scales/zeros (BS_kv, K, N/g), out (BS, 1, N).
 * K is the reduction length (IC), N the output length (OC).  For qK^T:
 * K = D, N = Tq.  For sV: K = Tv, N = D.
 */
int kivi_oracle_bmm_fA_qB_outer(const uint16_t* fA, int64_t fa_stride, const int32_t* qB,
                                const uint16_t* scales, const uint16_t* zeros, uint16_t* out, int64_t BS,
                                int64_t K, int64_t N, int bits, int g, int nh, int nh_kv, int use_fma) {
    if (!(bits == 2 || bits == 4)) return -1;
    if (nh_kv <= 0 || nh % nh_kv) return -2;
    const int pf = 32 / bits;
    const int ratio = nh / nh_kv;
    const int64_t nw = N / pf, ng = N / g;
    for (int64_t b = 0; b < BS; b++) {
        const int64_t bk = b / ratio;
        for (int64_t r = 0; r < nw; r++)
            gemv_row(fA + b * fa_stride, (const uint32_t*)qB + bk * K * nw, 1, nw, scales + bk * K * ng,
                     zeros + bk * K * ng, 1, ng, K, N, r, g, bits, use_fma, out + b * N);
    }
    return 0;
}

/*
 * Fake-quant GEMV of the reference's test procedure (quant/gemv.py:70-74,
 * :118-126; quant/test.py:187-195): dequantise to fp16 with two roundings,
 * then a matmul with fp32 accumulation in plain ascending-k order and one
 * fp16 rounding.  This is the CPU timing baseline, NOT the parity oracle
 * (the two reference paths differ by > 1e-3).
 */
int kivi_oracle_fakequant_bmm(const uint16_t* fA, int64_t fa_stride, const int32_t* qB,
                              const uint16_t* scales, const uint16_t* zeros, uint16_t* out, int64_t BS,
                              int64_t K, int64_t N, int bits, int g, int nh, int nh_kv) {
    if (!(bits == 2 || bits == 4 || bits == 8)) return -1;
    if (nh_kv <= 0 || nh % nh_kv) return -2;
    const int pf = 32 / bits;
    const int mask = 0xFF >> (8 - bits);
    const int ratio = nh / nh_kv;
    const int64_t nw = N / pf, ng = N / g;
    float* acc = (float*)malloc(sizeof(float) * (size_t)N);
    if (!acc) return -3;
    for (int64_t b = 0; b < BS; b++) {
        const int64_t bk = b / ratio;
        for (int64_t n = 0; n < N; n++) acc[n] = 0.0f;
        for (int64_t k = 0; k < K; k++) {
            float a = h2f(fA[b * fa_stride + k]);
            const int32_t* wrow = qB + (bk * K + k) * nw;
            const uint16_t* srow = scales + (bk * K + k) * ng;
            const uint16_t* zrow = zeros + (bk * K + k) * ng;
            for (int64_t n = 0; n < N; n++) {
                int q = (wrow[n / pf] >> (bits * (int)(n % pf))) & mask;
                uint16_t p = f2h(h2f(f2h((float)q)) * h2f(srow[n / g]));
                uint16_t wv = f2h(h2f(p) + h2f(zrow[n / g]));
                acc[n] += a * h2f(wv);
            }
        }
        for (int64_t n = 0; n < N; n++) out[b * N + n] = f2h(acc[n]);
    }
    free(acc);
    return 0;
}

/*
 * gemv_forward_cuda -> gemv_kernel_g64 / gemv_kernel_g128 (gemv_cuda.cu:60-246): legacy AWQ-style inner-dim
 * 4-bit GEMV.  32 lanes, each pass covers 1024 input channels: lane l owns words 4l..4l+3 (32 codes) of the pass
 * (:81, :153), scale / zero index = pass * (1024/g) + l / (g/32) (:84-85, :156-157), fp32 fma accumulation in
 * (pass, ic_0, ic_1) order, shuffle-down tree, one RN rounding.  Scale rows use the caller's pitch (the reference
 * hard-codes a padded pitch that equals IC/g whenever IC is a multiple of 1024).
 */
int kivi_oracle_gemv_awq(const uint16_t* in, const int32_t* kernel, const uint16_t* scale, const uint16_t* zeros,
                         uint16_t* out, int64_t B, int64_t IC, int64_t OC, int g, int64_t sz_pitch, int use_fma) {
    if (!(g == 64 || g == 128) || IC % g) return -1;
    const int64_t ww = IC / 8;
    const int64_t npass = (IC + 1023) / 1024;
    const int lanes_per_group = g / 32;
    for (int64_t b = 0; b < B; b++)
        for (int64_t oc = 0; oc < OC; oc++) {
            float psum[32];
            for (int l = 0; l < 32; l++) psum[l] = 0.0f;
            for (int64_t p = 0; p < npass; p++)
                for (int l = 0; l < 32; l++) {
                    const int64_t gi = p * (1024 / g) + l / lanes_per_group;
                    for (int i0 = 0; i0 < 4; i0++) {
                        const int64_t w = p * 128 + l * 4 + i0;
                        if (w >= ww) continue; /* inputs guard (:93, :165) */
                        uint32_t word = (uint32_t)kernel[oc * ww + w];
                        const float cs = h2f(scale[oc * sz_pitch + gi]), cz = h2f(zeros[oc * sz_pitch + gi]);
                        for (int i1 = 0; i1 < 8; i1++) {
                            const float wf = (float)(word & 0xFu);
                            const float x = h2f(in[b * IC + w * 8 + i1]);
                            if (use_fma) psum[l] = fmaf(fmaf(cs, wf, cz), x, psum[l]);
                            else psum[l] = psum[l] + (cs * wf + cz) * x;
                            word >>= 4;
                        }
                    }
                }
            for (int off = 16; off >= 1; off >>= 1)
                for (int l = 0; l + off < 32; l++) psum[l] += psum[l + off];
            out[b * OC + oc] = f2h(psum[0]);
        }
    return 0;
}

/*
 * CPU oracle (test infrastructure only, see orc_common.h) for the EXL3 (ExLlamaV3 trellis) path.
 *
 * Follows the reference's vendored kernels:
 *   crates/core/kernels/exl3/exl3_dq.cuh:5-31      16-bit window of weight t in the tile's MSB-first, cyclic bit stream
 *   crates/core/kernels/exl3/codebook.cuh:25-56    three "3-instruction" codebooks (default / mcg / mul1), fp16 arithmetic
 *   crates/core/kernels/exl3/reconstruct.cu:36-72  position of weight t inside the 16x16 tile
 *   crates/core/kernels/exl3/hadamard_inner.cuh:14-41,100-144,213-276  128-point Hadamard, rounding points
 *   crates/core/src/quantization/exl3_cuda.rs:271-416,1581-1591         y = had(had(x*suh) @ W) * svh
 * The reference holds no ExLlamaV3-generated golden vectors: only the zero-trellis constant
 * (exl3_cuda.rs:1397-1432) and self-consistency tests (gemm == had . matmul . reconstruct within 10 %,
 * :1542-1633), which tests/test_oracle_exl3.py restates.  Parity beyond those is "unpinned".
 */
#include "orc_common.h"

/* fp16 add / mul / fma with one rounding (operands and result binary16) */
static inline uint16_t h_add(uint16_t a, uint16_t b) { return orc_f32_to_f16(orc_f16_to_f32(a) + orc_f16_to_f32(b)); }
static inline uint16_t h_mul(uint16_t a, uint16_t b) { return orc_f32_to_f16(orc_f16_to_f32(a) * orc_f16_to_f32(b)); }
static inline uint16_t h_fma(uint16_t a, uint16_t b, uint16_t c) {
    /* a*b is exact in double (11+11 bits), the sum too (exponent range of fp16): one rounding */
    double r = (double)orc_f16_to_f32(a) * (double)orc_f16_to_f32(b) + (double)orc_f16_to_f32(c);
    /* double -> half with a single rounding: go through float only if exact there; products of halves
       plus a half need at most 22+ bits, exact in float's 24 when exponents are close; use long path */
    float f = (float)r;
    if ((double)f == r) return orc_f32_to_f16(f);
    /* inexact in float: nudge towards r so that the final half rounding sees the right side (sticky) */
    uint32_t u;
    memcpy(&u, &f, 4);
    if (((double)f < r) == (f >= 0)) u |= 1u; else { u -= 1u; u |= 1u; }
    memcpy(&f, &u, 4);
    return orc_f32_to_f16(f);
}

/* codebook.cuh:25-56.  w = 16-bit window. */
uint16_t orc_exl3_decode(uint32_t w, int cb) {
    uint32_t x = w;
    if (cb == 0 || cb == 1) {
        if (cb == 0) { x *= 89226354u; x += 64248484u; }
        else x *= 0xCBAC1FEDu;
        x = (x & 0x8fff8fffu) ^ 0x3b603b60u;
        return h_add((uint16_t)(x & 0xffff), (uint16_t)(x >> 16));
    }
    x *= 0x83DCD12Du;
    uint32_t sum = (x & 0xff) + ((x >> 8) & 0xff) + ((x >> 16) & 0xff) + (x >> 24) + 0x6400u;
    return h_fma((uint16_t)sum, 0x1eee, 0xc931);
}

/* 16-bit window of weight t (0..255) in a tile of 8*bits u32 words: stream bit i is bit (31 - i%32) of word
 * i/32; the window ends at stream bit (t+257)*bits (exclusive), cyclically. */
static uint32_t window(const uint32_t *words, int bits, int t) {
    const int L = 256 * bits;
    int b0 = ((t + 257) * bits - 16) % L;
    uint32_t w = 0;
    for (int j = 0; j < 16; j++) {
        int i = (b0 + j) % L;
        w = (w << 1) | ((words[i / 32] >> (31 - i % 32)) & 1u);
    }
    return w;
}

/* trellis: u16 [K/16][N/16][16*bits] (pairs of u16 = little-endian u32 words) -> dense fp16 [K][N]. */
int orc_exl3_reconstruct(const uint16_t *trellis, int64_t K, int64_t N, int bits, int cb, uint16_t *dense) {
    if (K % 16 || N % 16 || bits < 1 || bits > 8) return -1;
    const int64_t tk = K / 16, tn = N / 16;
    const int tile_u16 = 16 * bits;
    for (int64_t kt = 0; kt < tk; kt++)
        for (int64_t nt = 0; nt < tn; nt++) {
            const uint16_t *p = trellis + (kt * tn + nt) * tile_u16;
            uint32_t words[64];
            for (int i = 0; i < 8 * bits; i++) words[i] = (uint32_t)p[2 * i] | ((uint32_t)p[2 * i + 1] << 16);
            for (int t = 0; t < 256; t++) {
                const int lane = t >> 3, i = t & 7;
                const int c = lane >> 3, h = (lane >> 2) & 1, q = lane & 3;
                const int row = 2 * q + (i & 1) + 8 * ((i >> 1) & 1), col = 2 * c + h + 8 * (i >> 2);
                dense[(kt * 16 + row) * N + nt * 16 + col] = orc_exl3_decode(window(words, bits, t), cb);
            }
        }
    return 0;
}

/* One 128-point block: f32 butterflies in stride order 1,2,4,...,64 with v' = (bit ? -v : v) + partner. */
static void had128_f32(float v[128]) {
    for (int s = 1; s < 128; s <<= 1) {
        float o[128];
        for (int i = 0; i < 128; i++) o[i] = ((i & s) ? -v[i] : v[i]) + v[i ^ s];
        memcpy(v, o, sizeof(o));
    }
}

/* had_hf_r_128 (hadamard_inner.cuh:100-144): fp16 in/out; mode 0 none, 1 pre-scale (fp16 mul before),
 * 2 post-scale (fp16 mul after the rounding).  x [rows][cols], cols % 128 == 0, scale [cols]. */
int orc_exl3_had_fp16(const uint16_t *x, const uint16_t *scale, int64_t rows, int64_t cols, int mode, uint16_t *y) {
    if (cols % 128) return -1;
    const float r = 0.088388347648f;
    for (int64_t rr = 0; rr < rows; rr++)
        for (int64_t b = 0; b < cols; b += 128) {
            float v[128];
            for (int i = 0; i < 128; i++) {
                uint16_t e = x[rr * cols + b + i];
                if (mode == 1) e = h_mul(e, scale[b + i]);
                v[i] = orc_f16_to_f32(e);
            }
            had128_f32(v);
            for (int i = 0; i < 128; i++) {
                uint16_t o = orc_f32_to_f16(v[i] * r);
                if (mode == 2) o = h_mul(o, scale[b + i]);
                y[rr * cols + b + i] = o;
            }
        }
    return 0;
}

/* had_fh_r_128<false,true> (hadamard_inner.cuh:213-276): f32 in, fp16 out, optional post-scale. */
int orc_exl3_had_f32_to_fp16(const float *x, const uint16_t *post_scale, int64_t rows, int64_t cols, uint16_t *y) {
    if (cols % 128) return -1;
    const float r = 0.088388347648f;
    for (int64_t rr = 0; rr < rows; rr++)
        for (int64_t b = 0; b < cols; b += 128) {
            float v[128];
            memcpy(v, x + rr * cols + b, sizeof(v));
            had128_f32(v);
            for (int i = 0; i < 128; i++) {
                uint16_t o = orc_f32_to_f16(v[i] * r);
                if (post_scale) o = h_mul(o, post_scale[b + i]);
                y[rr * cols + b + i] = o;
            }
        }
    return 0;
}

/* Deterministic restatement of exl3_gemm (exl3_cuda.rs:271-416): A_had = had_pre(x, suh) in fp16; C = A_had @ W with
 * exact fp16 products accumulated in f32 (sequential over k); y = had_post(C) in f32 -> fp16 -> * svh. */
int orc_exl3_gemm(const uint16_t *x, const uint16_t *trellis, const uint16_t *suh, const uint16_t *svh, int64_t M, int64_t K,
                  int64_t N, int bits, int cb, uint16_t *y, uint16_t *w_scratch, uint16_t *a_scratch, float *c_scratch) {
    if (K % 128 || N % 128) return -1;
    if (orc_exl3_reconstruct(trellis, K, N, bits, cb, w_scratch)) return -1;
    if (suh) orc_exl3_had_fp16(x, suh, M, K, 1, a_scratch);
    else orc_exl3_had_fp16(x, NULL, M, K, 0, a_scratch);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++)
        for (int64_t m = 0; m < M; m++) {
            float acc = 0.0f;
            for (int64_t k = 0; k < K; k++) acc += orc_f16_to_f32(a_scratch[m * K + k]) * orc_f16_to_f32(w_scratch[k * N + n]);
            c_scratch[m * N + n] = acc;
        }
    return orc_exl3_had_f32_to_fp16(c_scratch, svh, M, N, y);
}

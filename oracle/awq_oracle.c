/*
 * CPU oracle (test infrastructure only, see orc_common.h) for the packed int4
 * paths: AWQ load-time repacks, AWQ/GPTQ dequantisation and the int4 GEMV.
 *
 * Follows:
 *   crates/core/src/quantization/awq_marlin.rs:126-151  repack_awq_nibbles
 *   crates/core/src/quantization/awq_marlin.rs:170-240  awq_to_gptq_qweight
 *   crates/core/src/quantization/awq_marlin.rs:372-391  undo-pack shifts / awq_to_gptq_u32
 *   crates/core/src/quantization/awq_marlin.rs:635      transpose to [N, K/8]
 *   crates/core/kernels/awq_gemv.cu:70-117              GEMV arithmetic (f32 acc, bf16 out)
 *   crates/core/src/quantization/marlin_cuda.rs:2091-2112  scalar CPU reference of the GEMV test
 *   crates/core/kernels/marlin_gemm.cu:43-48            GPTQ symmetric (nib - 8) * scale
 * Pinned by the reference's known-answer tests (awq_marlin.rs:727-738,761-799)
 * restated in tests/test_oracle_awq.py.
 */
#include "orc_common.h"

/* AWQ word: output lane i sits at nibble position undo_pack[i] = {0,4,1,5,2,6,3,7}. */
static const uint32_t AWQ_SHIFT[8] = {0, 16, 4, 20, 8, 24, 12, 28};

uint32_t orc_awq_to_gptq_u32(uint32_t w) {
    uint32_t out = 0;
    for (int i = 0; i < 8; i++) out |= ((w >> AWQ_SHIFT[i]) & 0xFu) << (4 * i);
    return out;
}

/* Same-shape nibble reorder (used for qzeros [G, N/8]). */
void orc_repack_awq_nibbles(const uint32_t *in, int64_t n_words, uint32_t *out) {
    for (int64_t i = 0; i < n_words; i++) out[i] = orc_awq_to_gptq_u32(in[i]);
}

/* AWQ qweight [K, N/8] (packed along N, interleaved) -> GPTQ-sequential [K/8, N]
 * (packed along K, nibble i of a word = input row 8*kb + i). */
int orc_awq_to_gptq_qweight(const uint32_t *awq, int64_t K, int64_t N, uint32_t *out) {
    if (K % 8 || N % 8) return -1;
    int64_t pn = N / 8, pk = K / 8;
    for (int64_t n = 0; n < N; n++) {
        uint32_t sh = AWQ_SHIFT[n % 8];
        for (int64_t kb = 0; kb < pk; kb++) {
            uint32_t word = 0;
            for (int i = 0; i < 8; i++) {
                uint32_t src = awq[(kb * 8 + i) * pn + n / 8];
                word |= ((src >> sh) & 0xFu) << (4 * i);
            }
            out[kb * N + n] = word;
        }
    }
    return 0;
}

/* [R, C] -> [C, R] */
void orc_transpose_u32(const uint32_t *in, int64_t R, int64_t C, uint32_t *out) {
    for (int64_t r = 0; r < R; r++)
        for (int64_t c = 0; c < C; c++) out[c * R + r] = in[r * C + c];
}

/* Dense f32 weight [K, N] from GPTQ-sequential qweight [K/8, N], scales (f32
 * values, already bf16-rounded by the caller) [G, N], qzeros [G, N/8] (sequential
 * nibbles) or NULL -> zero point 8 (awq_gemv.cu:85; marlin_gemm.cu:43-48). */
int orc_int4_dequant(const uint32_t *qweight, const float *scales, const uint32_t *qzeros, int64_t K,
                     int64_t N, int64_t group, float *w) {
    if (K % 8 || N % 8 || group <= 0 || K % group) return -1;
    for (int64_t k = 0; k < K; k++) {
        int64_t g = k / group;
        for (int64_t n = 0; n < N; n++) {
            int nib = (int)((qweight[(k / 8) * N + n] >> (4 * (k % 8))) & 0xFu);
            int zp = 8;
            if (qzeros) zp = (int)((qzeros[g * (N / 8) + n / 8] >> (4 * (n % 8))) & 0xFu);
            w[k * N + n] = (float)(nib - zp) * scales[g * N + n];
        }
    }
    return 0;
}

/* The reference test's scalar GEMV (marlin_cuda.rs:2091-2112): f32 inputs,
 * acc += x * ((nib - zp) * scale) sequentially over k, f32 result. */
int orc_int4_gemv_f32(const float *x, const uint32_t *qweight, const float *scales,
                      const uint32_t *qzeros, int64_t M, int64_t K, int64_t N, int64_t group, float *y) {
    if (K % 8 || N % 8 || group <= 0 || K % group) return -1;
    for (int64_t m = 0; m < M; m++)
        for (int64_t n = 0; n < N; n++) {
            float acc = 0.0f;
            for (int64_t k = 0; k < K; k++) {
                int64_t g = k / group;
                int nib = (int)((qweight[(k / 8) * N + n] >> (4 * (k % 8))) & 0xFu);
                int zp = 8;
                if (qzeros) zp = (int)((qzeros[g * (N / 8) + n / 8] >> (4 * (n % 8))) & 0xFu);
                float wv = (float)(nib - zp) * scales[g * N + n];
                acc += x[m * K + k] * wv;
            }
            y[m * N + n] = acc;
        }
    return 0;
}

/* The kernel's contract (awq_gemv.cu:46-118): bf16 activations, bf16 scales,
 * optional bf16 bias, f32 accumulation, bf16 output (round-to-nearest-even).
 * qweight here is GPTQ-sequential [K/8, N]. */
int orc_int4_gemv_bf16(const uint16_t *x, const uint32_t *qweight, const uint16_t *scales,
                       const uint32_t *qzeros, const uint16_t *bias, int64_t M, int64_t K, int64_t N,
                       int64_t group, uint16_t *y) {
    if (K % 8 || N % 8 || group <= 0 || K % group) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        for (int64_t m = 0; m < M; m++) {
            float acc = 0.0f;
            for (int64_t kp = 0; kp < K / 8; kp++) {
                int64_t g = (kp * 8) / group;
                float sc = orc_bf16_to_f32(scales[g * N + n]);
                float zp = 8.0f;
                if (qzeros) zp = (float)((qzeros[g * (N / 8) + n / 8] >> (4 * (n % 8))) & 0xFu);
                uint32_t wd = qweight[kp * N + n];
                for (int i = 0; i < 8; i++) {
                    float nib = (float)((wd >> (4 * i)) & 0xFu);
                    float wv = (nib - zp) * sc;
                    acc += orc_bf16_to_f32(x[m * K + kp * 8 + i]) * wv;
                }
            }
            if (bias) acc += orc_bf16_to_f32(bias[n]);
            y[m * N + n] = orc_f32_to_bf16(acc);
        }
    }
    return 0;
}

/* bf16 <-> f32 array helpers exported for the Python side. */
void orc_f32_to_bf16_array(const float *in, int64_t n, uint16_t *out) {
    for (int64_t i = 0; i < n; i++) out[i] = orc_f32_to_bf16(in[i]);
}
void orc_bf16_to_f32_array(const uint16_t *in, int64_t n, float *out) {
    for (int64_t i = 0; i < n; i++) out[i] = orc_bf16_to_f32(in[i]);
}
void orc_f32_to_f16_array(const float *in, int64_t n, uint16_t *out) {
    for (int64_t i = 0; i < n; i++) out[i] = orc_f32_to_f16(in[i]);
}
void orc_f16_to_f32_array(const uint16_t *in, int64_t n, float *out) {
    for (int64_t i = 0; i < n; i++) out[i] = orc_f16_to_f32(in[i]);
}

/* GPTQ 8-bit dequantisation (crates/core/kernels/gptq_dequant.cu:128-170): qweight u32 [K/4][N] (k % 4 selects the byte, low
 * first), scales bf16 [groups][N], qzeros u32 [groups][ceil(N/4)] (n % 4 selects the byte); out bf16 [K][N] = bf16((q - z) * s)
 * with s read as bf16 -> f32.  group <= 0: a single group. */
int orc_gptq_dequant_int8(const uint32_t *qweight, const uint16_t *scales, const uint32_t *qzeros, int64_t K, int64_t N,
                          int64_t group, uint16_t *out) {
    if (K % 4) return -1;
    int64_t zpr = (N + 3) / 4;
    for (int64_t k = 0; k < K; k++) {
        int64_t g = group > 0 ? k / group : 0;
        for (int64_t n = 0; n < N; n++) {
            int q = (int)((qweight[(k / 4) * N + n] >> (8 * (k % 4))) & 0xFF);
            int z = (int)((qzeros[g * zpr + n / 4] >> (8 * (n % 4))) & 0xFF);
            uint32_t sb = (uint32_t)scales[g * N + n] << 16;
            float sf;
            memcpy(&sf, &sb, 4);
            float w = (float)(q - z) * sf;
            uint32_t u;
            memcpy(&u, &w, 4);
            u += 0x7FFFu + ((u >> 16) & 1u);   /* round to nearest even (finite values only here) */
            out[k * N + n] = (uint16_t)(u >> 16);
        }
    }
    return 0;
}

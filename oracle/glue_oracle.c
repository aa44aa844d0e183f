/*
 * CPU oracle (test infrastructure only, see orc_common.h) for the per-token
 * glue operators of the decode path: RMSNorm, RoPE, SiLU*mul, argmax, KV-cache
 * scatter, FP8 e4m3 codec and paged-attention decode.
 *
 * Follows:
 *   crates/core/kernels/layernorm.cu:59-129,232-270   rms_norm_bf16 / fused_add_rms_norm_bf16
 *   crates/core/kernels/rope.cu:18-114                 rotary_embedding_neox_bf16
 *   crates/core/kernels/activations.cu:33-35,148-164   silu, silu_and_mul_separate_bf16
 *   crates/core/kernels/sampling.cu:47-94              argmax_bf16
 *   crates/core/kernels/cache_ops.cu:28-62             reshape_and_cache_bf16
 *   crates/core/kernels/paged_attention.cu:558-954     paged_attention_v2 + reduce
 *   crates/core/src/kv_cache/quantization.rs:287-303,408-452  FP8 quantise / CPU codec
 * The reference holds no golden vectors for these operators (only shape and
 * self-consistency tests), so apart from the FP8 known answers
 * (kv_cache/quantization.rs:654-692) parity here is "unpinned": the oracle is a
 * line-by-line restatement of the kernels' arithmetic in f32.
 */
#include "orc_common.h"
#include <float.h>
#include <stdlib.h>

/* out = bf16(x * inv_rms * w), inv_rms = 1/sqrt(sum(x^2)/H + eps); gemma_plus_one
 * selects the (1 + w) weight convention of Gemma (layers/normalization.rs:17-205). */
void orc_rms_norm_bf16(const uint16_t *x, const uint16_t *w, float eps, int64_t tokens, int64_t H,
                       int gemma_plus_one, uint16_t *out) {
    for (int64_t t = 0; t < tokens; t++) {
        const uint16_t *xr = x + t * H;
        float var = 0.0f;
        for (int64_t i = 0; i < H; i++) {
            float v = orc_bf16_to_f32(xr[i]);
            var += v * v;
        }
        float inv = 1.0f / sqrtf(var / (float)H + eps);
        for (int64_t i = 0; i < H; i++) {
            float wv = w ? orc_bf16_to_f32(w[i]) : 1.0f;
            if (gemma_plus_one) wv = 1.0f + wv;
            out[t * H + i] = orc_f32_to_bf16(orc_bf16_to_f32(xr[i]) * inv * wv);
        }
    }
}

/* residual = bf16(residual + x); out = bf16(residual * inv_rms * w); the variance
 * is taken over the unrounded f32 sums (layernorm.cu:246-251). */
void orc_fused_add_rms_norm_bf16(const uint16_t *x, uint16_t *residual, const uint16_t *w, float eps,
                                 int64_t tokens, int64_t H, uint16_t *out) {
    for (int64_t t = 0; t < tokens; t++) {
        float var = 0.0f;
        for (int64_t i = 0; i < H; i++) {
            float r = orc_bf16_to_f32(residual[t * H + i]) + orc_bf16_to_f32(x[t * H + i]);
            residual[t * H + i] = orc_f32_to_bf16(r);
            var += r * r;
        }
        float inv = 1.0f / sqrtf(var / (float)H + eps);
        for (int64_t i = 0; i < H; i++) {
            float r = orc_bf16_to_f32(residual[t * H + i]);
            out[t * H + i] = orc_f32_to_bf16(r * inv * orc_bf16_to_f32(w[i]));
        }
    }
}

/* NeoX RoPE in place on [tokens, heads, head_size] bf16; cos_sin [max_pos, rot_dim]
 * f32 = cos[rot/2] then sin[rot/2]. */
void orc_rope_neox_bf16(const int32_t *positions, uint16_t *q, uint16_t *k, const float *cos_sin,
                        int64_t tokens, int rot_dim, int head_size, int n_heads, int n_kv_heads) {
    int half = rot_dim / 2;
    for (int64_t t = 0; t < tokens; t++) {
        const float *cs = cos_sin + (int64_t)positions[t] * rot_dim;
        for (int pass = 0; pass < 2; pass++) {
            uint16_t *base = pass == 0 ? q : k;
            int nh = pass == 0 ? n_heads : n_kv_heads;
            if (!base) continue;
            for (int h = 0; h < nh; h++) {
                uint16_t *a = base + (t * nh + h) * head_size;
                for (int i = 0; i < half; i++) {
                    float xv = orc_bf16_to_f32(a[i]), yv = orc_bf16_to_f32(a[half + i]);
                    float c = cs[i], s = cs[half + i];
                    a[i] = orc_f32_to_bf16(xv * c - yv * s);
                    a[half + i] = orc_f32_to_bf16(yv * c + xv * s);
                }
            }
        }
    }
}

void orc_silu_and_mul_bf16(const uint16_t *gate, const uint16_t *up, int64_t n, uint16_t *out) {
    for (int64_t i = 0; i < n; i++) {
        float g = orc_bf16_to_f32(gate[i]), u = orc_bf16_to_f32(up[i]);
        float s = g / (1.0f + expf(-g));
        out[i] = orc_f32_to_bf16(s * u);
    }
}

/* argmax per row; ties: the reference's merge keeps the incumbent on equality and
 * its order over threads is unspecified, so only the VALUE is comparable on ties.
 * This returns the first index of the maximum. */
void orc_argmax_bf16(const uint16_t *logits, int64_t rows, int64_t vocab, int32_t *out_idx) {
    for (int64_t r = 0; r < rows; r++) {
        float best = -FLT_MAX;
        int32_t bi = 0;
        for (int64_t i = 0; i < vocab; i++) {
            float v = orc_bf16_to_f32(logits[r * vocab + i]);
            if (v > best) { best = v; bi = (int32_t)i; }
        }
        out_idx[r] = bi;
    }
}

/* ---- FP8 e4m3fn ---- */
float orc_fp8_e4m3_decode(uint8_t b) {
    int sign = b & 0x80, exp = (b >> 3) & 0xF, man = b & 7;
    float v;
    if (exp == 0xF && man == 7) v = NAN;
    else if (exp == 0) v = ldexpf((float)man / 8.0f, -6);
    else v = ldexpf(1.0f + (float)man / 8.0f, exp - 7);
    return sign ? -v : v;
}

/* Round-to-nearest-even, saturating to +-448 (the hardware conversion the GPU
 * path uses: crates/core/kernels/fp8_quant.cu static quant). */
uint8_t orc_fp8_e4m3_encode_rne(float f) {
    uint8_t sign = signbit(f) ? 0x80 : 0;
    float a = fabsf(f);
    if (isnan(a)) return (uint8_t)(sign | 0x7F);
    if (a >= 448.0f) return (uint8_t)(sign | 0x7E);
    if (a < ldexpf(1.0f, -10)) { /* below half the smallest subnormal (2^-9): 0, tie 2^-10 -> even 0 */
        return sign;
    }
    int e;
    float m = frexpf(a, &e); /* a = m * 2^e, m in [0.5,1) */
    int exp = e - 1;         /* a = (2m) * 2^exp, 2m in [1,2) */
    if (exp < -6) {          /* subnormal: units of 2^-9 */
        float q = a * 512.0f;
        float r = nearbyintf(q); /* RNE in default rounding mode */
        int qi = (int)r;
        if (qi >= 8) return (uint8_t)(sign | 0x08);
        return (uint8_t)(sign | qi);
    }
    float frac = (2.0f * m - 1.0f) * 8.0f; /* in [0,8) */
    int qi = (int)nearbyintf(frac);
    int be = exp + 7;
    if (qi == 8) { qi = 0; be++; }
    if (be > 15 || (be == 15 && qi == 7)) return (uint8_t)(sign | 0x7E);
    return (uint8_t)(sign | (be << 3) | qi);
}

/* The reference's CPU encoder (mantissa truncation, kv_cache/quantization.rs:408-430). */
uint8_t orc_fp8_e4m3_encode_trunc(float val) {
    if (val == 0.0f) return 0;
    uint8_t sign = val < 0.0f ? 0x80 : 0;
    float a = fabsf(val);
    if (a > 448.0f) a = 448.0f;
    if (a < 1.0f / 512.0f) return sign;
    uint32_t bits;
    memcpy(&bits, &a, 4);
    int e = (int)((bits >> 23) & 0xFF) - 127 + 7;
    if (e < 0) e = 0;
    if (e > 15) e = 15;
    return (uint8_t)(sign | (e << 3) | ((bits & 0x7FFFFF) >> 20));
}

/* KV FP8 store as the GPU path does it: f32(x)/scale, clamp +-448, round to bf16,
 * then bf16 -> e4m3 RNE (cache_engine.rs:202-223, quantization.rs:287-303,459-473). */
uint8_t orc_kv_fp8_quant(uint16_t x_bf16, float scale) {
    float v = orc_bf16_to_f32(x_bf16) / scale;
    if (v > 448.0f) v = 448.0f;
    if (v < -448.0f) v = -448.0f;
    return orc_fp8_e4m3_encode_rne(orc_round_bf16(v));
}

/* GPT-J RoPE (rope.cu:34-45,119-141): pair (2i, 2i + 1) turns by angle i; same cos|sin table layout as NeoX. */
void orc_rope_gptj_bf16(const int32_t *positions, uint16_t *q, uint16_t *k, const float *cos_sin,
                        int64_t tokens, int rot_dim, int head_size, int n_heads, int n_kv_heads) {
    int half = rot_dim / 2;
    for (int64_t t = 0; t < tokens; t++) {
        const float *cs = cos_sin + (int64_t)positions[t] * rot_dim;
        for (int pass = 0; pass < 2; pass++) {
            uint16_t *base = pass == 0 ? q : k;
            int nh = pass == 0 ? n_heads : n_kv_heads;
            if (!base) continue;
            for (int h = 0; h < nh; h++) {
                uint16_t *a = base + (t * nh + h) * head_size;
                for (int i = 0; i < half; i++) {
                    float xv = orc_bf16_to_f32(a[2 * i]), yv = orc_bf16_to_f32(a[2 * i + 1]);
                    float c = cs[i], s = cs[half + i];
                    a[2 * i] = orc_f32_to_bf16(xv * c - yv * s);
                    a[2 * i + 1] = orc_f32_to_bf16(yv * c + xv * s);
                }
            }
        }
    }
}

/* reshape_and_cache_hnd_* (cache_ops.cu:106-176): cache [blocks, kv_heads, block_size, head_dim], 16-bit elements */
void orc_reshape_and_cache_hnd(const uint16_t *key, const uint16_t *value, uint16_t *k_cache, uint16_t *v_cache,
                               const int32_t *slots, int64_t tokens, int kv_heads, int head_dim, int block_size) {
    for (int64_t t = 0; t < tokens; t++) {
        int32_t slot = slots[t];
        if (slot < 0) continue;
        int64_t blk = slot / block_size, off = slot % block_size;
        for (int h = 0; h < kv_heads; h++) {
            int64_t src = (t * kv_heads + h) * head_dim;
            int64_t dst = ((blk * kv_heads + h) * block_size + off) * head_dim;
            for (int d = 0; d < head_dim; d++) {
                k_cache[dst + d] = key[src + d];
                v_cache[dst + d] = value[src + d];
            }
        }
    }
}

/* copy_blocks_* (cache_ops.cu:183-202): pairs (src, dst) in order, whole blocks, both caches */
void orc_copy_blocks(uint8_t *k_cache, uint8_t *v_cache, const int32_t *mapping, int64_t pairs, int64_t bytes_per_block) {
    for (int64_t p = 0; p < pairs; p++) {
        int64_t src = (int64_t)mapping[2 * p] * bytes_per_block, dst = (int64_t)mapping[2 * p + 1] * bytes_per_block;
        for (int64_t i = 0; i < bytes_per_block; i++) {
            k_cache[dst + i] = k_cache[src + i];
            v_cache[dst + i] = v_cache[src + i];
        }
    }
}

static inline uint8_t orc_kv_int8_quant(uint16_t x_bf16, float scale) {
    float a = orc_bf16_to_f32(x_bf16) / scale;
    a = a < -127.0f ? -127.0f : (a > 127.0f ? 127.0f : a);
    return (uint8_t)((int)roundf(a) + 128); /* Tensor::round = round half away from zero */
}

/* FP8 e5m2 is the top byte of an IEEE half: decode through fp16 (load_kv_to_f32<fp8_e5m2_byte>, paged_attention.cu:77-82).
 * Encode: x / scale clamped to the largest finite value 57344 and rounded through bf16 as the e4m3 writer does, then
 * round-to-nearest-even on the 8 dropped bits of the half (exact: bf16 -> half is exact inside e5m2's range except for
 * values under half's normal range, where the double rounding is avoided by rounding from the float directly). */
float orc_fp8_e5m2_decode(uint8_t b) { return orc_f16_to_f32((uint16_t)((uint16_t)b << 8)); }

uint8_t orc_fp8_e5m2_encode_rne(float v) {
    uint32_t u; memcpy(&u, &v, 4);
    uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    float a = fabsf(v);
    if (a != a) return (uint8_t)(sign | 0x7Fu);
    if (a > 57344.0f) a = 57344.0f;
    /* e5m2 grid: normals 2^e * (1 + m/4), e in [-14, 15]; subnormals m * 2^-16 */
    int e; float fr = frexpf(a, &e);          /* a = fr * 2^e, fr in [0.5, 1) */
    if (a == 0.0f) return sign;
    int ex = e - 1;                            /* a = (2 fr) * 2^ex, 2 fr in [1, 2) */
    if (ex < -14) ex = -14;                    /* subnormal spacing */
    float step = ldexpf(1.0f, ex - 2);
    float q = a / step;                        /* exact: power-of-two scaling */
    float r = nearbyintf(q);                   /* default rounding mode: ties to even */
    float val = r * step;
    (void)fr;
    /* val is on the grid (possibly the next binade); read its bits back through half */
    uint16_t h = orc_f32_to_f16(val);
    return (uint8_t)(sign | (uint8_t)(h >> 8));
}

static inline uint8_t orc_kv_e5m2_quant(uint16_t x_bf16, float scale) {
    float a = orc_bf16_to_f32(x_bf16) / scale;
    a = a < -57344.0f ? -57344.0f : (a > 57344.0f ? 57344.0f : a);
    return orc_fp8_e5m2_encode_rne(orc_bf16_to_f32(orc_f32_to_bf16(a)));
}

/* Scatter K/V rows [tokens, kv_heads*head_dim] into the paged cache
 * [blocks, block_size, kv_heads, head_dim]; slot < 0 skips.  kv_dtype 0 = bf16
 * copy, 1 = FP8 e4m3 with per-tensor scales, 2 = INT8 + 128, 4 = FP8 e5m2. */
void orc_reshape_and_cache(const uint16_t *key, const uint16_t *value, void *k_cache, void *v_cache,
                           const int32_t *slots, int64_t tokens, int kv_heads, int head_dim,
                           int block_size, int kv_dtype, float k_scale, float v_scale) {
    int64_t row = (int64_t)kv_heads * head_dim;
    for (int64_t t = 0; t < tokens; t++) {
        int32_t slot = slots[t];
        if (slot < 0) continue;
        int64_t dst = (int64_t)slot * row; /* block*bs*row + off*row == slot*row */
        (void)block_size;
        for (int64_t i = 0; i < row; i++) {
            if (kv_dtype == 0) {
                ((uint16_t *)k_cache)[dst + i] = key[t * row + i];
                ((uint16_t *)v_cache)[dst + i] = value[t * row + i];
            } else if (kv_dtype == 2) { /* quantize_int8 (kv_cache/quantization.rs:330-345): clamp(x / scale, +-127), round, + 128 */
                ((uint8_t *)k_cache)[dst + i] = orc_kv_int8_quant(key[t * row + i], k_scale);
                ((uint8_t *)v_cache)[dst + i] = orc_kv_int8_quant(value[t * row + i], v_scale);
            } else if (kv_dtype == 4) {
                ((uint8_t *)k_cache)[dst + i] = orc_kv_e5m2_quant(key[t * row + i], k_scale);
                ((uint8_t *)v_cache)[dst + i] = orc_kv_e5m2_quant(value[t * row + i], v_scale);
            } else {
                ((uint8_t *)k_cache)[dst + i] = orc_kv_fp8_quant(key[t * row + i], k_scale);
                ((uint8_t *)v_cache)[dst + i] = orc_kv_fp8_quant(value[t * row + i], v_scale);
            }
        }
    }
}

static inline float kv_load(const void *cache, int64_t idx, int kv_dtype, float scale) {
    if (kv_dtype == 0) return orc_bf16_to_f32(((const uint16_t *)cache)[idx]);
    if (kv_dtype == 2) return ((float)((const uint8_t *)cache)[idx] - 128.0f) * scale; /* paged_attention.cu:83-95 */
    if (kv_dtype == 3) return orc_f16_to_f32(((const uint16_t *)cache)[idx]);               /* the _f16 kernel variants, :59-61 */
    if (kv_dtype == 4) return orc_fp8_e5m2_decode(((const uint8_t *)cache)[idx]) * scale;   /* :77-82 */
    return orc_fp8_e4m3_decode(((const uint8_t *)cache)[idx]) * scale;
}

/* Paged-attention decode, V2 structure: per (seq, head) the window
 * [start, seq_len) is cut into partitions of partition_size tokens, each
 * normalised locally with 1/(sum + 1e-6), then merged with the log-sum-exp
 * rule and a second 1/(sum + 1e-6) (paged_attention.cu:750-953).  A single
 * partition is copied through unchanged (:846-855).  out is bf16. */
void orc_paged_attention_v2(const uint16_t *q, const void *k_cache, const void *v_cache,
                            const int32_t *block_tables, const int32_t *seq_lens, float scale,
                            int64_t num_seqs, int num_heads, int num_kv_heads, int max_blocks_per_seq,
                            int head_dim, int block_size, int partition_size, int window, int kv_dtype,
                            float k_scale, float v_scale, uint16_t *out) {
    int qpk = num_heads / num_kv_heads;
    int64_t tok_stride = (int64_t)num_kv_heads * head_dim;
    for (int64_t s = 0; s < num_seqs; s++) {
        int seq_len = seq_lens[s];
        if (seq_len == 0) continue;
        int start = (window > 0 && seq_len > window) ? seq_len - window : 0;
        int eff = seq_len - start;
        int nparts = (eff + partition_size - 1) / partition_size;
        for (int h = 0; h < num_heads; h++) {
            int kvh = h / qpk;
            const uint16_t *qv = q + (s * num_heads + h) * head_dim;
            float *logits = (float *)malloc(sizeof(float) * (size_t)partition_size);
            float gmax = -FLT_MAX;
            float *all_max = (float *)malloc(sizeof(float) * (size_t)nparts);
            float *all_sum = (float *)malloc(sizeof(float) * (size_t)nparts);
            float *all_tmp = (float *)malloc(sizeof(float) * (size_t)nparts * head_dim);
            for (int p = 0; p < nparts; p++) {
                int t0 = start + p * partition_size;
                int t1 = t0 + partition_size < seq_len ? t0 + partition_size : seq_len;
                float m = -FLT_MAX;
                for (int t = t0; t < t1; t++) {
                    int blk = block_tables[s * max_blocks_per_seq + t / block_size];
                    int64_t base = ((int64_t)blk * block_size + t % block_size) * tok_stride + (int64_t)kvh * head_dim;
                    float qk = 0.0f;
                    for (int d = 0; d < head_dim; d++)
                        qk += (kv_dtype == 3 ? orc_f16_to_f32(qv[d]) : orc_bf16_to_f32(qv[d])) * kv_load(k_cache, base + d, kv_dtype, k_scale);
                    float l = qk * scale;
                    logits[t - t0] = l;
                    if (l > m) m = l;
                }
                float sum = 0.0f;
                for (int t = t0; t < t1; t++) {
                    float e = expf(logits[t - t0] - m);
                    logits[t - t0] = e;
                    sum += e;
                }
                float inv = 1.0f / (sum + 1e-6f);
                for (int d = 0; d < head_dim; d++) {
                    float a = 0.0f;
                    for (int t = t0; t < t1; t++) {
                        int blk = block_tables[s * max_blocks_per_seq + t / block_size];
                        int64_t base = ((int64_t)blk * block_size + t % block_size) * tok_stride + (int64_t)kvh * head_dim;
                        a += (logits[t - t0] * inv) * kv_load(v_cache, base + d, kv_dtype, v_scale);
                    }
                    all_tmp[(int64_t)p * head_dim + d] = a;
                }
                all_max[p] = m;
                all_sum[p] = sum;
                if (m > gmax) gmax = m;
            }
            uint16_t *o = out + (s * num_heads + h) * head_dim;
            if (nparts == 1) {
                for (int d = 0; d < head_dim; d++) o[d] = kv_dtype == 3 ? orc_f32_to_f16(all_tmp[d]) : orc_f32_to_bf16(all_tmp[d]);
            } else {
                float gsum = 0.0f;
                for (int p = 0; p < nparts; p++) {
                    all_sum[p] = all_sum[p] * expf(all_max[p] - gmax);
                    gsum += all_sum[p];
                }
                float inv = 1.0f / (gsum + 1e-6f);
                for (int d = 0; d < head_dim; d++) {
                    float a = 0.0f;
                    for (int p = 0; p < nparts; p++) a += all_tmp[(int64_t)p * head_dim + d] * all_sum[p];
                    o[d] = kv_dtype == 3 ? orc_f32_to_f16(a * inv) : orc_f32_to_bf16(a * inv);
                }
            }
            free(all_tmp); free(all_sum); free(all_max); free(logits);
        }
    }
}

/* per_seq_decode_with_lse (crates/core/src/layers/attention/naive.rs:151-205): lse[s][h] = max + ln sum exp(scores - max) over
 * the (windowed) context, scores = scale * q . k; double accumulation here, the test carries the tolerance.  An empty
 * context gives -inf. */
void orc_paged_attention_lse(const uint16_t *q, const void *k_cache, const int32_t *block_tables, const int32_t *seq_lens, float scale,
                             int64_t num_seqs, int num_heads, int num_kv_heads, int max_blocks_per_seq, int head_dim, int block_size,
                             int window, int kv_dtype, float k_scale, double *lse) {
    int qpk = num_heads / num_kv_heads;
    int64_t tok_stride = (int64_t)num_kv_heads * head_dim;
    for (int64_t s = 0; s < num_seqs; s++) {
        int seq_len = seq_lens[s];
        int start = (window > 0 && seq_len > window) ? seq_len - window : 0;
        for (int h = 0; h < num_heads; h++) {
            int kvh = h / qpk;
            const uint16_t *qv = q + (s * num_heads + h) * head_dim;
            double m = -INFINITY, sum = 0.0;
            double *sc = (double *)malloc(sizeof(double) * (size_t)(seq_len > 0 ? seq_len : 1));
            for (int t = start; t < seq_len; t++) {
                int blk = block_tables[s * max_blocks_per_seq + t / block_size];
                int64_t base = ((int64_t)blk * block_size + t % block_size) * tok_stride + (int64_t)kvh * head_dim;
                double qk = 0.0;
                for (int d = 0; d < head_dim; d++) qk += (double)orc_bf16_to_f32(qv[d]) * (double)kv_load(k_cache, base + d, kv_dtype, k_scale);
                sc[t] = qk * (double)scale;
                if (sc[t] > m) m = sc[t];
            }
            for (int t = start; t < seq_len; t++) sum += exp(sc[t] - m);
            lse[s * num_heads + h] = seq_len > start ? m + log(sum) : -INFINITY;
            free(sc);
        }
    }
}

/* Q6_K super-block (210 B / 256 weights): ql[128], qh[64], scales i8[16], f16 d
 * (crates/core/kernels/gather_dequant.cu:52-82). */
void orc_q6k_dequant(const uint8_t *data, int64_t n_blocks, float *out) {
    for (int64_t b = 0; b < n_blocks; b++) {
        const uint8_t *blk = data + b * 210;
        const uint8_t *ql = blk, *qh = blk + 128;
        const int8_t *sc = (const int8_t *)(blk + 192);
        float d = orc_f16_to_f32((uint16_t)(blk[208] | (blk[209] << 8)));
        float *y = out + b * 256;
        for (int half = 0; half < 2; half++) {
            for (int l = 0; l < 32; l++) {
                int is = l / 16;
                int q1 = (int)((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                int q2 = (int)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                int q3 = (int)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                int q4 = (int)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l] = d * (float)sc[is] * (float)q1;
                y[l + 32] = d * (float)sc[is + 2] * (float)q2;
                y[l + 64] = d * (float)sc[is + 4] * (float)q3;
                y[l + 96] = d * (float)sc[is + 6] * (float)q4;
            }
            y += 128; ql += 64; qh += 32; sc += 8;
        }
    }
}

/* ---- FP8 weight path (crates/core/kernels/fp8_gemm.cu:38-70, fp8_quant.cu:89-112,190-262) ---- */

/* y[m,n] = bf16(sum_k x[m,k] * (e4m3(W[n,k]) * scale_n) + bias[n]), f32 sequential accumulate. */
void orc_fp8_gemm_bf16(const uint16_t *x, const uint8_t *W, const float *scale, const uint16_t *bias, int64_t M,
                       int64_t N, int64_t K, int per_channel, uint16_t *y) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        float ws = per_channel ? scale[n] : scale[0];
        for (int64_t m = 0; m < M; m++) {
            float acc = 0.0f;
            for (int64_t k = 0; k < K; k++) acc += orc_bf16_to_f32(x[m * K + k]) * (orc_fp8_e4m3_decode(W[n * K + k]) * ws);
            if (bias) acc += orc_bf16_to_f32(bias[n]);
            y[m * N + n] = orc_f32_to_bf16(acc);
        }
    }
}

/* dynamic per-token: scale = max(absmax/448, 1e-12); out = e4m3_satfinite_rne(x * (1/scale)).
 * static: scale given (per token or single). */
void orc_fp8_quant_bf16(const uint16_t *in, const float *scale_in, int64_t tokens, int64_t hidden, int dynamic,
                        int per_token, uint8_t *out, float *scales_out) {
    for (int64_t t = 0; t < tokens; t++) {
        float sc;
        if (dynamic) {
            float mx = 0.0f;
            for (int64_t i = 0; i < hidden; i++) {
                float v = fabsf(orc_bf16_to_f32(in[t * hidden + i]));
                if (v > mx) mx = v;
            }
            sc = mx / 448.0f;
            if (sc < 1e-12f) sc = 1e-12f;
            scales_out[t] = sc;
        } else {
            sc = per_token ? scale_in[t] : scale_in[0];
        }
        float inv = 1.0f / sc;
        for (int64_t i = 0; i < hidden; i++) out[t * hidden + i] = orc_fp8_e4m3_encode_rne(orc_bf16_to_f32(in[t * hidden + i]) * inv);
    }
}

void orc_fp8_dequant_bf16(const uint8_t *in, const float *scale, int64_t rows, int64_t cols, int per_row, uint16_t *out) {
    for (int64_t r = 0; r < rows; r++)
        for (int64_t c = 0; c < cols; c++)
            out[r * cols + c] = orc_f32_to_bf16(orc_fp8_e4m3_decode(in[r * cols + c]) * (per_row ? scale[r] : scale[0]));
}

/* fp8_scaled_mm_{bf16,f32} (crates/core/kernels/fp8_gemm.cu:332-466): both operands e4m3, each element multiplied by its scale
 * before the product, f32 accumulate in k order, + bias, -> bf16 (or f32).  a_mode / b_mode: 0 one scale, 1 per row / column. */
void orc_fp8_scaled_mm(const uint8_t *A, const uint8_t *B, const float *a_scale, const float *b_scale, const uint16_t *bias, int64_t M,
                       int64_t N, int64_t K, int a_mode, int b_mode, int out_f32, void *out) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        const float sb = b_mode ? b_scale[n] : b_scale[0];
        for (int64_t m = 0; m < M; m++) {
            const float sa = a_mode ? a_scale[m] : a_scale[0];
            float acc = 0.0f;
            for (int64_t k = 0; k < K; k++) acc += (orc_fp8_e4m3_decode(A[m * K + k]) * sa) * (orc_fp8_e4m3_decode(B[n * K + k]) * sb);
            if (bias) acc += orc_bf16_to_f32(bias[n]);
            if (out_f32) ((float *)out)[m * N + n] = acc;
            else ((uint16_t *)out)[m * N + n] = orc_f32_to_bf16(acc);
        }
    }
}

/* Block-scaled weights as the reference documents them (crates/core/src/quantization/fp8.rs:305-342):
 * W_bf16[i][j] = bf16(e4m3(w[i][j]) * scale[i / block_n][j / block_k]). */
void orc_fp8_dequant_block_bf16(const uint8_t *in, const float *scale, int64_t rows, int64_t cols, int64_t block_n, int64_t block_k,
                                uint16_t *out) {
    const int64_t sc = (cols + block_k - 1) / block_k;
    for (int64_t r = 0; r < rows; r++)
        for (int64_t c = 0; c < cols; c++)
            out[r * cols + c] = orc_f32_to_bf16(orc_fp8_e4m3_decode(in[r * cols + c]) * scale[(r / block_n) * sc + c / block_k]);
}

/* fused_rmsnorm_fp8 (crates/core/kernels/layernorm_quant.cu:33-134): normed = x * rsqrt(mean(x^2) + eps) * w in f32 (never rounded
 * to bf16), per-row scale = max(absmax(normed) / 448, 1e-12), out = e4m3_satfinite_rne(normed * (1 / scale)). */
void orc_fused_rmsnorm_fp8(const uint16_t *in, const uint16_t *w, int64_t rows, int64_t hidden, float eps, uint8_t *out, float *scales) {
    for (int64_t r = 0; r < rows; r++) {
        /* the kernel sums 256 strided per-thread partials and then a tree, all in f32; a double sum rounded once sits within that
         * order's noise (a sequential f32 sum does not: its error grows with the row length) */
        double ssd = 0.0;
        for (int64_t i = 0; i < hidden; i++) {
            const float v = orc_bf16_to_f32(in[r * hidden + i]);
            ssd += (double)(v * v);
        }
        const float ss = (float)ssd;
        const float rms = 1.0f / sqrtf(ss / (float)hidden + eps);
        float mx = 0.0f;
        for (int64_t i = 0; i < hidden; i++) {
            const float v = fabsf(orc_bf16_to_f32(in[r * hidden + i]) * rms * orc_bf16_to_f32(w[i]));
            if (v > mx) mx = v;
        }
        float sc = mx / 448.0f;
        if (sc < 1e-12f) sc = 1e-12f;
        scales[r] = sc;
        const float inv = 1.0f / sc;
        for (int64_t i = 0; i < hidden; i++)
            out[r * hidden + i] = orc_fp8_e4m3_encode_rne(orc_bf16_to_f32(in[r * hidden + i]) * rms * orc_bf16_to_f32(w[i]) * inv);
    }
}

/* Dense bf16 linear without bias, one row at a time: y[m,n] = bf16(sum_k x[m,k] * W[n,k]), f32 sequential accumulate
 * over k.  This is the arithmetic of the reference's unquantized Linear on the decode path (candle matmul of bf16
 * operands with f32 accumulation, result cast to bf16): the lm_head of the quantized models
 * (crates/core/src/models/qwen3_quantized.rs:1258-1302) and the projections of an unquantized checkpoint
 * (crates/core/src/models/qwen3.rs).  bf16 x bf16 products are exact in f32; only the summation order is a choice. */
void orc_dense_gemv_bf16(const uint16_t *x, const uint16_t *W, int64_t M, int64_t N, int64_t K, uint16_t *y) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        for (int64_t m = 0; m < M; m++) {
            float acc = 0.0f;
            for (int64_t k = 0; k < K; k++) acc += orc_bf16_to_f32(x[m * K + k]) * orc_bf16_to_f32(W[n * K + k]);
            y[m * N + n] = orc_f32_to_bf16(acc);
        }
    }
}

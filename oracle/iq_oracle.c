/*
 * CPU oracle (test infrastructure only, see orc_common.h) for the GGUF I-quant
 * path: block dequantisation, q8_1 activation quantisation and the q8_1 MMVQ
 * integer dot product.
 *
 * Follows /root/reference/crates/core/src/quantization/gguf/iq/mod.rs:
 *   block layouts + float dequant        :136-294
 *   integer dots (vec_dot_iq*_q8_1)      :556-658
 *   q8_1 quantiser                       :663-682
 *   MMVQ accumulation order              :689-725
 * Pinned by the reference's gguf-py golden vectors (tests/golden/iq*.json,
 * copied from crates/core/src/quantization/gguf/iq/testdata/).
 *
 * Restatement: every 32-weight sub-block is first decoded to 32 signed small
 * integers q[j] (codebook byte times sign) plus its scale fields; the float
 * dequant and the integer dot are both written over q[].  Integer results are
 * identical to the reference's dp4a/vsub4 formulation because int32 addition
 * is associative; float dequant keeps the reference's evaluation order
 * (d * scale_term) * factor, then * |q|, then * sign.
 */
#include "orc_common.h"
#include "iq_tables_gen.h"

enum { QK_K = 256 };

int orc_iq_type_size(int ggml_type) {
    switch (ggml_type) {
        case 17: return 74;   /* IQ2_XS  */
        case 22: return 82;   /* IQ2_S   */
        case 18: return 98;   /* IQ3_XXS */
        case 21: return 110;  /* IQ3_S   */
        case 23: return 136;  /* IQ4_XS  */
        default: return -1;
    }
}

static inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* Scale fields of one 32-weight sub-block. */
typedef struct {
    int ls0, ls1; /* IQ2_*: 4-bit scales of the two 16-weight halves;
                     IQ3_XXS: ls0 = 4-bit scale; IQ3_S: ls0 = nibble;
                     IQ4_XS: ls0 = 6-bit scale (before the -32 bias) */
} sub_scale_t;

/* Decode sub-block ib (0..7) of one super-block into 32 signed integers. */
static void decode_sub(int type, const uint8_t *b, int ib, int8_t q[32], sub_scale_t *sc) {
    sc->ls0 = sc->ls1 = 0;
    switch (type) {
    case 17: { /* IQ2_XS: qs u16[32]@2, scales u8[8]@66 (mod.rs:136-158) */
        for (int l = 0; l < 4; l++) {
            uint16_t v = rd16(b + 2 + 2 * (4 * ib + l));
            uint64_t grid = iq2xs_grid[v & 511];
            uint8_t signs = ksigns_iq2xs[v >> 9];
            for (int j = 0; j < 8; j++) {
                int g = (int)((grid >> (8 * j)) & 0xff);
                q[8 * l + j] = (int8_t)((signs & kmask_iq2xs[j]) ? -g : g);
            }
        }
        sc->ls0 = b[66 + ib] & 0xf;
        sc->ls1 = b[66 + ib] >> 4;
        break;
    }
    case 22: { /* IQ2_S: qs u8[32]@2, signs u8[32]@34, qh u8[8]@66, scales u8[8]@74 (mod.rs:161-192) */
        int qh = b[66 + ib];
        for (int l = 0; l < 4; l++) {
            int idx = b[2 + 4 * ib + l] | ((qh << (8 - 2 * l)) & 0x300);
            uint64_t grid = iq2s_grid[idx];
            uint8_t signs = b[34 + 4 * ib + l];
            for (int j = 0; j < 8; j++) {
                int g = (int)((grid >> (8 * j)) & 0xff);
                q[8 * l + j] = (int8_t)((signs & kmask_iq2xs[j]) ? -g : g);
            }
        }
        sc->ls0 = b[74 + ib] & 0xf;
        sc->ls1 = b[74 + ib] >> 4;
        break;
    }
    case 18: { /* IQ3_XXS: qs u8[64]@2, scales_and_signs u32[8]@66 (mod.rs:195-220) */
        uint32_t aux = rd32(b + 66 + 4 * ib);
        for (int l = 0; l < 4; l++) {
            uint8_t signs = ksigns_iq2xs[(aux >> (7 * l)) & 127];
            uint32_t g1 = iq3xxs_grid[b[2 + 8 * ib + 2 * l]];
            uint32_t g2 = iq3xxs_grid[b[2 + 8 * ib + 2 * l + 1]];
            for (int j = 0; j < 4; j++) {
                int a = (int)((g1 >> (8 * j)) & 0xff);
                int c = (int)((g2 >> (8 * j)) & 0xff);
                q[8 * l + j] = (int8_t)((signs & kmask_iq2xs[j]) ? -a : a);
                q[8 * l + j + 4] = (int8_t)((signs & kmask_iq2xs[j + 4]) ? -c : c);
            }
        }
        sc->ls0 = (int)(aux >> 28);
        break;
    }
    case 21: { /* IQ3_S: qs u8[64]@2, qh u8[8]@66, signs u8[32]@74, scales u8[4]@106 (mod.rs:223-271) */
        int qh = b[66 + ib];
        for (int l = 0; l < 4; l++) {
            uint32_t g1 = iq3s_grid[b[2 + 8 * ib + 2 * l] | ((qh << (8 - 2 * l)) & 256)];
            uint32_t g2 = iq3s_grid[b[2 + 8 * ib + 2 * l + 1] | ((qh << (7 - 2 * l)) & 256)];
            uint8_t signs = b[74 + 4 * ib + l];
            for (int j = 0; j < 4; j++) {
                int a = (int)((g1 >> (8 * j)) & 0xff);
                int c = (int)((g2 >> (8 * j)) & 0xff);
                q[8 * l + j] = (int8_t)((signs & kmask_iq2xs[j]) ? -a : a);
                q[8 * l + j + 4] = (int8_t)((signs & kmask_iq2xs[j + 4]) ? -c : c);
            }
        }
        sc->ls0 = (b[106 + ib / 2] >> (4 * (ib & 1))) & 0xf;
        break;
    }
    case 23: { /* IQ4_XS: scales_h u16@2, scales_l u8[4]@4, qs u8[128]@8 (mod.rs:274-294) */
        uint16_t scales_h = rd16(b + 2);
        for (int j = 0; j < 16; j++) {
            uint8_t v = b[8 + 16 * ib + j];
            q[j] = kvalues_iq4nl[v & 0xf];
            q[16 + j] = kvalues_iq4nl[v >> 4];
        }
        sc->ls0 = ((b[4 + ib / 2] >> (4 * (ib % 2))) & 0xf) | (((scales_h >> (2 * ib)) & 3) << 4);
        break;
    }
    default: break;
    }
}

/* Float dequantisation of n_blocks super-blocks -> n_blocks*256 f32. */
int orc_iq_dequant(int type, const uint8_t *data, int64_t n_blocks, float *out) {
    int ts = orc_iq_type_size(type);
    if (ts < 0) return -1;
    for (int64_t blk = 0; blk < n_blocks; blk++) {
        const uint8_t *b = data + blk * ts;
        float d = orc_f16_to_f32(rd16(b));
        float *y = out + blk * QK_K;
        for (int ib = 0; ib < 8; ib++) {
            int8_t q[32];
            sub_scale_t sc;
            decode_sub(type, b, ib, q, &sc);
            float db[2];
            switch (type) {
            case 17: case 22:
                db[0] = d * (0.5f + (float)sc.ls0) * 0.25f;
                db[1] = d * (0.5f + (float)sc.ls1) * 0.25f;
                break;
            case 18:
                db[0] = db[1] = d * (0.5f + (float)sc.ls0) * 0.5f;
                break;
            case 21:
                db[0] = db[1] = d * (float)(1 + 2 * sc.ls0);
                break;
            default: /* 23 */
                db[0] = db[1] = d * (float)(sc.ls0 - 32);
                break;
            }
            for (int j = 0; j < 32; j++) {
                float dl = db[j / 16];
                int v = q[j];
                if (type == 23) {
                    y[32 * ib + j] = dl * (float)v;
                } else {
                    float mag = dl * (float)(v < 0 ? -v : v);
                    y[32 * ib + j] = v < 0 ? mag * -1.0f : mag * 1.0f;
                }
            }
        }
    }
    return 0;
}

/* Integer dot of sub-block ib with 32 int8 activations, including the
 * per-type integer scale step with C truncating division (mod.rs:556-658). */
int32_t orc_iq_subdot(int type, const uint8_t *block, int ib, const int8_t *a) {
    int8_t q[32];
    sub_scale_t sc;
    decode_sub(type, block, ib, q, &sc);
    int32_t s0 = 0, s1 = 0;
    for (int j = 0; j < 16; j++) s0 += (int32_t)q[j] * (int32_t)a[j];
    for (int j = 16; j < 32; j++) s1 += (int32_t)q[j] * (int32_t)a[j];
    switch (type) {
    case 17: case 22:
        return (s0 * sc.ls0 + s1 * sc.ls1 + (s0 + s1) / 2) / 4;
    case 18: {
        int32_t sumi = s0 + s1;
        return (sc.ls0 * sumi + sumi / 2) / 2;
    }
    case 21:
        return (s0 + s1) * (1 + 2 * sc.ls0);
    case 23:
        return (s0 + s1) * (sc.ls0 - 32);
    default:
        return 0;
    }
}

/* q8_1 row quantiser: per 32 values d = amax/127, q = roundf(x/d)
 * (half away from zero); an all-zero block gives d = 0, q = 0 (mod.rs:663-682). */
void orc_quantize_q8_1(const float *x, int64_t m, int64_t n_in, int8_t *aq, float *ad) {
    int64_t n_blocks = m * (n_in / 32);
    for (int64_t blk = 0; blk < n_blocks; blk++) {
        const float *xb = x + blk * 32;
        float amax = 0.0f;
        for (int i = 0; i < 32; i++) {
            float v = fabsf(xb[i]);
            if (v > amax) amax = v;
        }
        float d = amax / 127.0f;
        ad[blk] = d;
        for (int i = 0; i < 32; i++) {
            float r = 0.0f;
            if (amax != 0.0f) r = roundf(xb[i] / d);
            if (r > 127.0f) r = 127.0f;
            if (r < -128.0f) r = -128.0f;
            aq[blk * 32 + i] = (int8_t)r;
        }
    }
}

/* All integer sub-dots: out[(o*m + mi)*nblk32 + g].  For bit-exact checks. */
int orc_iq_subdots(int type, const uint8_t *w, const int8_t *aq, int64_t n_out, int64_t n_in,
                   int64_t m, int32_t *out) {
    int ts = orc_iq_type_size(type);
    if (ts < 0 || n_in % QK_K) return -1;
    int64_t nb256 = n_in / QK_K, nblk32 = n_in / 32;
    for (int64_t o = 0; o < n_out; o++)
        for (int64_t mi = 0; mi < m; mi++)
            for (int64_t g = 0; g < nblk32; g++) {
                const uint8_t *wb = w + (o * nb256 + (g >> 3)) * ts;
                out[(o * m + mi) * nblk32 + g] = orc_iq_subdot(type, wb, (int)(g & 7), aq + mi * n_in + g * 32);
            }
    return 0;
}

/* MMVQ with pre-quantised activations, sequential f32 accumulation in the
 * reference's order: acc += (w_d * a_d) * (float)subdot  (mod.rs:702-722). */
int orc_iq_mmvq_q8(int type, const uint8_t *w, const int8_t *aq, const float *ad, int64_t n_out,
                   int64_t n_in, int64_t m, float *y) {
    int ts = orc_iq_type_size(type);
    if (ts < 0 || n_in % QK_K) return -1;
    int64_t nb256 = n_in / QK_K, nblk32 = n_in / 32;
    for (int64_t o = 0; o < n_out; o++) {
        for (int64_t mi = 0; mi < m; mi++) {
            float acc = 0.0f;
            for (int64_t g = 0; g < nblk32; g++) {
                const uint8_t *wb = w + (o * nb256 + (g >> 3)) * ts;
                float w_d = orc_f16_to_f32(rd16(wb));
                int32_t sub = orc_iq_subdot(type, wb, (int)(g & 7), aq + mi * n_in + g * 32);
                acc += w_d * ad[mi * nblk32 + g] * (float)sub;
            }
            y[mi * n_out + o] = acc;
        }
    }
    return 0;
}

/* Y[m, n_out] = X[m, n_in] @ W^T through q8_1 (q8_1_mmvq_cpu, mod.rs:689-725).
 * aq/ad are caller scratch of m*n_in and m*n_in/32 elements. */
int orc_iq_matmul(int type, const uint8_t *w, const float *x, int64_t n_out, int64_t n_in, int64_t m,
                  float *y, int8_t *aq, float *ad) {
    if (n_in % QK_K) return -1;
    orc_quantize_q8_1(x, m, n_in, aq, ad);
    return orc_iq_mmvq_q8(type, w, aq, ad, n_out, n_in, m, y);
}

/* Row-parallel variant for the timed CPU baseline (OpenMP over output rows,
 * identical per-row arithmetic). */
int orc_iq_mmvq_q8_mt(int type, const uint8_t *w, const int8_t *aq, const float *ad, int64_t n_out,
                      int64_t n_in, int64_t m, float *y) {
    int ts = orc_iq_type_size(type);
    if (ts < 0 || n_in % QK_K) return -1;
    int64_t nb256 = n_in / QK_K, nblk32 = n_in / 32;
#pragma omp parallel for schedule(static)
    for (int64_t o = 0; o < n_out; o++) {
        for (int64_t mi = 0; mi < m; mi++) {
            float acc = 0.0f;
            for (int64_t g = 0; g < nblk32; g++) {
                const uint8_t *wb = w + (o * nb256 + (g >> 3)) * ts;
                float w_d = orc_f16_to_f32(rd16(wb));
                int32_t sub = orc_iq_subdot(type, wb, (int)(g & 7), aq + mi * n_in + g * 32);
                acc += w_d * ad[mi * nblk32 + g] * (float)sub;
            }
            y[mi * n_out + o] = acc;
        }
    }
    return 0;
}

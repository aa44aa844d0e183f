/*
 * CPU oracle (test infrastructure only, see orc_common.h): GGML legacy / K-quant block dequantisation,
 * written as ggml's dequantize_row_* loops (public ggml block specs; the reference gets this arithmetic
 * from candle-core 0.10.2, which is not in the reference tree -> "parity unpinned": no golden vectors).
 * Q6_K lives in glue_oracle.c (restated from crates/core/kernels/gather_dequant.cu:52-82).
 */
#include "orc_common.h"

static inline float h2f(const uint8_t *p) { return orc_f16_to_f32((uint16_t)(p[0] | (p[1] << 8))); }

void orc_q4_0_dequant(const uint8_t *data, int64_t nb, float *y) {
    for (int64_t i = 0; i < nb; i++, data += 18, y += 32) {
        float d = h2f(data);
        for (int j = 0; j < 16; j++) {
            y[j] = (float)((data[2 + j] & 0xF) - 8) * d;
            y[j + 16] = (float)((data[2 + j] >> 4) - 8) * d;
        }
    }
}

void orc_q8_0_dequant(const uint8_t *data, int64_t nb, float *y) {
    for (int64_t i = 0; i < nb; i++, data += 34, y += 32) {
        float d = h2f(data);
        for (int j = 0; j < 32; j++) y[j] = (float)(int8_t)data[2 + j] * d;
    }
}

static void get_scale_min_k4(int j, const uint8_t *q, uint8_t *d, uint8_t *m) {
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else { *d = (uint8_t)((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4)); *m = (uint8_t)((q[j + 4] >> 4) | ((q[j] >> 6) << 4)); }
}

void orc_q4_k_dequant(const uint8_t *data, int64_t nb, float *y) {
    for (int64_t i = 0; i < nb; i++, data += 144) {
        const float d = h2f(data), dmin = h2f(data + 2);
        const uint8_t *scales = data + 4, *q = data + 16;
        int is = 0;
        for (int j = 0; j < 256; j += 64) {
            uint8_t sc, m;
            get_scale_min_k4(is + 0, scales, &sc, &m);
            const float d1 = d * sc, m1 = dmin * m;
            get_scale_min_k4(is + 1, scales, &sc, &m);
            const float d2 = d * sc, m2 = dmin * m;
            for (int l = 0; l < 32; l++) *y++ = d1 * (float)(q[l] & 0xF) - m1;
            for (int l = 0; l < 32; l++) *y++ = d2 * (float)(q[l] >> 4) - m2;
            q += 32;
            is += 2;
        }
    }
}

void orc_q3_k_dequant(const uint8_t *data, int64_t nb, float *y) {
    const uint32_t kmask1 = 0x03030303u, kmask2 = 0x0f0f0f0fu;
    for (int64_t i = 0; i < nb; i++, data += 110) {
        const uint8_t *hm = data, *q = data + 32;
        const float d_all = h2f(data + 108);
        uint32_t aux[4];
        memcpy(aux, data + 96, 12);
        uint32_t tmp = aux[2];
        aux[2] = ((aux[0] >> 4) & kmask2) | (((tmp >> 4) & kmask1) << 4);
        aux[3] = ((aux[1] >> 4) & kmask2) | (((tmp >> 6) & kmask1) << 4);
        aux[0] = (aux[0] & kmask2) | (((tmp >> 0) & kmask1) << 4);
        aux[1] = (aux[1] & kmask2) | (((tmp >> 2) & kmask1) << 4);
        const int8_t *scales = (const int8_t *)aux;
        int is = 0;
        uint8_t m = 1;
        for (int n = 0; n < 256; n += 128) {
            int shift = 0;
            for (int j = 0; j < 4; j++) {
                float dl = d_all * (float)(scales[is++] - 32);
                for (int l = 0; l < 16; l++) *y++ = dl * (float)((int8_t)((q[l + 0] >> shift) & 3) - ((hm[l + 0] & m) ? 0 : 4));
                dl = d_all * (float)(scales[is++] - 32);
                for (int l = 0; l < 16; l++) *y++ = dl * (float)((int8_t)((q[l + 16] >> shift) & 3) - ((hm[l + 16] & m) ? 0 : 4));
                shift += 2;
                m <<= 1;
            }
            q += 32;
        }
    }
}

void orc_q4_1_dequant(const uint8_t *data, int64_t nb, float *y) {
    for (int64_t i = 0; i < nb; i++, data += 20, y += 32) {
        const float d = h2f(data), m = h2f(data + 2);
        for (int j = 0; j < 16; j++) {
            y[j] = (float)(data[4 + j] & 0xF) * d + m;
            y[j + 16] = (float)(data[4 + j] >> 4) * d + m;
        }
    }
}

void orc_q5_0_dequant(const uint8_t *data, int64_t nb, float *y) {
    for (int64_t i = 0; i < nb; i++, data += 22, y += 32) {
        const float d = h2f(data);
        uint32_t qh;
        memcpy(&qh, data + 2, 4);
        for (int j = 0; j < 16; j++) {
            const uint8_t xh_0 = (uint8_t)(((qh >> (j + 0)) << 4) & 0x10), xh_1 = (uint8_t)((qh >> (j + 12)) & 0x10);
            const int x0 = (int)((data[6 + j] & 0x0F) | xh_0) - 16, x1 = (int)((data[6 + j] >> 4) | xh_1) - 16;
            y[j] = (float)x0 * d;
            y[j + 16] = (float)x1 * d;
        }
    }
}

void orc_q5_1_dequant(const uint8_t *data, int64_t nb, float *y) {
    for (int64_t i = 0; i < nb; i++, data += 24, y += 32) {
        const float d = h2f(data), m = h2f(data + 2);
        uint32_t qh;
        memcpy(&qh, data + 4, 4);
        for (int j = 0; j < 16; j++) {
            const uint8_t xh_0 = (uint8_t)(((qh >> (j + 0)) << 4) & 0x10), xh_1 = (uint8_t)((qh >> (j + 12)) & 0x10);
            const int x0 = (data[8 + j] & 0x0F) | xh_0, x1 = (data[8 + j] >> 4) | xh_1;
            y[j] = (float)x0 * d + m;
            y[j + 16] = (float)x1 * d + m;
        }
    }
}

/* block_q2_K: scales[16] @0, qs[64] @16, d @80, dmin @82 */
void orc_q2_k_dequant(const uint8_t *data, int64_t nb, float *y) {
    for (int64_t i = 0; i < nb; i++, data += 84) {
        const float d = h2f(data + 80), min = h2f(data + 82);
        const uint8_t *q = data + 16, *scales = data;
        int is = 0;
        for (int n = 0; n < 256; n += 128) {
            int shift = 0;
            for (int j = 0; j < 4; j++) {
                uint8_t sc = scales[is++];
                float dl = d * (float)(sc & 0xF), ml = min * (float)(sc >> 4);
                for (int l = 0; l < 16; l++) *y++ = dl * (float)((int8_t)((q[l] >> shift) & 3)) - ml;
                sc = scales[is++];
                dl = d * (float)(sc & 0xF); ml = min * (float)(sc >> 4);
                for (int l = 0; l < 16; l++) *y++ = dl * (float)((int8_t)((q[l + 16] >> shift) & 3)) - ml;
                shift += 2;
            }
            q += 32;
        }
    }
}

/* block_q5_K: d @0, dmin @2, scales[12] @4, qh[32] @16, qs[128] @48 */
void orc_q5_k_dequant(const uint8_t *data, int64_t nb, float *y) {
    for (int64_t i = 0; i < nb; i++, data += 176) {
        const float d = h2f(data), min = h2f(data + 2);
        const uint8_t *scales = data + 4, *qh = data + 16, *ql = data + 48;
        int is = 0;
        uint8_t u1 = 1, u2 = 2;
        for (int j = 0; j < 256; j += 64) {
            uint8_t sc, m;
            get_scale_min_k4(is + 0, scales, &sc, &m);
            const float d1 = d * sc, m1 = min * m;
            get_scale_min_k4(is + 1, scales, &sc, &m);
            const float d2 = d * sc, m2 = min * m;
            for (int l = 0; l < 32; l++) *y++ = d1 * (float)((ql[l] & 0xF) + ((qh[l] & u1) ? 16 : 0)) - m1;
            for (int l = 0; l < 32; l++) *y++ = d2 * (float)((ql[l] >> 4) + ((qh[l] & u2) ? 16 : 0)) - m2;
            ql += 32;
            is += 2;
            u1 <<= 2;
            u2 <<= 2;
        }
    }
}

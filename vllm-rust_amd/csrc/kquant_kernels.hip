// GGML K-quant / legacy block formats: dequantisation and a dequant-fused small-M GEMV.
// In the reference these tensors go through Candle's QMatMul (crates/core/src/quantization/gguf/mod.rs:114-258),
// whose kernels live in candle-kernels 0.10.2 and are not part of the reference tree; the block layouts and
// decode formulas below are the public ggml ones (ggml-common.h / ggml-quants.c): Q4_0 18 B/32, Q8_0 34 B/32,
// Q4_1 20 B/32, Q5_0 22 B/32, Q5_1 24 B/32, Q2_K 84 B/256, Q3_K 110 B/256, Q4_K 144 B/256, Q5_K 176 B/256, Q6_K 210 B/256 (Q6_K is also restated in crates/core/kernels/gather_dequant.cu:52-82).
// The GEMV multiplies the f32 activations with the exactly dequantised weights in f32 (no q8_1/Q8_K
// activation quantisation step as in ggml's mul_mat_vec_q): parity with the third-party kernel is
// tolerance-only by construction (the reference's own gate is rel-RMS < 3 % against a dense product).
#include "vmx_internal.h"

namespace {

struct __attribute__((packed, aligned(2))) u16a2 { uint16_t v; };
__device__ __forceinline__ float ldh(const uint8_t *p) { return vmx_f16_to_f32(reinterpret_cast<const u16a2 *>(p)->v); }

template <int T> struct KTraits;
template <> struct KTraits<2> { static constexpr int TS = 18, QK = 32; };    // Q4_0
template <> struct KTraits<3> { static constexpr int TS = 20, QK = 32; };    // Q4_1
template <> struct KTraits<6> { static constexpr int TS = 22, QK = 32; };    // Q5_0
template <> struct KTraits<7> { static constexpr int TS = 24, QK = 32; };    // Q5_1
template <> struct KTraits<8> { static constexpr int TS = 34, QK = 32; };    // Q8_0
template <> struct KTraits<10> { static constexpr int TS = 84, QK = 256; };  // Q2_K
template <> struct KTraits<13> { static constexpr int TS = 176, QK = 256; }; // Q5_K
template <> struct KTraits<11> { static constexpr int TS = 110, QK = 256; }; // Q3_K
template <> struct KTraits<12> { static constexpr int TS = 144, QK = 256; }; // Q4_K
template <> struct KTraits<14> { static constexpr int TS = 210, QK = 256; }; // Q6_K

// Decode the 32 weights of sub-block s (s < QK/32) of block b into f[32] (element order of the block).
template <int T> __device__ __forceinline__ void kdecode32(const uint8_t *b, int s, float f[32]) {
    if constexpr (T == 2) {
        const float d = ldh(b);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int q = b[2 + j];
            f[j] = (float)((q & 0xF) - 8) * d;
            f[j + 16] = (float)((q >> 4) - 8) * d;
        }
    } else if constexpr (T == 3) {   // d f16 @0, m f16 @2, qs[16] @4
        const float d = ldh(b), mn = ldh(b + 2);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int q = b[4 + j];
            f[j] = __fadd_rn(__fmul_rn((float)(q & 0xF), d), mn);
            f[j + 16] = __fadd_rn(__fmul_rn((float)(q >> 4), d), mn);
        }
    } else if constexpr (T == 6 || T == 7) {   // Q5_0: d, qh u32 @2, qs[16] @6;  Q5_1: d, m, qh u32 @4, qs[16] @8
        const float d = ldh(b), mn = T == 7 ? ldh(b + 2) : 0.f;
        const uint8_t *hp = b + (T == 7 ? 4 : 2), *qs = b + (T == 7 ? 8 : 6);
        const uint32_t qh = hp[0] | (hp[1] << 8) | (hp[2] << 16) | ((uint32_t)hp[3] << 24);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int x0 = (qs[j] & 0xF) | (((qh >> j) << 4) & 0x10), x1 = (qs[j] >> 4) | ((qh >> (j + 12)) & 0x10);
            if constexpr (T == 6) {
                f[j] = (float)(x0 - 16) * d;
                f[j + 16] = (float)(x1 - 16) * d;
            } else {
                f[j] = __fadd_rn(__fmul_rn((float)x0, d), mn);
                f[j + 16] = __fadd_rn(__fmul_rn((float)x1, d), mn);
            }
        }
    } else if constexpr (T == 8) {
        const float d = ldh(b);
#pragma unroll
        for (int j = 0; j < 32; j++) f[j] = (float)(int8_t)b[2 + j] * d;
    } else if constexpr (T == 10) {  // scales[16] @0, qs[64] @16, d f16 @80, dmin f16 @82
        const float d = ldh(b + 80), dmin = ldh(b + 82);
        const int n = s >> 2, j = s & 3;
        const uint8_t *q = b + 16 + 32 * n;
#pragma unroll
        for (int l = 0; l < 32; l++) {
            const int sc = b[8 * n + 2 * j + (l >> 4)];
            const float dl = d * (float)(sc & 0xF), ml = dmin * (float)(sc >> 4);
            f[l] = __fsub_rn(__fmul_rn(dl, (float)((q[l] >> (2 * j)) & 3)), ml);
        }
    } else if constexpr (T == 13) {  // d @0, dmin @2, scales[12] @4, qh[32] @16, qs[128] @48
        const float d = ldh(b), dmin = ldh(b + 2);
        const uint8_t *sc = b + 4;
        int scv, mv;
        if (s < 4) { scv = sc[s] & 63; mv = sc[s + 4] & 63; }
        else { scv = (sc[s + 4] & 0xF) | ((sc[s - 4] >> 6) << 4); mv = (sc[s + 4] >> 4) | ((sc[s] >> 6) << 4); }
        const float dl = d * (float)scv, ml = dmin * (float)mv;
        const uint8_t *ql = b + 48 + 32 * (s >> 1), *qh = b + 16;
        const int sh = (s & 1) * 4;
#pragma unroll
        for (int l = 0; l < 32; l++) f[l] = __fsub_rn(__fmul_rn(dl, (float)(((ql[l] >> sh) & 0xF) + (((qh[l] >> s) & 1) << 4))), ml);
    } else if constexpr (T == 12) {  // d f16 @0, dmin f16 @2, scales[12] @4, qs[128] @16
        const float d = ldh(b), dmin = ldh(b + 2);
        const uint8_t *sc = b + 4;
        int scv, mv;
        if (s < 4) { scv = sc[s] & 63; mv = sc[s + 4] & 63; }
        else { scv = (sc[s + 4] & 0xF) | ((sc[s - 4] >> 6) << 4); mv = (sc[s + 4] >> 4) | ((sc[s] >> 6) << 4); }
        const float dl = d * (float)scv, ml = dmin * (float)mv;
        const uint8_t *q = b + 16 + 32 * (s >> 1);
        const int sh = (s & 1) * 4;
#pragma unroll
        for (int l = 0; l < 32; l++) f[l] = dl * (float)((q[l] >> sh) & 0xF) - ml;
    } else if constexpr (T == 14) {  // ql[128] @0, qh[64] @128, scales i8[16] @192, d f16 @208
        const float d = ldh(b + 208);
        const int half = s >> 2, quarter = s & 3;  // 128-weight half; 32-weight group inside it
        const uint8_t *ql = b + 64 * half + 32 * (quarter & 1), *qh = b + 128 + 32 * half;
        const int8_t *scs = reinterpret_cast<const int8_t *>(b + 192) + 8 * half + 2 * quarter;
        const int lsh = (quarter >> 1) * 4, hsh = 2 * quarter;
#pragma unroll
        for (int l = 0; l < 32; l++) {
            const int q = (int)(((ql[l] >> lsh) & 0xF) | (((qh[l] >> hsh) & 3) << 4)) - 32;
            f[l] = d * (float)scs[l >> 4] * (float)q;
        }
    } else {  // Q3_K: hmask[32] @0, qs[64] @32, scales[12] @96, d f16 @108
        const float d = ldh(b + 108);
        uint32_t a0 = b[96] | (b[97] << 8) | (b[98] << 16) | ((uint32_t)b[99] << 24);
        uint32_t a1 = b[100] | (b[101] << 8) | (b[102] << 16) | ((uint32_t)b[103] << 24);
        const uint32_t tmp = b[104] | (b[105] << 8) | (b[106] << 16) | ((uint32_t)b[107] << 24);
        uint32_t aux[4];
        aux[2] = ((a0 >> 4) & 0x0f0f0f0fu) | (((tmp >> 4) & 0x03030303u) << 4);
        aux[3] = ((a1 >> 4) & 0x0f0f0f0fu) | (((tmp >> 6) & 0x03030303u) << 4);
        aux[0] = (a0 & 0x0f0f0f0fu) | (((tmp >> 0) & 0x03030303u) << 4);
        aux[1] = (a1 & 0x0f0f0f0fu) | (((tmp >> 2) & 0x03030303u) << 4);
        // sub-block s: 128-half n = s>>2, j = s&3 (shift 2j, mask bit 4n + j), 16-weight scales 2s and 2s+1
        const int n = s >> 2, j = s & 3;
        const uint8_t *q = b + 32 + 32 * n, *hm = b;
        const int m = 1 << (4 * n + j);
#pragma unroll
        for (int l = 0; l < 32; l++) {
            const int si = 2 * s + (l >> 4);
            const int scale = (int)(int8_t)((aux[si >> 2] >> (8 * (si & 3))) & 0xFF) - 32;
            const int v = ((q[l] >> (2 * j)) & 3) - ((hm[l] & m) ? 0 : 4);
            f[l] = d * (float)scale * (float)v;
        }
    }
}

template <int T>
__global__ __launch_bounds__(256) void kdequant_kernel(const uint8_t *__restrict__ blocks, float *__restrict__ out, int64_t n_blocks) {
    constexpr int SUB = KTraits<T>::QK / 32;
    const int64_t total = n_blocks * SUB;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float f[32];
        kdecode32<T>(blocks + (i / SUB) * KTraits<T>::TS, (int)(i % SUB), f);
        float4 *y = reinterpret_cast<float4 *>(out + i * 32);
#pragma unroll
        for (int k = 0; k < 8; k++) y[k] = make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]);
    }
}

// y[m, o] = sum_k x[m, k] * dequant(W)[o, k]; wave per output row, lane per 32-weight sub-block.
template <int T, int MT>
__global__ __launch_bounds__(256) void kgemv_kernel(const uint8_t *__restrict__ w, const float *__restrict__ x, float *__restrict__ y,
                                                     int n_out, int n_in, int m) {
    constexpr int SUB = KTraits<T>::QK / 32, TS = KTraits<T>::TS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nsub = n_in / 32, nblk = n_in / KTraits<T>::QK;
    for (int row = blockIdx.x * 4 + wave; row < n_out; row += gridDim.x * 4) {
        float acc[MT];
#pragma unroll
        for (int mi = 0; mi < MT; mi++) acc[mi] = 0.f;
        for (int g = lane; g < nsub; g += 64) {
            float f[32];
            kdecode32<T>(w + ((int64_t)row * nblk + g / SUB) * TS, g % SUB, f);
#pragma unroll
            for (int mi = 0; mi < MT; mi++) {
                if (mi < m) {
                    const float4 *xp = reinterpret_cast<const float4 *>(x + (int64_t)mi * n_in + g * 32);
                    float a = acc[mi];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const float4 v = xp[k];
                        a = fmaf(f[4 * k], v.x, a); a = fmaf(f[4 * k + 1], v.y, a); a = fmaf(f[4 * k + 2], v.z, a); a = fmaf(f[4 * k + 3], v.w, a);
                    }
                    acc[mi] = a;
                }
            }
        }
#pragma unroll
        for (int mi = 0; mi < MT; mi++) {
            const float v = vmx_wave_sum(acc[mi]);
            if (lane == 0 && mi < m) y[(int64_t)mi * n_out + row] = v;
        }
    }
}

// Q3_K x f32 vector, one activation row (the tied lm_head of a GGUF Gemma: 262144 x 3840, 432 MB): a lane owns 4
// consecutive positions l0..l0+3 of one super-block's 32-wide pattern, i.e. the 32 weights {128n + 32j + l0 + i}: they
// share one hmask dword and two qs dwords, so a block is fetched exactly once (8 lanes x 12 B + scales) and the lane's 32
// activations (and their per-group sums, for the "- 4" offset) are loaded once and stay in registers for every row.
// A wave covers 8 super-blocks per pass (P passes: rows of up to 8 P blocks) and streams its rows through a ring of D
// row slots.  sum = d * sum_{n,j} scale[n,j,hi] * (sum_i t_i x_i - 4 sum_i x_i), t = q2 | hbit << 2.
struct __attribute__((packed, aligned(2))) u32a2 { uint32_t v; };
__device__ __forceinline__ uint32_t ldw(const uint8_t *p) { return reinterpret_cast<const u32a2 *>(p)->v; }

struct Q3Raw { uint32_t hm, q0, q1, s0, s1, s2, d; };

template <int P, int D>
__global__ __launch_bounds__(256) void q3k_gemv_m1_kernel(const uint8_t *__restrict__ w, const float *__restrict__ x, float *__restrict__ y,
                                                           int n_out, int nb, int rows_per_wave) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lb = lane >> 3, l0 = (lane & 7) * 4, hi = l0 >> 4;
    float xv[P][8][4], xs[P][8];
    int boff[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int blk = lb + 8 * p;
        const bool live = blk < nb;
        boff[p] = min(blk, nb - 1) * 110;
#pragma unroll
        for (int g = 0; g < 8; g++) {   // g = 4n + j
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) v = *reinterpret_cast<const float4 *>(x + blk * 256 + 32 * g + l0);
            xv[p][g][0] = v.x; xv[p][g][1] = v.y; xv[p][g][2] = v.z; xv[p][g][3] = v.w;
            xs[p][g] = 4.0f * (v.x + v.y + v.z + v.w);
        }
    }
    const int row0 = gw * rows_per_wave;
    const int64_t row_bytes = (int64_t)nb * 110;
    auto load_row = [&](int k, Q3Raw (&r)[P]) {   // rows past the wave's range re-read its last row (cheap, never stored)
        const uint8_t *rp = w + (int64_t)min(row0 + min(k, rows_per_wave - 1), n_out - 1) * row_bytes;
#pragma unroll
        for (int p = 0; p < P; p++) {
            const uint8_t *b = rp + boff[p];
            r[p].hm = ldw(b + l0); r[p].q0 = ldw(b + 32 + l0); r[p].q1 = ldw(b + 64 + l0);
            r[p].s0 = ldw(b + 96); r[p].s1 = ldw(b + 100); r[p].s2 = ldw(b + 104);
            r[p].d = reinterpret_cast<const u16a2 *>(b + 108)->v;
        }
    };
    auto compute_row = [&](int k, const Q3Raw (&r)[P]) {
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < P; p++) {
            uint32_t aux[4];
            aux[0] = (r[p].s0 & 0x0f0f0f0fu) | (((r[p].s2 >> 0) & 0x03030303u) << 4);
            aux[1] = (r[p].s1 & 0x0f0f0f0fu) | (((r[p].s2 >> 2) & 0x03030303u) << 4);
            aux[2] = ((r[p].s0 >> 4) & 0x0f0f0f0fu) | (((r[p].s2 >> 4) & 0x03030303u) << 4);
            aux[3] = ((r[p].s1 >> 4) & 0x0f0f0f0fu) | (((r[p].s2 >> 6) & 0x03030303u) << 4);
            float part = 0.f;
#pragma unroll
            for (int g = 0; g < 8; g++) {
                const int n = g >> 2, j = g & 3;
                const uint32_t q = n ? r[p].q1 : r[p].q0;
                const uint32_t t = ((q >> (2 * j)) & 0x03030303u) | (((r[p].hm >> g) & 0x01010101u) << 2);
                // scale index 2g + hi -> word (2g + hi) >> 2 = g >> 1, byte ((2g & 3) + hi)
                const int sc = (int)((aux[g >> 1] >> (8 * (((2 * g) & 3) + hi))) & 0xFFu) - 32;
                float dot = (float)(t & 0xFFu) * xv[p][g][0];
                dot = fmaf((float)((t >> 8) & 0xFFu), xv[p][g][1], dot);
                dot = fmaf((float)((t >> 16) & 0xFFu), xv[p][g][2], dot);
                dot = fmaf((float)(t >> 24), xv[p][g][3], dot);
                part = fmaf((float)sc, dot - xs[p][g], part);
            }
            acc = fmaf(vmx_f16_to_f32((uint16_t)r[p].d), part, acc);
        }
        const float v = vmx_wave_sum(acc);
        const int row = row0 + k;
        if (lane == 0 && k < rows_per_wave && row < n_out) y[row] = v;
    };
    Q3Raw ring[D][P];
#pragma unroll
    for (int j = 0; j < D; j++) load_row(j, ring[j]);
    for (int k0 = 0; k0 < rows_per_wave; k0 += D) {
#pragma unroll
        for (int j = 0; j < D; j++) {
            compute_row(k0 + j, ring[j]);
            load_row(k0 + j + D, ring[j]);
        }
    }
}

// Q3_K table x 2..16 bf16 activation rows on the matrix cores (the tied lm_head of a GGUF Gemma at several concurrent
// sequences; the one-row kernel above would stream the 432 MB table once per row).  A weight is d * scale * (t - 4) with
// t = q2 | hbit << 2 in 0..7: bf16(t - 4) is exact and comes from two byte-table lookups (v_perm_b32), so a 16-weight group
// (one scale) is one v_mfma_f32_16x16x16_bf16 with exact products, and  d * scale  is applied to its f32 sums.  One
// 16-wave workgroup per compute unit keeps all activation rows in LDS (16 x K bf16) and every wave walks 16-row tiles of
// the table, the next super-block's bytes requested before the current one is decoded.  f32 accumulation in group order.
__global__ __launch_bounds__(1024) void q3k_gemm16_kernel(const uint8_t *__restrict__ w, const uint16_t *__restrict__ x, float *__restrict__ y,
                                                           int n_out, int nb, int M) {
    extern __shared__ __align__(16) uint16_t q3_xs[];
    typedef short q3_s4 __attribute__((ext_vector_type(4)));
    typedef float q3_f2 __attribute__((ext_vector_type(2)));
    const int K = nb * 256, XROW = K + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int cpr = K >> 3, nch = 16 * cpr;
    for (int c0 = 0; c0 < nch; c0 += 8 * 1024) {   // eight chunks per thread requested before the first is stored
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int c = c0 + i * 1024 + tid, m = c / cpr, cc = c % cpr;
            v[i] = make_uint4(0, 0, 0, 0);
            if (c < nch && m < M) v[i] = *reinterpret_cast<const uint4 *>(x + (int64_t)m * K + cc * 8);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int c = c0 + i * 1024 + tid, m = c / cpr, cc = c % cpr;
            if (c < nch) *reinterpret_cast<uint4 *>(q3_xs + m * XROW + cc * 8) = v[i];
        }
    }
    __syncthreads();
    const int tiles = (n_out + 15) >> 4, waves_total = gridDim.x * 16;
    const uint16_t *xa = q3_xs + r * XROW + 4 * kg;   // A operand: lane (activation row r, k-group kg)
    const int64_t row_bytes = (int64_t)nb * 110;
    struct Raw { uint32_t qs[2][2], hm[2], s0, s1, s2, d; };
    for (int tile = blockIdx.x * 16 + wave; tile < tiles; tile += waves_total) {
        const uint8_t *rp = w + (int64_t)min(tile * 16 + r, n_out - 1) * row_bytes;   // B operand: lane (table row r of the tile, kg)
        auto load = [&](int sb, Raw &o) {
            const uint8_t *b = rp + sb * 110;
#pragma unroll
            for (int hi = 0; hi < 2; hi++) {
                o.hm[hi] = ldw(b + 16 * hi + 4 * kg);
#pragma unroll
                for (int n = 0; n < 2; n++) o.qs[n][hi] = ldw(b + 32 + 32 * n + 16 * hi + 4 * kg);
            }
            o.s0 = ldw(b + 96); o.s1 = ldw(b + 100); o.s2 = ldw(b + 104);
            o.d = reinterpret_cast<const u16a2 *>(b + 108)->v;
        };
        q3_f2 fin_lo = {0.f, 0.f}, fin_hi = {0.f, 0.f};
        Raw nxt;
        load(0, nxt);
        for (int sb = 0; sb < nb; sb++) {
            const Raw cur = nxt;
            if (sb + 1 < nb) load(sb + 1, nxt);
            uint32_t aux[4];
            aux[0] = (cur.s0 & 0x0f0f0f0fu) | (((cur.s2 >> 0) & 0x03030303u) << 4);
            aux[1] = (cur.s1 & 0x0f0f0f0fu) | (((cur.s2 >> 2) & 0x03030303u) << 4);
            aux[2] = ((cur.s0 >> 4) & 0x0f0f0f0fu) | (((cur.s2 >> 4) & 0x03030303u) << 4);
            aux[3] = ((cur.s1 >> 4) & 0x0f0f0f0fu) | (((cur.s2 >> 6) & 0x03030303u) << 4);
            const float dd = vmx_f16_to_f32((uint16_t)cur.d), dd32 = -32.0f * dd;
            const uint16_t *xsb = xa + sb * 256;
#pragma unroll
            for (int n = 0; n < 2; n++) {   // eight independent MFMAs in flight, then their eight scalings: no wait per group
                vmx_f32x4 acc[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {   // u = 2 j + hi; g = 4 n + j: weights 128 n + 32 j + 16 hi + (0..15), scale 2 g + hi
                    const int j = u >> 1, hi = u & 1, g = 4 * n + j;
                    const uint32_t q4 = (cur.qs[n][hi] >> (2 * j)) & 0x03030303u;
                    const uint32_t t4 = q4 | (((cur.hm[hi] >> g) & 0x01010101u) << 2);
                    const uint32_t hib = __builtin_amdgcn_perm(0x40403F00u, 0xBFC0C0C0u, t4);   // high / low bytes of bf16(t - 4)
                    const uint32_t lob = __builtin_amdgcn_perm(0x40008000u, 0x80004080u, t4);
                    const uint2 bw = make_uint2(__builtin_amdgcn_perm(hib, lob, 0x05010400u), __builtin_amdgcn_perm(hib, lob, 0x07030602u));
                    const uint2 aw = *reinterpret_cast<const uint2 *>(xsb + 32 * g + 16 * hi);
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(q3_s4, aw), __builtin_bit_cast(q3_s4, bw),
                                                                       vmx_f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int j = u >> 1, hi = u & 1, g = 4 * n + j;
                    // d * (scale - 32) is exact in f32 (11-bit x 6-bit), so byte -> float -> one fma gives the same bits
                    const float fsc = fmaf((float)((aux[g >> 1] >> (8 * (((2 * g) & 3) + hi))) & 0xFFu), dd, dd32);
                    const q3_f2 f2 = {fsc, fsc};
                    fin_lo = __builtin_elementwise_fma(f2, acc[u].lo, fin_lo);   // v_pk_fma_f32
                    fin_hi = __builtin_elementwise_fma(f2, acc[u].hi, fin_hi);
                }
            }
        }
        const int col = tile * 16 + r;   // lane holds activation rows 4 kg + i of table row `col`
        if (col < n_out) {
            const float fin[4] = {fin_lo.x, fin_lo.y, fin_hi.x, fin_hi.y};
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (4 * kg + i < M) y[(int64_t)(4 * kg + i) * n_out + col] = fin[i];
        }
    }
}

// out[t, :] = bf16(bf16(dequant(table row ids[t])) * scale): QuantizedEmbedding::forward (gather the row's blocks,
// dequantise, cast to the model dtype; crates/core/src/quantization/gguf/mod.rs:546-600) followed by the model's
// embedding normaliser in bf16.  Out-of-range ids give a zero row.  One thread per 32-weight sub-block.
template <int T>
__global__ __launch_bounds__(256) void kgather_bf16_kernel(const uint8_t *__restrict__ table, const int32_t *__restrict__ ids,
                                                            uint16_t *__restrict__ out, int tokens, int dim, int vocab, float scale) {
    constexpr int SUB = KTraits<T>::QK / 32;
    const int subs_per_row = dim / 32;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= tokens * subs_per_row) return;
    const int t = i / subs_per_row, g = i % subs_per_row;
    const int id = ids[t];
    float f[32];
    if (id >= 0 && id < vocab) {
        kdecode32<T>(table + ((int64_t)id * (dim / KTraits<T>::QK) + g / SUB) * KTraits<T>::TS, g % SUB, f);
    } else {
#pragma unroll
        for (int k = 0; k < 32; k++) f[k] = 0.f;
    }
    uint16_t *y = out + (int64_t)t * dim + g * 32;
#pragma unroll
    for (int k = 0; k < 32; k++) y[k] = vmx_f32_to_bf16(vmx_bf16_to_f32(vmx_f32_to_bf16(f[k])) * scale);
}

}  // namespace

#define K_DISPATCH(type, CALL)                                                                  \
    switch (type) {                                                                             \
        case 2: CALL(2); break;                                                                 \
        case 3: CALL(3); break;                                                                 \
        case 6: CALL(6); break;                                                                 \
        case 7: CALL(7); break;                                                                 \
        case 8: CALL(8); break;                                                                 \
        case 10: CALL(10); break;                                                               \
        case 11: CALL(11); break;                                                               \
        case 12: CALL(12); break;                                                               \
        case 13: CALL(13); break;                                                               \
        case 14: CALL(14); break;                                                               \
        default:                                                                                \
            vmx_set_error("unsupported ggml type id %d (supported: 2 Q4_0, 3 Q4_1, 6 Q5_0, 7 Q5_1, 8 Q8_0, 10 Q2_K, 11 Q3_K, 12 Q4_K, 13 Q5_K, 14 Q6_K)", type); \
            return VMX_EUNSUPPORTED;                                                            \
    }

extern "C" int vmx_ggml_type_size(int t) {
    switch (t) {
        case 2: return 18; case 3: return 20; case 6: return 22; case 7: return 24; case 8: return 34;
        case 10: return 84; case 11: return 110; case 12: return 144; case 13: return 176; case 14: return 210;
        default: return vmx_iq_type_size(t);
    }
}

extern "C" int vmx_ggml_block_elems(int t) { return (t == 2 || t == 3 || t == 6 || t == 7 || t == 8) ? 32 : 256; }

extern "C" int vmx_kquant_dequantize_f32(int ggml_type, const void *blocks, float *out, int64_t n_blocks, void *stream) {
    VMX_REQUIRE(blocks && out && n_blocks >= 0, "kquant_dequantize: bad arguments");
    if (n_blocks == 0) return VMX_OK;
    int64_t g = (n_blocks * 8 + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
#define CALL(T) hipLaunchKernelGGL((kdequant_kernel<T>), dim3((unsigned)g), dim3(256), 0, vmx_stream(stream), (const uint8_t *)blocks, out, n_blocks)
    K_DISPATCH(ggml_type, CALL)
#undef CALL
    VMX_CHECK_LAUNCH("kquant_dequantize");
    return VMX_OK;
}

extern "C" int vmx_kquant_gemv_f32(int ggml_type, const void *w, const float *x, float *y, int n_out, int n_in, int m, void *stream) {
    VMX_REQUIRE(w && x && y, "kquant_gemv: null pointer");
    VMX_REQUIRE(m >= 1 && m <= 8, "kquant_gemv: m (%d) must be in 1..8 (larger m takes the dequant + GEMM path)", m);
    VMX_REQUIRE(n_out > 0 && n_in > 0 && n_in % vmx_ggml_block_elems(ggml_type) == 0, "kquant_gemv: n_in (%d) not a multiple of the block size",
                n_in);
    hipStream_t st = vmx_stream(stream);
    if (ggml_type == 11 && n_in <= 4096 && (m == 1 || n_out >= 65536)) {
        // lm_head-sized Q3_K: the one-row kernel (activations resident in registers, every weight byte fetched once), one launch
        // per activation row -- for a vocabulary-sized table m passes of it are several times faster than the generic m-row kernel
        const int nb = n_in / 256;
        int g = (n_out + 3) / 4;
        const int cap = vmx_num_cus() * 3;
        if (g > cap) g = cap;
        const int rpw = (n_out + g * 4 - 1) / (g * 4);
        g = (n_out + rpw * 4 - 1) / (rpw * 4);
        for (int r = 0; r < m; r++) {
            const float *xr = x + (int64_t)r * n_in;
            float *yr = y + (int64_t)r * n_out;
            if (nb <= 8) hipLaunchKernelGGL((q3k_gemv_m1_kernel<1, 3>), dim3(g), dim3(256), 0, st, (const uint8_t *)w, xr, yr, n_out, nb, rpw);
            else hipLaunchKernelGGL((q3k_gemv_m1_kernel<2, 3>), dim3(g), dim3(256), 0, st, (const uint8_t *)w, xr, yr, n_out, nb, rpw);
        }
        VMX_CHECK_LAUNCH("kquant_gemv(q3_k)");
        return VMX_OK;
    }
    int grid = (n_out + 3) / 4;
    if (grid > 8192) grid = 8192;
#define CALL(T)                                                                                                                  \
    do {                                                                                                                         \
        if (m == 1) hipLaunchKernelGGL((kgemv_kernel<T, 1>), dim3(grid), dim3(256), 0, st, (const uint8_t *)w, x, y, n_out, n_in, m); \
        else if (m <= 4) hipLaunchKernelGGL((kgemv_kernel<T, 4>), dim3(grid), dim3(256), 0, st, (const uint8_t *)w, x, y, n_out, n_in, m); \
        else hipLaunchKernelGGL((kgemv_kernel<T, 8>), dim3(grid), dim3(256), 0, st, (const uint8_t *)w, x, y, n_out, n_in, m);   \
    } while (0)
    K_DISPATCH(ggml_type, CALL)
#undef CALL
    VMX_CHECK_LAUNCH("kquant_gemv");
    return VMX_OK;
}

// ---- internal (C++) entry point used by the Gemma 4 decoder
int vmx_kquant_gather_scaled_bf16(int ggml_type, const void *table, const int32_t *ids, uint16_t *out, int tokens, int dim, int vocab,
                                  float scale, hipStream_t st) {
    if (dim <= 0 || dim % vmx_ggml_block_elems(ggml_type) != 0) {
        vmx_set_error("kquant_gather: dim %d not a multiple of the block size", dim);
        return VMX_EINVAL;
    }
    const int n = tokens * (dim / 32);
#define CALL(T) hipLaunchKernelGGL((kgather_bf16_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, st, (const uint8_t *)table, ids, out, tokens, dim, vocab, scale)
    K_DISPATCH(ggml_type, CALL)
#undef CALL
    VMX_CHECK_LAUNCH("kquant_gather");
    return VMX_OK;
}

// y [m][n_out] f32 = x [m][n_in] (bf16) x dequant(W)^T for a Q3_K table, 2..16 rows, n_in <= 4096 (the activation rows stay in
// LDS): see q3k_gemm16_kernel.  Internal (the Gemma 4 executor's tied lm_head at several concurrent sequences).
bool vmx_q3k_gemm16_ok(int n_out, int n_in, int m) { return m >= 2 && m <= 16 && n_in % 256 == 0 && n_in <= 4096 && n_out >= 16 * vmx_num_cus(); }
int vmx_q3k_gemm16_bf16(const void *w, const uint16_t *x, float *y, int n_out, int n_in, int m, hipStream_t st) {
    VMX_REQUIRE(w && x && y && m >= 1 && m <= 16 && n_out >= 1 && n_in >= 256 && n_in % 256 == 0 && n_in <= 4096,
                "q3k_gemm16: shape [%d x %d -> %d] not supported (1..16 rows, n_in a multiple of 256 up to 4096)", m, n_in, n_out);
    const int lds = 16 * (n_in + 16) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        VMX_CHECK_HIP(hipFuncSetAttribute((const void *)q3k_gemm16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * (4096 + 16) * 2));
        attr_set = true;
    }
    hipLaunchKernelGGL(q3k_gemm16_kernel, dim3(vmx_num_cus()), dim3(1024), lds, st, (const uint8_t *)w, x, y, n_out, n_in / 256, m);
    VMX_CHECK_LAUNCH("q3k_gemm16");
    return VMX_OK;
}

extern "C" int vmx_q3k_rows16_bf16(const void *w, const uint16_t *x, float *y, int n_out, int n_in, int m, void *stream) {
    return vmx_q3k_gemm16_bf16(w, x, y, n_out, n_in, m, (hipStream_t)stream);
}

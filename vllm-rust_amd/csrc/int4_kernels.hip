// AWQ / GPTQ int4 kernels for gfx950: load-time repacks and the small-M
// dequant-fused GEMM that replaces the reference's awq_gemv_* / marlin_gemm_int4*
// kernels (crates/core/kernels/awq_gemv.cu:46-665, marlin_gemm.cu:67-187,301-430).
//
// Arithmetic contract (awq_gemv.cu:70-117): y[m,n] = bf16(sum_k x[m,k] * ((nib - zp) * s) + bias),
// bf16 x and s, f32 accumulation.  This kernel evaluates the same sum re-associated per
// quantisation group on the matrix cores:
//     sum_k x*(nib - zp)*s = s * ( sum_k x*(128 + nib)  -  (128 + zp) * sum_k x )
// 0x4300|nib is the bf16 value 128+nib exactly, so a packed word unpacks to an MFMA
// B-fragment with one and-or per pair of weights; products are exact in f32 and only the
// summation order differs from the reference.
//
// HBM layout ("tile stream", built once at load by vmx_int4_pack):
//   W  [N/16][K/128][64][4] u32 : for column tile nt and k-block kb one contiguous 1 KiB unit;
//       lane l of a wave reads its dwordx4 at l*16.  Word i of lane l holds the 8 weights
//       k = kb*128 + 32*i + 8*(l>>4) + j (j = 0..7) of column nt*16 + (l&15), weight j at
//       nibble position {0,4,1,5,2,6,3,7}[j], so (w >> 4t) & 0x000F000F yields weights 2t, 2t+1.
//   SZ [N/16][G][16] u32 : low half = bf16 scale bits, high half = bf16 bits of 128 + zp.
// One wave-load is therefore 1 KiB contiguous = 16 columns x one 128-deep group slice,
// and feeds four v_mfma_f32_16x16x32_bf16 (A = activations, rows = m).
#include "vmx_kernels.h"

namespace {

constexpr int KB = 128;     // k-values per unit
constexpr int NT = 16;      // columns per unit

__host__ __device__ inline size_t sz_offset_bytes(int K, int N) { return ((size_t)K * N / 2 + 255) & ~(size_t)255; }

// ---------------------------------------------------------------- repacks
// AWQ [K, N/8] (interleaved along N) -> GPTQ-sequential [K/8, N] (awq_marlin.rs:170-240).
__global__ __launch_bounds__(256) void awq_to_gptq_kernel(const uint32_t *__restrict__ awq, uint32_t *__restrict__ out, int K,
                                                           int N, int ldo) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [K/8, N]
    if (idx >= (int64_t)(K / 8) * N) return;
    const int n = (int)(idx % N), kb = (int)(idx / N);
    const int shift = ((n & 1) * 16) + ((n & 7) >> 1) * 4;  // {0,16,4,20,8,24,12,28}[n%8]
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t src = awq[(int64_t)(kb * 8 + i) * (N / 8) + (n >> 3)];
        word |= ((src >> shift) & 0xFu) << (4 * i);
    }
    out[(int64_t)kb * ldo + n] = word;
}

// rows x cols words; out row stride ldo (for writing into a wider, merged tensor)
__global__ __launch_bounds__(256) void awq_nibble_reorder_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                                  int64_t n, int cols, int ldo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = in[i];
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int shift = ((j & 1) * 16) + (j >> 1) * 4;
        r |= ((w >> shift) & 0xFu) << (4 * j);
    }
    out[(i / cols) * ldo + (i % cols)] = r;
}

__global__ __launch_bounds__(256) void int4_pack_w_kernel(const uint32_t *__restrict__ qweight, uint32_t *__restrict__ W, int K,
                                                           int N) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over K*N/8 output words
    if (idx >= (int64_t)K * N / 8) return;
    const int i = (int)(idx & 3), l = (int)((idx >> 2) & 63);
    const int64_t unit = idx >> 8;
    const int kblocks = K / KB;
    const int kb = (int)(unit % kblocks), nt = (int)(unit / kblocks);
    const int n = nt * NT + (l & 15);
    const int c = 4 * i + (l >> 4);  // 8-wide k chunk inside the unit
    const uint32_t src = qweight[(int64_t)(kb * 16 + c) * N + n];
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int pos = (j >> 1) + 4 * (j & 1);  // {0,4,1,5,2,6,3,7}[j]
        r |= ((src >> (4 * j)) & 0xFu) << (4 * pos);
    }
    W[idx] = r;
}

__global__ __launch_bounds__(256) void int4_pack_sz_kernel(const uint16_t *__restrict__ scales, const uint32_t *__restrict__ qzeros,
                                                            uint32_t *__restrict__ SZ, int G, int N) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [N/16][G][16]
    if (idx >= (int64_t)G * N) return;
    const int nl = (int)(idx & 15);
    const int g = (int)((idx >> 4) % G), nt = (int)((idx >> 4) / G);
    const int n = nt * NT + nl;
    uint32_t zp = 8;
    if (qzeros) zp = (qzeros[(int64_t)g * (N / 8) + (n >> 3)] >> (4 * (n & 7))) & 0xFu;
    SZ[idx] = (uint32_t)scales[(int64_t)g * N + n] | ((0x4300u | zp) << 16);
}

// tile stream -> dense bf16 [K, N]; value bf16((nib - zp) * scale) as awq_marlin_dequant.cu:35-69.
__global__ __launch_bounds__(256) void int4_dequant_kernel(const uint32_t *__restrict__ W, const uint32_t *__restrict__ SZ,
                                                            uint16_t *__restrict__ dense, int K, int N, int group) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)K * N / 8) return;
    const int i = (int)(idx & 3), l = (int)((idx >> 2) & 63);
    const int64_t unit = idx >> 8;
    const int kblocks = K / KB, G = K / group;
    const int kb = (int)(unit % kblocks), nt = (int)(unit / kblocks);
    const int n = nt * NT + (l & 15);
    const int k0 = kb * KB + 32 * i + 8 * (l >> 4);
    const uint32_t sz = SZ[((int64_t)nt * G + k0 / group) * 16 + (l & 15)];
    const float s = vmx_bf16_to_f32((uint16_t)(sz & 0xFFFFu));
    const int zp = (int)((sz >> 16) & 0xFu);
    const uint32_t w = W[idx];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int pos = (j >> 1) + 4 * (j & 1);
        const int nib = (int)((w >> (4 * pos)) & 0xFu);
        dense[(int64_t)(k0 + j) * N + n] = vmx_f32_to_bf16((float)(nib - zp) * s);
    }
}

// ---------------------------------------------------------------- small-M GEMM
// Work item = (split-K slice ks, column tile nt): 16 columns x (4*UPW) k-blocks; the 4 waves of a
// workgroup own UPW consecutive k-blocks each and sum their results through LDS.  A workgroup keeps
// ks fixed and walks the tiles nt = first, first + tile_stride, ...: the activations of its K slice
// are staged once, and the weight units of item j+1 are requested before item j is computed, so
// HBM requests stay in flight under the unpack/MFMA work (register double buffer, static indices).
//
// The activation source is a template parameter, so that the small vector operators that sit
// between two linears in a decoder layer run in the prologue of the consuming linear instead of
// in launches of their own:
//   X_BF16   x is a bf16 tensor [M, K].
//   X_ACT    x = bf16(silu(bf16(g)) * bf16(u)), g|u being the f32 split-K slices of a merged gate|up
//            linear, row layout [2*K] (silu_and_mul_separate_bf16 semantics).
//   X_NORM   h = bf16(h + bf16(sum of f32 split-K slices of the previous linear)) (or h as is),
//            x = bf16(h * rsqrt(mean(h^2) + eps) * w): add_bf16 followed by rms_norm_bf16.  Every
//            workgroup reduces the whole row (K == hidden) for the variance; workgroups with the
//            first tile write their K slice of the updated h back.
//   X_ATTN   x = merged paged-attention partitions (paged_attention_v2_reduce semantics) of the
//            f32 per-partition outputs, k = head * head_dim + d.
enum XMode { X_BF16 = 0, X_ACT = 1, X_NORM = 2, X_ATTN = 3 };

struct XSource {
    const uint16_t *x;        // X_BF16: [M, K]
    const float *part;        // X_ACT / X_NORM: f32 slices [splits][M][row]; X_ATTN: tmp_out [M][H][P][hd]
    int part_splits;          // slices in `part`
    uint16_t *h;              // X_NORM: residual stream [M, K], updated in place when part != NULL
    const uint16_t *norm_w;   // X_NORM: [K]
    float eps;                // X_NORM
    const float *exp_sums;    // X_ATTN: [M][H][P]
    const float *max_logits;  // X_ATTN
    const int32_t *seq_lens;  // X_ATTN: [M]
    int head_dim, max_parts, window, part_tokens;  // X_ATTN
};

constexpr int SPLIT_MAX = 8;

__device__ __forceinline__ float rbf(float v) { return vmx_bf16_to_f32(vmx_f32_to_bf16(v)); }

__device__ __forceinline__ void bf16x8_to_f32(const uint4 v, float f[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f[2 * j] = __uint_as_float(w[j] << 16);
        f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
    }
}

__device__ __forceinline__ uint4 f32_to_bf16x8(const float f[8]) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = (uint32_t)vmx_f32_to_bf16(f[2 * j]) | ((uint32_t)vmx_f32_to_bf16(f[2 * j + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Sum of up to SPLIT_MAX f32 slices, 8 consecutive elements; all loads independent.
__device__ __forceinline__ void slices8(const float *__restrict__ p, int splits, int64_t stride, int64_t idx, float out[8]) {
    float4 a[SPLIT_MAX], b[SPLIT_MAX];
#pragma unroll
    for (int s = 0; s < SPLIT_MAX; s++)
        if (s < splits) {
            a[s] = *reinterpret_cast<const float4 *>(p + s * stride + idx);
            b[s] = *reinterpret_cast<const float4 *>(p + s * stride + idx + 4);
        }
#pragma unroll
    for (int j = 0; j < 8; j++) out[j] = 0.f;
#pragma unroll
    for (int s = 0; s < SPLIT_MAX; s++)
        if (s < splits) {
            out[0] += a[s].x; out[1] += a[s].y; out[2] += a[s].z; out[3] += a[s].w;
            out[4] += b[s].x; out[5] += b[s].y; out[6] += b[s].z; out[7] += b[s].w;
        }
}

// Stage the workgroup's K slice [k0, k0 + KS) of the M activation rows into xs (bf16, row stride
// XROW) and the per-(k-block, row) sums into sumx.  Ends with a barrier.
template <int XMODE, int KS, int XROW>
__device__ __forceinline__ void stage_x(const XSource &src, uint16_t *xs, float *sumx, float *red, int M, int K, int k0,
                                        bool writes_h) {
    constexpr int CPR = KS / 8;
    const int tid = threadIdx.x;
    if constexpr (XMODE == X_NORM) {
        // pass A: whole row -> updated h, variance; the slice's h values parked in xs as bf16
        const int nchunk = K >> 3;
        for (int m = 0; m < M; m++) {
            float var = 0.f;
            for (int c = tid; c < nchunk; c += 256) {
                const int k = c * 8;
                float hv[8];
                bf16x8_to_f32(*reinterpret_cast<const uint4 *>(src.h + (int64_t)m * K + k), hv);
                if (src.part) {
                    float o[8];
                    slices8(src.part, src.part_splits, (int64_t)M * K, (int64_t)m * K + k, o);
#pragma unroll
                    for (int j = 0; j < 8; j++) hv[j] = rbf(hv[j] + rbf(o[j]));
                }
#pragma unroll
                for (int j = 0; j < 8; j++) var += hv[j] * hv[j];
                if (k >= k0 && k < k0 + KS) {
                    const uint4 packed = f32_to_bf16x8(hv);
                    *reinterpret_cast<uint4 *>(xs + m * XROW + (k - k0)) = packed;
                    if (writes_h && src.part) *reinterpret_cast<uint4 *>(src.h + (int64_t)m * K + k) = packed;
                }
            }
            var = vmx_block_sum_256(var, red);
            if (tid == 0) red[4 + m] = rsqrtf(var / (float)K + src.eps);
        }
        __syncthreads();
        // pass B: normalise the slice in place
        for (int c = tid; c < M * CPR; c += 256) {
            const int m = c / CPR, cc = c % CPR, k = k0 + cc * 8;
            if (k < K) {
                float hv[8], wv[8];
                bf16x8_to_f32(*reinterpret_cast<const uint4 *>(xs + m * XROW + cc * 8), hv);
                bf16x8_to_f32(*reinterpret_cast<const uint4 *>(src.norm_w + k), wv);
                const float inv = red[4 + m];
#pragma unroll
                for (int j = 0; j < 8; j++) hv[j] = hv[j] * inv * wv[j];
                *reinterpret_cast<uint4 *>(xs + m * XROW + cc * 8) = f32_to_bf16x8(hv);
            } else {
                *reinterpret_cast<uint4 *>(xs + m * XROW + cc * 8) = make_uint4(0, 0, 0, 0);
            }
        }
    } else {
        for (int c = tid; c < M * CPR; c += 256) {
            const int m = c / CPR, cc = c % CPR, k = k0 + cc * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (k < K) {
                if constexpr (XMODE == X_BF16) {
                    v = *reinterpret_cast<const uint4 *>(src.x + (int64_t)m * K + k);
                } else if constexpr (XMODE == X_ACT) {
                    float g[8], u[8];
                    slices8(src.part, src.part_splits, (int64_t)M * 2 * K, (int64_t)m * 2 * K + k, g);
                    slices8(src.part, src.part_splits, (int64_t)M * 2 * K, (int64_t)m * 2 * K + K + k, u);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const float gg = rbf(g[j]);
                        g[j] = (gg / (1.0f + expf(-gg))) * rbf(u[j]);
                    }
                    v = f32_to_bf16x8(g);
                } else {  // X_ATTN
                    const int hd = src.head_dim, H = K / hd, head = k / hd, d = k % hd;
                    const int seq_len = src.seq_lens[m];
                    const int eff = (src.window > 0 && seq_len > src.window) ? src.window : seq_len;
                    const int np = (eff + src.part_tokens - 1) / src.part_tokens;
                    const int64_t mo = ((int64_t)m * H + head) * src.max_parts;
                    const float *tp = src.part + mo * hd + d;
                    float o[8];
                    if (np <= 1) {
                        const float4 a = *reinterpret_cast<const float4 *>(tp), b = *reinterpret_cast<const float4 *>(tp + 4);
                        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
                        if (seq_len == 0)
#pragma unroll
                            for (int j = 0; j < 8; j++) o[j] = 0.f;
                    } else {
                        float gmax = -3.402823466e+38f;
                        for (int p = 0; p < np; p++) gmax = fmaxf(gmax, src.max_logits[mo + p]);
                        float gsum = 0.f;
#pragma unroll
                        for (int j = 0; j < 8; j++) o[j] = 0.f;
                        for (int p = 0; p < np; p++) {
                            const float w = src.exp_sums[mo + p] * __expf(src.max_logits[mo + p] - gmax);
                            gsum += w;
                            const float4 a = *reinterpret_cast<const float4 *>(tp + (int64_t)p * hd);
                            const float4 b = *reinterpret_cast<const float4 *>(tp + (int64_t)p * hd + 4);
                            o[0] += a.x * w; o[1] += a.y * w; o[2] += a.z * w; o[3] += a.w * w;
                            o[4] += b.x * w; o[5] += b.y * w; o[6] += b.z * w; o[7] += b.w * w;
                        }
                        const float inv = 1.0f / (gsum + 1e-6f);
#pragma unroll
                        for (int j = 0; j < 8; j++) o[j] *= inv;
                    }
                    v = f32_to_bf16x8(o);
                }
            }
            *reinterpret_cast<uint4 *>(xs + m * XROW + cc * 8) = v;
        }
    }
    __syncthreads();
    // per-(k-block, row) sums of the staged bf16 values: 16 consecutive chunks = one (row, k-block)
    for (int c0 = 0; c0 < M * CPR; c0 += 256) {
        const int c = c0 + tid;
        float s = 0.f;
        int m = 0, cc = 0;
        if (c < M * CPR) {
            m = c / CPR;
            cc = c % CPR;
            float f[8];
            bf16x8_to_f32(*reinterpret_cast<const uint4 *>(xs + m * XROW + cc * 8), f);
#pragma unroll
            for (int j = 0; j < 8; j++) s += f[j];
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        if (c < M * CPR && (tid & 15) == 0) sumx[(cc >> 4) * 16 + m] = s;
    }
    __syncthreads();
}

template <int UPW, int XMODE>
__global__ __launch_bounds__(256) void int4_gemm_kernel(const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ, XSource src,
                                                         const uint16_t *__restrict__ bias, uint16_t *__restrict__ y,
                                                         float *__restrict__ partial, int M, int N, int K, int group, int splits,
                                                         int tile_workers) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int KS = 4 * UPW * KB;       // k-values per workgroup
    constexpr int XROW = KS + 8;           // bf16 per staged row (+16 B pad against bank conflicts)
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);                     // [M][XROW]
    float *sumx = reinterpret_cast<float *>(smem + (size_t)M * XROW * 2);  // [4*UPW][16]
    float *red = sumx + 4 * UPW * 16;                                      // [2][4][64][4] cross-wave sums (double buffered)
    float *misc = red + 2 * 4 * 64 * 4;                                    // [4 + 16] block-reduce scratch, per-row 1/rms

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = N / NT, kblocks = K / KB, G = K / group;
    const int ks = blockIdx.x / tile_workers, first_tile = blockIdx.x % tile_workers;
    const int kb_wg = ks * 4 * UPW;        // first k-block of this workgroup
    const int nl = lane & 15, h = lane >> 4;
    const int n_items = (n_tiles - first_tile + tile_workers - 1) / tile_workers;

    uint4 wbuf[2][UPW];
    uint32_t szbuf[2][UPW];
    auto request = [&](int which, int nt) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UPW; u++) {
            const int kb = kb_wg + wave * UPW + u;
            if (kb < kblocks) {
                wbuf[which][u] = W[((int64_t)nt * kblocks + kb) * 64 + lane];
                szbuf[which][u] = SZ[((int64_t)nt * G + (kb * KB) / group) * 16 + nl];
            } else {
                wbuf[which][u] = make_uint4(0, 0, 0, 0);
                szbuf[which][u] = 0;
            }
        }
    };
    request(0, first_tile);

    stage_x<XMODE, KS, XROW>(src, xs, sumx, misc, M, K, kb_wg * KB, first_tile == 0);

    const bool arow = nl < M;
    auto compute = [&](int which, int nt, int parity) __attribute__((always_inline)) {
        vmx_f32x4 fin = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UPW; u++) {
            const int kbl = wave * UPW + u;  // k-block inside the workgroup slice
            if (kb_wg + kbl < kblocks) {     // wave-uniform
                vmx_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const uint32_t w4[4] = {wbuf[which][u].x, wbuf[which][u].y, wbuf[which][u].z, wbuf[which][u].w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t w = w4[i];
                    uint4 bu;
                    bu.x = (w & 0x000F000Fu) | 0x43004300u;
                    bu.y = ((w >> 4) & 0x000F000Fu) | 0x43004300u;
                    bu.z = ((w >> 8) & 0x000F000Fu) | 0x43004300u;
                    bu.w = ((w >> 12) & 0x000F000Fu) | 0x43004300u;
                    uint4 au = make_uint4(0, 0, 0, 0);
                    if (arow) au = *reinterpret_cast<const uint4 *>(xs + nl * XROW + kbl * KB + 32 * i + 8 * h);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, au),
                                                                  __builtin_bit_cast(vmx_bf16x8, bu), acc, 0, 0, 0);
                }
                const float s = __uint_as_float(szbuf[which][u] << 16);
                const float zo = __uint_as_float(szbuf[which][u] & 0xFFFF0000u);
                const float4 sx = *reinterpret_cast<const float4 *>(sumx + kbl * 16 + 4 * h);
                fin[0] += s * (acc[0] - zo * sx.x);
                fin[1] += s * (acc[1] - zo * sx.y);
                fin[2] += s * (acc[2] - zo * sx.z);
                fin[3] += s * (acc[3] - zo * sx.w);
            }
        }
        // sum the 4 waves.  Lane holds rows m = 4h + r, column n = nt*16 + nl.
        float *rb = red + parity * (4 * 64 * 4);
        *reinterpret_cast<float4 *>(rb + (wave * 64 + lane) * 4) = make_float4(fin[0], fin[1], fin[2], fin[3]);
        __syncthreads();
        if (wave == 0) {
            float4 a = *reinterpret_cast<const float4 *>(rb + lane * 4);
#pragma unroll
            for (int wv = 1; wv < 4; wv++) {
                const float4 b = *reinterpret_cast<const float4 *>(rb + (wv * 64 + lane) * 4);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            const float out4[4] = {a.x, a.y, a.z, a.w};
            const int n = nt * NT + nl;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int mrow = 4 * h + r;
                if (mrow < M) {
                    if (y) {
                        float v = out4[r];
                        if (bias) v += vmx_bf16_to_f32(bias[n]);
                        y[(int64_t)mrow * N + n] = vmx_f32_to_bf16(v);
                    } else {
                        partial[((int64_t)ks * M + mrow) * N + n] = out4[r];
                    }
                }
            }
        }
    };

    for (int j = 0; j < n_items; j += 2) {
        const int nt0 = first_tile + j * tile_workers;
        if (j + 1 < n_items) request(1, nt0 + tile_workers);
        compute(0, nt0, 0);
        if (j + 1 < n_items) {
            if (j + 2 < n_items) request(0, nt0 + 2 * tile_workers);
            compute(1, nt0 + tile_workers, 1);
        }
    }
}

__global__ __launch_bounds__(256) void int4_reduce_kernel(const float *__restrict__ partial, const uint16_t *__restrict__ bias,
                                                           uint16_t *__restrict__ y, int M, int N, int splits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    float v = 0.f;
    for (int s = 0; s < splits; s++) v += partial[(int64_t)s * M * N + i];
    if (bias) v += vmx_bf16_to_f32(bias[i % N]);
    y[i] = vmx_f32_to_bf16(v);
}

struct GemmPlan { int upw; int splits; int tile_workers; };

// Units per wave (-> k-blocks per workgroup and split-K slices) and tile workers per slice.  Aim:
// about four workgroups per CU, two or more items per workgroup, at most SPLIT_MAX slices.
inline GemmPlan plan_gemm(int N, int K) {
    const int n_tiles = N / NT, kblocks = K / KB;
    auto splits_for = [&](int upw) { return (kblocks + 4 * upw - 1) / (4 * upw); };
    int upw = 4;
    while (upw > 1 && ((int64_t)n_tiles * splits_for(upw) < 1536 || 4 * upw > kblocks)) upw >>= 1;
    while (upw < 8 && splits_for(upw) > SPLIT_MAX) upw <<= 1;
    GemmPlan p;
    p.upw = upw;
    p.splits = splits_for(upw);
    const int64_t items = (int64_t)n_tiles * p.splits;
    int per_wg = (int)((items + 512) / 1024);
    if (per_wg < 1) per_wg = 1;
    p.tile_workers = (n_tiles + per_wg - 1) / per_wg;
    return p;
}

inline size_t gemm_lds_bytes(int M, int upw) {
    return (size_t)M * (4 * upw * KB + 8) * 2 + (size_t)4 * upw * 16 * 4 + 2 * 4 * 64 * 4 * 4 + (4 + 16) * 4;
}

template <int XMODE>
int launch_gemm(const GemmPlan &p, const uint4 *W, const uint32_t *SZ, const XSource &src, const uint16_t *bias, uint16_t *y,
                float *partial, int M, int N, int K, int group, hipStream_t st) {
    const size_t lds = gemm_lds_bytes(M, p.upw);
    if (lds > 160 * 1024) {
        vmx_set_error("int4_gemm: LDS need %zu B exceeds 160 KiB", lds);
        return VMX_EINVAL;
    }
    const dim3 grid((unsigned)(p.tile_workers * p.splits)), block(256);
#define LAUNCH(U)                                                                                                              \
    do {                                                                                                                       \
        if (lds > 64 * 1024)                                                                                                   \
            VMX_CHECK_HIP(hipFuncSetAttribute((const void *)int4_gemm_kernel<U, XMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              (int)lds));                                                                      \
        hipLaunchKernelGGL((int4_gemm_kernel<U, XMODE>), grid, block, lds, st, W, SZ, src, bias, y, partial, M, N, K, group, p.splits, \
                           p.tile_workers);                                                                                    \
    } while (0)
    switch (p.upw) {
        case 8: LAUNCH(8); break;
        case 4: LAUNCH(4); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(1); break;
    }
#undef LAUNCH
    VMX_CHECK_LAUNCH("int4_gemm");
    return VMX_OK;
}

}  // namespace

extern "C" int vmx_awq_to_gptq_qweight(const uint32_t *awq, uint32_t *out, int K, int N, void *stream) {
    VMX_REQUIRE(awq && out, "awq_to_gptq_qweight: null pointer");
    VMX_REQUIRE(K > 0 && K % 8 == 0, "awq_to_gptq_qweight: in_features (%d) must be divisible by 8", K);
    VMX_REQUIRE(N > 0 && N % 8 == 0, "awq_to_gptq_qweight: out_features (%d) must be divisible by 8", N);
    const int64_t total = (int64_t)(K / 8) * N;
    hipLaunchKernelGGL(awq_to_gptq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vmx_stream(stream), awq, out, K, N, N);
    VMX_CHECK_LAUNCH("awq_to_gptq_qweight");
    return VMX_OK;
}

extern "C" int vmx_awq_repack_nibbles(const uint32_t *in, uint32_t *out, int64_t n_words, void *stream) {
    VMX_REQUIRE(in && out && n_words >= 0 && n_words < ((int64_t)1 << 31), "awq_repack_nibbles: bad arguments");
    if (n_words == 0) return VMX_OK;
    hipLaunchKernelGGL(awq_nibble_reorder_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, vmx_stream(stream), in, out,
                       n_words, (int)n_words, (int)n_words);
    VMX_CHECK_LAUNCH("awq_repack_nibbles");
    return VMX_OK;
}

static int check_int4_shape(const char *who, int K, int N, int group) {
    VMX_REQUIRE(K > 0 && K % KB == 0, "%s: K (%d) must be a multiple of 128", who, K);
    VMX_REQUIRE(N > 0 && N % NT == 0, "%s: N (%d) must be a multiple of 16", who, N);
    VMX_REQUIRE(group > 0 && group % KB == 0 && K % group == 0, "%s: group_size (%d) must be a multiple of 128 dividing K (%d)", who,
                group, K);
    return VMX_OK;
}

extern "C" size_t vmx_int4_packed_bytes(int K, int N, int group) {
    if (K <= 0 || N <= 0 || group <= 0) return 0;
    return sz_offset_bytes(K, N) + (size_t)(K / group) * N * 4;
}

extern "C" int vmx_int4_pack(const uint32_t *qweight, const uint16_t *scales_bf16, const uint32_t *qzeros, void *packed, int K, int N,
                             int group, void *stream) {
    VMX_REQUIRE(qweight && scales_bf16 && packed, "int4_pack: null pointer");
    if (int rc = check_int4_shape("int4_pack", K, N, group)) return rc;
    uint32_t *W = (uint32_t *)packed;
    uint32_t *SZ = (uint32_t *)((char *)packed + sz_offset_bytes(K, N));
    const int64_t words = (int64_t)K * N / 8;
    hipLaunchKernelGGL(int4_pack_w_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, vmx_stream(stream), qweight, W, K, N);
    VMX_CHECK_LAUNCH("int4_pack(W)");
    const int G = K / group;
    const int64_t szn = (int64_t)G * N;
    hipLaunchKernelGGL(int4_pack_sz_kernel, dim3((unsigned)((szn + 255) / 256)), dim3(256), 0, vmx_stream(stream), scales_bf16, qzeros,
                       SZ, G, N);
    VMX_CHECK_LAUNCH("int4_pack(SZ)");
    return VMX_OK;
}

extern "C" int vmx_int4_dequant_bf16(const void *packed, uint16_t *dense, int K, int N, int group, void *stream) {
    VMX_REQUIRE(packed && dense, "int4_dequant: null pointer");
    if (int rc = check_int4_shape("int4_dequant", K, N, group)) return rc;
    const uint32_t *W = (const uint32_t *)packed;
    const uint32_t *SZ = (const uint32_t *)((const char *)packed + sz_offset_bytes(K, N));
    const int64_t words = (int64_t)K * N / 8;
    hipLaunchKernelGGL(int4_dequant_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, vmx_stream(stream), W, SZ, dense, K, N,
                       group);
    VMX_CHECK_LAUNCH("int4_dequant");
    return VMX_OK;
}

// ---- internal (C++) entry points used by the decoder ---------------------------------------
int vmx_int4_splits(int N, int K) { return plan_gemm(N, K).splits; }

// partial[splits][M][N] f32, also when splits == 1.  Arguments must already satisfy check_int4_shape.
// mode: 0 bf16 x; 1 silu*mul of gate|up slices; 2 residual add + RMSNorm; 3 paged-attention merge.
int vmx_int4_gemm_fused(const void *packed, int mode, const VmxXSource *xsrc, float *partial, int M, int N, int K, int group,
                        hipStream_t st) {
    const GemmPlan p = plan_gemm(N, K);
    const uint4 *W = (const uint4 *)packed;
    const uint32_t *SZ = (const uint32_t *)((const char *)packed + sz_offset_bytes(K, N));
    XSource src;
    src.x = xsrc->x; src.part = xsrc->part; src.part_splits = xsrc->part_splits; src.h = xsrc->h; src.norm_w = xsrc->norm_w;
    src.eps = xsrc->eps; src.exp_sums = xsrc->exp_sums; src.max_logits = xsrc->max_logits; src.seq_lens = xsrc->seq_lens;
    src.head_dim = xsrc->head_dim; src.max_parts = xsrc->max_parts; src.window = xsrc->window; src.part_tokens = xsrc->part_tokens;
    if (src.part_splits > SPLIT_MAX) {
        vmx_set_error("int4_gemm: %d input slices exceed %d", src.part_splits, SPLIT_MAX);
        return VMX_EINVAL;
    }
    switch (mode) {
        case X_BF16: return launch_gemm<X_BF16>(p, W, SZ, src, nullptr, nullptr, partial, M, N, K, group, st);
        case X_ACT: return launch_gemm<X_ACT>(p, W, SZ, src, nullptr, nullptr, partial, M, N, K, group, st);
        case X_NORM: return launch_gemm<X_NORM>(p, W, SZ, src, nullptr, nullptr, partial, M, N, K, group, st);
        case X_ATTN: return launch_gemm<X_ATTN>(p, W, SZ, src, nullptr, nullptr, partial, M, N, K, group, st);
        default: vmx_set_error("int4_gemm: bad activation mode %d", mode); return VMX_EINVAL;
    }
}

int vmx_int4_gemm_partial(const void *packed, const uint16_t *x, float *partial, int M, int N, int K, int group, hipStream_t st) {
    VmxXSource xs = {};
    xs.x = x;
    return vmx_int4_gemm_fused(packed, X_BF16, &xs, partial, M, N, K, group, st);
}

// AWQ checkpoint tensors -> columns [col0, col0+N) of merged GPTQ-sequential tensors with Ntot columns.
int vmx_awq_convert_into(const uint32_t *awq_qweight, const uint32_t *awq_qzeros, uint32_t *qweight_out, uint32_t *qzeros_out,
                         int K, int N, int group, int Ntot, int col0, hipStream_t st) {
    const int64_t total = (int64_t)(K / 8) * N;
    hipLaunchKernelGGL(awq_to_gptq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, awq_qweight, qweight_out + col0, K,
                       N, Ntot);
    VMX_CHECK_LAUNCH("awq_to_gptq_qweight");
    const int64_t zw = (int64_t)(K / group) * (N / 8);
    hipLaunchKernelGGL(awq_nibble_reorder_kernel, dim3((unsigned)((zw + 255) / 256)), dim3(256), 0, st, awq_qzeros,
                       qzeros_out + col0 / 8, zw, N / 8, Ntot / 8);
    VMX_CHECK_LAUNCH("awq_repack_nibbles");
    return VMX_OK;
}

extern "C" size_t vmx_int4_gemm_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % KB || N % NT) return 0;
    const GemmPlan p = plan_gemm(N, K);
    return p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
}

extern "C" int vmx_int4_gemm_bf16(uint16_t *y, const uint16_t *x, const void *packed, const uint16_t *bias, int M, int N, int K,
                                  int group, void *workspace, size_t workspace_bytes, void *stream) {
    VMX_REQUIRE(y && x && packed, "int4_gemm: null pointer");
    VMX_REQUIRE(M >= 1 && M <= 16, "int4_gemm: M (%d) must be in 1..16 (larger M takes the dequant + GEMM path)", M);
    if (int rc = check_int4_shape("int4_gemm", K, N, group)) return rc;
    const GemmPlan p = plan_gemm(N, K);
    const size_t need = p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
    VMX_REQUIRE(workspace_bytes >= need && (need == 0 || workspace), "int4_gemm: workspace %zu B < required %zu B", workspace_bytes,
                need);
    const uint4 *W = (const uint4 *)packed;
    const uint32_t *SZ = (const uint32_t *)((const char *)packed + sz_offset_bytes(K, N));
    hipStream_t st = vmx_stream(stream);
    float *partial = (float *)workspace;
    XSource src = {};
    src.x = x;
    if (int rc = launch_gemm<X_BF16>(p, W, SZ, src, bias, p.splits == 1 ? y : nullptr, partial, M, N, K, group, st)) return rc;
    if (p.splits > 1) {
        const int64_t total = (int64_t)M * N;
        hipLaunchKernelGGL(int4_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partial, bias, y, M, N,
                           p.splits);
        VMX_CHECK_LAUNCH("int4_reduce");
    }
    return VMX_OK;
}

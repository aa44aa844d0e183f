// Per-token glue operators of the decode path for gfx950: RMSNorm (+ fused residual add),
// NeoX RoPE, SiLU*mul, argmax, KV-cache scatter (bf16 copy or FP8 e4m3 quantise).
// Semantics: crates/core/kernels/{layernorm.cu:59-129,232-270, rope.cu:18-114,
// activations.cu:33-35,148-164, sampling.cu:47-94, cache_ops.cu:28-62} and the FP8 branch of
// crates/core/src/kv_cache/cache_engine.rs:196-247.  All are HBM/launch-bound: 16-byte
// vector accesses, one pass over the data, wave64 butterflies for the reductions.
#include <cmath>
#include "vmx_kernels.h"
#include <float.h>

namespace {

__device__ __forceinline__ void unpack8(const uint4 v, float f[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f[2 * j] = __uint_as_float(w[j] << 16);
        f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
    }
}

__device__ __forceinline__ uint4 pack8(const float f[8]) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = (uint32_t)vmx_f32_to_bf16(f[2 * j]) | ((uint32_t)vmx_f32_to_bf16(f[2 * j + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// One workgroup per token.  hidden % 8 == 0.  With `residual`: residual = bf16(residual + x),
// variance over the unrounded f32 sums, normalisation from the rounded residual (layernorm.cu:246-268).
__global__ __launch_bounds__(256) void rms_norm_kernel(uint16_t *__restrict__ out, const uint16_t *__restrict__ x,
                                                        uint16_t *__restrict__ residual, const uint16_t *__restrict__ weight,
                                                        float eps, int hidden, int plus_one) {
    __shared__ float red[4];
    const int64_t row = (int64_t)blockIdx.x * hidden;
    const int nvec = hidden >> 3;
    float var = 0.f;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        float a[8];
        unpack8(*reinterpret_cast<const uint4 *>(x + row + 8 * i), a);
        if (residual) {
            float r[8];
            unpack8(*reinterpret_cast<const uint4 *>(residual + row + 8 * i), r);
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = r[j] + a[j];
            *reinterpret_cast<uint4 *>(residual + row + 8 * i) = pack8(a);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) var += a[j] * a[j];
    }
    var = vmx_block_sum_256(var, red);
    const float inv = rsqrtf(var / (float)hidden + eps);
    const uint16_t *src = residual ? residual : x;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        float a[8], w[8];
        unpack8(*reinterpret_cast<const uint4 *>(src + row + 8 * i), a);
        if (weight) unpack8(*reinterpret_cast<const uint4 *>(weight + 8 * i), w);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float wj = weight ? w[j] : 1.0f;
            if (plus_one) wj = 1.0f + wj;
            a[j] = a[j] * inv * wj;
        }
        *reinterpret_cast<uint4 *>(out + row + 8 * i) = pack8(a);
    }
}

// One workgroup per token; thread i -> (head, pair index).
__global__ __launch_bounds__(256) void rope_neox_kernel(const int32_t *__restrict__ positions, uint16_t *__restrict__ q,
                                                         uint16_t *__restrict__ k, const float *__restrict__ cos_sin, int rot_dim,
                                                         int head_size, int num_heads, int num_kv_heads) {
    const int t = blockIdx.x;
    const int half = rot_dim >> 1;
    const float *cs = cos_sin + (int64_t)positions[t] * rot_dim;
    const int nq = num_heads * half, nk = k ? num_kv_heads * half : 0;
    for (int i = threadIdx.x; i < nq + nk; i += 256) {
        uint16_t *base;
        int hi, ri;
        if (i < nq) { hi = i / half; ri = i % half; base = q + ((int64_t)t * num_heads + hi) * head_size; }
        else { const int j = i - nq; hi = j / half; ri = j % half; base = k + ((int64_t)t * num_kv_heads + hi) * head_size; }
        const float c = cs[ri], s = cs[half + ri];
        const float xv = vmx_bf16_to_f32(base[ri]), yv = vmx_bf16_to_f32(base[half + ri]);
        base[ri] = vmx_f32_to_bf16(xv * c - yv * s);
        base[half + ri] = vmx_f32_to_bf16(yv * c + xv * s);
    }
}

// GPT-J pairing: element pair (2i, 2i + 1) of the rotated part turns by angle i; the table row is cos[rot/2] | sin[rot/2] as for NeoX.
__global__ __launch_bounds__(256) void rope_gptj_kernel(const int32_t *__restrict__ positions, uint16_t *__restrict__ q,
                                                         uint16_t *__restrict__ k, const float *__restrict__ cos_sin, int rot_dim,
                                                         int head_size, int num_heads, int num_kv_heads) {
    const int t = blockIdx.x;
    const int half = rot_dim >> 1;
    const float *cs = cos_sin + (int64_t)positions[t] * rot_dim;
    const int nq = num_heads * half, nk = k ? num_kv_heads * half : 0;
    for (int i = threadIdx.x; i < nq + nk; i += 256) {
        uint32_t *pair;   // the two bf16 values of a pair share an aligned dword (head_size is even)
        int ri;
        if (i < nq) { ri = i % half; pair = reinterpret_cast<uint32_t *>(q + ((int64_t)t * num_heads + i / half) * head_size) + ri; }
        else { const int j = i - nq; ri = j % half; pair = reinterpret_cast<uint32_t *>(k + ((int64_t)t * num_kv_heads + j / half) * head_size) + ri; }
        const float c = cs[ri], s = cs[half + ri];
        const uint32_t w = *pair;
        const float xv = vmx_bf16_to_f32((uint16_t)(w & 0xFFFFu)), yv = vmx_bf16_to_f32((uint16_t)(w >> 16));
        *pair = (uint32_t)vmx_f32_to_bf16(xv * c - yv * s) | ((uint32_t)vmx_f32_to_bf16(yv * c + xv * s) << 16);
    }
}

// K/V rows [tokens, kv_heads, head_dim] (16-bit elements) into a cache laid out [blocks, kv_heads, block_size, head_dim].
// grid (tokens, kv_heads); 8 elements per thread step.
__global__ __launch_bounds__(64) void reshape_and_cache_hnd_kernel(const uint16_t *__restrict__ key, const uint16_t *__restrict__ value,
                                                                    uint16_t *__restrict__ k_cache, uint16_t *__restrict__ v_cache,
                                                                    const int32_t *__restrict__ slots, int kv_heads, int head_dim,
                                                                    int block_size) {
    const int t = blockIdx.x, h = blockIdx.y;
    const int slot = slots[t];
    if (slot < 0) return;
    const int64_t src = ((int64_t)t * kv_heads + h) * head_dim;
    const int64_t dst = (((int64_t)(slot / block_size) * kv_heads + h) * block_size + slot % block_size) * head_dim;
    for (int d = threadIdx.x * 8; d < head_dim; d += 64 * 8) {
        *reinterpret_cast<uint4 *>(k_cache + dst + d) = *reinterpret_cast<const uint4 *>(key + src + d);
        *reinterpret_cast<uint4 *>(v_cache + dst + d) = *reinterpret_cast<const uint4 *>(value + src + d);
    }
}

// (src, dst) block pairs; grid (pairs, 2 caches x chunks); 16 bytes per thread step.  Pairs are independent unless the
// caller lists a block both as a source and as a destination (the reference has the same rule: cache_ops.cu:183-202).
__global__ __launch_bounds__(256) void copy_blocks_kernel(unsigned char *__restrict__ k_cache, unsigned char *__restrict__ v_cache,
                                                           const int32_t *__restrict__ mapping, int64_t bytes_per_block) {
    const int pair = blockIdx.x;
    unsigned char *cache = (blockIdx.y & 1) ? v_cache : k_cache;
    const int chunk = blockIdx.y >> 1, nchunk = gridDim.y >> 1;
    const int64_t src = (int64_t)mapping[2 * pair] * bytes_per_block, dst = (int64_t)mapping[2 * pair + 1] * bytes_per_block;
    const int64_t n16 = bytes_per_block >> 4;
    for (int64_t i = (int64_t)chunk * 256 + threadIdx.x; i < n16; i += (int64_t)nchunk * 256)
        reinterpret_cast<uint4 *>(cache + dst)[i] = reinterpret_cast<const uint4 *>(cache + src)[i];
}

__device__ __forceinline__ float silu(float g) { return g / (1.0f + expf(-g)); }

__global__ __launch_bounds__(256) void silu_mul_kernel(uint16_t *__restrict__ out, const uint16_t *__restrict__ gate,
                                                        const uint16_t *__restrict__ up, int64_t n) {
    const int64_t nvec = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float g[8], u[8];
        unpack8(reinterpret_cast<const uint4 *>(gate)[i], g);
        unpack8(reinterpret_cast<const uint4 *>(up)[i], u);
#pragma unroll
        for (int j = 0; j < 8; j++) g[j] = silu(g[j]) * u[j];
        reinterpret_cast<uint4 *>(out)[i] = pack8(g);
    }
    if (blockIdx.x == 0)
        for (int64_t i = (nvec << 3) + threadIdx.x; i < n; i += 256)
            out[i] = vmx_f32_to_bf16(silu(vmx_bf16_to_f32(gate[i])) * vmx_bf16_to_f32(up[i]));
}

// Argmax: key = (order-preserving float bits << 32) | ~index, so a u64 max picks the largest
// value and, among equal values, the smallest index (the reference leaves ties unspecified).
__device__ __forceinline__ unsigned long long argmax_key(float v, uint32_t idx) {
    uint32_t b = __float_as_uint(v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)b << 32) | (unsigned long long)(~idx);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = ((unsigned long long)(unsigned)vmx_lane_xor_m((int)(v >> 32), off) << 32) | (unsigned)vmx_lane_xor_m((int)v, off);
        v = o > v ? o : v;
    }
    return v;
}

// grid (chunks, rows): each workgroup reduces one slice of a row to a key in part[row][chunk].
template <bool F16>
__global__ __launch_bounds__(256) void argmax_partial_kernel(const uint16_t *__restrict__ logits, unsigned long long *__restrict__ part,
                                                              int vocab, uint32_t index_offset) {
    __shared__ unsigned long long red[4];
    const int row = blockIdx.y, chunks = gridDim.x;
    const uint16_t *lg = logits + (int64_t)row * vocab;
    const int per = (((vocab + chunks - 1) / chunks) + 7) & ~7;
    const int lo = blockIdx.x * per, hi = min(vocab, lo + per);
    unsigned long long best = 0;
    const bool vec_ok = (((uintptr_t)lg) & 15) == 0;
    int i = lo + threadIdx.x * 8;
    for (; vec_ok && i + 8 <= hi; i += 256 * 8) {
        float f[8];
        vmx_unpack16x8<F16>(*reinterpret_cast<const uint4 *>(lg + i), f);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned long long k = argmax_key(f[j], index_offset + (uint32_t)(i + j));
            best = k > best ? k : best;
        }
    }
    if (!vec_ok) i = lo + threadIdx.x * 8;
    for (; i < hi; i += 256 * 8)  // tail (or unaligned row): scalar
        for (int j = i; j < min(i + 8, hi); j++) {
            const unsigned long long k = argmax_key(vmx_ld16<F16>(lg[j]), index_offset + (uint32_t)j);
            best = k > best ? k : best;
        }
    best = wave_max_u64(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = red[0];
        for (int w = 1; w < 4; w++) b = red[w] > b ? red[w] : b;
        part[(int64_t)row * chunks + blockIdx.x] = b;
    }
}

// out != NULL: token id; keys_out != NULL: the winning key (for a cross-rank max)
__global__ __launch_bounds__(64) void argmax_final_kernel(const unsigned long long *__restrict__ part, int32_t *__restrict__ out,
                                                           unsigned long long *__restrict__ keys_out, int chunks) {
    const int row = blockIdx.x;
    unsigned long long best = 0;
    for (int c = threadIdx.x; c < chunks; c += 64) {
        const unsigned long long k = part[(int64_t)row * chunks + c];
        best = k > best ? k : best;
    }
    best = wave_max_u64(best);
    if (threadIdx.x == 0) {
        if (out) out[row] = (int32_t)(~(uint32_t)(best & 0xFFFFFFFFull));
        if (keys_out) keys_out[row] = best;
    }
}

// e4m3fn, round-to-nearest-even, saturating (no inf): the OCP conversion gfx950 has in hardware.
__device__ __forceinline__ uint8_t f32_to_e4m3(float v) {
    const uint32_t packed = __builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false);
    return (uint8_t)(packed & 0xFFu);
}

// One workgroup per token.  FP8: f32(x)/scale, clamp +-448, round to bf16, then e4m3 RNE
// (the reference casts the scaled f32 tensor to bf16 before its static-quant kernel).
__global__ __launch_bounds__(256) void reshape_and_cache_kernel(const uint16_t *__restrict__ key, const uint16_t *__restrict__ value,
                                                                 void *__restrict__ k_cache, void *__restrict__ v_cache,
                                                                 const int32_t *__restrict__ slots, int row_elems, int fp8,
                                                                 const float *__restrict__ k_scale, const float *__restrict__ v_scale) {
    const int t = blockIdx.x;
    const int slot = slots[t];
    if (slot < 0) return;
    const uint16_t *ks = key + (int64_t)t * row_elems, *vs = value + (int64_t)t * row_elems;
    const int64_t dst = (int64_t)slot * row_elems;
    if (!fp8) {
        for (int i = threadIdx.x; i < (row_elems >> 3); i += 256) {
            reinterpret_cast<uint4 *>((uint16_t *)k_cache + dst)[i] = reinterpret_cast<const uint4 *>(ks)[i];
            reinterpret_cast<uint4 *>((uint16_t *)v_cache + dst)[i] = reinterpret_cast<const uint4 *>(vs)[i];
        }
    } else if (fp8 == 2) {   // INT8: round(clamp(x / scale, +-127)) + 128, round half away from zero (quantize_int8, kv_cache/quantization.rs:330-345)
        const float ksc = k_scale ? *k_scale : 1.0f, vsc = v_scale ? *v_scale : 1.0f;
        for (int i = threadIdx.x; i < row_elems; i += 256) {
            const float a = fminf(fmaxf(vmx_bf16_to_f32(ks[i]) / ksc, -127.0f), 127.0f);
            const float b = fminf(fmaxf(vmx_bf16_to_f32(vs[i]) / vsc, -127.0f), 127.0f);
            ((uint8_t *)k_cache)[dst + i] = (uint8_t)((int)roundf(a) + 128);
            ((uint8_t *)v_cache)[dst + i] = (uint8_t)((int)roundf(b) + 128);
        }
    } else if (fp8 == 3) {   // FP8 e5m2 (max finite 57344): x / scale, clamped, through bf16 like the e4m3 path, round to nearest even
        const float ksc = k_scale ? *k_scale : 1.0f, vsc = v_scale ? *v_scale : 1.0f;
        for (int i = threadIdx.x; i < row_elems; i += 256) {
            float a = vmx_bf16_to_f32(ks[i]) / ksc, b = vmx_bf16_to_f32(vs[i]) / vsc;
            a = vmx_bf16_to_f32(vmx_f32_to_bf16(fminf(fmaxf(a, -57344.0f), 57344.0f)));
            b = vmx_bf16_to_f32(vmx_f32_to_bf16(fminf(fmaxf(b, -57344.0f), 57344.0f)));
            ((uint8_t *)k_cache)[dst + i] = (uint8_t)(__builtin_amdgcn_cvt_pk_bf8_f32(a, a, 0, false) & 0xFFu);
            ((uint8_t *)v_cache)[dst + i] = (uint8_t)(__builtin_amdgcn_cvt_pk_bf8_f32(b, b, 0, false) & 0xFFu);
        }
    } else {
        const float ksc = k_scale ? *k_scale : 1.0f, vsc = v_scale ? *v_scale : 1.0f;
        for (int i = threadIdx.x; i < row_elems; i += 256) {
            float a = vmx_bf16_to_f32(ks[i]) / ksc, b = vmx_bf16_to_f32(vs[i]) / vsc;
            a = fminf(fmaxf(a, -448.0f), 448.0f);
            b = fminf(fmaxf(b, -448.0f), 448.0f);
            a = vmx_bf16_to_f32(vmx_f32_to_bf16(a));
            b = vmx_bf16_to_f32(vmx_f32_to_bf16(b));
            ((uint8_t *)k_cache)[dst + i] = f32_to_e4m3(a);
            ((uint8_t *)v_cache)[dst + i] = f32_to_e4m3(b);
        }
    }
}

}  // namespace

extern "C" int vmx_rms_norm_bf16(uint16_t *out, const uint16_t *x, uint16_t *residual, const uint16_t *weight, float eps, int tokens,
                                 int hidden, int weight_plus_one, void *stream) {
    VMX_REQUIRE(out && x, "rms_norm: null pointer");
    VMX_REQUIRE(tokens >= 0 && hidden > 0 && hidden % 8 == 0, "rms_norm: hidden (%d) must be a positive multiple of 8", hidden);
    if (tokens == 0) return VMX_OK;
    hipLaunchKernelGGL(rms_norm_kernel, dim3(tokens), dim3(256), 0, vmx_stream(stream), out, x, residual, weight, eps, hidden,
                       weight_plus_one);
    VMX_CHECK_LAUNCH("rms_norm");
    return VMX_OK;
}

extern "C" int vmx_rope_neox_bf16(const int32_t *positions, uint16_t *q, uint16_t *k, const float *cos_sin_cache, int tokens,
                                  int rot_dim, int head_size, int num_heads, int num_kv_heads, void *stream) {
    VMX_REQUIRE(positions && q && cos_sin_cache, "rope: null pointer");
    VMX_REQUIRE(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rope: rot_dim %d invalid for head_size %d", rot_dim,
                head_size);
    if (tokens <= 0) return VMX_OK;
    hipLaunchKernelGGL(rope_neox_kernel, dim3(tokens), dim3(256), 0, vmx_stream(stream), positions, q, k, cos_sin_cache, rot_dim,
                       head_size, num_heads, num_kv_heads);
    VMX_CHECK_LAUNCH("rope");
    return VMX_OK;
}

extern "C" int vmx_rope_gptj_bf16(const int32_t *positions, uint16_t *q, uint16_t *k, const float *cos_sin_cache, int tokens,
                                  int rot_dim, int head_size, int num_heads, int num_kv_heads, void *stream) {
    VMX_REQUIRE(positions && q && cos_sin_cache, "rope: null pointer");
    VMX_REQUIRE(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size && head_size % 2 == 0, "rope: rot_dim %d invalid for head_size %d",
                rot_dim, head_size);
    VMX_REQUIRE(((uintptr_t)q & 3) == 0 && ((uintptr_t)k & 3) == 0, "rope(gptj): q and k must be 4-byte aligned");
    if (tokens <= 0) return VMX_OK;
    hipLaunchKernelGGL(rope_gptj_kernel, dim3(tokens), dim3(256), 0, vmx_stream(stream), positions, q, k, cos_sin_cache, rot_dim,
                       head_size, num_heads, num_kv_heads);
    VMX_CHECK_LAUNCH("rope(gptj)");
    return VMX_OK;
}

extern "C" int vmx_reshape_and_cache_hnd(const uint16_t *key, const uint16_t *value, void *k_cache, void *v_cache,
                                         const int32_t *slot_mapping, int tokens, int kv_heads, int head_dim, int block_size, void *stream) {
    VMX_REQUIRE(key && value && k_cache && v_cache && slot_mapping, "reshape_and_cache_hnd: null pointer");
    VMX_REQUIRE(kv_heads > 0 && block_size > 0 && head_dim > 0 && head_dim % 8 == 0, "reshape_and_cache_hnd: head_dim %d not a multiple of 8",
                head_dim);
    VMX_REQUIRE((((uintptr_t)key | (uintptr_t)value | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15) == 0,
                "reshape_and_cache_hnd: pointers must be 16-byte aligned");
    if (tokens <= 0) return VMX_OK;
    hipLaunchKernelGGL(reshape_and_cache_hnd_kernel, dim3(tokens, kv_heads), dim3(64), 0, vmx_stream(stream), key, value,
                       (uint16_t *)k_cache, (uint16_t *)v_cache, slot_mapping, kv_heads, head_dim, block_size);
    VMX_CHECK_LAUNCH("reshape_and_cache_hnd");
    return VMX_OK;
}

extern "C" int vmx_copy_blocks(void *k_cache, void *v_cache, const int32_t *block_mapping, int num_pairs, int64_t bytes_per_block,
                               void *stream) {
    VMX_REQUIRE(k_cache && v_cache && (block_mapping || num_pairs == 0), "copy_blocks: null pointer");
    VMX_REQUIRE(num_pairs >= 0 && bytes_per_block > 0 && bytes_per_block % 16 == 0, "copy_blocks: bytes_per_block %lld not a multiple of 16",
                (long long)bytes_per_block);
    VMX_REQUIRE((((uintptr_t)k_cache | (uintptr_t)v_cache) & 15) == 0, "copy_blocks: caches must be 16-byte aligned");
    if (num_pairs == 0) return VMX_OK;
    int64_t chunks = (bytes_per_block / 16 + 255) / 256;
    chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
    hipLaunchKernelGGL(copy_blocks_kernel, dim3(num_pairs, (unsigned)(2 * chunks)), dim3(256), 0, vmx_stream(stream), (unsigned char *)k_cache,
                       (unsigned char *)v_cache, block_mapping, bytes_per_block);
    VMX_CHECK_LAUNCH("copy_blocks");
    return VMX_OK;
}

extern "C" int vmx_silu_and_mul_bf16(uint16_t *out, const uint16_t *gate, const uint16_t *up, int64_t n, void *stream) {
    VMX_REQUIRE(out && gate && up && n >= 0, "silu_and_mul: bad arguments");
    if (n == 0) return VMX_OK;
    int64_t blocks = ((n >> 3) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, vmx_stream(stream), out, gate, up, n);
    VMX_CHECK_LAUNCH("silu_and_mul");
    return VMX_OK;
}

static int argmax_chunks(int vocab) {
    int c = (vocab + 4095) / 4096;
    return c < 1 ? 1 : (c > 64 ? 64 : c);
}

extern "C" size_t vmx_argmax_workspace_bytes(int rows, int vocab) {
    if (rows <= 0 || vocab <= 0) return 0;
    return (size_t)rows * argmax_chunks(vocab) * sizeof(unsigned long long);
}

static int argmax_impl(int32_t *out_ids, const uint16_t *logits, int rows, int vocab, void *workspace, size_t workspace_bytes, void *stream,
                       bool f16) {
    VMX_REQUIRE(out_ids && logits && workspace, "argmax: null pointer");
    VMX_REQUIRE(rows >= 0 && vocab > 0, "argmax: bad shape");
    VMX_REQUIRE(workspace_bytes >= vmx_argmax_workspace_bytes(rows, vocab), "argmax: workspace too small");
    if (rows == 0) return VMX_OK;
    const int chunks = argmax_chunks(vocab);
    if (f16)
        hipLaunchKernelGGL(argmax_partial_kernel<true>, dim3(chunks, rows), dim3(256), 0, vmx_stream(stream), logits,
                           (unsigned long long *)workspace, vocab, 0u);
    else
        hipLaunchKernelGGL(argmax_partial_kernel<false>, dim3(chunks, rows), dim3(256), 0, vmx_stream(stream), logits,
                           (unsigned long long *)workspace, vocab, 0u);
    VMX_CHECK_LAUNCH("argmax(partial)");
    hipLaunchKernelGGL(argmax_final_kernel, dim3(rows), dim3(64), 0, vmx_stream(stream), (const unsigned long long *)workspace, out_ids,
                       (unsigned long long *)nullptr, chunks);
    VMX_CHECK_LAUNCH("argmax(final)");
    return VMX_OK;
}

extern "C" int vmx_argmax_bf16(int32_t *out_ids, const uint16_t *logits, int rows, int vocab, void *workspace, size_t workspace_bytes,
                               void *stream) {
    return argmax_impl(out_ids, logits, rows, vocab, workspace, workspace_bytes, stream, false);
}

// The same over fp16 logits (models that run in fp16)
extern "C" int vmx_argmax_f16(int32_t *out_ids, const uint16_t *logits, int rows, int vocab, void *workspace, size_t workspace_bytes,
                              void *stream) {
    return argmax_impl(out_ids, logits, rows, vocab, workspace, workspace_bytes, stream, true);
}

extern "C" int vmx_reshape_and_cache(const uint16_t *key, const uint16_t *value, void *k_cache, void *v_cache,
                                     const int32_t *slot_mapping, int tokens, int kv_heads, int head_dim, int block_size, int kv_dtype,
                                     const float *k_scale, const float *v_scale, void *stream) {
    VMX_REQUIRE(key && value && k_cache && v_cache && slot_mapping, "reshape_and_cache: null pointer");
    VMX_REQUIRE(kv_heads > 0 && head_dim > 0 && (kv_heads * head_dim) % 8 == 0 && block_size > 0, "reshape_and_cache: bad geometry");
    VMX_REQUIRE(kv_dtype == VMX_KV_AUTO || kv_dtype == VMX_KV_FP8_E4M3 || kv_dtype == VMX_KV_INT8 || kv_dtype == VMX_KV_FP8_E5M2,
                "reshape_and_cache: unsupported kv dtype %d", kv_dtype);
    if (tokens <= 0) return VMX_OK;
    hipLaunchKernelGGL(reshape_and_cache_kernel, dim3(tokens), dim3(256), 0, vmx_stream(stream), key, value, k_cache, v_cache,
                       slot_mapping, kv_heads * head_dim, kv_dtype == VMX_KV_FP8_E5M2 ? 3 : (kv_dtype == VMX_KV_INT8 ? 2 : (kv_dtype == VMX_KV_FP8_E4M3 ? 1 : 0)),
                       k_scale, v_scale);
    VMX_CHECK_LAUNCH("reshape_and_cache");
    return VMX_OK;
}

// ---- KVScales: first-write calibration of the FP8 KV cache's per-tensor scales (kv_cache/quantization.rs:80-277) ----
// Device-resident state, 32 bytes: f32 k_scale, f32 v_scale, u32 calibrated, u32 amax bits of K and V (scratch, zero between
// calls), 3 words of padding.  The first two words are what every kernel here takes as k_scale / v_scale pointers.
namespace {
struct KvScalesState {
    float k_scale, v_scale;
    uint32_t calibrated, amax_k, amax_v, pad[3];
};

__global__ __launch_bounds__(256) void kv_scales_amax_kernel(KvScalesState *__restrict__ st, const uint16_t *__restrict__ k, int64_t nk,
                                                              const uint16_t *__restrict__ v, int64_t nv) {
    __shared__ float red[4];
    if (st->calibrated) return;   // pinned scales never move again (calibrate_if_needed, :200-203)
    const bool is_v = blockIdx.y == 1;
    const uint16_t *x = is_v ? v : k;
    const int64_t n = is_v ? nv : nk;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        m = fmaxf(m, fabsf(vmx_bf16_to_f32(x[i])));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(is_v ? &st->amax_v : &st->amax_k, __float_as_uint(m));   // non-negative floats order as their bit patterns
    }
}

// scale = max(amax * headroom / 448, 1e-12) in f32, the division as the multiply by f32(1/448) the reference's affine op does
// (compute_scale_with_headroom, :272-277)
__global__ void kv_scales_finish_kernel(KvScalesState *st, float headroom) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!st->calibrated) {
        const float inv_max = (float)(1.0 / 448.0);
        st->k_scale = fmaxf((__uint_as_float(st->amax_k) * headroom) * inv_max, 1e-12f);
        st->v_scale = fmaxf((__uint_as_float(st->amax_v) * headroom) * inv_max, 1e-12f);
        st->calibrated = 1u;
    }
    st->amax_k = 0u;
    st->amax_v = 0u;
}
}  // namespace

extern "C" size_t vmx_kv_scales_state_bytes(void) { return sizeof(KvScalesState); }

extern "C" int vmx_kv_scales_init(void *state, float k_scale, float v_scale, int calibrated, void *stream) {
    VMX_REQUIRE(state, "kv_scales_init: null pointer");
    VMX_REQUIRE(std::isfinite(k_scale) && std::isfinite(v_scale) && k_scale > 0.f && v_scale > 0.f,
                "kv_scales_init: scales must be positive and finite (got %g, %g)", (double)k_scale, (double)v_scale);
    KvScalesState h = {k_scale, v_scale, calibrated ? 1u : 0u, 0u, 0u, {0u, 0u, 0u}};
    // the source is this stack frame: wait for the copy before returning
    VMX_CHECK_HIP(hipMemcpyAsync(state, &h, sizeof(h), hipMemcpyHostToDevice, vmx_stream(stream)));
    VMX_CHECK_HIP(hipStreamSynchronize(vmx_stream(stream)));
    return VMX_OK;
}

extern "C" int vmx_kv_scales_calibrate_if_needed(void *state, const uint16_t *k, int64_t nk, const uint16_t *v, int64_t nv, float headroom,
                                                 void *stream) {
    VMX_REQUIRE(state && k && v, "kv_scales_calibrate_if_needed: null pointer");
    VMX_REQUIRE(nk >= 0 && nv >= 0, "kv_scales_calibrate_if_needed: negative element count");
    VMX_REQUIRE(std::isfinite(headroom) && headroom > 0.f,
                "KV scale headroom factor must be a positive finite value, got %g", (double)headroom);   // set_headroom_factor, :143-151
    const int64_t n = nk > nv ? nk : nv;
    int64_t blocks = (n + 256 * 8 - 1) / (256 * 8);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(kv_scales_amax_kernel, dim3((unsigned)blocks, 2), dim3(256), 0, vmx_stream(stream), (KvScalesState *)state, k, nk, v, nv);
    hipLaunchKernelGGL(kv_scales_finish_kernel, dim3(1), dim3(64), 0, vmx_stream(stream), (KvScalesState *)state, headroom);
    VMX_CHECK_LAUNCH("kv_scales_calibrate_if_needed");
    return VMX_OK;
}

namespace {
__global__ void argmax_keys_to_ids_kernel(const unsigned long long *__restrict__ keys, int32_t *__restrict__ out, int rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) out[r] = (int32_t)(~(uint32_t)(keys[r] & 0xFFFFFFFFull));
}
}  // namespace

// Vocab-parallel argmax, step 1: per-row winning key over this rank's logits slice (global index =
// index_offset + local index).  keys: u64 [rows]; workspace as vmx_argmax_bf16, plus the keys.
int vmx_argmax_keys(unsigned long long *keys, const uint16_t *logits, int rows, int vocab, uint32_t index_offset, void *workspace,
                    hipStream_t st, int f16) {
    const int chunks = argmax_chunks(vocab);
    if (f16)
        hipLaunchKernelGGL(argmax_partial_kernel<true>, dim3(chunks, rows), dim3(256), 0, st, logits, (unsigned long long *)workspace, vocab,
                           index_offset);
    else
        hipLaunchKernelGGL(argmax_partial_kernel<false>, dim3(chunks, rows), dim3(256), 0, st, logits, (unsigned long long *)workspace, vocab,
                           index_offset);
    hipLaunchKernelGGL(argmax_final_kernel, dim3(rows), dim3(64), 0, st, (const unsigned long long *)workspace, (int32_t *)nullptr, keys,
                       chunks);
    VMX_CHECK_LAUNCH("argmax(keys)");
    return VMX_OK;
}

// step 2 (after the cross-rank max of the keys): keys -> token ids.
int vmx_argmax_keys_to_ids(const unsigned long long *keys, int32_t *out_ids, int rows, hipStream_t st) {
    hipLaunchKernelGGL(argmax_keys_to_ids_kernel, dim3((rows + 63) / 64), dim3(64), 0, st, keys, out_ids, rows);
    VMX_CHECK_LAUNCH("argmax(keys_to_ids)");
    return VMX_OK;
}

// ---------------------------------------------------------------- more elementwise / fused glue
namespace {

// grid = batch*(Hq+Hkv)*seq heads, block 64: per-head RMSNorm (eps 1e-6) and NeoX RoPE in one pass with a
// single rounding at the store (crates/core/kernels/qknorm_rope.cu:47-136).  Layout [batch, heads, seq, hd].
__global__ __launch_bounds__(64) void fused_qknorm_rope_kernel(uint16_t *__restrict__ out_q, uint16_t *__restrict__ out_k,
                                                                const uint16_t *__restrict__ in_q, const uint16_t *__restrict__ in_k,
                                                                const uint16_t *__restrict__ qw, const uint16_t *__restrict__ kw,
                                                                const float *__restrict__ cos_cache, const float *__restrict__ sin_cache,
                                                                const int32_t *__restrict__ positions, int batch, int Hq, int Hkv, int seq,
                                                                int hd) {
    const int total_q = batch * Hq * seq;
    int id = blockIdx.x;
    const bool is_q = id < total_q;
    if (!is_q) id -= total_q;
    const int H = is_q ? Hq : Hkv;
    const int s = id % seq, b = id / (H * seq);
    const int64_t off = (int64_t)id * hd;  // ((b*H + head)*seq + s)*hd == id*hd
    const uint16_t *in = (is_q ? in_q : in_k) + off;
    uint16_t *out = (is_q ? out_q : out_k) + off;
    const uint16_t *w = is_q ? qw : kw;
    const int half = hd >> 1, lane = threadIdx.x;
    float ss = 0.f;
    for (int i = lane; i < hd; i += 64) { const float v = vmx_bf16_to_f32(in[i]); ss += v * v; }
    ss = vmx_wave_sum(ss);
    const float inv = rsqrtf(ss / (float)hd + 1e-6f);
    const int pos = positions[b * seq + s];
    const float *cr = cos_cache + (int64_t)pos * half, *sr = sin_cache + (int64_t)pos * half;
    for (int i = lane; i < half; i += 64) {
        const float lo = vmx_bf16_to_f32(in[i]) * inv * vmx_bf16_to_f32(w[i]);
        const float hi = vmx_bf16_to_f32(in[i + half]) * inv * vmx_bf16_to_f32(w[i + half]);
        const float c = cr[i], sn = sr[i];
        out[i] = vmx_f32_to_bf16(lo * c - hi * sn);
        out[i + half] = vmx_f32_to_bf16(hi * c + lo * sn);
    }
}

// mode 0: out = a + b;  1: gelu_erf(a) * b;  2: gelu_tanh(a) * b   (activations.cu:16-31,37-77, add_bf16 :96-110)
__global__ __launch_bounds__(256) void binary_bf16_kernel(uint16_t *__restrict__ out, const uint16_t *__restrict__ a,
                                                           const uint16_t *__restrict__ b, int64_t n, int mode) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = vmx_bf16_to_f32(a[i]), y = vmx_bf16_to_f32(b[i]);
        float r;
        if (mode == 0) r = x + y;
        else if (mode == 1) r = (x * 0.5f * (1.0f + erff(x * 0.7071067811865476f))) * y;
        else r = (0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)))) * y;
        out[i] = vmx_f32_to_bf16(r);
    }
}

}  // namespace

extern "C" int vmx_fused_qknorm_rope_bf16(uint16_t *out_q, uint16_t *out_k, const uint16_t *in_q, const uint16_t *in_k,
                                          const uint16_t *q_weight, const uint16_t *k_weight, const float *cos_cache,
                                          const float *sin_cache, const int32_t *positions, int batch, int num_q_heads, int num_kv_heads,
                                          int seq_len, int head_dim, void *stream) {
    VMX_REQUIRE(out_q && out_k && in_q && in_k && q_weight && k_weight && cos_cache && sin_cache && positions,
                "fused_qknorm_rope: null pointer");
    VMX_REQUIRE(batch >= 0 && seq_len >= 0 && num_q_heads > 0 && num_kv_heads > 0 && head_dim > 0 && head_dim % 2 == 0,
                "fused_qknorm_rope: bad geometry");
    const int total = batch * (num_q_heads + num_kv_heads) * seq_len;
    if (total == 0) return VMX_OK;
    hipLaunchKernelGGL(fused_qknorm_rope_kernel, dim3(total), dim3(64), 0, vmx_stream(stream), out_q, out_k, in_q, in_k, q_weight,
                       k_weight, cos_cache, sin_cache, positions, batch, num_q_heads, num_kv_heads, seq_len, head_dim);
    VMX_CHECK_LAUNCH("fused_qknorm_rope");
    return VMX_OK;
}

static int launch_binary(uint16_t *out, const uint16_t *a, const uint16_t *b, int64_t n, int mode, void *stream, const char *who) {
    VMX_REQUIRE(out && a && b && n >= 0, "%s: bad arguments", who);
    if (n == 0) return VMX_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(binary_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, vmx_stream(stream), out, a, b, n, mode);
    VMX_CHECK_LAUNCH(who);
    return VMX_OK;
}

extern "C" int vmx_add_bf16(uint16_t *out, const uint16_t *a, const uint16_t *b, int64_t n, void *stream) {
    return launch_binary(out, a, b, n, 0, stream, "add_bf16");
}

extern "C" int vmx_gelu_and_mul_bf16(uint16_t *out, const uint16_t *gate, const uint16_t *up, int64_t n, int tanh_approx, void *stream) {
    return launch_binary(out, gate, up, n, tanh_approx ? 2 : 1, stream, "gelu_and_mul");
}

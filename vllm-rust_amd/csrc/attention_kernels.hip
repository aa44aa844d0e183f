// Paged-attention decode for gfx950 (replaces paged_attention_v2_* and its reduce kernel,
// crates/core/kernels/paged_attention.cu:558-954).
//
// Semantics kept from the reference: logits (q.k)*scale, softmax with exp(l - max), split of the
// attended range [start, seq_len) into partitions that are normalised locally and merged with
// the log-sum-exp rule; GQA kv_head = head / (H / KVH); sliding window start = seq_len - window;
// FP8 e4m3 caches dequantised as float(byte) * scale.
//
// Structure (gfx950's own): one workgroup = one (sequence, KV head, partition of 256 tokens) and
// serves ALL query heads of that KV head, so each K/V byte is read from HBM once instead of once
// per query head.  Phase 1 is token-per-lane (each lane streams one K row in 16-byte pieces and
// dots it against the q vectors held in LDS, which are read as broadcasts); phase 2 is a
// per-head softmax over the 256 logits; phase 3 is dim-per-lane (a wave reads one V row as one
// coalesced access and the four waves split the tokens).
#include "vmx_kernels.h"
#include <float.h>

namespace {

constexpr int PART = 64;  // tokens per partition = lanes of the wave that owns it

template <bool FP8> struct KvElem;
template <> struct KvElem<false> { using T = uint16_t; };
template <> struct KvElem<true> { using T = uint8_t; };

// 8 consecutive cache elements -> f32
template <bool FP8> __device__ __forceinline__ void load8(const void *base, int64_t idx, float scale, float f[8]) {
    if constexpr (!FP8) {
        const uint4 v = *reinterpret_cast<const uint4 *>((const uint16_t *)base + idx);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f[2 * j] = __uint_as_float(w[j] << 16);
            f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
        }
    } else {
        const uint2 v = *reinterpret_cast<const uint2 *>((const uint8_t *)base + idx);
        f[0] = __builtin_amdgcn_cvt_f32_fp8(v.x, 0) * scale;
        f[1] = __builtin_amdgcn_cvt_f32_fp8(v.x, 1) * scale;
        f[2] = __builtin_amdgcn_cvt_f32_fp8(v.x, 2) * scale;
        f[3] = __builtin_amdgcn_cvt_f32_fp8(v.x, 3) * scale;
        f[4] = __builtin_amdgcn_cvt_f32_fp8(v.y, 0) * scale;
        f[5] = __builtin_amdgcn_cvt_f32_fp8(v.y, 1) * scale;
        f[6] = __builtin_amdgcn_cvt_f32_fp8(v.y, 2) * scale;
        f[7] = __builtin_amdgcn_cvt_f32_fp8(v.y, 3) * scale;
    }
}

// 2 consecutive cache elements -> f32
template <bool FP8> __device__ __forceinline__ void load2(const void *base, int64_t idx, float scale, float &a, float &b) {
    if constexpr (!FP8) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>((const uint16_t *)base + idx);
        a = __uint_as_float(w << 16);
        b = __uint_as_float(w & 0xFFFF0000u);
    } else {
        const uint32_t w = *reinterpret_cast<const uint16_t *>((const uint8_t *)base + idx);
        a = __builtin_amdgcn_cvt_f32_fp8(w, 0) * scale;
        b = __builtin_amdgcn_cvt_f32_fp8(w, 1) * scale;
    }
}

// Raw (unconverted) cache accesses, so that many rows can wait in registers cheaply.
template <bool FP8> struct Raw8 { using T = uint4; };
template <> struct Raw8<true> { using T = uint2; };

template <bool FP8> __device__ __forceinline__ typename Raw8<FP8>::T load8_raw(const void *base, int64_t idx) {
    if constexpr (!FP8) return *reinterpret_cast<const uint4 *>((const uint16_t *)base + idx);
    else return *reinterpret_cast<const uint2 *>((const uint8_t *)base + idx);
}

template <bool FP8> __device__ __forceinline__ void cvt8(const typename Raw8<FP8>::T v, float scale, float f[8]) {
    if constexpr (!FP8) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f[2 * j] = __uint_as_float(w[j] << 16);
            f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
        }
    } else {
        f[0] = __builtin_amdgcn_cvt_f32_fp8(v.x, 0) * scale;
        f[1] = __builtin_amdgcn_cvt_f32_fp8(v.x, 1) * scale;
        f[2] = __builtin_amdgcn_cvt_f32_fp8(v.x, 2) * scale;
        f[3] = __builtin_amdgcn_cvt_f32_fp8(v.x, 3) * scale;
        f[4] = __builtin_amdgcn_cvt_f32_fp8(v.y, 0) * scale;
        f[5] = __builtin_amdgcn_cvt_f32_fp8(v.y, 1) * scale;
        f[6] = __builtin_amdgcn_cvt_f32_fp8(v.y, 2) * scale;
        f[7] = __builtin_amdgcn_cvt_f32_fp8(v.y, 3) * scale;
    }
}

template <bool FP8> __device__ __forceinline__ uint32_t load2_raw(const void *base, int64_t idx) {
    if constexpr (!FP8) return *reinterpret_cast<const uint32_t *>((const uint16_t *)base + idx);
    else return *reinterpret_cast<const uint16_t *>((const uint8_t *)base + idx);
}

template <bool FP8> __device__ __forceinline__ void cvt2(uint32_t w, float scale, float &a, float &b) {
    if constexpr (!FP8) {
        a = __uint_as_float(w << 16);
        b = __uint_as_float(w & 0xFFFF0000u);
    } else {
        a = __builtin_amdgcn_cvt_f32_fp8(w, 0) * scale;
        b = __builtin_amdgcn_cvt_f32_fp8(w, 1) * scale;
    }
}

// grid (kv_heads, seqs, partitions); block 64: ONE wave per (sequence, KV head, 64-token partition),
// serving all QPK query heads of the KV head.  Everything the wave needs from HBM is requested up
// front so a partition costs about one memory round trip:
//   phase 1  token per lane: the lane's K row arrives as 16-byte pieces (up to 16 in flight) and is
//            dotted against the q vectors, which are read with scalar loads (wave-uniform addresses);
//   phase 2  softmax over the 64 lanes with butterflies (no LDS);
//   phase 3  dim per lane: V rows (one coalesced row per wave-instruction, requested before phase 1
//            for head_dim <= 128) are weighted by p[t], broadcast from lane t with v_readlane.
template <int HD, int QPK, bool FP8>
__global__ __launch_bounds__(64) void paged_attn_partition_kernel(
    float *__restrict__ tmp_out, float *__restrict__ exp_sums, float *__restrict__ max_logits, const uint16_t *__restrict__ q,
    const void *__restrict__ k_cache, const void *__restrict__ v_cache, const int32_t *__restrict__ block_tables,
    const int32_t *__restrict__ seq_lens, float scale, int num_heads, int num_kv_heads, int max_blocks_per_seq, int block_size,
    int max_parts, int window, const float *__restrict__ k_scale_ptr, const float *__restrict__ v_scale_ptr) {
    __shared__ int64_t tok_base[PART];
    __shared__ __align__(16) float p_s[PART * QPK];  // [token][head] softmax weights

    const int kvh = blockIdx.x, seq = blockIdx.y, part = blockIdx.z;
    const int lane = threadIdx.x;
    const int seq_len = seq_lens[seq];
    if (seq_len == 0) return;
    const int start = (window > 0 && seq_len > window) ? seq_len - window : 0;
    const int t0 = start + part * PART;
    if (t0 >= seq_len) return;
    const int ntok = min(PART, seq_len - t0);
    const float ksc = (FP8 && k_scale_ptr) ? *k_scale_ptr : 1.0f;
    const float vsc = (FP8 && v_scale_ptr) ? *v_scale_ptr : 1.0f;
    const int64_t tok_stride = (int64_t)num_kv_heads * HD;
    const int head0 = kvh * QPK;

    // this lane's token row (lanes past the end re-read the last valid row; their logits are masked)
    const int t = t0 + min(lane, ntok - 1);
    const int blk = block_tables[(int64_t)seq * max_blocks_per_seq + t / block_size];
    const int64_t base = ((int64_t)blk * block_size + t % block_size) * tok_stride + (int64_t)kvh * HD;
    tok_base[lane] = base;

    constexpr int NP = HD >= 128 ? HD / 128 : 1;       // V: pairs of dims per lane
    constexpr int VB = NP == 1 ? 32 : (NP == 2 ? 16 : 8);   // V rows requested per batch (32 registers)
    constexpr bool V_EARLY = true;                      // first V batch is requested before the K pass
    const bool lane_live = (HD >= 128) || (lane < 32);

    __syncthreads();
    // q rows of this KV head's query group: addresses are wave-uniform, so the reads below become
    // scalar loads and the dot products take q from SGPRs (no LDS traffic, no VGPRs for q).
    const uint16_t *qg = q + ((int64_t)seq * num_heads + head0) * HD;

    uint32_t vraw[VB][NP];
    auto load_v_batch = [&](int b0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < VB; j++) {
            const int64_t rb = tok_base[min(b0 + j, ntok - 1)];
#pragma unroll
            for (int p = 0; p < NP; p++) {
                vraw[j][p] = 0;
                if (lane_live) vraw[j][p] = load2_raw<FP8>(v_cache, rb + 2 * lane + 128 * p);
            }
        }
    };
    if (V_EARLY) load_v_batch(0);

    // ---- phase 1
    float acc[QPK];
#pragma unroll
    for (int h = 0; h < QPK; h++) acc[h] = 0.f;
    constexpr int KCH = HD / 8 < 16 ? HD / 8 : 16;  // K pieces in flight
    for (int c0 = 0; c0 < HD / 8; c0 += KCH) {
        typename Raw8<FP8>::T kraw[KCH];
#pragma unroll
        for (int c = 0; c < KCH; c++) kraw[c] = load8_raw<FP8>(k_cache, base + 8 * (c0 + c));
#pragma unroll
        for (int c = 0; c < KCH; c++) {
            float kf[8];
            cvt8<FP8>(kraw[c], ksc, kf);
#pragma unroll
            for (int h = 0; h < QPK; h++) {
                const uint4 qv = *reinterpret_cast<const uint4 *>(qg + h * HD + 8 * (c0 + c));
                const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
                float a = acc[h];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    a = fmaf(kf[2 * j], __uint_as_float(qw[j] << 16), a);
                    a = fmaf(kf[2 * j + 1], __uint_as_float(qw[j] & 0xFFFF0000u), a);
                }
                acc[h] = a;
            }
        }
    }

    // ---- phase 2: p[h] = exp(l - max) / (sum + 1e-6) for this lane's token
    float pr[QPK];
#pragma unroll
    for (int h = 0; h < QPK; h++) {
        const float l = lane < ntok ? acc[h] * scale : -FLT_MAX;
        const float m = vmx_wave_max(l);
        const float e = lane < ntok ? __expf(l - m) : 0.f;
        const float ssum = vmx_wave_sum(e);
        pr[h] = e * (1.0f / (ssum + 1e-6f));
        p_s[lane * QPK + h] = pr[h];
        if (lane == 0) {
            const int64_t mo = ((int64_t)seq * num_heads + head0 + h) * max_parts + part;
            max_logits[mo] = m;
            exp_sums[mo] = ssum;
        }
    }

    __syncthreads();

    // ---- phase 3
    float o[QPK][NP][2];
#pragma unroll
    for (int h = 0; h < QPK; h++)
#pragma unroll
        for (int p = 0; p < NP; p++) o[h][p][0] = o[h][p][1] = 0.f;
#pragma unroll
    for (int b0 = 0; b0 < PART; b0 += VB) {
        if (b0 >= ntok) break;
        if (!V_EARLY || b0 > 0) load_v_batch(b0);
#pragma unroll
        for (int j = 0; j < VB; j++) {
            if (b0 + j < ntok) {  // wave-uniform
                float va[NP], vb[NP];
#pragma unroll
                for (int p = 0; p < NP; p++) cvt2<FP8>(vraw[j][p], vsc, va[p], vb[p]);
#pragma unroll
                for (int h = 0; h < QPK; h++) {
                    const float w = p_s[(b0 + j) * QPK + h];  // same address in every lane: LDS broadcast
#pragma unroll
                    for (int p = 0; p < NP; p++) { o[h][p][0] += w * va[p]; o[h][p][1] += w * vb[p]; }
                }
                asm volatile("" ::: "memory");  // do not hoist the weights of later tokens (register pressure)
            }
        }
    }
    if (lane_live) {
#pragma unroll
        for (int h = 0; h < QPK; h++)
#pragma unroll
            for (int p = 0; p < NP; p++) {
                float *dst = tmp_out + (((int64_t)seq * num_heads + head0 + h) * max_parts + part) * HD + 2 * lane + 128 * p;
                *reinterpret_cast<float2 *>(dst) = make_float2(o[h][p][0], o[h][p][1]);
            }
    }
}

// grid (heads, seqs); block 128.  Merge as paged_attention.cu:840-953 (single partition = copy).
__global__ __launch_bounds__(128) void paged_attn_reduce_kernel(uint16_t *__restrict__ out, const float *__restrict__ tmp_out,
                                                                 const float *__restrict__ exp_sums,
                                                                 const float *__restrict__ max_logits,
                                                                 const int32_t *__restrict__ seq_lens, int num_heads, int head_dim,
                                                                 int max_parts, int window) {
    const int head = blockIdx.x, seq = blockIdx.y, tid = threadIdx.x;
    const int seq_len = seq_lens[seq];
    if (seq_len == 0) return;
    const int eff = (window > 0 && seq_len > window) ? window : seq_len;
    const int np = (eff + PART - 1) / PART;
    const int64_t mo = ((int64_t)seq * num_heads + head) * max_parts;
    const float *tp = tmp_out + mo * head_dim;
    uint16_t *o = out + ((int64_t)seq * num_heads + head) * head_dim;
    if (np == 1) {
        for (int d = tid; d < head_dim; d += 128) o[d] = vmx_f32_to_bf16(tp[d]);
        return;
    }
    float gmax = -FLT_MAX;
    for (int p = 0; p < np; p++) gmax = fmaxf(gmax, max_logits[mo + p]);
    float gsum = 0.f;
    for (int p = 0; p < np; p++) gsum += exp_sums[mo + p] * __expf(max_logits[mo + p] - gmax);
    const float inv = 1.0f / (gsum + 1e-6f);
    for (int d = tid; d < head_dim; d += 128) {
        float acc = 0.f;
        for (int p = 0; p < np; p++) acc += tp[(int64_t)p * head_dim + d] * (exp_sums[mo + p] * __expf(max_logits[mo + p] - gmax));
        o[d] = vmx_f32_to_bf16(acc * inv);
    }
}

template <int HD, int QPK, bool FP8>
int launch_part(dim3 grid, hipStream_t st, float *tmp_out, float *exp_sums, float *max_logits, const uint16_t *q, const void *kc,
                const void *vc, const int32_t *bt, const int32_t *sl, float scale, int H, int KVH, int mbps, int bs, int max_parts,
                int window, const float *ks, const float *vs) {
    hipLaunchKernelGGL((paged_attn_partition_kernel<HD, QPK, FP8>), grid, dim3(64), 0, st, tmp_out, exp_sums, max_logits, q, kc, vc,
                       bt, sl, scale, H, KVH, mbps, bs, max_parts, window, ks, vs);
    VMX_CHECK_LAUNCH("paged_attention(partition)");
    return VMX_OK;
}

static int parts_for_impl(int max_seq_len) { return max_seq_len <= 0 ? 1 : (max_seq_len + PART - 1) / PART; }
}  // namespace


extern "C" size_t vmx_paged_attention_workspace_bytes(int num_seqs, int num_heads, int head_dim, int max_seq_len) {
    if (num_seqs <= 0 || num_heads <= 0 || head_dim <= 0) return 0;
    const size_t P = (size_t)parts_for_impl(max_seq_len);
    return (size_t)num_seqs * num_heads * P * ((size_t)head_dim + 2) * sizeof(float);
}

// Internal: launch only the per-partition kernel; the caller merges the partitions (either with
// paged_attn_reduce_kernel or inside the prologue of the consuming linear).
int vmx_paged_attention_parts(const uint16_t *q, const void *k_cache, const void *v_cache, const int32_t *block_tables,
                              const int32_t *seq_lens, float scale, int num_seqs, int num_heads, int num_kv_heads,
                              int max_blocks_per_seq, int max_seq_len, int head_dim, int block_size, int window, int kv_dtype,
                              const float *k_scale, const float *v_scale, void *workspace, VmxAttnParts *out, hipStream_t st) {
    const int P = parts_for_impl(max_seq_len);
    const int qpk = num_heads / num_kv_heads;
    float *tmp_out = (float *)workspace;
    float *exp_sums = tmp_out + (size_t)num_seqs * num_heads * P * head_dim;
    float *max_logits = exp_sums + (size_t)num_seqs * num_heads * P;
    out->tmp_out = tmp_out;
    out->exp_sums = exp_sums;
    out->max_logits = max_logits;
    out->max_parts = P;
    out->part_tokens = PART;
    const dim3 grid(num_kv_heads, num_seqs, P);
    const bool fp8 = kv_dtype == VMX_KV_FP8_E4M3;
    int rc = VMX_EUNSUPPORTED;
#define ARGS grid, st, tmp_out, exp_sums, max_logits, q, k_cache, v_cache, block_tables, seq_lens, scale, num_heads, num_kv_heads, \
             max_blocks_per_seq, block_size, P, window, k_scale, v_scale
#define BY_QPK(HD)                                                                                                         \
    switch (qpk) {                                                                                                         \
        case 1: rc = fp8 ? launch_part<HD, 1, true>(ARGS) : launch_part<HD, 1, false>(ARGS); break;                        \
        case 2: rc = fp8 ? launch_part<HD, 2, true>(ARGS) : launch_part<HD, 2, false>(ARGS); break;                        \
        case 4: rc = fp8 ? launch_part<HD, 4, true>(ARGS) : launch_part<HD, 4, false>(ARGS); break;                        \
        case 8: rc = fp8 ? launch_part<HD, 8, true>(ARGS) : launch_part<HD, 8, false>(ARGS); break;                        \
        case 16: rc = fp8 ? launch_part<HD, 16, true>(ARGS) : launch_part<HD, 16, false>(ARGS); break;                     \
        default: vmx_set_error("paged_attention: %d query heads per KV head unsupported (1,2,4,8,16)", qpk); return VMX_EUNSUPPORTED; \
    }
    switch (head_dim) {
        case 64: BY_QPK(64) break;
        case 128: BY_QPK(128) break;
        case 256: BY_QPK(256) break;
        case 512: BY_QPK(512) break;
        default:
            vmx_set_error("paged_attention: head_dim %d unsupported (64, 128, 256, 512)", head_dim);
            return VMX_EUNSUPPORTED;
    }
#undef BY_QPK
#undef ARGS
    return rc;
}

int vmx_paged_attention_merge(uint16_t *out, const VmxAttnParts *parts, const int32_t *seq_lens, int num_seqs, int num_heads,
                              int head_dim, int window, hipStream_t st) {
    hipLaunchKernelGGL(paged_attn_reduce_kernel, dim3(num_heads, num_seqs), dim3(128), 0, st, out, parts->tmp_out, parts->exp_sums,
                       parts->max_logits, seq_lens, num_heads, head_dim, parts->max_parts, window);
    VMX_CHECK_LAUNCH("paged_attention(reduce)");
    return VMX_OK;
}

extern "C" int vmx_paged_attention_bf16(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                                        const int32_t *block_tables, const int32_t *seq_lens, float scale, int num_seqs, int num_heads,
                                        int num_kv_heads, int max_blocks_per_seq, int max_seq_len, int head_dim, int block_size,
                                        int window, int kv_dtype, const float *k_scale, const float *v_scale, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    VMX_REQUIRE(out && q && k_cache && v_cache && block_tables && seq_lens && workspace, "paged_attention: null pointer");
    VMX_REQUIRE(num_seqs >= 0 && num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
                "paged_attention: num_heads (%d) must be a multiple of num_kv_heads (%d)", num_heads, num_kv_heads);
    VMX_REQUIRE(block_size > 0 && max_blocks_per_seq > 0 && window >= 0, "paged_attention: bad cache geometry");
    VMX_REQUIRE(kv_dtype == VMX_KV_AUTO || kv_dtype == VMX_KV_FP8_E4M3, "paged_attention: unsupported kv dtype %d", kv_dtype);
    VMX_REQUIRE(workspace_bytes >= vmx_paged_attention_workspace_bytes(num_seqs, num_heads, head_dim, max_seq_len),
                "paged_attention: workspace too small");
    if (num_seqs == 0) return VMX_OK;
    hipStream_t st = vmx_stream(stream);
    VmxAttnParts parts;
    if (int rc = vmx_paged_attention_parts(q, k_cache, v_cache, block_tables, seq_lens, scale, num_seqs, num_heads, num_kv_heads,
                                           max_blocks_per_seq, max_seq_len, head_dim, block_size, window, kv_dtype, k_scale, v_scale,
                                           workspace, &parts, st))
        return rc;
    hipLaunchKernelGGL(paged_attn_reduce_kernel, dim3(num_heads, num_seqs), dim3(128), 0, st, out, parts.tmp_out, parts.exp_sums,
                       parts.max_logits, seq_lens, num_heads, head_dim, parts.max_parts, window);
    VMX_CHECK_LAUNCH("paged_attention(reduce)");
    return VMX_OK;
}

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
#include "vmx_internal.h"
#include <float.h>

namespace {

constexpr int PART = 256;  // tokens per partition = threads per workgroup

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

// grid (kv_heads, seqs, partitions); block 256.  QPK = query heads per KV head (compile time).
// LDS: stat | q f32 [QPK][HD] | logits f32 [QPK][PART]; the cross-wave sums reuse the q/logits region.
template <int HD, int QPK, bool FP8>
__global__ __launch_bounds__(256) void paged_attn_partition_kernel(
    float *__restrict__ tmp_out, float *__restrict__ exp_sums, float *__restrict__ max_logits, const uint16_t *__restrict__ q,
    const void *__restrict__ k_cache, const void *__restrict__ v_cache, const int32_t *__restrict__ block_tables,
    const int32_t *__restrict__ seq_lens, float scale, int num_heads, int num_kv_heads, int max_blocks_per_seq, int block_size,
    int max_parts, int window, const float *__restrict__ k_scale_ptr, const float *__restrict__ v_scale_ptr) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int A_FLOATS = QPK * HD + QPK * PART, B_FLOATS = 4 * QPK * HD;
    float *stat = reinterpret_cast<float *>(smem);   // [QPK][2] max, sum (+pad to 16 B)
    float *q_s = stat + ((2 * QPK + 3) & ~3);        // [QPK][HD]
    float *lg = q_s + QPK * HD;                      // [QPK][PART]
    float *red = q_s;                                // [4][QPK][HD] cross-wave sums; reuses q_s/lg after a barrier
    static_assert(A_FLOATS > 0 && B_FLOATS > 0, "");

    const int kvh = blockIdx.x, seq = blockIdx.y, part = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    for (int i = tid; i < QPK * HD; i += 256)
        q_s[i] = vmx_bf16_to_f32(q[((int64_t)seq * num_heads + head0 + i / HD) * HD + i % HD]);
    __syncthreads();

    // ---- phase 1: token per lane
    {
        float acc[QPK];
#pragma unroll
        for (int h = 0; h < QPK; h++) acc[h] = 0.f;
        if (tid < ntok) {
            const int t = t0 + tid;
            const int blk = block_tables[(int64_t)seq * max_blocks_per_seq + t / block_size];
            const int64_t base = ((int64_t)blk * block_size + t % block_size) * tok_stride + (int64_t)kvh * HD;
#pragma unroll 4
            for (int c = 0; c < HD / 8; c++) {
                float kf[8];
                load8<FP8>(k_cache, base + 8 * c, ksc, kf);
#pragma unroll
                for (int h = 0; h < QPK; h++) {
                    const float4 qa = *reinterpret_cast<const float4 *>(q_s + h * HD + 8 * c);
                    const float4 qb = *reinterpret_cast<const float4 *>(q_s + h * HD + 8 * c + 4);
                    acc[h] += kf[0] * qa.x + kf[1] * qa.y + kf[2] * qa.z + kf[3] * qa.w + kf[4] * qb.x + kf[5] * qb.y +
                              kf[6] * qb.z + kf[7] * qb.w;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < QPK; h++) lg[h * PART + tid] = tid < ntok ? acc[h] * scale : -FLT_MAX;
    }
    __syncthreads();

    // ---- phase 2: softmax per head (wave w takes heads w, w+4, ...)
    for (int h = wave; h < QPK; h += 4) {
        float v[4], m = -FLT_MAX;
#pragma unroll
        for (int j = 0; j < 4; j++) { v[j] = lg[h * PART + lane + 64 * j]; m = fmaxf(m, v[j]); }
        m = vmx_wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float e = (lane + 64 * j < ntok) ? __expf(v[j] - m) : 0.f;
            v[j] = e;
            s += e;
        }
        s = vmx_wave_sum(s);
        const float inv = 1.0f / (s + 1e-6f);
#pragma unroll
        for (int j = 0; j < 4; j++) lg[h * PART + lane + 64 * j] = v[j] * inv;
        if (lane == 0) { stat[2 * h] = m; stat[2 * h + 1] = s; }
    }
    __syncthreads();

    // ---- phase 3: dim per lane; wave w takes tokens w, w+4, ...
    constexpr int DPL = HD / 64;  // dims per lane (HD >= 128: pairs), handled as DPL/2 pairs
    constexpr int NP = DPL >= 2 ? DPL / 2 : 1;
    float o[QPK][NP][2];
#pragma unroll
    for (int h = 0; h < QPK; h++)
#pragma unroll
        for (int p = 0; p < NP; p++) o[h][p][0] = o[h][p][1] = 0.f;
    const bool lane_live = (HD >= 128) || (lane < 32);  // HD = 64: 32 lanes x 2 dims
    for (int i = wave; i < ntok; i += 4) {
        const int t = t0 + i;
        const int blk = block_tables[(int64_t)seq * max_blocks_per_seq + t / block_size];
        const int64_t base = ((int64_t)blk * block_size + t % block_size) * tok_stride + (int64_t)kvh * HD;
        float va[NP], vb[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) {
            va[p] = vb[p] = 0.f;
            if (lane_live) load2<FP8>(v_cache, base + 2 * lane + 128 * p, vsc, va[p], vb[p]);
        }
#pragma unroll
        for (int h = 0; h < QPK; h++) {
            const float pr = lg[h * PART + i];
#pragma unroll
            for (int p = 0; p < NP; p++) { o[h][p][0] += pr * va[p]; o[h][p][1] += pr * vb[p]; }
        }
    }
    __syncthreads();  // every wave is done reading lg before red overwrites it
    if (lane_live) {
#pragma unroll
        for (int h = 0; h < QPK; h++)
#pragma unroll
            for (int p = 0; p < NP; p++) {
                red[(wave * QPK + h) * HD + 2 * lane + 128 * p] = o[h][p][0];
                red[(wave * QPK + h) * HD + 2 * lane + 128 * p + 1] = o[h][p][1];
            }
    }
    __syncthreads();
    for (int i = tid; i < QPK * HD; i += 256) {
        const int h = i / HD, d = i % HD;
        const float v = red[(0 * QPK + h) * HD + d] + red[(1 * QPK + h) * HD + d] + red[(2 * QPK + h) * HD + d] +
                        red[(3 * QPK + h) * HD + d];
        tmp_out[(((int64_t)seq * num_heads + head0 + h) * max_parts + part) * HD + d] = v;
    }
    if (tid < QPK) {
        const int64_t mo = ((int64_t)seq * num_heads + head0 + tid) * max_parts + part;
        max_logits[mo] = stat[2 * tid];
        exp_sums[mo] = stat[2 * tid + 1];
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

template <int HD, int QPK, bool FP8> size_t part_lds_bytes() {
    const size_t a = (size_t)QPK * HD + (size_t)QPK * PART, b = (size_t)4 * QPK * HD;
    return (((size_t)2 * QPK + 3) & ~(size_t)3) * sizeof(float) + (a > b ? a : b) * sizeof(float);
}

template <int HD, int QPK, bool FP8>
int launch_part(dim3 grid, hipStream_t st, float *tmp_out, float *exp_sums, float *max_logits, const uint16_t *q, const void *kc,
                const void *vc, const int32_t *bt, const int32_t *sl, float scale, int H, int KVH, int mbps, int bs, int max_parts,
                int window, const float *ks, const float *vs) {
    const size_t lds = part_lds_bytes<HD, QPK, FP8>();
    if (lds > 64 * 1024)
        VMX_CHECK_HIP(hipFuncSetAttribute((const void *)paged_attn_partition_kernel<HD, QPK, FP8>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((paged_attn_partition_kernel<HD, QPK, FP8>), grid, dim3(256), lds, st, tmp_out, exp_sums, max_logits, q, kc, vc,
                       bt, sl, scale, H, KVH, mbps, bs, max_parts, window, ks, vs);
    VMX_CHECK_LAUNCH("paged_attention(partition)");
    return VMX_OK;
}

}  // namespace

static int parts_for(int max_seq_len) { return max_seq_len <= 0 ? 1 : (max_seq_len + PART - 1) / PART; }

extern "C" size_t vmx_paged_attention_workspace_bytes(int num_seqs, int num_heads, int head_dim, int max_seq_len) {
    if (num_seqs <= 0 || num_heads <= 0 || head_dim <= 0) return 0;
    const size_t P = (size_t)parts_for(max_seq_len);
    return (size_t)num_seqs * num_heads * P * ((size_t)head_dim + 2) * sizeof(float);
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
    const int P = parts_for(max_seq_len);
    const int qpk = num_heads / num_kv_heads;
    float *tmp_out = (float *)workspace;
    float *exp_sums = tmp_out + (size_t)num_seqs * num_heads * P * head_dim;
    float *max_logits = exp_sums + (size_t)num_seqs * num_heads * P;
    hipStream_t st = vmx_stream(stream);
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
    if (rc) return rc;
    hipLaunchKernelGGL(paged_attn_reduce_kernel, dim3(num_heads, num_seqs), dim3(128), 0, st, out, tmp_out, exp_sums, max_logits,
                       seq_lens, num_heads, head_dim, P, window);
    VMX_CHECK_LAUNCH("paged_attention(reduce)");
    return VMX_OK;
}

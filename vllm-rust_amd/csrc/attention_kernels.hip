// Paged-attention decode for gfx950 (replaces paged_attention_v2_* and its reduce kernel,
// crates/core/kernels/paged_attention.cu:558-954).
//
// Semantics kept from the reference: logits (q.k)*scale, softmax with exp(l - max), split of the
// attended range [start, seq_len) into partitions that are normalised locally and merged with
// the log-sum-exp rule; GQA kv_head = head / (H / KVH); sliding window start = seq_len - window;
// FP8 e4m3 caches dequantised as float(byte) * scale.
//
// Structure (gfx950's own): one wave = one (sequence, KV head, partition) and serves ALL query heads
// of that KV head, so each K/V byte is read from HBM once instead of once per query head; QK^T runs
// on the matrix cores over 16-token tiles, PV on the vector ALU with one V row per wave access
// (details at the kernel).  At most 64 partitions per sequence; a second kernel (or the consumer)
// merges them.
#include "vmx_kernels.h"
#include <float.h>
#include <mutex>

// Split-KV partials of vmx_paged_attention_bf16_static: one array per GPU, held until the process ends.  In host code this
// name is only a handle for hipGetSymbolAddress; the device address of each GPU's copy is kept in g_attn_scratch.
constexpr size_t VMX_ATTN_SCRATCH_BYTES = (size_t)32 << 20;
static __device__ __align__(256) unsigned char scratch[VMX_ATTN_SCRATCH_BYTES];
constexpr int VMX_MAX_DEVICES = 64;
static void *g_attn_scratch[VMX_MAX_DEVICES];
static std::mutex g_attn_scratch_mu;

// Called by vmx_init with `device` current: resolve this GPU's copy of the array once.
int vmx_attention_scratch_init(int device) {
    VMX_REQUIRE(device >= 0 && device < VMX_MAX_DEVICES, "vmx_init: device %d beyond the %d this library tracks", device, VMX_MAX_DEVICES);
    std::lock_guard<std::mutex> lk(g_attn_scratch_mu);
    if (g_attn_scratch[device]) return VMX_OK;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
        vmx_set_error("vmx_init: hipGetSymbolAddress(attention scratch) failed on device %d: %s", device, hipGetErrorString(e));
        return VMX_ELAUNCH;
    }
    g_attn_scratch[device] = ptr;
    return VMX_OK;
}

namespace {


// Raw (unconverted) cache accesses, so that many rows can wait in registers cheaply.
// FP8: the cache's element type as a template value -- 0 bf16, 1 FP8 e4m3 bytes, 2 INT8 bytes stored as value + 128
// (load_kv_to_f32<int8_t>, crates/core/kernels/paged_attention.cu:83-95), 3 fp16 with fp16 queries (the paged_attention_*_f16
// variants: load_kv_to_f32<__half>, :59-61), 4 FP8 e5m2 bytes (load_kv_to_f32<fp8_e5m2_byte>, :77-82); callers that know only
// bf16 / e4m3 pass false / true.
template <int FP8> struct Raw8 { using T = uint2; };
template <> struct Raw8<0> { using T = uint4; };
template <> struct Raw8<3> { using T = uint4; };
template <int FP8> constexpr bool kv16() { return FP8 == 0 || FP8 == 3; }
__device__ __forceinline__ float attn_h2f(uint32_t bits16) { return (float)__builtin_bit_cast(_Float16, (uint16_t)bits16); }

template <int FP8> __device__ __forceinline__ typename Raw8<FP8>::T load8_raw(const void *base, int64_t idx) {
    if constexpr (kv16<FP8>()) return *reinterpret_cast<const uint4 *>((const uint16_t *)base + idx);
    else return *reinterpret_cast<const uint2 *>((const uint8_t *)base + idx);
}

template <int FP8> __device__ __forceinline__ void cvt8(const typename Raw8<FP8>::T v, float scale, float f[8]) {
    if constexpr (FP8 == 3) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f[2 * j] = attn_h2f(w[j] & 0xFFFFu);
            f[2 * j + 1] = attn_h2f(w[j] >> 16);
        }
    } else if constexpr (!FP8) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f[2 * j] = __uint_as_float(w[j] << 16);
            f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
        }
    } else if constexpr (FP8 == 2) {
        const uint32_t w[2] = {v.x, v.y};
#pragma unroll
        for (int j = 0; j < 8; j++) f[j] = ((float)((w[j >> 2] >> (8 * (j & 3))) & 0xFFu) - 128.0f) * scale;
    } else if constexpr (FP8 == 4) {
        f[0] = __builtin_amdgcn_cvt_f32_bf8(v.x, 0) * scale;
        f[1] = __builtin_amdgcn_cvt_f32_bf8(v.x, 1) * scale;
        f[2] = __builtin_amdgcn_cvt_f32_bf8(v.x, 2) * scale;
        f[3] = __builtin_amdgcn_cvt_f32_bf8(v.x, 3) * scale;
        f[4] = __builtin_amdgcn_cvt_f32_bf8(v.y, 0) * scale;
        f[5] = __builtin_amdgcn_cvt_f32_bf8(v.y, 1) * scale;
        f[6] = __builtin_amdgcn_cvt_f32_bf8(v.y, 2) * scale;
        f[7] = __builtin_amdgcn_cvt_f32_bf8(v.y, 3) * scale;
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

template <int FP8> __device__ __forceinline__ uint32_t load2_raw(const void *base, int64_t idx) {
    if constexpr (kv16<FP8>()) return *reinterpret_cast<const uint32_t *>((const uint16_t *)base + idx);
    else return *reinterpret_cast<const uint16_t *>((const uint8_t *)base + idx);
}

template <int FP8> __device__ __forceinline__ void cvt2(uint32_t w, float scale, float &a, float &b) {
    if constexpr (FP8 == 3) {
        a = attn_h2f(w & 0xFFFFu);
        b = attn_h2f(w >> 16);
    } else if constexpr (!FP8) {
        a = __uint_as_float(w << 16);
        b = __uint_as_float(w & 0xFFFF0000u);
    } else if constexpr (FP8 == 2) {
        a = ((float)(w & 0xFFu) - 128.0f) * scale;
        b = ((float)((w >> 8) & 0xFFu) - 128.0f) * scale;
    } else if constexpr (FP8 == 4) {
        a = __builtin_amdgcn_cvt_f32_bf8(w, 0) * scale;
        b = __builtin_amdgcn_cvt_f32_bf8(w, 1) * scale;
    } else {
        a = __builtin_amdgcn_cvt_f32_fp8(w, 0) * scale;
        b = __builtin_amdgcn_cvt_f32_fp8(w, 1) * scale;
    }
}

// One 16-token tile of K and V as fetched from the cache (unconverted), waiting in registers.
template <int HD, int FP8> struct AttnTile {
    typename Raw8<FP8>::T kraw[HD / 32];
    uint32_t vraw[16][HD >= 128 ? HD / 128 : 1];
};

// Request tile [t0, min(t0 + 16, p1)): lane (token tl, dim group dg) takes 16-byte pieces of its token's K row, and every
// V row of the tile is one coalesced wave access (row base broadcast from the lane that owns the token).  Lanes past the
// end re-read the last valid row; their logits are masked later.
// BTREG: lane l of bt_reg holds the sequence's block-table entry l (read once at kernel start) and the lookup is a lane
// exchange instead of a dependent global load.
template <int HD, int FP8, bool BTREG = false>
__device__ __forceinline__ void attn_tile_load(AttnTile<HD, FP8> &t, const void *__restrict__ k_cache, const void *__restrict__ v_cache,
                                               const int32_t *__restrict__ bt_row, int t0, int p1, int block_size, int64_t tok_stride,
                                               int kvh, int lane, int bt_reg = 0) {
    constexpr int KSTEPS = HD / 32, NP = HD >= 128 ? HD / 128 : 1;
    const int tl = lane & 15, dg = lane >> 4;
    const bool lane_live = (HD >= 128) || (lane < 32);
    const int ntok = min(16, p1 - t0);
    const int tok = t0 + min(tl, ntok - 1);
    const int blk = BTREG ? __shfl(bt_reg, tok / block_size, 64) : bt_row[tok / block_size];
    const int64_t base = ((int64_t)blk * block_size + tok % block_size) * tok_stride + (int64_t)kvh * HD;
#pragma unroll
    for (int i = 0; i < KSTEPS; i++) t.kraw[i] = load8_raw<FP8>(k_cache, base + 32 * i + 8 * dg);
    const uint32_t base_lo = (uint32_t)base, base_hi = (uint32_t)(base >> 32);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int64_t rb = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)base_hi, j) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)base_lo, j));
#pragma unroll
        for (int p = 0; p < NP; p++) {
            t.vraw[j][p] = 0;
            if (lane_live) t.vraw[j][p] = load2_raw<FP8>(v_cache, rb + 2 * lane + 128 * p);
        }
    }
}

// Fold one tile into the running softmax state.  QK^T: one v_mfma_f32_16x16x32_bf16 per 32 head dims (A = q rows, row =
// head; B = K^T, column = token); softmax over the tile's tokens with 4 butterfly steps inside each 16-lane row group;
// PV on the vector ALU, p[head][token] broadcast with v_readlane.
template <int HD, int QPK, int FP8>
__device__ __forceinline__ void attn_tile_fold(const AttnTile<HD, FP8> &t, int ntok, const uint4 (&qa)[HD / 32], float lscale, float vsc,
                                               int lane, float (&m_run)[4], float (&s_run)[4],
                                               float (&o)[QPK][HD >= 128 ? HD / 128 : 1][2]) {
    constexpr int KSTEPS = HD / 32, NP = HD >= 128 ? HD / 128 : 1;
    const int tl = lane & 15;
    vmx_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KSTEPS; i++) {
        uint4 kb;
        if constexpr (FP8 == 3) {   // fp16 queries and keys: the fp16 form of the same matrix instruction
            typedef _Float16 attn_f16x8 __attribute__((ext_vector_type(8)));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(attn_f16x8, qa[i]), __builtin_bit_cast(attn_f16x8, t.kraw[i]), acc, 0, 0, 0);
            continue;
        } else if constexpr (!FP8) {
            kb = t.kraw[i];
        } else {  // e4m3 -> bf16 is exact
            float f[8];
            cvt8<FP8>(t.kraw[i], 1.0f, f);
            kb.x = (uint32_t)vmx_f32_to_bf16(f[0]) | ((uint32_t)vmx_f32_to_bf16(f[1]) << 16);
            kb.y = (uint32_t)vmx_f32_to_bf16(f[2]) | ((uint32_t)vmx_f32_to_bf16(f[3]) << 16);
            kb.z = (uint32_t)vmx_f32_to_bf16(f[4]) | ((uint32_t)vmx_f32_to_bf16(f[5]) << 16);
            kb.w = (uint32_t)vmx_f32_to_bf16(f[6]) | ((uint32_t)vmx_f32_to_bf16(f[7]) << 16);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, qa[i]), __builtin_bit_cast(vmx_bf16x8, kb), acc, 0, 0, 0);
    }
    float pw[4], corr[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float l = tl < ntok ? acc[r] * lscale : -FLT_MAX;
        const float mx = vmx_row16_max(l);
        const float m_new = fmaxf(m_run[r], mx);
        const float e = tl < ntok ? __expf(l - m_new) : 0.f;
        const float sm = vmx_row16_sum(e);
        corr[r] = __expf(m_run[r] - m_new);  // 0 on the first tile (m_run = -FLT_MAX)
        s_run[r] = s_run[r] * corr[r] + sm;
        m_run[r] = m_new;
        pw[r] = e;
    }
#pragma unroll
    for (int h = 0; h < QPK; h++) {
        const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, corr[h & 3]), 16 * (h >> 2)));
#pragma unroll
        for (int p = 0; p < NP; p++) { o[h][p][0] *= c; o[h][p][1] *= c; }
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (j < ntok) {  // wave-uniform
            float va[NP], vb[NP];
#pragma unroll
            for (int p = 0; p < NP; p++) cvt2<FP8>(t.vraw[j][p], vsc, va[p], vb[p]);
#pragma unroll
            for (int h = 0; h < QPK; h++) {
                const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pw[h & 3]), 16 * (h >> 2) + j));
#pragma unroll
                for (int p = 0; p < NP; p++) { o[h][p][0] += w * va[p]; o[h][p][1] += w * vb[p]; }
            }
        }
    }
}

// Online-softmax walk over the tokens [p0, p1) of one sequence for the QPK query heads of one KV head (one wave).
// qa: the q rows as MFMA A fragments (lane (row = head tl, dim group dg) holds dims [32*i + 8*dg, +8)); bt_row: the
// sequence's block-table row.  `first` holds tile p0 when first_loaded (requested by the caller ahead of other work).
// Returns running max / sum per head row 4*dg + r and the un-normalised output accumulator.
template <int HD, int QPK, int FP8, bool BTREG = false>
__device__ __forceinline__ void attn_walk(const uint4 (&qa)[HD / 32], const void *__restrict__ k_cache, const void *__restrict__ v_cache,
                                          const int32_t *__restrict__ bt_row, int p0, int p1, int block_size, int64_t tok_stride,
                                          int kvh, float lscale, float vsc, int lane, float (&m_run)[4], float (&s_run)[4],
                                          float (&o)[QPK][HD >= 128 ? HD / 128 : 1][2], AttnTile<HD, FP8> &first, bool first_loaded,
                                          int bt_reg = 0) {
    constexpr int NP = HD >= 128 ? HD / 128 : 1;
#pragma unroll
    for (int r = 0; r < 4; r++) { m_run[r] = -FLT_MAX; s_run[r] = 0.f; }
#pragma unroll
    for (int h = 0; h < QPK; h++)
#pragma unroll
        for (int p = 0; p < NP; p++) o[h][p][0] = o[h][p][1] = 0.f;
    if (!first_loaded) attn_tile_load<HD, FP8, BTREG>(first, k_cache, v_cache, bt_row, p0, p1, block_size, tok_stride, kvh, lane, bt_reg);
    for (int t0 = p0; t0 < p1; t0 += 16) {   // the next tile is requested before the current one is folded
        AttnTile<HD, FP8> next;
        const bool more = t0 + 16 < p1;
        if (more) attn_tile_load<HD, FP8, BTREG>(next, k_cache, v_cache, bt_row, t0 + 16, p1, block_size, tok_stride, kvh, lane, bt_reg);
        attn_tile_fold<HD, QPK, FP8>(first, min(16, p1 - t0), qa, lscale, vsc, lane, m_run, s_run, o);
        if (more) first = next;
    }
}

// grid (kv_heads, seqs, partitions); block 64: ONE wave per (sequence, KV head, partition), serving
// all QPK (<= 16) query heads of the KV head.  A partition is `tiles_per_part` tiles of 16 tokens,
// walked with an online softmax (running max / sum / rescaled accumulator).  Per tile:
//   QK^T  one v_mfma_f32_16x16x32_bf16 per 32 head dims: A = the q rows of the group (row = head),
//         B = K^T (column = token); lane (token, dim group) feeds 16-byte pieces of its token's K row
//         straight from the cache, so the 16x16 logit tile costs HD/32 matrix instructions and no
//         cross-lane reduction.  bf16 products are exact in f32, as in the reference's FMA chain.
//   softmax over the tile's 16 tokens with 4 butterfly steps inside each 16-lane row group;
//   PV    dim per lane: a V row is one coalesced wave access; p[head][token] is broadcast from the
//         lane that holds it with v_readlane (wave-uniform, no LDS).
// Everything a tile needs from HBM (K pieces, V rows, q) is requested before its arithmetic starts;
// with several tiles per partition the next tile's block-table entry is fetched one tile ahead.
template <int HD, int QPK, int FP8>
__global__ __launch_bounds__(64) void paged_attn_partition_kernel(
    float *__restrict__ tmp_out, float *__restrict__ exp_sums, float *__restrict__ max_logits, const uint16_t *__restrict__ q,
    const void *__restrict__ k_cache, const void *__restrict__ v_cache, const int32_t *__restrict__ block_tables,
    const int32_t *__restrict__ seq_lens, float scale, int num_heads, int num_kv_heads, int max_blocks_per_seq, int block_size,
    int max_parts, int window, const float *__restrict__ k_scale_ptr, const float *__restrict__ v_scale_ptr, int tiles_per_part, int nq_all,
    int hsplit) {
    // nq = query heads of this wave (<= QPK, the compiled capacity: rows nq..QPK-1 of the A fragment stay zero).  hsplit > 1:
    // the nq_all query heads of a KV head are dealt to hsplit waves (blockIdx.x = kv head * hsplit + group), each re-reading
    // the K/V tiles -- for wide heads with many query heads per KV head and few partitions (Gemma 4 full-attention layers: one
    // 512-wide KV head, 16 query heads, 25 tiles at 400 tokens = 25 waves on the whole chip otherwise)
    constexpr int KSTEPS = HD / 32;                 // MFMA k-steps per tile
    constexpr int NP = HD >= 128 ? HD / 128 : 1;    // V: pairs of dims per lane
    constexpr int HG = (QPK + 3) / 4;               // 16-lane row groups that hold live heads
    const int kvh = blockIdx.x / hsplit, hgrp = blockIdx.x % hsplit, seq = blockIdx.y, part = blockIdx.z;
    const int nq = nq_all / hsplit;
    const int lane = threadIdx.x, tl = lane & 15, dg = lane >> 4;
    const int seq_len = seq_lens[seq];
    if (seq_len == 0) return;
    const int start = (window > 0 && seq_len > window) ? seq_len - window : 0;
    const int part_tokens = tiles_per_part * 16;
    const int p0 = start + part * part_tokens;
    if (p0 >= seq_len) return;
    const int p1 = min(p0 + part_tokens, seq_len);
    const float ksc = (FP8 && k_scale_ptr) ? *k_scale_ptr : 1.0f;
    const float vsc = (FP8 && v_scale_ptr) ? *v_scale_ptr : 1.0f;
    const float lscale = scale * ksc;
    const int64_t tok_stride = (int64_t)num_kv_heads * HD;
    const int head0 = kvh * nq_all + hgrp * nq;
    const bool lane_live = (HD >= 128) || (lane < 32);

    // q rows as MFMA A fragments: lane (row = head tl, dim group dg) holds dims [32*i + 8*dg, +8)
    uint4 qa[KSTEPS];
#pragma unroll
    for (int i = 0; i < KSTEPS; i++) {
        qa[i] = make_uint4(0, 0, 0, 0);
        if (tl < nq) qa[i] = *reinterpret_cast<const uint4 *>(q + ((int64_t)seq * num_heads + head0 + tl) * HD + 32 * i + 8 * dg);
    }

    float m_run[4], s_run[4];  // per head row 4*dg + r (replicated over the 16 lanes of the group)
    float o[QPK][NP][2];
    AttnTile<HD, FP8> tile;
    attn_walk<HD, QPK, FP8>(qa, k_cache, v_cache, block_tables + (int64_t)seq * max_blocks_per_seq, p0, p1, block_size, tok_stride, kvh,
                            lscale, vsc, lane, m_run, s_run, o, tile, false);

    // ---- partition result: locally normalised output, max and sum per head
#pragma unroll
    for (int h = 0; h < QPK; h++) {
        if (h < nq) {   // wave-uniform
            const float sh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s_run[h & 3]), 16 * (h >> 2)));
            const float mh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m_run[h & 3]), 16 * (h >> 2)));
            const float inv = 1.0f / (sh + 1e-6f);
            const int64_t mo = ((int64_t)seq * num_heads + head0 + h) * max_parts + part;
            if (lane == 0) { max_logits[mo] = mh; exp_sums[mo] = sh; }
            if (lane_live) {
#pragma unroll
                for (int p = 0; p < NP; p++)
                    *reinterpret_cast<float2 *>(tmp_out + mo * HD + 2 * lane + 128 * p) = make_float2(o[h][p][0] * inv, o[h][p][1] * inv);
            }
        }
    }
    (void)HG;
}

// The same partition walked by four waves (long contexts: with at most 64 partitions per sequence a partition of a
// 8192-token context is 8 tiles, and one wave per partition leaves the chip with a few hundred waves).  Wave w takes the
// w-th quarter of the partition's tiles; the four states are merged through LDS with the log-sum-exp rule of the reduce
// kernel and the workgroup writes one partition result, so tmp_out / exp_sums / max_logits keep their meaning.
template <int HD, int QPK, int FP8>
__global__ __launch_bounds__(256) void paged_attn_partition4_kernel(
    float *__restrict__ tmp_out, float *__restrict__ exp_sums, float *__restrict__ max_logits, const uint16_t *__restrict__ q,
    const void *__restrict__ k_cache, const void *__restrict__ v_cache, const int32_t *__restrict__ block_tables,
    const int32_t *__restrict__ seq_lens, float scale, int num_heads, int num_kv_heads, int max_blocks_per_seq, int block_size,
    int max_parts, int window, const float *__restrict__ k_scale_ptr, const float *__restrict__ v_scale_ptr, int tiles_per_part, int nq) {
    constexpr int KSTEPS = HD / 32, NP = HD >= 128 ? HD / 128 : 1, NW = 4;
    __shared__ float mo[NW * QPK * HD];
    __shared__ float mm[NW * QPK], ms[NW * QPK];
    const int kvh = blockIdx.x, seq = blockIdx.y, part = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tl = lane & 15, dg = lane >> 4;
    const int seq_len = seq_lens[seq];
    if (seq_len == 0) return;
    const int start = (window > 0 && seq_len > window) ? seq_len - window : 0;
    const int part_tokens = tiles_per_part * 16;
    const int q0 = start + part * part_tokens;
    if (q0 >= seq_len) return;                      // workgroup-uniform
    const int q1 = min(q0 + part_tokens, seq_len);
    const int nt = (q1 - q0 + 15) >> 4, tpw = (nt + NW - 1) / NW;   // tiles of this partition / per wave
    const int p0 = q0 + 16 * wave * tpw, p1 = min(p0 + 16 * tpw, q1);
    const bool active = p0 < q1;
    const float ksc = (FP8 && k_scale_ptr) ? *k_scale_ptr : 1.0f;
    const float vsc = (FP8 && v_scale_ptr) ? *v_scale_ptr : 1.0f;
    const int head0 = kvh * nq;
    const bool lane_live = (HD >= 128) || (lane < 32);
    float m_run[4], s_run[4];
    float o[QPK][NP][2];
    if (active) {
        uint4 qa[KSTEPS];
#pragma unroll
        for (int i = 0; i < KSTEPS; i++) {
            qa[i] = make_uint4(0, 0, 0, 0);
            if (tl < nq) qa[i] = *reinterpret_cast<const uint4 *>(q + ((int64_t)seq * num_heads + head0 + tl) * HD + 32 * i + 8 * dg);
        }
        AttnTile<HD, FP8> tile;
        attn_walk<HD, QPK, FP8>(qa, k_cache, v_cache, block_tables + (int64_t)seq * max_blocks_per_seq, p0, p1, block_size,
                                (int64_t)num_kv_heads * HD, kvh, scale * ksc, vsc, lane, m_run, s_run, o, tile, false);
    }
#pragma unroll
    for (int h = 0; h < QPK; h++) {
        float sh = 0.f, mh = -FLT_MAX;
        if (active) {
            sh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s_run[h & 3]), 16 * (h >> 2)));
            mh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m_run[h & 3]), 16 * (h >> 2)));
        }
        if (lane == 0) { mm[wave * QPK + h] = mh; ms[wave * QPK + h] = sh; }
        if (active && lane_live) {
            const float inv = 1.0f / (sh + 1e-6f);
#pragma unroll
            for (int p = 0; p < NP; p++)
                *reinterpret_cast<float2 *>(mo + (wave * QPK + h) * HD + 2 * lane + 128 * p) = make_float2(o[h][p][0] * inv, o[h][p][1] * inv);
        }
    }
    __syncthreads();
    const int nwv = min(NW, (nt + tpw - 1) / tpw);   // waves that saw tokens
    for (int idx = tid; idx < nq * HD; idx += 64 * NW) {
        const int h = idx / HD, d = idx % HD;
        float val, gmax = -FLT_MAX, gsum = 0.f;
        if (nwv == 1) {
            val = mo[h * HD + d];
            gmax = mm[h];
            gsum = ms[h];
        } else {
            for (int w = 0; w < nwv; w++) gmax = fmaxf(gmax, mm[w * QPK + h]);
            float acc = 0.f;
            for (int w = 0; w < nwv; w++) {
                const float wt = ms[w * QPK + h] * __expf(mm[w * QPK + h] - gmax);
                gsum += wt;
                acc += mo[(w * QPK + h) * HD + d] * wt;
            }
            val = acc * (1.0f / (gsum + 1e-6f));
        }
        const int64_t po = ((int64_t)seq * num_heads + head0 + h) * max_parts + part;
        tmp_out[po * HD + d] = val;
        if (d == 0) { max_logits[po] = gmax; exp_sums[po] = gsum; }
    }
}

// grid (heads, seqs); block 128.  Merge as paged_attention.cu:840-953 (single partition = copy):
// out = sum_p tmp_p * w_p / (sum_p w_p + 1e-6), w_p = exp_sum_p * exp(max_p - gmax).  np <= 64, so the
// weights are computed by one lane each and shared through LDS; the partition rows are then read
// with independent loads.
__global__ __launch_bounds__(512) void paged_attn_reduce_kernel(uint16_t *__restrict__ out, const float *__restrict__ tmp_out,
                                                                 const float *__restrict__ exp_sums,
                                                                 const float *__restrict__ max_logits,
                                                                 const int32_t *__restrict__ seq_lens, int num_heads, int head_dim,
                                                                 int max_parts, int window, int part_tokens, int8_t *__restrict__ aq,
                                                                 float *__restrict__ ad, int out_f16, float *__restrict__ lse = nullptr) {
    // lse (optional) [seqs][heads]: natural-log sum of exp of the head's scaled logits over its whole context, the statistic
    // batched_decode_attention_with_lse hands to the decode-context-parallel merge (layers/attention/backend.rs:99-137)
    __shared__ float w_s[64];
    __shared__ float inv_s;
    __shared__ float acc_s[512];
    const int head = blockIdx.x, seq = blockIdx.y, tid = threadIdx.x;
    const int seq_len = seq_lens[seq];
    if (seq_len == 0) {   // no context: the output row stays as the caller left it; the statistic of an empty sum is -inf
        if (lse && tid == 0) lse[(int64_t)seq * num_heads + head] = -INFINITY;
        return;
    }
    const int eff = (window > 0 && seq_len > window) ? window : seq_len;
    // part_tokens == 0: the producer dealt the tiles to max_parts balanced ranges (split_decode_attn_kernel; a range without
    // tokens carries sum 0 and weighs nothing)
    const int np = part_tokens > 0 ? (eff + part_tokens - 1) / part_tokens : max_parts;
    const int64_t mo = ((int64_t)seq * num_heads + head) * max_parts;
    const float *tp = tmp_out + mo * head_dim;
    const int64_t obase = ((int64_t)seq * num_heads + head) * head_dim;
    uint16_t *o = out + obase;
    // 512 threads = G groups of head_dim threads; group g takes the batches g, g + G, ... of eight partition rows, the first
    // of them requested before the weights are known (the launch is a chain of memory round trips); the groups' sums are
    // added in group order through LDS.
    const int G = blockDim.x / head_dim > 0 ? blockDim.x / head_dim : 1;
    const int g = tid / head_dim, d = tid - g * head_dim;
    const bool in_grid = g < G;                 // blockDim.x not a multiple of head_dim: surplus threads only take the barriers
    float r[8];
#pragma unroll
    for (int u = 0; u < 8; u++) r[u] = (in_grid && np > 1 && 8 * g < np) ? tp[(int64_t)min(8 * g + u, np - 1) * head_dim + d] : 0.f;
    float inv = 1.0f;
    if (np > 1) {
        if (tid < 64) {
            const float mx = tid < np ? max_logits[mo + tid] : -FLT_MAX;
            const float sm = tid < np ? exp_sums[mo + tid] : 0.f;
            const float gmax = vmx_wave_max(mx);
            const float w = tid < np ? sm * __expf(mx - gmax) : 0.f;
            const float gsum = vmx_wave_sum(w);
            w_s[tid] = w;
            if (tid == 0) {
                inv_s = 1.0f / (gsum + 1e-6f);
                if (lse) lse[(int64_t)seq * num_heads + head] = gmax + logf(gsum);
            }
        }
        __syncthreads();
        inv = inv_s;
        float acc = 0.f;
        if (in_grid) {
            for (int p0 = 8 * g; p0 < np; p0 += 8 * G) {   // rows past np were read as the last one and weigh 0 (w_s is zero there)
#pragma unroll
                for (int u = 0; u < 8; u++) acc += r[u] * w_s[p0 + u];
                if (p0 + 8 * G < np) {
#pragma unroll
                    for (int u = 0; u < 8; u++) r[u] = tp[(int64_t)min(p0 + 8 * G + u, np - 1) * head_dim + d];
                }
            }
            acc_s[tid] = acc;
        }
        __syncthreads();
    }
    if (g == 0) {
        float val;
        if (np == 1) {
            val = tp[d];
            if (lse && d == 0) lse[(int64_t)seq * num_heads + head] = max_logits[mo] + logf(exp_sums[mo]);
        } else {
            float acc = 0.f;
            for (int k = 0; k < G; k++) acc += acc_s[k * head_dim + d];
            val = acc * inv;
        }
        const uint16_t vb = out_f16 ? __builtin_bit_cast(uint16_t, (_Float16)val) : vmx_f32_to_bf16(val);
        o[d] = vb;
        if (aq) {   // q8_1 of the bf16 output for an I-quant o_proj: 32 consecutive lanes hold one block (iq_dequant.cu:253-267)
            const float v = vmx_bf16_to_f32(vb);
            float amax = fabsf(v);
            amax = vmx_row32_max(amax);
            const float dq = amax / 127.0f;
            aq[obase + d] = amax != 0.0f ? (int8_t)(int)roundf(v / dq) : (int8_t)0;
            if ((d & 31) == 0) ad[(obase + d) >> 5] = dq;
        }
    }
}

// The same merge for the few balanced ranges of split_decode_attn_kernel (NP <= 8 partitions, every one of them written):
// all addresses are known at entry, so the partition rows, maxima and sums are requested in one go instead of after the
// weights (the kernel is a chain of memory round trips and little else).  grid (heads, seqs), block 128, EPT = head_dim / 128.
template <int NP, int EPT>
__global__ __launch_bounds__(128) void paged_attn_reduce_small_kernel(uint16_t *__restrict__ out, const float *__restrict__ tmp_out,
                                                                       const float *__restrict__ exp_sums,
                                                                       const float *__restrict__ max_logits,
                                                                       const int32_t *__restrict__ seq_lens, int num_heads,
                                                                       int8_t *__restrict__ aq, float *__restrict__ ad) {
    constexpr int HD = 128 * EPT;
    const int head = blockIdx.x, seq = blockIdx.y, tid = threadIdx.x;
    const int64_t mo = ((int64_t)seq * num_heads + head) * NP;
    const float *tp = tmp_out + mo * HD;
    float mx[NP], sm[NP], row[NP][EPT];
    const int seq_len = seq_lens[seq];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        mx[p] = max_logits[mo + p];
        sm[p] = exp_sums[mo + p];
#pragma unroll
        for (int e = 0; e < EPT; e++) row[p][e] = tp[p * HD + tid + 128 * e];
    }
    if (seq_len == 0) return;
    float gmax = -FLT_MAX;
#pragma unroll
    for (int p = 0; p < NP; p++) gmax = fmaxf(gmax, mx[p]);
    float w[NP], gsum = 0.f;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        w[p] = sm[p] * __expf(mx[p] - gmax);
        gsum += w[p];
    }
    const float inv = 1.0f / (gsum + 1e-6f);
    const int64_t obase = ((int64_t)seq * num_heads + head) * HD;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < NP; p++) acc += row[p][e] * w[p];
        const int d = tid + 128 * e;
        const uint16_t vb = vmx_f32_to_bf16(NP == 1 ? row[0][e] : acc * inv);
        out[obase + d] = vb;
        if (aq) {
            const float v = vmx_bf16_to_f32(vb);
            float amax = fabsf(v);
            amax = vmx_row32_max(amax);
            const float dq = amax / 127.0f;
            aq[obase + d] = amax != 0.0f ? (int8_t)(int)roundf(v / dq) : (int8_t)0;
            if ((d & 31) == 0) ad[(obase + d) >> 5] = dq;
        }
    }
}

// Output transform of an EXL3 linear (exl3.rs:256-286) on one 128-wide head held as elements lane, lane + 64: H_128 in f32
// (element bits 0..5 = lane bits, bit 6 = the in-lane pair, in the stage order of the separate transform kernel),
// * 1/sqrt(128), round to fp16, * svh in fp16.
template <int M> __device__ __forceinline__ void attn_had_stage(float v[2], int lane) {
    const float p0 = vmx_lane_xor<M>(v[0]), p1 = vmx_lane_xor<M>(v[1]);
    const bool neg = (lane & M) != 0;
    v[0] = (neg ? -v[0] : v[0]) + p0;
    v[1] = (neg ? -v[1] : v[1]) + p1;
}
__device__ __forceinline__ void attn_exl3_out_transform(float v[2], int lane, uint16_t sv0, uint16_t sv1) {
    attn_had_stage<1>(v, lane);
    attn_had_stage<2>(v, lane);
    attn_had_stage<4>(v, lane);
    attn_had_stage<8>(v, lane);
    attn_had_stage<16>(v, lane);
    attn_had_stage<32>(v, lane);
    const float s0 = v[0] + v[1], d0 = v[0] - v[1];
    v[0] = (float)((_Float16)(s0 * 0.088388347648f) * __builtin_bit_cast(_Float16, sv0));
    v[1] = (float)((_Float16)(d0 * 0.088388347648f) * __builtin_bit_cast(_Float16, sv1));
}

// Sum of the split-K slices of this lane's EPL elements (element j at p[sp * slice + 64 * j]), slices added in index order.
// Eight (four) slices are requested before the first add: a plain loop over a run-time count waits for every load in turn.
template <int EPL>
__device__ __forceinline__ void attn_sum_splits(const float *__restrict__ p, int splits, int64_t slice, float v[EPL]) {
#pragma unroll
    for (int j = 0; j < EPL; j++) v[j] = 0.f;
    constexpr int UB = EPL > 2 ? 4 : 8;   // 256-wide heads: four at a time (registers)
    for (int s0 = 0; s0 < splits; s0 += UB) {
        float t[UB][EPL];
#pragma unroll
        for (int u = 0; u < UB; u++)
#pragma unroll
            for (int j = 0; j < EPL; j++) t[u][j] = p[min(s0 + u, splits - 1) * slice + 64 * j];
#pragma unroll
        for (int u = 0; u < UB; u++)
#pragma unroll
            for (int j = 0; j < EPL; j++) v[j] += s0 + u < splits ? t[u][j] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Short-context decode in ONE launch: grid (kv_heads, seqs), block = NW waves.  A decode step is a chain of dependent
// launches and each link costs ~4-5 us whatever it computes, so for contexts of up to NW*64 tokens the three links
// "q/k/v post-processing -> split-KV attention -> merge" become one:
//   A  waves 0..QPK+1: one head each (QPK query heads, the k head, the v head) from the merged qkv linear output (f32
//      split-K slices): bf16 round, optional per-head RMSNorm, NeoX RoPE (q, k) with the rounding points of the separate
//      kernel; q stays in LDS, k and v go to the paged cache (bf16 or FP8 e4m3);
//   B  after a workgroup barrier (stores of a workgroup are visible to its own waves: one CU, one vector L1) every wave
//      walks its share of [start, seq_len) exactly like a partition of the split kernel;
//   C  the partitions are merged through LDS with the same log-sum-exp rule (paged_attention.cu:840-953).
template <int HD, int QPK, int FP8, int NW>
__global__ __launch_bounds__(64 * NW) void fused_decode_attn_kernel(
    uint16_t *__restrict__ out, const float *__restrict__ partial, int splits, int tokens, void *__restrict__ k_cache,
    void *__restrict__ v_cache, const uint16_t *__restrict__ q_norm_w, const uint16_t *__restrict__ k_norm_w, float eps,
    const float *__restrict__ cos_sin, const int32_t *__restrict__ positions, const int32_t *__restrict__ slots,
    const int32_t *__restrict__ block_tables, const int32_t *__restrict__ seq_lens, float scale, int Hq, int Hkv, int rot_dim,
    int max_blocks_per_seq, int block_size, int window, const float *__restrict__ k_scale_ptr, const float *__restrict__ v_scale_ptr,
    int8_t *__restrict__ aq, float *__restrict__ ad, int v_norm, const uint16_t *__restrict__ had_svh) {
    constexpr int KSTEPS = HD / 32, NP = HD >= 128 ? HD / 128 : 1, EPL = HD / 64;   // EPL: elements per lane in phase A
    const int nq = Hq / Hkv;   // query heads per KV head (<= QPK, the compiled capacity)
    extern __shared__ __align__(16) unsigned char smem_attn[];
    uint16_t *qs = reinterpret_cast<uint16_t *>(smem_attn);                      // [QPK][HD] bf16
    float *mo = reinterpret_cast<float *>(smem_attn + QPK * HD * 2);             // [NW][QPK][HD]
    float *mm = mo + NW * QPK * HD;                                              // [NW][QPK] max
    float *ms = mm + NW * QPK;                                                   // [NW][QPK] sum
    const int kvh = blockIdx.x, seq = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tl = lane & 15, dg = lane >> 4;
    const int seq_len = seq_lens[seq];
    const float ksc = (FP8 && k_scale_ptr) ? *k_scale_ptr : 1.0f;
    const float vsc = (FP8 && v_scale_ptr) ? *v_scale_ptr : 1.0f;
    // Partitions are handed out from the last wave down, so that the waves busy with phase A get the trailing (often
    // empty) ones; a wave without phase-A work whose first tile lies wholly before the new token requests it right away.
    const int start = (window > 0 && seq_len > window) ? seq_len - window : 0;
    const int tiles = (seq_len - start + 15) >> 4;
    const int part_tokens = ((tiles + NW - 1) / NW) * 16;
    const int part = NW - 1 - wave;
    const int p0 = start + part * part_tokens, p1 = min(p0 + part_tokens, seq_len);
    const bool active = seq_len > 0 && p0 < seq_len;
    const int32_t *bt_row = block_tables + (int64_t)seq * max_blocks_per_seq;
    AttnTile<HD, FP8> tile;
    const bool pre = active && wave >= nq + 2 && p0 + 16 < seq_len;
    if (pre) attn_tile_load<HD, FP8>(tile, k_cache, v_cache, bt_row, p0, p1, block_size, (int64_t)Hkv * HD, kvh, lane);

    // ---- A: this token's q / k / v heads
    if (wave < nq + 2) {
        const bool is_q = wave < nq, is_k = wave == nq;
        const int head = is_q ? kvh * nq + wave : (is_k ? Hq + kvh : Hq + Hkv + kvh);
        const int ntot = (Hq + 2 * Hkv) * HD;
        const int64_t slice = (int64_t)tokens * ntot, base = (int64_t)seq * ntot + (int64_t)head * HD;
        float v[EPL];
        float ss = 0.f;
        attn_sum_splits<EPL>(partial + base + lane, splits, slice, v);
        if constexpr (HD == 128) {   // EXL3: `partial` is the trellis GEMM before its output transform; a head is one Hadamard block
            if (had_svh) attn_exl3_out_transform(v, lane, had_svh[(int64_t)head * HD + lane], had_svh[(int64_t)head * HD + lane + 64]);
        }
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            v[j] = vmx_rnd16<FP8 == 3>((v[j]));
            ss += v[j] * v[j];
        }
        const uint16_t *nw = is_q ? q_norm_w : (is_k ? k_norm_w : nullptr);
        if (nw || (!is_q && !is_k && v_norm)) {   // v_norm: RMSNorm without weight on the value head (gemma4_quantized.rs:1311-1424)
            ss = vmx_wave_sum(ss);
            const float inv = rsqrtf(ss / (float)HD + eps);
#pragma unroll
            for (int j = 0; j < EPL; j++)
                v[j] = vmx_rnd16<FP8 == 3>((v[j] * inv * (nw ? vmx_ld16<FP8 == 3>(nw[lane + 64 * j]) : 1.0f)));
        }
        if (is_q || is_k) {   // NeoX rotation of the first rot_dim dims: pairs (i, i + rot_dim/2)
            const int half = rot_dim >> 1;
            const float *cs = cos_sin + (int64_t)positions[seq] * rot_dim;
            if ((half & 63) == 0) {   // partner in the same lane: element j pairs with j + half/64
                const int hj = half >> 6;
#pragma unroll
                for (int j = 0; j < EPL / 2; j++) {
                    if (j < hj) {
                        const int i = lane + 64 * j;
                        const float xv = v[j], yv = v[j + hj], c = cs[i], sn = cs[half + i];
                        v[j] = vmx_rnd16<FP8 == 3>((xv * c - yv * sn));
                        v[j + hj] = vmx_rnd16<FP8 == 3>((yv * c + xv * sn));
                    }
                }
            } else {                  // half == 32 (checked on the host): partner is lane ^ 32 of element 0
                const float other = vmx_lane_xor<32>(v[0]);
                const int i = lane & 31;
                const float c = cs[i], sn = cs[32 + i];
                v[0] = vmx_rnd16<FP8 == 3>((lane < 32 ? v[0] * c - other * sn : v[0] * c + other * sn));
            }
        }
        if (is_q) {
#pragma unroll
            for (int j = 0; j < EPL; j++) qs[wave * HD + lane + 64 * j] = vmx_st16<FP8 == 3>(v[j]);
        } else {
            const int slot = slots[seq];
            if (slot >= 0) {
                const int64_t dst = ((int64_t)slot * Hkv + kvh) * HD;
                void *cache = is_k ? k_cache : v_cache;
                if constexpr (kv16<FP8>()) {
#pragma unroll
                    for (int j = 0; j < EPL; j++) ((uint16_t *)cache)[dst + lane + 64 * j] = vmx_st16<FP8 == 3>(v[j]);
                } else {
                    const float sc = is_k ? ksc : vsc;
#pragma unroll
                    for (int j = 0; j < EPL; j++) {
                        float a = v[j] / sc;
                        a = fminf(fmaxf(a, -448.0f), 448.0f);
                        a = vmx_bf16_to_f32(vmx_f32_to_bf16(a));
                        ((uint8_t *)cache)[dst + lane + 64 * j] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(a, a, 0, false) & 0xFFu);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (seq_len == 0) return;   // uniform over the workgroup

    // ---- B: wave = partition
    float m_run[4], s_run[4];
    float o[QPK][NP][2];
    if (active) {   // wave-uniform
        uint4 qa[KSTEPS];
#pragma unroll
        for (int i = 0; i < KSTEPS; i++) {
            qa[i] = make_uint4(0, 0, 0, 0);
            if (tl < nq) qa[i] = *reinterpret_cast<const uint4 *>(qs + tl * HD + 32 * i + 8 * dg);
        }
        attn_walk<HD, QPK, FP8>(qa, k_cache, v_cache, bt_row, p0, p1, block_size, (int64_t)Hkv * HD, kvh, scale * ksc, vsc, lane, m_run,
                                s_run, o, tile, pre);
    }
    const bool lane_live = (HD >= 128) || (lane < 32);
#pragma unroll
    for (int h = 0; h < QPK; h++) {
        float sh = 0.f, mh = -FLT_MAX;
        if (active) {
            sh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s_run[h & 3]), 16 * (h >> 2)));
            mh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m_run[h & 3]), 16 * (h >> 2)));
        }
        if (lane == 0) { mm[part * QPK + h] = mh; ms[part * QPK + h] = sh; }
        if (active && lane_live) {
            const float inv = 1.0f / (sh + 1e-6f);
#pragma unroll
            for (int p = 0; p < NP; p++)
                *reinterpret_cast<float2 *>(mo + (part * QPK + h) * HD + 2 * lane + 128 * p) = make_float2(o[h][p][0] * inv, o[h][p][1] * inv);
        }
    }
    __syncthreads();

    // ---- C: merge the partitions (a partition that saw no token has sum 0 and contributes nothing)
    const int np = min(NW, (seq_len - start + part_tokens - 1) / part_tokens);
    for (int idx = tid; idx < nq * HD; idx += 64 * NW) {   // nq * HD is a multiple of 64: a wave is all in or all out
        const int h = idx / HD, d = idx % HD;
        float val;
        if (np == 1) {
            val = mo[h * HD + d];
        } else {
            float gmax = -FLT_MAX;
            for (int p = 0; p < np; p++) gmax = fmaxf(gmax, mm[p * QPK + h]);
            float gsum = 0.f, acc = 0.f;
            for (int p = 0; p < np; p++) {
                const float w = ms[p * QPK + h] * __expf(mm[p * QPK + h] - gmax);
                gsum += w;
                acc += mo[(p * QPK + h) * HD + d] * w;
            }
            val = acc * (1.0f / (gsum + 1e-6f));
        }
        const int64_t ob = ((int64_t)seq * Hq + kvh * nq + h) * HD + d;
        const uint16_t vb = vmx_st16<FP8 == 3>(val);
        out[ob] = vb;
        if (aq) {   // q8_1 of the bf16 output for an I-quant o_proj: 32 consecutive lanes hold one block
            const float vq = vmx_ld16<FP8 == 3>(vb);
            float amax = fabsf(vq);
            amax = vmx_row32_max(amax);
            const float dq = amax / 127.0f;
            aq[ob] = amax != 0.0f ? (int8_t)(int)roundf(vq / dq) : (int8_t)0;
            if ((d & 31) == 0) ad[ob >> 5] = dq;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The one-launch kernel above puts a KV head's whole context on ONE compute unit, and a CU can only keep so many cache
// lines in flight: at a few hundred tokens it spends most of its time waiting for K/V rows (measured with in-kernel
// clock stamps: ~5 us of a 13 us launch until the tiles have landed, the rest a chain of dependent round trips).  This
// variant spreads a KV head over P workgroups -- grid (kv_heads, seqs, P), NW waves each, P * kv_heads CUs pulling rows --
// and leaves the merge of the P partition results to the consumer (the o_proj linear merges them while it stages its
// activations: vmx_int4_gemm_resid_attn), so it is still one launch:
//   A  every workgroup derives the q heads of its KV head from the qkv linear's split-K sums (same arithmetic and
//      rounding points as above); only the workgroup whose range holds the new token also derives k and v and writes them
//      to the cache -- no other workgroup reads that row;
//   B  the tiles of [start, seq_len) are dealt to the workgroups in P contiguous ranges (balanced from seq_len), inside a
//      workgroup to its waves in contiguous sub-ranges, each walked with the online softmax of attn_walk;
//   C  the waves' states are merged through LDS and the partition's normalised output, max logit and sum go to
//      tmp_out / max_logits / exp_sums [seq][head][P] (a partition without tokens writes sum 0 and a zero row).
#ifdef VMX_ATTN_STAMPS
// build-time measurement variant: 100 MHz wall-clock stamps of the last split_decode_attn_kernel launch, 8 u64 per wave
__device__ unsigned long long vmx_attn_stamp_buf[4096 * 8];
#define ATTN_STAMP(i)                                                                                   \
    do {                                                                                                \
        unsigned long long t__;                                                                         \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                 \
        if ((threadIdx.x & 63) == 0) {                                                                  \
            const int w__ = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + (threadIdx.x >> 6); \
            if (w__ < 4096) vmx_attn_stamp_buf[w__ * 8 + (i)] = t__;                                    \
        }                                                                                               \
    } while (0)
#else
#define ATTN_STAMP(i) do {} while (0)
#endif
template <int HD, int QPK, int FP8, int NW>
__global__ __launch_bounds__(64 * NW) void split_decode_attn_kernel(
    float *__restrict__ tmp_out, float *__restrict__ exp_sums, float *__restrict__ max_logits, const float *__restrict__ partial, int splits,
    int tokens, void *__restrict__ k_cache, void *__restrict__ v_cache, const uint16_t *__restrict__ q_norm_w,
    const uint16_t *__restrict__ k_norm_w, float eps, const float *__restrict__ cos_sin, const int32_t *__restrict__ positions,
    const int32_t *__restrict__ slots, const int32_t *__restrict__ block_tables, const int32_t *__restrict__ seq_lens, float scale, int Hq,
    int Hkv, int rot_dim, int max_blocks_per_seq, int block_size, int window, const float *__restrict__ k_scale_ptr,
    const float *__restrict__ v_scale_ptr, int v_norm, const uint16_t *__restrict__ had_svh) {
    constexpr int KSTEPS = HD / 32, NP = HD >= 128 ? HD / 128 : 1, EPL = HD / 64;
    const int nq = Hq / Hkv, P = gridDim.z;
    extern __shared__ __align__(16) unsigned char smem_attn[];
    uint16_t *qs = reinterpret_cast<uint16_t *>(smem_attn);                      // [QPK][HD] bf16
    float *mo = reinterpret_cast<float *>(smem_attn + QPK * HD * 2);             // [NW][QPK][HD]
    float *mm = mo + NW * QPK * HD;                                              // [NW][QPK] max
    float *ms = mm + NW * QPK;                                                   // [NW][QPK] sum
    const int kvh = blockIdx.x, seq = blockIdx.y, part = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tl = lane & 15, dg = lane >> 4;
    // The launch is a chain of memory round trips, so whatever has a known address is requested at once: the three per-row
    // scalars, the block-table row (lane l = entry l; tile lookups are then lane exchanges), and below the wave's head of the
    // qkv sums with its norm weights -- none of them waits for seq_len.
    const int seq_len = seq_lens[seq];
    const int pos = positions[seq];
    const int slot = slots[seq];
    const int32_t *bt_row = block_tables + (int64_t)seq * max_blocks_per_seq;
    const int bt_reg = bt_row[min(lane, max_blocks_per_seq - 1)];
    const float ksc = (FP8 && k_scale_ptr) ? *k_scale_ptr : 1.0f;
    const float vsc = (FP8 && v_scale_ptr) ? *v_scale_ptr : 1.0f;

    ATTN_STAMP(0);
    // ---- A (loads): head `wave` of this KV head's q heads, then k, then v (NW >= nq + 2 waves, checked on the host)
    const bool has_head = wave < nq + 2;
    const bool is_q = wave < nq, is_k = wave == nq;
    float v[EPL], nwf[EPL];
    uint16_t sv0 = 0x3c00, sv1 = 0x3c00;   // EXL3 output scales of this lane's two elements (HD == 128)
    {
        const int head = is_q ? kvh * nq + wave : (is_k ? Hq + kvh : Hq + Hkv + kvh);
        if constexpr (HD == 128) {
            if (had_svh) { sv0 = had_svh[(int64_t)(has_head ? head : 0) * HD + lane]; sv1 = had_svh[(int64_t)(has_head ? head : 0) * HD + lane + 64]; }
        }
        const int ntot = (Hq + 2 * Hkv) * HD;
        const int64_t slice = (int64_t)tokens * ntot, base = (int64_t)seq * ntot + (int64_t)(has_head ? head : 0) * HD;
        const uint16_t *nw = is_q ? q_norm_w : (is_k ? k_norm_w : nullptr);
#pragma unroll
        for (int j = 0; j < EPL; j++) nwf[j] = nw ? vmx_ld16<FP8 == 3>(nw[lane + 64 * j]) : 1.0f;
        attn_sum_splits<EPL>(partial + base + lane, splits, slice, v);
    }
    asm volatile("" ::"v"(v[0]));   // the split sums have landed
    ATTN_STAMP(1);
    // rotation factors of this lane's pairs (position known after the first round trip)
    const int half = rot_dim >> 1;
    const float *cs = cos_sin + (int64_t)pos * rot_dim;
    float rc[EPL], rs[EPL];   // (half & 63) == 0: cos / sin of pair j < half/64; half == 32: of pair lane & 31 in [0]
#pragma unroll
    for (int j = 0; j < EPL; j++) { rc[j] = 1.f; rs[j] = 0.f; }
    if ((half & 63) == 0) {
#pragma unroll
        for (int j = 0; j < EPL / 2; j++)
            if (j < (half >> 6)) { rc[j] = cs[lane + 64 * j]; rs[j] = cs[half + lane + 64 * j]; }
    } else {
        rc[0] = cs[lane & 31];
        rs[0] = cs[32 + (lane & 31)];
    }

    // ---- tiles of this workgroup / wave (needs seq_len), and the wave's first tile if it lies before the new token
    const int start = (window > 0 && seq_len > window) ? seq_len - window : 0;
    const int T = (seq_len - start + 15) >> 4;                  // tiles of this sequence
    // this workgroup's tiles: P contiguous ranges of T/P tiles, rounded down or up (fewer tiles than workgroups: one each for
    // the first T); inside the workgroup the same deal over its waves, so no wave walks more than one tile above the average
    const int g0 = T >= P ? (int)((int64_t)part * T / P) : min(part, T), g1 = T >= P ? (int)((int64_t)(part + 1) * T / P) : min(part + 1, T);
    const bool owner = seq_len > 0 && g0 < T && g1 == T;        // its range ends with the tile of the new token
    const int ng = g1 - g0;
    const int w0 = ng >= NW ? g0 + wave * ng / NW : g0 + min(wave, ng), w1 = ng >= NW ? g0 + (wave + 1) * ng / NW : g0 + min(wave + 1, ng);
    const bool active = seq_len > 0 && w0 < w1;
    const int p0 = start + 16 * w0, p1 = min(start + 16 * w1, seq_len);
    // block-table entries in a register: every token of the walk lies in the first 64 blocks
    const bool bt_regs = (seq_len + block_size - 1) / block_size <= 64;
    AttnTile<HD, FP8> tile;
    const bool pre = active && p0 + 16 < seq_len;
    if (pre) {
        if (bt_regs) attn_tile_load<HD, FP8, true>(tile, k_cache, v_cache, bt_row, p0, p1, block_size, (int64_t)Hkv * HD, kvh, lane, bt_reg);
        else attn_tile_load<HD, FP8, false>(tile, k_cache, v_cache, bt_row, p0, p1, block_size, (int64_t)Hkv * HD, kvh, lane);
    }

    ATTN_STAMP(2);
    // ---- A (arithmetic): bf16 round, optional per-head RMSNorm, NeoX RoPE; q -> LDS, k and v -> cache (owner only)
    if (has_head && g0 < T && (is_q || owner)) {
        if constexpr (HD == 128) {   // EXL3: the sums are the trellis GEMM before its output transform; a head is one Hadamard block
            if (had_svh) attn_exl3_out_transform(v, lane, sv0, sv1);
        }
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            v[j] = vmx_rnd16<FP8 == 3>((v[j]));
            ss += v[j] * v[j];
        }
        const bool normed = is_q ? q_norm_w != nullptr : (is_k ? k_norm_w != nullptr : v_norm != 0);
        if (normed) {   // v_norm: RMSNorm without weight on the value head (gemma4_quantized.rs:1311-1424)
            ss = vmx_wave_sum(ss);
            const float inv = rsqrtf(ss / (float)HD + eps);
#pragma unroll
            for (int j = 0; j < EPL; j++) v[j] = vmx_rnd16<FP8 == 3>((v[j] * inv * nwf[j]));
        }
        if (is_q || is_k) {   // NeoX rotation of the first rot_dim dims: pairs (i, i + rot_dim/2)
            if ((half & 63) == 0) {   // partner in the same lane: element j pairs with j + half/64
                const int hj = half >> 6;
#pragma unroll
                for (int j = 0; j < EPL / 2; j++) {
                    if (j < hj) {
                        const float xv = v[j], yv = v[j + hj];
                        v[j] = vmx_rnd16<FP8 == 3>((xv * rc[j] - yv * rs[j]));
                        v[j + hj] = vmx_rnd16<FP8 == 3>((yv * rc[j] + xv * rs[j]));
                    }
                }
            } else {                  // half == 32 (checked on the host): partner is lane ^ 32 of element 0
                const float other = vmx_lane_xor<32>(v[0]);
                v[0] = vmx_rnd16<FP8 == 3>((lane < 32 ? v[0] * rc[0] - other * rs[0] : v[0] * rc[0] + other * rs[0]));
            }
        }
        if (is_q) {
#pragma unroll
            for (int j = 0; j < EPL; j++) qs[wave * HD + lane + 64 * j] = vmx_st16<FP8 == 3>(v[j]);
        } else if (slot >= 0) {
            const int64_t dst = ((int64_t)slot * Hkv + kvh) * HD;
            void *cache = is_k ? k_cache : v_cache;
            if constexpr (kv16<FP8>()) {
#pragma unroll
                for (int j = 0; j < EPL; j++) ((uint16_t *)cache)[dst + lane + 64 * j] = vmx_st16<FP8 == 3>(v[j]);
            } else {
                const float sc = is_k ? ksc : vsc;
#pragma unroll
                for (int j = 0; j < EPL; j++) {
                    float a = v[j] / sc;
                    a = fminf(fmaxf(a, -448.0f), 448.0f);
                    a = vmx_bf16_to_f32(vmx_f32_to_bf16(a));
                    ((uint8_t *)cache)[dst + lane + 64 * j] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(a, a, 0, false) & 0xFFu);
                }
            }
        }
    }
    ATTN_STAMP(3);
    __syncthreads();

    // ---- B
    ATTN_STAMP(4);
    float m_run[4], s_run[4];
    float o[QPK][NP][2];
    if (active) {   // wave-uniform
        uint4 qa[KSTEPS];
#pragma unroll
        for (int i = 0; i < KSTEPS; i++) {
            qa[i] = make_uint4(0, 0, 0, 0);
            if (tl < nq) qa[i] = *reinterpret_cast<const uint4 *>(qs + tl * HD + 32 * i + 8 * dg);
        }
        if (bt_regs)
            attn_walk<HD, QPK, FP8, true>(qa, k_cache, v_cache, bt_row, p0, p1, block_size, (int64_t)Hkv * HD, kvh, scale * ksc, vsc, lane,
                                          m_run, s_run, o, tile, pre, bt_reg);
        else
            attn_walk<HD, QPK, FP8, false>(qa, k_cache, v_cache, bt_row, p0, p1, block_size, (int64_t)Hkv * HD, kvh, scale * ksc, vsc, lane,
                                           m_run, s_run, o, tile, pre);
    }
    asm volatile("" ::"v"(o[0][0][0]), "v"(s_run[0]));
    ATTN_STAMP(5);
    const bool lane_live = (HD >= 128) || (lane < 32);
#pragma unroll
    for (int h = 0; h < QPK; h++) {
        float sh = 0.f, mh = -FLT_MAX;
        if (active) {
            sh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s_run[h & 3]), 16 * (h >> 2)));
            mh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m_run[h & 3]), 16 * (h >> 2)));
        }
        if (lane == 0) { mm[wave * QPK + h] = mh; ms[wave * QPK + h] = sh; }
        if (active && lane_live) {
            const float inv = 1.0f / (sh + 1e-6f);
#pragma unroll
            for (int p = 0; p < NP; p++)
                *reinterpret_cast<float2 *>(mo + (wave * QPK + h) * HD + 2 * lane + 128 * p) = make_float2(o[h][p][0] * inv, o[h][p][1] * inv);
        }
    }
    __syncthreads();

    // ---- C: this workgroup's partition result
    ATTN_STAMP(6);
    const int nwv = min(NW, ng);   // waves that saw tokens (the first ones)
    for (int idx = tid; idx < nq * HD; idx += 64 * NW) {
        const int h = idx / HD, d = idx % HD;
        float val = 0.f, gmax = -FLT_MAX, gsum = 0.f;
        if (nwv == 1) {
            val = mo[h * HD + d];
            gmax = mm[h];
            gsum = ms[h];
        } else if (nwv > 1) {   // every wave's entry is read at once (a loop over nwv waits for each LDS read in turn)
            float mw[NW], sw[NW], ow[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const bool on = w < nwv;
                mw[w] = on ? mm[w * QPK + h] : -FLT_MAX;
                sw[w] = on ? ms[w * QPK + h] : 0.f;
                ow[w] = on ? mo[(w * QPK + h) * HD + d] : 0.f;
            }
#pragma unroll
            for (int w = 0; w < NW; w++) gmax = fmaxf(gmax, mw[w]);
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const float wt = sw[w] * __expf(mw[w] - gmax);
                gsum += wt;
                acc += ow[w] * wt;
            }
            val = acc * (1.0f / (gsum + 1e-6f));
        }
        const int64_t po = ((int64_t)seq * Hq + kvh * nq + h) * P + part;
        tmp_out[po * HD + d] = val;
        if (d == 0) { max_logits[po] = gmax; exp_sums[po] = gsum; }
    }
    ATTN_STAMP(7);
}

// waves per workgroup: 16 where a thread may hold 128 registers' worth of tiles, 8 for 256-wide heads; context bound = 4 tiles
// per wave
constexpr int fused_nw(int hd) { return hd <= 128 ? 16 : 8; }
constexpr int fused_max_ctx(int hd) { return fused_nw(hd) * 64; }

template <int HD, int QPK, int FP8>
int launch_fused(dim3 grid, hipStream_t st, uint16_t *out, const float *partial, int splits, int tokens, void *kc, void *vc,
                 const uint16_t *qn, const uint16_t *kn, float eps, const float *cos_sin, const int32_t *positions, const int32_t *slots,
                 const int32_t *bt, const int32_t *sl, float scale, int Hq, int Hkv, int rot_dim, int mbps, int bs, int window,
                 const float *ks, const float *vs, int8_t *aq, float *ad, int v_norm, const uint16_t *had_svh) {
    constexpr int FUSED_NW = fused_nw(HD);
    const size_t lds = (size_t)QPK * HD * 2 + (size_t)FUSED_NW * QPK * HD * 4 + (size_t)2 * FUSED_NW * QPK * 4;
    if (lds > 64 * 1024)
        VMX_CHECK_HIP(hipFuncSetAttribute((const void *)fused_decode_attn_kernel<HD, QPK, FP8, FUSED_NW>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((fused_decode_attn_kernel<HD, QPK, FP8, FUSED_NW>), grid, dim3(64 * FUSED_NW), lds, st, out, partial, splits, tokens,
                       kc, vc, qn, kn, eps, cos_sin, positions, slots, bt, sl, scale, Hq, Hkv, rot_dim, mbps, bs, window, ks, vs, aq, ad, v_norm,
                       had_svh);
    VMX_CHECK_LAUNCH("fused_decode_attention");
    return VMX_OK;
}

template <int HD, int QPK, int FP8, int NW>
int launch_split(dim3 grid, hipStream_t st, float *tmp_out, float *exp_sums, float *max_logits, const float *partial, int splits, int tokens,
                 void *kc, void *vc, const uint16_t *qn, const uint16_t *kn, float eps, const float *cos_sin, const int32_t *positions,
                 const int32_t *slots, const int32_t *bt, const int32_t *sl, float scale, int Hq, int Hkv, int rot_dim, int mbps, int bs,
                 int window, const float *ks, const float *vs, int v_norm, const uint16_t *had_svh) {
    const size_t lds = (size_t)QPK * HD * 2 + (size_t)NW * QPK * HD * 4 + (size_t)2 * NW * QPK * 4;
    if (lds > 64 * 1024)
        VMX_CHECK_HIP(hipFuncSetAttribute((const void *)split_decode_attn_kernel<HD, QPK, FP8, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds));
    hipLaunchKernelGGL((split_decode_attn_kernel<HD, QPK, FP8, NW>), grid, dim3(64 * NW), lds, st, tmp_out, exp_sums, max_logits, partial,
                       splits, tokens, kc, vc, qn, kn, eps, cos_sin, positions, slots, bt, sl, scale, Hq, Hkv, rot_dim, mbps, bs, window, ks,
                       vs, v_norm, had_svh);
    VMX_CHECK_LAUNCH("split_decode_attention");
    return VMX_OK;
}

template <int HD, int QPK, int FP8>
int launch_part(dim3 grid, hipStream_t st, float *tmp_out, float *exp_sums, float *max_logits, const uint16_t *q, const void *kc,
                const void *vc, const int32_t *bt, const int32_t *sl, float scale, int H, int KVH, int mbps, int bs, int max_parts,
                int window, const float *ks, const float *vs, int tiles_per_part, int hsplit = 1) {
    // four waves per partition from four tiles per partition on (contexts of 4096 tokens and more); 64- and 128-wide heads,
    // bf16 / FP8 cache, up to 8 query heads per KV head
    if constexpr ((HD == 64 || HD == 128) && (FP8 == 0 || FP8 == 1) && QPK <= 8) {
        static const bool off = vmx_tune("VMX_NO_PARTITION4", 0) != 0;
        if (tiles_per_part >= 4 && !off && hsplit == 1) {
            hipLaunchKernelGGL((paged_attn_partition4_kernel<HD, QPK, FP8>), grid, dim3(256), 0, st, tmp_out, exp_sums, max_logits, q, kc, vc,
                               bt, sl, scale, H, KVH, mbps, bs, max_parts, window, ks, vs, tiles_per_part, H / KVH);
            VMX_CHECK_LAUNCH("paged_attention(partition, four waves)");
            return VMX_OK;
        }
    }
    hipLaunchKernelGGL((paged_attn_partition_kernel<HD, QPK, FP8>), grid, dim3(64), 0, st, tmp_out, exp_sums, max_logits, q, kc, vc,
                       bt, sl, scale, H, KVH, mbps, bs, max_parts, window, ks, vs, tiles_per_part, H / KVH, hsplit);
    VMX_CHECK_LAUNCH("paged_attention(partition)");
    return VMX_OK;
}

// Compiled query-heads-per-KV-head capacities are 1, 2, 4, 8, 16; a ratio in between runs on the next one up with the
// surplus rows of the q fragment left zero (kv_head = head / (H / KVH) for any ratio, paged_attention.cu:558-600).
static int qpk_capacity(int ratio) { return ratio <= 1 ? 1 : ratio <= 2 ? 2 : ratio <= 4 ? 4 : ratio <= 8 ? 8 : ratio <= 16 ? 16 : 0; }

// Partition geometry: 16-token tiles; at most 64 partitions per sequence, so a partition is
// ceil(tiles / 64) tiles.
static int tiles_per_part_for(int max_seq_len) {
    const int tiles = max_seq_len <= 0 ? 1 : (max_seq_len + 15) / 16;
    return (tiles + 63) / 64;
}
static int parts_for_impl(int max_seq_len) {
    const int tiles = max_seq_len <= 0 ? 1 : (max_seq_len + 15) / 16, tpp = tiles_per_part_for(max_seq_len);
    return (tiles + tpp - 1) / tpp;
}
// ------------------------------------------------------------------ prefill over the paged cache
// Causal (optionally windowed) attention of q_len new tokens at positions offset .. offset + q_len - 1 over everything the
// paged cache holds for the sequence up to and including themselves (the caller wrote the new K/V rows first, as the
// reference's naive backend does: write, read back, softmax(QK^T * scale + mask) V; layers/attention/naive.rs:38-83).
// Flash-style on the matrix cores.  A workgroup = 4 waves = up to 4 query heads of one KV head x 16-token query tiles;
// KV tiles of 16 tokens are staged through LDS once per workgroup (K row-major, V transposed), then per wave
//   S^T = K Q^T      v_mfma_f32_16x16x32_bf16: A = K rows (row = token), B = the wave's q rows (column = query token);
//                    the accumulator then holds, per lane, 4 consecutive tokens of ONE query row, which is
//   O^T += V^T P^T   exactly the B fragment of v_mfma_f32_16x16x16_bf16 (k = token); A = V^T from LDS (row = head dim).
// O^T keeps a query row in one accumulator column, so the online-softmax rescale is a per-lane multiply.
typedef short vmx_s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 vmx_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 vmx_f16x8 __attribute__((ext_vector_type(8)));

template <int HD, int QPK, int FP8>
__global__ __launch_bounds__(256) void prefill_attn_kernel(uint16_t *__restrict__ out, const uint16_t *__restrict__ q,
                                                            const void *__restrict__ k_cache, const void *__restrict__ v_cache,
                                                            const int32_t *__restrict__ block_table, int q_len, int offset, float scale,
                                                            int num_heads, int num_kv_heads, int block_size, int window,
                                                            const float *__restrict__ k_scale_ptr, const float *__restrict__ v_scale_ptr) {
    constexpr int HPW = QPK < 4 ? QPK : 4;   // query heads per workgroup
    constexpr int TT = 4 / HPW;              // 16-token query tiles per workgroup
    constexpr int KSTEPS = HD / 32, DT = HD / 16;
    constexpr int KROW = HD + 8;             // bf16 per K row in LDS (16 B pad)
    constexpr int VROW = 20;                 // bf16 per V^T row: 16 tokens + pad (rows stay 8-byte aligned)
    __shared__ __align__(16) uint16_t Ks[16 * KROW];
    __shared__ __align__(16) uint16_t Vt[HD * VROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int hh = wave % HPW, tt = wave / HPW;
    constexpr int HGROUPS = QPK / HPW;
    const int kvh = blockIdx.y / HGROUPS, hgrp = blockIdx.y % HGROUPS;
    const int nq = num_heads / num_kv_heads;           // query heads per KV head (<= QPK, the compiled capacity)
    const bool head_ok = hgrp * HPW + hh < nq;         // a wave past the last real head only helps staging the KV tiles
    const int head = kvh * nq + min(hgrp * HPW + hh, nq - 1);
    const int wg_q0 = blockIdx.x * TT * 16, wg_q1 = min(q_len, wg_q0 + TT * 16);
    const int qt0 = wg_q0 + tt * 16;                     // the wave's first query token
    const int qi = qt0 + n;                              // this lane's query row
    const bool q_valid = qi < q_len && head_ok;
    const int qpos = offset + min(qi, q_len - 1);
    const float ksc = (FP8 && k_scale_ptr) ? *k_scale_ptr : 1.0f, vsc = (FP8 && v_scale_ptr) ? *v_scale_ptr : 1.0f;
    const float lscale = scale * ksc;

    uint4 qb[KSTEPS];
#pragma unroll
    for (int c = 0; c < KSTEPS; c++)
        qb[c] = *reinterpret_cast<const uint4 *>(q + ((int64_t)min(qi, q_len - 1) * num_heads + head) * HD + 32 * c + 8 * g);

    const int hi = offset + wg_q1;                                                    // tokens [lo, hi) are needed by the workgroup
    int lo = window > 0 ? max(0, offset + wg_q0 - window + 1) : 0;
    lo &= ~15;
    const bool wave_live = qt0 < q_len && head_ok;
    const int wave_last_pos = offset + min(qt0 + 15, q_len - 1), wave_first_pos = offset + qt0;
    const int64_t tok_stride = (int64_t)num_kv_heads * HD;

    float m_run = -FLT_MAX, l_part = 0.f;
    vmx_f32x4 acc[DT];
#pragma unroll
    for (int d = 0; d < DT; d++) acc[d] = vmx_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t0 = lo; t0 < hi; t0 += 16) {
        __syncthreads();   // the previous tile is no longer read
        for (int c = tid; c < 2 * HD; c += 256) {   // 16 tokens x HD / 8 chunks of 8 elements
            const int tok = c / (HD / 8), dc = c % (HD / 8);
            const int t = min(t0 + tok, hi - 1);    // rows past the end repeat the last valid row (their weights are 0)
            const int blk = block_table[t / block_size];
            const int64_t base = ((int64_t)blk * block_size + t % block_size) * tok_stride + (int64_t)kvh * HD + dc * 8;
            if constexpr (FP8 == 3) {   // fp16 cache, fp16 matrix instructions: the bits go to LDS as they are
                const uint4 kr = load8_raw<FP8>(k_cache, base), vr = load8_raw<FP8>(v_cache, base);
                *reinterpret_cast<uint4 *>(&Ks[tok * KROW + dc * 8]) = kr;
                const uint32_t vw[4] = {vr.x, vr.y, vr.z, vr.w};
#pragma unroll
                for (int e = 0; e < 8; e++) Vt[(dc * 8 + e) * VROW + tok] = (uint16_t)(e & 1 ? vw[e >> 1] >> 16 : vw[e >> 1] & 0xFFFFu);
            } else {
            float kf[8], vf[8];
            cvt8<FP8>(load8_raw<FP8>(k_cache, base), 1.0f, kf);
            cvt8<FP8>(load8_raw<FP8>(v_cache, base), 1.0f, vf);
            uint32_t kw[4];
#pragma unroll
            for (int j = 0; j < 4; j++) kw[j] = (__float_as_uint(kf[2 * j]) >> 16) | (__float_as_uint(kf[2 * j + 1]) & 0xFFFF0000u);
            *reinterpret_cast<uint4 *>(&Ks[tok * KROW + dc * 8]) = make_uint4(kw[0], kw[1], kw[2], kw[3]);
#pragma unroll
            for (int e = 0; e < 8; e++) Vt[(dc * 8 + e) * VROW + tok] = (uint16_t)(__float_as_uint(vf[e]) >> 16);
            }
        }
        __syncthreads();
        // wave-uniform: does this KV tile intersect the wave's causal / window range at all
        const bool need = wave_live && t0 <= wave_last_pos && (window <= 0 || t0 + 15 > wave_first_pos - window);
        if (need) {
            vmx_f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < KSTEPS; c++) {
                const uint4 ka = *reinterpret_cast<const uint4 *>(&Ks[n * KROW + 32 * c + 8 * g]);
                if constexpr (FP8 == 3)
                    sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vmx_f16x8, ka), __builtin_bit_cast(vmx_f16x8, qb[c]), sacc, 0, 0, 0);
                else
                sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, ka), __builtin_bit_cast(vmx_bf16x8, qb[c]), sacc, 0, 0, 0);
            }
            float sv[4], tmax = -FLT_MAX;
            bool ok[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int tpos = t0 + 4 * g + i;
                ok[i] = tpos <= qpos && (window <= 0 || tpos > qpos - window);
                sv[i] = sacc[i] * lscale;
                if (ok[i]) tmax = fmaxf(tmax, sv[i]);
            }
            tmax = fmaxf(tmax, vmx_lane_xor<16>(tmax));
            tmax = fmaxf(tmax, vmx_lane_xor<32>(tmax));
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __expf(m_run - m_new);   // both -FLT_MAX (nothing seen yet): exp(0) = 1 on zero sums
            m_run = m_new;
            float p[4], psum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                p[i] = ok[i] ? __expf(sv[i] - m_new) : 0.f;
                psum += p[i];
            }
            l_part = l_part * alpha + psum;
            vmx_s16x4 pb;
#pragma unroll
            for (int i = 0; i < 4; i++) pb[i] = (short)vmx_st16<FP8 == 3>(p[i]);
#pragma unroll
            for (int d = 0; d < DT; d++) {
                acc[d] *= alpha;
                const vmx_s16x4 va = *reinterpret_cast<const vmx_s16x4 *>(&Vt[(16 * d + n) * VROW + 4 * g]);
                if constexpr (FP8 == 3)
                    acc[d] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(vmx_f16x4, va), __builtin_bit_cast(vmx_f16x4, pb), acc[d], 0, 0, 0);
                else
                acc[d] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va, pb, acc[d], 0, 0, 0);
            }
        }
    }
    float l = l_part + vmx_lane_xor<16>(l_part);
    l += vmx_lane_xor<32>(l);
    if (q_valid) {
        const float inv = vsc / l;
        uint16_t *o = out + ((int64_t)qi * num_heads + head) * HD;
#pragma unroll
        for (int d = 0; d < DT; d++) {   // acc[d][i] = O[query row n][dim 16 d + 4 g + i]
            const uint32_t w0 = (uint32_t)vmx_st16<FP8 == 3>(acc[d][0] * inv) | ((uint32_t)vmx_st16<FP8 == 3>(acc[d][1] * inv) << 16);
            const uint32_t w1 = (uint32_t)vmx_st16<FP8 == 3>(acc[d][2] * inv) | ((uint32_t)vmx_st16<FP8 == 3>(acc[d][3] * inv) << 16);
            *reinterpret_cast<uint2 *>(o + 16 * d + 4 * g) = make_uint2(w0, w1);
        }
    }
}

template <int HD, int QPK, int FP8>
int launch_prefill(hipStream_t st, uint16_t *out, const uint16_t *q, const void *kc, const void *vc, const int32_t *bt, int q_len, int offset,
                   float scale, int H, int KVH, int bs, int window, const float *ks, const float *vs) {
    constexpr int HPW = QPK < 4 ? QPK : 4, TT = 4 / HPW;
    const dim3 grid((q_len + 16 * TT - 1) / (16 * TT), KVH * (QPK / HPW));
    hipLaunchKernelGGL((prefill_attn_kernel<HD, QPK, FP8>), grid, dim3(256), 0, st, out, q, kc, vc, bt, q_len, offset, scale, H, KVH, bs, window,
                       ks, vs);
    VMX_CHECK_LAUNCH("prefill_attention");
    return VMX_OK;
}
}  // namespace

static int prefill_attention_impl(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                                  const int32_t *block_table, int q_len, int seqlen_offset, float scale, int num_heads,
                                  int num_kv_heads, int head_dim, int block_size, int window, int kv_dtype, const float *k_scale,
                                  const float *v_scale, void *stream, bool f16) {
    VMX_REQUIRE(out && q && k_cache && v_cache && block_table, "prefill_attention: null pointer");
    VMX_REQUIRE(q_len >= 0 && seqlen_offset >= 0, "prefill_attention: q_len %d / seqlen_offset %d", q_len, seqlen_offset);
    VMX_REQUIRE(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
                "prefill_attention: num_heads (%d) must be a multiple of num_kv_heads (%d)", num_heads, num_kv_heads);
    VMX_REQUIRE(block_size > 0 && window >= 0, "prefill_attention: bad cache geometry");
    VMX_REQUIRE(kv_dtype == VMX_KV_AUTO || kv_dtype == VMX_KV_FP8_E4M3, "prefill_attention: unsupported kv dtype %d", kv_dtype);
    if (q_len == 0) return VMX_OK;
    hipStream_t st = vmx_stream(stream);
    const int qpk = qpk_capacity(num_heads / num_kv_heads);   // any ratio up to 16 runs on the next compiled capacity
    const bool fp8 = kv_dtype == VMX_KV_FP8_E4M3;
    int rc = vmx_prefill_attention_tiled(out, q, k_cache, v_cache, block_table, q_len, seqlen_offset, scale, num_heads, num_kv_heads, head_dim,
                                         block_size, window, f16 ? 3 : (fp8 ? 1 : 0), k_scale, v_scale, st);   // 64- / 128-wide heads
    if (rc != VMX_EUNSUPPORTED) return rc;
#define ARGS st, out, q, k_cache, v_cache, block_table, q_len, seqlen_offset, scale, num_heads, num_kv_heads, block_size, window, k_scale, v_scale
#define PF_ONE(HD, Q) rc = f16 ? launch_prefill<HD, Q, 3>(ARGS) : fp8 ? launch_prefill<HD, Q, 1>(ARGS) : launch_prefill<HD, Q, 0>(ARGS)
#define BY_QPK(HD)                                                                                                          \
    switch (qpk) {                                                                                                          \
        case 1: PF_ONE(HD, 1); break;                                                                                       \
        case 2: PF_ONE(HD, 2); break;                                                                                       \
        case 4: PF_ONE(HD, 4); break;                                                                                       \
        case 8: PF_ONE(HD, 8); break;                                                                                       \
        case 16: PF_ONE(HD, 16); break;                                                                                     \
        default: vmx_set_error("prefill_attention: %d query heads per KV head unsupported (1..16)", num_heads / num_kv_heads); return VMX_EUNSUPPORTED; \
    }
    switch (head_dim) {
        case 64: BY_QPK(64) break;
        case 128: BY_QPK(128) break;
        case 256: BY_QPK(256) break;
        case 512: BY_QPK(512) break;
        default:
            vmx_set_error("prefill_attention: head_dim %d unsupported (64, 128, 256, 512)", head_dim);
            return VMX_EUNSUPPORTED;
    }
#undef BY_QPK
#undef PF_ONE
#undef ARGS
    return rc;
}

extern "C" int vmx_prefill_attention_bf16(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                                          const int32_t *block_table, int q_len, int seqlen_offset, float scale, int num_heads,
                                          int num_kv_heads, int head_dim, int block_size, int window, int kv_dtype, const float *k_scale,
                                          const float *v_scale, void *stream) {
    return prefill_attention_impl(out, q, k_cache, v_cache, block_table, q_len, seqlen_offset, scale, num_heads, num_kv_heads, head_dim,
                                  block_size, window, kv_dtype, k_scale, v_scale, stream, false);
}

// The same over fp16 queries and an fp16 cache, fp16 out (models that run in fp16): the fp16 forms of the matrix instructions,
// softmax weights rounded to fp16 where the bf16 form rounds them to bf16.
extern "C" int vmx_prefill_attention_f16(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                                         const int32_t *block_table, int q_len, int seqlen_offset, float scale, int num_heads,
                                         int num_kv_heads, int head_dim, int block_size, int window, void *stream) {
    return prefill_attention_impl(out, q, k_cache, v_cache, block_table, q_len, seqlen_offset, scale, num_heads, num_kv_heads, head_dim,
                                  block_size, window, VMX_KV_AUTO, nullptr, nullptr, stream, true);
}


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
    // wide heads, many query heads per KV head, few waves: the query heads of a KV head are dealt to four waves
    static const bool no_hsplit = vmx_tune("VMX_NO_ATTN_HEAD_SPLIT", 0) != 0;
    const int ratio = num_heads / num_kv_heads;
    const int hsplit = (!no_hsplit && head_dim >= 256 && ratio >= 8 && ratio % 4 == 0 && (int64_t)num_kv_heads * num_seqs * P < 1024) ? 4 : 1;
    const int qpk = qpk_capacity(ratio / hsplit);
    float *tmp_out = (float *)workspace;
    float *exp_sums = tmp_out + (size_t)num_seqs * num_heads * P * head_dim;
    float *max_logits = exp_sums + (size_t)num_seqs * num_heads * P;
    out->tmp_out = tmp_out;
    out->exp_sums = exp_sums;
    out->max_logits = max_logits;
    out->max_parts = P;
    const int tpp = tiles_per_part_for(max_seq_len);
    out->part_tokens = tpp * 16;
    const dim3 grid(num_kv_heads * hsplit, num_seqs, P);
    const int kvt = kv_dtype == VMX_KV_F16_INTERNAL ? 3 : (kv_dtype == VMX_KV_INT8 ? 2 : (kv_dtype == VMX_KV_FP8_E4M3 ? 1 : (kv_dtype == VMX_KV_FP8_E5M2 ? 4 : 0)));
    int rc = VMX_EUNSUPPORTED;
#define ARGS grid, st, tmp_out, exp_sums, max_logits, q, k_cache, v_cache, block_tables, seq_lens, scale, num_heads, num_kv_heads, \
             max_blocks_per_seq, block_size, P, window, k_scale, v_scale, tpp, hsplit
#define BY_KV(HD, Q)                                                                                                       \
    rc = kvt == 4 ? launch_part<HD, Q, 4>(ARGS) : kvt == 3 ? launch_part<HD, Q, 3>(ARGS)                                     \
                  : (kvt == 2 ? launch_part<HD, Q, 2>(ARGS) : (kvt == 1 ? launch_part<HD, Q, 1>(ARGS) : launch_part<HD, Q, 0>(ARGS)))
#define BY_QPK(HD)                                                                                                         \
    switch (qpk) {                                                                                                         \
        case 1: BY_KV(HD, 1); break;                                                                                       \
        case 2: BY_KV(HD, 2); break;                                                                                       \
        case 4: BY_KV(HD, 4); break;                                                                                       \
        case 8: BY_KV(HD, 8); break;                                                                                       \
        case 16: BY_KV(HD, 16); break;                                                                                     \
        default: vmx_set_error("paged_attention: %d query heads per KV head unsupported (1..16)", num_heads / num_kv_heads); return VMX_EUNSUPPORTED; \
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
#undef BY_KV
#undef ARGS
    return rc;
}

int vmx_paged_attention_merge(uint16_t *out, const VmxAttnParts *parts, const int32_t *seq_lens, int num_seqs, int num_heads,
                              int head_dim, int window, hipStream_t st, int8_t *aq, float *ad, int out_f16) {
    const dim3 grid(num_heads, num_seqs);
#define SMALL(NP, EPT)                                                                                                                  \
    hipLaunchKernelGGL((paged_attn_reduce_small_kernel<NP, EPT>), grid, dim3(128), 0, st, out, parts->tmp_out, parts->exp_sums,          \
                       parts->max_logits, seq_lens, num_heads, aq, ad)
    if (out_f16)
        hipLaunchKernelGGL(paged_attn_reduce_kernel, grid, dim3(512), 0, st, out, parts->tmp_out, parts->exp_sums, parts->max_logits,
                           seq_lens, num_heads, head_dim, parts->max_parts, window, parts->part_tokens, aq, ad, 1);
    else if (parts->part_tokens == 0 && parts->max_parts == 4 && head_dim == 256) SMALL(4, 2);
    else if (parts->part_tokens == 0 && parts->max_parts == 4 && head_dim == 128) SMALL(4, 1);
    else if (parts->part_tokens == 0 && parts->max_parts == 2 && head_dim == 256) SMALL(2, 2);
    else if (parts->part_tokens == 0 && parts->max_parts == 8 && head_dim == 256) SMALL(8, 2);
    else
        hipLaunchKernelGGL(paged_attn_reduce_kernel, grid, dim3(512), 0, st, out, parts->tmp_out, parts->exp_sums, parts->max_logits,
                           seq_lens, num_heads, head_dim, parts->max_parts, window, parts->part_tokens, aq, ad, 0);
#undef SMALL
    VMX_CHECK_LAUNCH("paged_attention(reduce)");
    return VMX_OK;
}

// Whether the one-launch short-context path covers this geometry (otherwise: qkv post kernel + parts + merge).
static int g_fused_attn_enabled = -1;   // -1: not decided yet (environment VMX_NO_FUSED_ATTN), 0 off, 1 on

// Measurement / parity aid: switch the one-launch short-context attention off (0) or on (1) for the calls that follow, so
// that both forms can be compared on the same inputs in one process.
#ifdef VMX_ATTN_STAMPS
extern "C" int vmx_debug_attn_stamps_read(void *dst, size_t bytes) {
    VMX_CHECK_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(vmx_attn_stamp_buf), bytes < sizeof(vmx_attn_stamp_buf) ? bytes : sizeof(vmx_attn_stamp_buf)));
    return VMX_OK;
}
#endif

extern "C" int vmx_debug_set_fused_attention(int enable) {
    g_fused_attn_enabled = enable ? 1 : 0;
    return VMX_OK;
}

bool vmx_fused_decode_attention_ok(int head_dim, int qpk, int rot_dim, int max_seq_len, int window) {
    if (g_fused_attn_enabled < 0) g_fused_attn_enabled = (vmx_tune("VMX_NO_FUSED_ATTN", 0) != 0) ? 0 : 1;
    if (!g_fused_attn_enabled) return false;
    // 256-wide heads were measured slower here than the three-launch path (8 waves, 3+ tiles each): not covered
    if (head_dim != 64 && head_dim != 128) return false;
    if (qpk < 1 || qpk > 8) return false;
    const int half = rot_dim / 2;
    if (rot_dim <= 0 || rot_dim > head_dim || !((half % 64) == 0 || (half == 32 && head_dim == 64))) return false;
    const int ctx = window > 0 && window < max_seq_len ? window : max_seq_len;
    return ctx > 0 && ctx <= fused_max_ctx(head_dim);
}

int vmx_fused_decode_attention(uint16_t *out, const float *qkv_partial, int splits, void *k_cache, void *v_cache,
                               const uint16_t *q_norm_w, const uint16_t *k_norm_w, float eps, const float *cos_sin,
                               const int32_t *positions, const int32_t *slots, const int32_t *block_tables, const int32_t *seq_lens,
                               float scale, int num_seqs, int num_heads, int num_kv_heads, int head_dim, int rot_dim,
                               int max_blocks_per_seq, int block_size, int window, int kv_dtype, const float *k_scale,
                               const float *v_scale, int8_t *aq, float *ad, int v_norm, hipStream_t st, const uint16_t *had_svh) {
    const int qpk = qpk_capacity(num_heads / num_kv_heads);
    const dim3 grid(num_kv_heads, num_seqs);
    const bool fp8 = kv_dtype == VMX_KV_FP8_E4M3, f16 = kv_dtype == VMX_KV_F16_INTERNAL;   // f16: fp16 activations and cache
    VMX_REQUIRE(!had_svh || head_dim == 128, "fused_decode_attention: the output Hadamard form needs 128-wide heads");
    int rc = VMX_EUNSUPPORTED;
#define FARGS grid, st, out, qkv_partial, splits, num_seqs, k_cache, v_cache, q_norm_w, k_norm_w, eps, cos_sin, positions, slots, block_tables, \
              seq_lens, scale, num_heads, num_kv_heads, rot_dim, max_blocks_per_seq, block_size, window, k_scale, v_scale, aq, ad, v_norm, had_svh
#define F_QPK(HD)                                                                                         \
    switch (qpk) {                                                                                        \
        case 1: rc = f16 ? launch_fused<HD, 1, 3>(FARGS) : fp8 ? launch_fused<HD, 1, 1>(FARGS) : launch_fused<HD, 1, 0>(FARGS); break;   \
        case 2: rc = f16 ? launch_fused<HD, 2, 3>(FARGS) : fp8 ? launch_fused<HD, 2, 1>(FARGS) : launch_fused<HD, 2, 0>(FARGS); break;   \
        case 4: rc = f16 ? launch_fused<HD, 4, 3>(FARGS) : fp8 ? launch_fused<HD, 4, 1>(FARGS) : launch_fused<HD, 4, 0>(FARGS); break;   \
        case 8: rc = f16 ? launch_fused<HD, 8, 3>(FARGS) : fp8 ? launch_fused<HD, 8, 1>(FARGS) : launch_fused<HD, 8, 0>(FARGS); break;   \
        default: break;                                                                                   \
    }
    if (head_dim == 64) { F_QPK(64) } else if (head_dim == 128) { F_QPK(128) }
#undef F_QPK
#undef FARGS
    if (rc == VMX_EUNSUPPORTED)
        vmx_set_error("fused_decode_attention: geometry not covered (head_dim %d, %d query heads per KV head)", head_dim, num_heads / num_kv_heads);
    return rc;
}

// ---- decode attention spread over P workgroups per KV head, merged by the consumer (split_decode_attn_kernel)
// Partition count for a context bound, 0 = not covered.  VMX_SPLIT_ATTN_P overrides the count (measurement), VMX_NO_SPLIT_ATTN=1 switches the path off.
// Context length from which the many-partition form with its own merge launch takes over from the few-partition form whose
// partitions the consuming linear merges (VMX_SPLIT_LONG_MIN_CTX overrides, for measurement)
static int split_long_min_ctx() {
    // Qwen3-8B AWQ, one row, same box: 300 tokens 563 (few partitions) vs 554 tok/s (many), 600 tokens 540 vs 551, 900 tokens 531 vs 550
    static const int v = vmx_tune("VMX_SPLIT_LONG_MIN_CTX", 512);
    return v;
}
static int split_parts_impl(int head_dim, int qpk, int rot_dim, int max_seq_len, int window, bool wide_heads) {
    static const bool off = vmx_tune("VMX_NO_SPLIT_ATTN", 0) != 0;
    static const int p_env = vmx_tune("VMX_SPLIT_ATTN_P", 0);
    const bool hd_ok = head_dim == 64 || head_dim == 128 || (wide_heads && head_dim == 256 && qpk <= 2);
    if (off || !hd_ok || qpk < 1 || qpk > 8) return 0;
    const int half = rot_dim / 2;
    if (rot_dim <= 0 || rot_dim > head_dim || !((half % 64) == 0 || (half == 32 && head_dim == 64))) return 0;
    const int ctx = window > 0 && window < max_seq_len ? window : max_seq_len;
    // Measured on MI355X (Qwen3-8B, batch 1): up to 1024 tokens 4 workgroups per KV head beat both the one-workgroup kernel
    // and the three-launch path (560 vs 530 tok/s at 300 tokens); beyond that the three-launch path with its 64 partitions
    // per sequence wins (2048 tokens: 478 vs 460 tok/s with 8 workgroups here), because the consumer's merge grows with P.
    // beyond that the many-partition form takes over where it exists (64- / 128-wide heads); 256-wide heads keep this one up to 1024
    if (ctx <= 0 || ctx > 1024 || (ctx > split_long_min_ctx() && (head_dim == 64 || head_dim == 128))) return 0;
    // 256-wide heads (Gemma 4 sliding layers, merged by paged_attn_reduce_small_kernel): 8 workgroups per KV head measured best
    // (Gemma-4-12B shape, 300-400 tokens: 2 / 4 / 8 workgroups = 297 / 305 / 310 tok/s, three-launch path 295)
    return p_env >= 1 && p_env <= 8 ? p_env : (head_dim == 256 ? 8 : 4);
}

int vmx_split_decode_attention_parts(int head_dim, int qpk, int rot_dim, int max_seq_len, int window) {
    return split_parts_impl(head_dim, qpk, rot_dim, max_seq_len, window, false);
}

// As above, also covering 256-wide heads with one or two query heads per KV head (Gemma 4 sliding layers), for callers that
// merge the partitions with vmx_paged_attention_merge rather than inside a linear's prologue.
int vmx_split_decode_attention_parts_wide(int head_dim, int qpk, int rot_dim, int max_seq_len, int window) {
    return split_parts_impl(head_dim, qpk, rot_dim, max_seq_len, window, true);
}

size_t vmx_split_decode_attention_workspace_bytes(int num_seqs, int num_heads, int head_dim) {
    return (size_t)num_seqs * num_heads * 8 * ((size_t)head_dim + 2) * 4;
}

// Contexts beyond the reach of the few-partition form (from 512 tokens on): the same kernel with 8 ... 64 workgroups of 8 waves per KV head (as many
// as give every compute unit one workgroup) and the separate merge (vmx_paged_attention_merge) behind it, in place of
// "q/k/v post-processing + one wave per partition + merge": at 8192 tokens the one-wave partitions put 464 waves on the
// chip for 24 us per layer (Qwen3-8B, one row).  Returns the partition count, 0 = not covered.  VMX_NO_SPLIT_ATTN_LONG=1
// switches the form off.
int vmx_split_decode_attention_long_parts(int head_dim, int qpk, int rot_dim, int max_seq_len, int window, int num_kv_heads, int num_seqs) {
    static const bool off = vmx_tune("VMX_NO_SPLIT_ATTN_LONG", 0) != 0;
    if (off || !(head_dim == 64 || head_dim == 128) || qpk < 1 || qpk > 8) return 0;
    const int half = rot_dim / 2;
    if (rot_dim <= 0 || rot_dim > head_dim || !((half % 64) == 0 || (half == 32 && head_dim == 64))) return 0;
    const int ctx = window > 0 && window < max_seq_len ? window : max_seq_len;
    if (ctx <= split_long_min_ctx()) return 0;
    static const int p_env = vmx_tune("VMX_SPLIT_LONG_P", 0);   // measurement: force the count
    if (p_env >= 8 && p_env <= 64) return p_env;
    // one workgroup per compute unit: measured on Qwen3-8B (8 KV heads, one row) at 1024 ... 32768 tokens, 32 partitions beat
    // 12 / 16 / 24 / 64 at every length (8192 tokens: 420 / 440 / 475 / 441 against 478 tok/s; 33 partitions = 264 workgroups
    // on 256 compute units cost 6 %)
    const int per = vmx_num_cus() / (num_kv_heads * num_seqs > 0 ? num_kv_heads * num_seqs : 1);
    return per < 8 ? 8 : (per > 64 ? 64 : per);
}
size_t vmx_split_decode_attention_long_workspace_bytes(int num_seqs, int num_heads, int head_dim, int parts) {
    return (size_t)num_seqs * num_heads * parts * ((size_t)head_dim + 2) * 4;
}

int vmx_split_decode_attention(VmxAttnParts *out, void *workspace, const float *qkv_partial, int splits, void *k_cache, void *v_cache,
                               const uint16_t *q_norm_w, const uint16_t *k_norm_w, float eps, const float *cos_sin,
                               const int32_t *positions, const int32_t *slots, const int32_t *block_tables, const int32_t *seq_lens,
                               float scale, int num_seqs, int num_heads, int num_kv_heads, int head_dim, int rot_dim,
                               int max_blocks_per_seq, int max_seq_len, int block_size, int window, int kv_dtype, const float *k_scale,
                               const float *v_scale, int v_norm, hipStream_t st, const uint16_t *had_svh, int long_parts) {
    VMX_REQUIRE(!had_svh || head_dim == 128, "split_decode_attention: the output Hadamard form needs 128-wide heads");
    const int qpk = qpk_capacity(num_heads / num_kv_heads);
    const int P = long_parts > 0 ? long_parts : split_parts_impl(head_dim, num_heads / num_kv_heads, rot_dim, max_seq_len, window, true);
    VMX_REQUIRE(P > 0 && workspace && out, "split_decode_attention: geometry not covered (head_dim %d, %d query heads per KV head, context %d)",
                head_dim, num_heads / num_kv_heads, max_seq_len);
    const int ctx = window > 0 && window < max_seq_len ? window : max_seq_len;
    const bool wide = ctx > 1024 && long_parts <= 0;   // 16 waves per workgroup for the long contexts, else 8 (one per head of phase A)
    const size_t rows = (size_t)num_seqs * num_heads * P;
    out->tmp_out = (float *)workspace;
    out->exp_sums = out->tmp_out + rows * head_dim;
    out->max_logits = out->exp_sums + rows;
    out->max_parts = P;
    out->part_tokens = 0;
    const dim3 grid(num_kv_heads, num_seqs, P);
    const bool fp8 = kv_dtype == VMX_KV_FP8_E4M3, f16 = kv_dtype == VMX_KV_F16_INTERNAL;   // f16: fp16 activations and cache
    int rc = VMX_EUNSUPPORTED;
#define SARGS grid, st, out->tmp_out, out->exp_sums, out->max_logits, qkv_partial, splits, num_seqs, k_cache, v_cache, q_norm_w, k_norm_w, eps, \
              cos_sin, positions, slots, block_tables, seq_lens, scale, num_heads, num_kv_heads, rot_dim, max_blocks_per_seq, block_size,     \
              window, k_scale, v_scale, v_norm, had_svh
#define S_ONE(HD, Q)                                                                                                          \
    rc = wide ? (f16 ? launch_split<HD, Q, 3, 16>(SARGS) : fp8 ? launch_split<HD, Q, 1, 16>(SARGS) : launch_split<HD, Q, 0, 16>(SARGS))       \
              : (f16   ? launch_split<HD, Q, 3, (Q + 2 > 8 ? 16 : 8)>(SARGS)                                                                 \
                 : fp8 ? launch_split<HD, Q, 1, (Q + 2 > 8 ? 16 : 8)>(SARGS)                                                                 \
                       : launch_split<HD, Q, 0, (Q + 2 > 8 ? 16 : 8)>(SARGS))
#define S_QPK(HD)                       \
    switch (qpk) {                      \
        case 1: S_ONE(HD, 1); break;    \
        case 2: S_ONE(HD, 2); break;    \
        case 4: S_ONE(HD, 4); break;    \
        case 8: S_ONE(HD, 8); break;    \
        default: break;                 \
    }
    if (head_dim == 64) { S_QPK(64) } else if (head_dim == 128) { S_QPK(128) }
    else if (head_dim == 256 && !wide && !f16) {   // 8 waves only: the 16-wave form would spill at 128 registers per lane
        if (qpk == 1) rc = fp8 ? launch_split<256, 1, true, 8>(SARGS) : launch_split<256, 1, false, 8>(SARGS);
        else if (qpk == 2) rc = fp8 ? launch_split<256, 2, true, 8>(SARGS) : launch_split<256, 2, false, 8>(SARGS);
    }
#undef S_QPK
#undef S_ONE
#undef SARGS
    if (rc == VMX_EUNSUPPORTED) vmx_set_error("split_decode_attention: geometry not covered");
    return rc;
}

static int paged_attention_impl(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                                const int32_t *block_tables, const int32_t *seq_lens, float scale, int num_seqs, int num_heads,
                                int num_kv_heads, int max_blocks_per_seq, int max_seq_len, int head_dim, int block_size,
                                int window, int kv_dtype, const float *k_scale, const float *v_scale, void *workspace,
                                size_t workspace_bytes, void *stream, int f16, float *lse = nullptr) {
    VMX_REQUIRE(out && q && k_cache && v_cache && block_tables && seq_lens && workspace, "paged_attention: null pointer");
    VMX_REQUIRE(num_seqs >= 0 && num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
                "paged_attention: num_heads (%d) must be a multiple of num_kv_heads (%d)", num_heads, num_kv_heads);
    VMX_REQUIRE(block_size > 0 && max_blocks_per_seq > 0 && window >= 0, "paged_attention: bad cache geometry");
    VMX_REQUIRE(kv_dtype == VMX_KV_AUTO || (!f16 && (kv_dtype == VMX_KV_FP8_E4M3 || kv_dtype == VMX_KV_INT8 || kv_dtype == VMX_KV_FP8_E5M2)),
                "paged_attention: unsupported kv dtype %d%s", kv_dtype, f16 ? " (the fp16 form takes an fp16 cache only)" : "");
    VMX_REQUIRE(workspace_bytes >= vmx_paged_attention_workspace_bytes(num_seqs, num_heads, head_dim, max_seq_len),
                "paged_attention: workspace too small");
    if (num_seqs == 0) return VMX_OK;
    hipStream_t st = vmx_stream(stream);
    VmxAttnParts parts;
    if (int rc = vmx_paged_attention_parts(q, k_cache, v_cache, block_tables, seq_lens, scale, num_seqs, num_heads, num_kv_heads,
                                           max_blocks_per_seq, max_seq_len, head_dim, block_size, window,
                                           f16 ? VMX_KV_F16_INTERNAL : kv_dtype, k_scale, v_scale, workspace, &parts, st))
        return rc;
    hipLaunchKernelGGL(paged_attn_reduce_kernel, dim3(num_heads, num_seqs), dim3(512), 0, st, out, parts.tmp_out, parts.exp_sums,
                       parts.max_logits, seq_lens, num_heads, head_dim, parts.max_parts, window, parts.part_tokens, (int8_t *)nullptr,
                       (float *)nullptr, f16, lse);
    VMX_CHECK_LAUNCH("paged_attention(reduce)");
    return VMX_OK;
}

extern "C" int vmx_paged_attention_bf16(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                                        const int32_t *block_tables, const int32_t *seq_lens, float scale, int num_seqs, int num_heads,
                                        int num_kv_heads, int max_blocks_per_seq, int max_seq_len, int head_dim, int block_size,
                                        int window, int kv_dtype, const float *k_scale, const float *v_scale, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    return paged_attention_impl(out, q, k_cache, v_cache, block_tables, seq_lens, scale, num_seqs, num_heads, num_kv_heads, max_blocks_per_seq,
                                max_seq_len, head_dim, block_size, window, kv_dtype, k_scale, v_scale, workspace, workspace_bytes, stream, 0);
}

// AttentionBackend::batched_decode_attention_with_lse (layers/attention/backend.rs:99-137): the same decode attention, plus
// lse [seqs, heads] f32 = log sum_i exp(scale * q . k_i) over each head's context (natural log)
extern "C" int vmx_paged_attention_bf16_with_lse(uint16_t *out, float *lse, const uint16_t *q, const void *k_cache, const void *v_cache,
                                                 const int32_t *block_tables, const int32_t *seq_lens, float scale, int num_seqs,
                                                 int num_heads, int num_kv_heads, int max_blocks_per_seq, int max_seq_len, int head_dim,
                                                 int block_size, int window, int kv_dtype, const float *k_scale, const float *v_scale,
                                                 void *workspace, size_t workspace_bytes, void *stream) {
    VMX_REQUIRE(lse, "paged_attention_with_lse: null pointer");
    return paged_attention_impl(out, q, k_cache, v_cache, block_tables, seq_lens, scale, num_seqs, num_heads, num_kv_heads, max_blocks_per_seq,
                                max_seq_len, head_dim, block_size, window, kv_dtype, k_scale, v_scale, workspace, workspace_bytes, stream, 0, lse);
}

// lse_correct_and_reduce, the local half (crates/core/src/distributed/context_parallel.rs:281-315): every decode-context-parallel
// rank holds attention over its own slice of the context plus that slice's log-sum-exp; with the statistics of all ranks gathered
// (lse_all [ranks][rows], rows = seqs * heads) the rank's share of the full result is attn * exp(lse_local - logsumexp_r lse_r).
// The shares are then summed by an all-reduce.  f32 out so that the sum over ranks rounds once; a row whose every rank is
// empty (-inf everywhere) gives zeros.  grid (rows), block 64.
__global__ __launch_bounds__(64) void lse_correct_kernel(float *__restrict__ out, const uint16_t *__restrict__ attn,
                                                        const float *__restrict__ lse_all, int ranks, int rank, int64_t rows, int head_dim) {
    const int64_t row = blockIdx.x;
    float mx = -INFINITY;
    for (int r = 0; r < ranks; r++) mx = fmaxf(mx, lse_all[(int64_t)r * rows + row]);
    float c = 0.0f;
    if (mx != -INFINITY) {
        float sum = 0.0f;
        for (int r = 0; r < ranks; r++) sum += expf(lse_all[(int64_t)r * rows + row] - mx);
        c = expf(lse_all[(int64_t)rank * rows + row] - (mx + logf(sum)));
    }
    for (int d = threadIdx.x; d < head_dim; d += 64) out[row * head_dim + d] = c != 0.0f ? vmx_bf16_to_f32(attn[row * head_dim + d]) * c : 0.0f;
}

extern "C" int vmx_lse_correct_f32(float *out, const uint16_t *attn_out, const float *lse_all, int ranks, int rank, int64_t rows,
                                   int head_dim, void *stream) {
    VMX_REQUIRE(out && attn_out && lse_all, "lse_correct: null pointer");
    VMX_REQUIRE(ranks > 0 && rank >= 0 && rank < ranks && rows >= 0 && head_dim > 0, "lse_correct: rank %d of %d, %lld rows of %d", rank, ranks,
                (long long)rows, head_dim);
    if (rows == 0) return VMX_OK;
    hipLaunchKernelGGL(lse_correct_kernel, dim3((unsigned)rows), dim3(64), 0, vmx_stream(stream), out, attn_out, lse_all, ranks, rank, rows,
                       head_dim);
    VMX_CHECK_LAUNCH("lse_correct");
    return VMX_OK;
}

// paged_attention_v2_f16 + paged_attention_v2_reduce_f16 (crates/core/kernels/paged_attention.cu): fp16 queries, fp16 cache, fp16 output
extern "C" int vmx_paged_attention_f16(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                                       const int32_t *block_tables, const int32_t *seq_lens, float scale, int num_seqs, int num_heads,
                                       int num_kv_heads, int max_blocks_per_seq, int max_seq_len, int head_dim, int block_size, int window,
                                       void *workspace, size_t workspace_bytes, void *stream) {
    return paged_attention_impl(out, q, k_cache, v_cache, block_tables, seq_lens, scale, num_seqs, num_heads, num_kv_heads, max_blocks_per_seq,
                                max_seq_len, head_dim, block_size, window, VMX_KV_AUTO, nullptr, nullptr, workspace, workspace_bytes, stream, 1);
}

extern "C" size_t vmx_paged_attention_static_max_bytes(void) { return VMX_ATTN_SCRATCH_BYTES; }

// vmx_paged_attention_bf16 with the split-KV partials in the library's per-GPU static array instead of a caller workspace:
// no allocation of any kind.  Calls on one GPU share that array, so calls that overlap on one GPU (different streams or
// threads) race; calls on one stream are ordered.  Nothing in the array has to start at zero: the merge kernel reads only the
// partition slots the partition kernel of the same call wrote.
extern "C" int vmx_paged_attention_bf16_static(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                                               const int32_t *block_tables, const int32_t *seq_lens, float scale, int num_seqs,
                                               int num_heads, int num_kv_heads, int max_blocks_per_seq, int max_seq_len, int head_dim,
                                               int block_size, int window, int kv_dtype, const float *k_scale, const float *v_scale,
                                               void *stream) {
    int dev = -1;
    VMX_CHECK_HIP(hipGetDevice(&dev));
    VMX_REQUIRE(dev >= 0 && dev < VMX_MAX_DEVICES && g_attn_scratch[dev],
                "paged_attention(static): vmx_init(%d) has not run on this device", dev);
    const size_t need = vmx_paged_attention_workspace_bytes(num_seqs, num_heads, head_dim, max_seq_len);
    VMX_REQUIRE(need <= VMX_ATTN_SCRATCH_BYTES,
                "paged_attention(static): %d seqs x %d heads x head_dim %d at max_seq_len %d needs %zu bytes of partials, the static "
                "array holds %zu; use vmx_paged_attention_bf16 with a workspace", num_seqs, num_heads, head_dim, max_seq_len, need,
                VMX_ATTN_SCRATCH_BYTES);
    return vmx_paged_attention_bf16(out, q, k_cache, v_cache, block_tables, seq_lens, scale, num_seqs, num_heads, num_kv_heads,
                                    max_blocks_per_seq, max_seq_len, head_dim, block_size, window, kv_dtype, k_scale, v_scale,
                                    g_attn_scratch[dev], VMX_ATTN_SCRATCH_BYTES, stream);
}

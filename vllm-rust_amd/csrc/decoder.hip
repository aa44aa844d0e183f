// Whole-token decode executor for the Qwen3-family quantized decoder (AWQ / GPTQ int4 linears,
// bf16 norms and lm_head, paged KV cache), mirroring the reference's
// QuantizedQwen3ForCausalLM::forward_decode_batch (crates/core/src/models/qwen3_quantized.rs:
// attention :694-927, MLP :532-552, layer :1036-1088, lm_head :1258-1302) behind a C ABI.
//
// Operator order and rounding points are the reference's: every linear output, residual sum,
// norm output, rotated q/k and activation product is rounded to bf16 exactly where the
// reference's separate kernels round it.  What differs is the launch structure: the linears
// leave split-K partial sums in f32 and the next vector kernel folds "sum partials -> bf16 ->
// (+bias) -> residual add -> RMSNorm" (or "-> per-head norm -> RoPE -> KV-cache write", or
// "-> SiLU*mul") into one pass, so a layer is 10 launches instead of the reference's ~17, and the
// whole step is a fixed launch sequence without allocation or host reads (hipGraph-capturable).
#include "vmx_kernels.h"

#include <float.h>
#include <stdlib.h>
#include <new>
#include <vector>

namespace {

__device__ __forceinline__ float round_bf16(float v) { return vmx_bf16_to_f32(vmx_f32_to_bf16(v)); }

__device__ __forceinline__ float sum_partials(const float *__restrict__ partial, int splits, int64_t slice_stride, int64_t idx) {
    // four slices requested before the first add (branch-free: a slice past the end re-reads the last one and is not added);
    // a plain loop over the run-time count waits for every load in turn
    float v = 0.f;
    for (int s0 = 0; s0 < splits; s0 += 4) {
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) t[u] = partial[min(s0 + u, splits - 1) * slice_stride + idx];
#pragma unroll
        for (int u = 0; u < 4; u++) v += s0 + u < splits ? t[u] : 0.f;
    }
    return v;
}

constexpr int SMAX = 4;  // most split-K slices any linear produces (int4 plan: K/128 / 4 <= ... checked at create)

// 8 consecutive f32 sums of the split-K slices at element offset idx (multiple of 8); all loads independent.
__device__ __forceinline__ void sum_partials8(const float *__restrict__ partial, int splits, int64_t slice_stride, int64_t idx,
                                              float out[8]) {
    float4 a[SMAX], b[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; s++) {   // branch-free: a slice past the end re-reads the last one and is not added
        const float *src = partial + max(min(s, splits - 1), 0) * slice_stride + idx;   // splits == 0: slice 0 is read, nothing is added
        a[s] = *reinterpret_cast<const float4 *>(src);
        b[s] = *reinterpret_cast<const float4 *>(src + 4);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) out[j] = 0.f;
#pragma unroll
    for (int s = 0; s < SMAX; s++) {
        const bool in = s < splits;
        out[0] += in ? a[s].x : 0.f; out[1] += in ? a[s].y : 0.f; out[2] += in ? a[s].z : 0.f; out[3] += in ? a[s].w : 0.f;
        out[4] += in ? b[s].x : 0.f; out[5] += in ? b[s].y : 0.f; out[6] += in ? b[s].z : 0.f; out[7] += in ? b[s].w : 0.f;
    }
}

// Output side of an EXL3 linear (exl3.rs:216-286; had_fh_r_128 of kernels/exl3/hadamard_inner.cuh) folded into the consumer
// of its split-K sums: `o` holds 8 consecutive columns (the sums of all slices, taken in slice order); the 16 lanes that
// hold one 128-column block apply the 128-point Hadamard in the stage order 1, 2, ..., 64 with v' = (bit ? -v : v) +
// partner (three stages inside the thread, four across lanes), then * 1/sqrt(128) -> fp16 -> * svh (fp16) -> the value the
// linear returns.  All 64 lanes of the wave must call it; col = first column of the thread's chunk.
__device__ __forceinline__ void exl3_out_transform8(const float *__restrict__ partial, int splits, int64_t slice_stride, int64_t idx,
                                                    const uint16_t *__restrict__ svh, int col, float o[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = 0.f;
    for (int s0 = 0; s0 < splits; s0 += 8) {   // eight slices requested before the first add (branch-free: a slice past the end re-reads the last one)
        float4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float *src = partial + min(s0 + u, splits - 1) * slice_stride + idx;
            a[u] = *reinterpret_cast<const float4 *>(src);
            b[u] = *reinterpret_cast<const float4 *>(src + 4);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool in = s0 + u < splits;
            o[0] += in ? a[u].x : 0.f; o[1] += in ? a[u].y : 0.f; o[2] += in ? a[u].z : 0.f; o[3] += in ? a[u].w : 0.f;
            o[4] += in ? b[u].x : 0.f; o[5] += in ? b[u].y : 0.f; o[6] += in ? b[u].z : 0.f; o[7] += in ? b[u].w : 0.f;
        }
    }
#pragma unroll
    for (int st = 1; st < 8; st <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (!(j & st)) {
                const float a = o[j], b = o[j | st];
                o[j] = a + b;
                o[j | st] = a - b;
            }
        }
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        const bool neg = (lane & m) != 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float p = vmx_lane_xor_m(o[j], m);
            o[j] = (neg ? -o[j] : o[j]) + p;
        }
    }
    const uint4 sv = *reinterpret_cast<const uint4 *>(svh + col);
    const uint32_t sw[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const _Float16 sc = __builtin_bit_cast(_Float16, (uint16_t)(j & 1 ? sw[j >> 1] >> 16 : sw[j >> 1] & 0xffffu));
        o[j] = (float)((_Float16)(o[j] * 0.088388347648f) * sc);
    }
}

__device__ __forceinline__ void unpack_bf16x8(const uint4 v, float f[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f[2 * j] = __uint_as_float(w[j] << 16);
        f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
    }
}

__device__ __forceinline__ uint4 pack_bf16x8(const float f[8]) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = (uint32_t)vmx_f32_to_bf16(f[2 * j]) | ((uint32_t)vmx_f32_to_bf16(f[2 * j + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// One workgroup per token.  h (residual stream in the activation dtype -- bf16, or fp16 with F16 --, in/out), optional incoming linear output as
// f32 partials [splits][tokens][hidden] (splits < 0: `partial` is instead a bf16 [tokens][hidden] tensor,
// the all-reduced output of a row-parallel linear):  o = bf16(sum partials); h = bf16(h + o);
// x = bf16(h * rsqrt(mean(h^2) + eps) * w)   (add_bf16 then rms_norm_bf16: variance over the ROUNDED h).
// Thread t owns the 8-element chunks t, t+256, ... (CH of them, hidden <= CH*2048) in registers.
template <int CH, bool F16>
__global__ __launch_bounds__(256) void resid_norm_kernel(uint16_t *__restrict__ h, uint16_t *__restrict__ x,
                                                          const float *__restrict__ partial, int splits, int tokens,
                                                          const uint16_t *__restrict__ w, float eps, int hidden,
                                                          int8_t *__restrict__ aq, float *__restrict__ ad,
                                                          const uint16_t *__restrict__ had_svh,
                                                          const uint16_t *__restrict__ emb_table, const int32_t *__restrict__ emb_ids,
                                                          int vocab) {
    __shared__ float red[4];
    const int t = blockIdx.x;
    const int64_t row = (int64_t)t * hidden;
    // emb_table != NULL (no partial): the row is the token's embedding (an id outside the table reads as zeros), stored to h as
    // the embedding kernel would have -- the first norm of a step without a launch for the lookup
    const int emb_id = emb_table ? emb_ids[t] : 0;
    const bool emb_ok = emb_id >= 0 && emb_id < vocab;
    const uint16_t *hsrc = emb_table ? emb_table + (int64_t)(emb_ok ? emb_id : 0) * hidden : h + row;
    const int64_t slice = (int64_t)tokens * hidden;
    const int nchunk = hidden >> 3;
    float hv[CH][8];
    float var = 0.f;
    uint4 wraw[CH];   // norm weights requested with the first loads: the kernel is a chain of latencies
#pragma unroll
    for (int c = 0; c < CH; c++) wraw[c] = *reinterpret_cast<const uint4 *>(w + 8 * min((int)threadIdx.x + 256 * c, nchunk - 1));
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int ci = threadIdx.x + 256 * c;
        if (256 * c < nchunk && had_svh) {   // block-uniform; hidden % 128 == 0, so a 16-lane group is all in or all out: `partial` holds an EXL3 linear's raw split-K sums
            float o[8];
            exl3_out_transform8(partial, splits, slice, row + 8 * min(ci, nchunk - 1), had_svh, 8 * min(ci, nchunk - 1), o);
            if (ci < nchunk) {
                vmx_unpack16x8<F16>(*reinterpret_cast<const uint4 *>(h + row + 8 * ci), hv[c]);
#pragma unroll
                for (int j = 0; j < 8; j++) hv[c][j] = vmx_rnd16<F16>(hv[c][j] + vmx_rnd16<F16>(o[j]));
                *reinterpret_cast<uint4 *>(h + row + 8 * ci) = vmx_pack16x8<F16>(hv[c]);
#pragma unroll
                for (int j = 0; j < 8; j++) var += hv[c][j] * hv[c][j];
            }
        } else if (ci < nchunk) {
            vmx_unpack16x8<F16>(emb_table && !emb_ok ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4 *>(hsrc + 8 * ci), hv[c]);
            if (emb_table) *reinterpret_cast<uint4 *>(h + row + 8 * ci) = vmx_pack16x8<F16>(hv[c]);
            if (partial) {
                float o[8];
                if (splits < 0) vmx_unpack16x8<F16>(*reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(partial) + row + 8 * ci), o);
                else sum_partials8(partial, splits, slice, row + 8 * ci, o);
#pragma unroll
                for (int j = 0; j < 8; j++) hv[c][j] = vmx_rnd16<F16>(hv[c][j] + vmx_rnd16<F16>(o[j]));
                *reinterpret_cast<uint4 *>(h + row + 8 * ci) = vmx_pack16x8<F16>(hv[c]);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) var += hv[c][j] * hv[c][j];
        }
    }
    var = vmx_block_sum_256(var, red);
    const float inv = rsqrtf(var / (float)hidden + eps);
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int ci = threadIdx.x + 256 * c;
        if (ci < nchunk) {
            float wv[8], o[8];
            vmx_unpack16x8<F16>(wraw[c], wv);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = hv[c][j] * inv * wv[j];
            const uint4 packed = vmx_pack16x8<F16>(o);
            *reinterpret_cast<uint4 *>(x + row + 8 * ci) = packed;
            if (aq) {  // q8_1 of the (bf16-rounded) normed row for an I-quant consumer; hidden % 32 == 0
                float xr[8];
                vmx_unpack16x8<F16>(packed, xr);
                vmx_q8_1_emit8(xr, aq + row + 8 * ci, ad + (row >> 5) + (ci >> 2), (ci & 3) == 0);
            }
        }
    }
}

// The residual half of resid_norm_kernel for an EXL3 producer at decode sizes, spread over hidden/512 workgroups per token
// (one workgroup there walks splits x hidden f32 -- 24 x 16 KB behind down_proj of Qwen3-8B -- on one CU).  A wave owns one
// 128-column Hadamard block: all its split-K loads are requested at once (two columns per lane), then the transform (same
// stage order and arithmetic as exl3_out_transform8), h = bf16(h + bf16(o)) and the block's sum of squares of the new h
// into blk_ssq [tokens][hidden / 128].  The consuming linear normalises its rows itself from those sums.
template <bool F16>
__global__ __launch_bounds__(256) void exl3_resid_stats_kernel(uint16_t *__restrict__ h, const float *__restrict__ partial, int splits,
                                                                int tokens, int hidden, const uint16_t *__restrict__ had_svh,
                                                                float *__restrict__ blk_ssq) {
    const int t = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nblk = hidden >> 7, blk = blockIdx.x * 4 + wave;
    const int64_t row = (int64_t)t * hidden, slice = (int64_t)tokens * hidden;
    if (blk >= nblk) return;
    const int col = blk * 128 + 2 * lane;
    const uint32_t hraw = *reinterpret_cast<const uint32_t *>(h + row + col);
    const uint32_t sv = *reinterpret_cast<const uint32_t *>(had_svh + col);
    float e0 = 0.f, e1 = 0.f;
    for (int s0 = 0; s0 < splits; s0 += 12) {   // twelve slices per round trip (24 behind down_proj of Qwen3-8B, 8 behind o_proj)
        float2 v[12];
#pragma unroll
        for (int u = 0; u < 12; u++) v[u] = *reinterpret_cast<const float2 *>(partial + min(s0 + u, splits - 1) * slice + row + col);
#pragma unroll
        for (int u = 0; u < 12; u++) {
            e0 += s0 + u < splits ? v[u].x : 0.f;
            e1 += s0 + u < splits ? v[u].y : 0.f;
        }
    }
    const float s = e0 + e1, d = e0 - e1;
    e0 = s;
    e1 = d;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float p0 = vmx_lane_xor_m(e0, m), p1 = vmx_lane_xor_m(e1, m);
        const bool neg = (lane & m) != 0;
        e0 = (neg ? -e0 : e0) + p0;
        e1 = (neg ? -e1 : e1) + p1;
    }
    const float o0 = (float)((_Float16)(e0 * 0.088388347648f) * __builtin_bit_cast(_Float16, (uint16_t)(sv & 0xffffu)));
    const float o1 = (float)((_Float16)(e1 * 0.088388347648f) * __builtin_bit_cast(_Float16, (uint16_t)(sv >> 16)));
    const float h0 = vmx_rnd16<F16>(vmx_ld16<F16>((uint16_t)(hraw & 0xffffu)) + vmx_rnd16<F16>(o0));
    const float h1 = vmx_rnd16<F16>(vmx_ld16<F16>((uint16_t)(hraw >> 16)) + vmx_rnd16<F16>(o1));
    *reinterpret_cast<uint32_t *>(h + row + col) = (uint32_t)vmx_st16<F16>(h0) | ((uint32_t)vmx_st16<F16>(h1) << 16);
    const float var = vmx_wave_sum(h0 * h0 + h1 * h1);
    if (lane == 0) blk_ssq[t * nblk + blk] = var;
}

// The three launches in front of a mixture-of-experts MLP as one (MoE layers with at most EMAX experts): one workgroup per token,
//   x = bf16(h * rsqrt(mean(h^2) + eps) * w)                                        (post_attention_layernorm, x also stored)
//   logit_e = bf16(sum_i x_i * router[e][i])                                        (the dense `gate`, mixtral_quantized.rs:118-119)
//   softmax over the experts rounded to bf16, top-k (ties: lower index), renormalised  (topk_softmax_kernel, renorm = 1)
// The router rows are requested together with h, ahead of the norm's reduction.
template <int CH, int EMAX>
__global__ __launch_bounds__(256) void norm_router_topk_kernel(const uint16_t *__restrict__ h, uint16_t *__restrict__ x,
                                                                const uint16_t *__restrict__ w, float eps, int hidden,
                                                                const uint16_t *__restrict__ router, int E, int k,
                                                                float *__restrict__ topk_w, int32_t *__restrict__ topk_ids) {
    __shared__ float red[4];
    __shared__ float pds[4][EMAX];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = (int64_t)t * hidden;
    const int nchunk = hidden >> 3;
    uint4 hraw[CH], wraw[CH], rraw[EMAX][CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int ci = min(tid + 256 * c, nchunk - 1);
        hraw[c] = *reinterpret_cast<const uint4 *>(h + row + 8 * ci);
        wraw[c] = *reinterpret_cast<const uint4 *>(w + 8 * ci);
    }
#pragma unroll
    for (int e = 0; e < EMAX; e++)
#pragma unroll
        for (int c = 0; c < CH; c++)
            rraw[e][c] = *reinterpret_cast<const uint4 *>(router + (int64_t)min(e, E - 1) * hidden + 8 * min(tid + 256 * c, nchunk - 1));
    float hv[CH][8], var = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        unpack_bf16x8(hraw[c], hv[c]);
        if (tid + 256 * c < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; j++) var += hv[c][j] * hv[c][j];
        }
    }
    var = vmx_block_sum_256(var, red);
    const float inv = rsqrtf(var / (float)hidden + eps);
    float pd[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; e++) pd[e] = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int ci = tid + 256 * c;
        if (ci < nchunk) {
            float wv[8], o[8], xr[8];
            unpack_bf16x8(wraw[c], wv);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = hv[c][j] * inv * wv[j];
            const uint4 packed = pack_bf16x8(o);
            *reinterpret_cast<uint4 *>(x + row + 8 * ci) = packed;
            unpack_bf16x8(packed, xr);
#pragma unroll
            for (int e = 0; e < EMAX; e++) {
                float rv[8];
                unpack_bf16x8(rraw[e][c], rv);
#pragma unroll
                for (int j = 0; j < 8; j++) pd[e] += xr[j] * rv[j];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EMAX; e++) {
        const float s = vmx_wave_sum(pd[e]);
        if (lane == 0) pds[wave][e] = s;
    }
    __syncthreads();
    if (wave != 0) return;
    // wave 0: lane e < E holds expert e
    float v = -FLT_MAX;
    if (lane < E) v = round_bf16((pds[0][lane] + pds[1][lane]) + (pds[2][lane] + pds[3][lane]));
    const float mx = vmx_wave_max(v);
    float p = lane < E ? expf(v - mx) : 0.f;
    const float sum = vmx_wave_sum(p);
    p = lane < E ? round_bf16(p / sum) : -FLT_MAX;
    float selected = 0.f, mine = 0.f;
    for (int ki = 0; ki < k; ki++) {
        float bv = p;
        int bi = lane < E ? lane : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = vmx_lane_xor_m(bv, off);
            const int oi = vmx_lane_xor_m(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == ki) { mine = bv; topk_ids[(int64_t)t * k + ki] = bi; }
        selected += bv;
        if (lane == bi) p = -FLT_MAX;
    }
    const float f = selected > 0.f ? 1.0f / selected : 1.0f;
    if (lane < k) topk_w[(int64_t)t * k + lane] = mine * f;
}

// grid (heads_q + 2*heads_kv, tokens), block 64.  Takes the merged qkv linear output as f32
// partials [splits][tokens][(Hq + 2*Hkv) * hd]; per head: bf16 round, optional per-head RMSNorm
// (bf16 round), NeoX RoPE (bf16 round); q -> q_out, k and v -> paged cache (bf16 or FP8 e4m3).
template <bool F16>
__global__ __launch_bounds__(64) void qkv_post_kernel(const float *__restrict__ partial, int splits, int tokens,
                                                       uint16_t *__restrict__ q_out, void *__restrict__ k_cache,
                                                       void *__restrict__ v_cache, const uint16_t *__restrict__ q_norm_w,
                                                       const uint16_t *__restrict__ k_norm_w, float eps,
                                                       const float *__restrict__ cos_sin, const int32_t *__restrict__ positions,
                                                       const int32_t *__restrict__ slots, int Hq, int Hkv, int hd, int rot_dim,
                                                       int fp8, const float *__restrict__ k_scale, const float *__restrict__ v_scale) {
    __shared__ float vals[512];
    const int head = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
    const int ntot = (Hq + 2 * Hkv) * hd;
    const int64_t slice = (int64_t)tokens * ntot;
    const int64_t base = (int64_t)t * ntot + (int64_t)head * hd;
    const bool is_q = head < Hq, is_k = !is_q && head < Hq + Hkv;
    float ss = 0.f;
    for (int i = lane; i < hd; i += 64) {
        const float v = vmx_rnd16<F16>(sum_partials(partial, splits, slice, base + i));
        vals[i] = v;
        ss += v * v;
    }
    const uint16_t *nw = is_q ? q_norm_w : (is_k ? k_norm_w : nullptr);
    if (nw) {  // per-head RMSNorm, output rounded to bf16 (rms_norm_bf16 on the [.., hd] view)
        ss = vmx_wave_sum(ss);
        const float inv = rsqrtf(ss / (float)hd + eps);
        for (int i = lane; i < hd; i += 64) vals[i] = vmx_rnd16<F16>(vals[i] * inv * vmx_ld16<F16>(nw[i]));
    }
    __syncthreads();
    if (is_q || is_k) {  // NeoX rotation of the first rot_dim dims
        const int half = rot_dim >> 1;
        const float *cs = cos_sin + (int64_t)positions[t] * rot_dim;
        for (int i = lane; i < half; i += 64) {
            const float xv = vals[i], yv = vals[half + i], c = cs[i], s = cs[half + i];
            vals[i] = vmx_rnd16<F16>(xv * c - yv * s);
            vals[half + i] = vmx_rnd16<F16>(yv * c + xv * s);
        }
        __syncthreads();
    }
    if (is_q) {
        for (int i = lane; i < hd; i += 64) q_out[((int64_t)t * Hq + head) * hd + i] = vmx_st16<F16>(vals[i]);
        return;
    }
    const int slot = slots[t];
    if (slot < 0) return;
    const int kvh = is_k ? head - Hq : head - Hq - Hkv;
    const int64_t dst = ((int64_t)slot * Hkv + kvh) * hd;
    void *cache = is_k ? k_cache : v_cache;
    if (!fp8) {
        for (int i = lane; i < hd; i += 64) ((uint16_t *)cache)[dst + i] = vmx_st16<F16>(vals[i]);
    } else {
        const float sc = is_k ? (k_scale ? *k_scale : 1.f) : (v_scale ? *v_scale : 1.f);
        for (int i = lane; i < hd; i += 64) {
            float a = vals[i] / sc;
            a = fminf(fmaxf(a, -448.0f), 448.0f);
            a = round_bf16(a);
            ((uint8_t *)cache)[dst + i] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(a, a, 0, false) & 0xFFu);
        }
    }
}

// a[t, i] = bf16(silu(bf16(gate)) * bf16(up)) from the merged gate|up linear output (f32 partials
// [splits][tokens][2*inter], gate columns first); 8 elements per thread.  Optionally also the q8_1 form of the
// result for an I-quant down projection (inter % 32 == 0).
template <bool F16>
__global__ __launch_bounds__(256) void gateup_act_kernel(const float *__restrict__ partial, int splits, int tokens,
                                                          uint16_t *__restrict__ act, int inter, int8_t *__restrict__ aq,
                                                          float *__restrict__ ad, const uint16_t *__restrict__ had_svh) {
    const int64_t chunks = (int64_t)tokens * inter / 8;
    const int64_t slice = (int64_t)tokens * 2 * inter;
    const int cpr = inter / 8;
    for (int64_t c0 = (int64_t)blockIdx.x * 256; c0 < chunks; c0 += (int64_t)gridDim.x * 256) {
        const int64_t c = c0 + threadIdx.x;
        if (c < chunks || had_svh) {  // chunks % 4 == 0, so 4-lane groups are all in or all out; chunks % 16 == 0 with had_svh
            const int64_t cc = c < chunks ? c : chunks - 1;
            const int64_t t = cc / cpr, col = (cc % cpr) * 8;
            float g[8], u[8];
            if (had_svh) {   // `partial` holds the raw split-K sums of an EXL3 gate|up launch (whole waves in or out: grid sized for it)
                exl3_out_transform8(partial, splits, slice, t * 2 * inter + col, had_svh, (int)col, g);
                exl3_out_transform8(partial, splits, slice, t * 2 * inter + inter + col, had_svh, inter + (int)col, u);
                if (c >= chunks) continue;
            } else {
            sum_partials8(partial, splits, slice, t * 2 * inter + col, g);
            sum_partials8(partial, splits, slice, t * 2 * inter + inter + col, u);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float gg = vmx_rnd16<F16>(g[j]);
                g[j] = (gg / (1.0f + expf(-gg))) * vmx_rnd16<F16>(u[j]);
            }
            const uint4 packed = vmx_pack16x8<F16>(g);
            *reinterpret_cast<uint4 *>(act + c * 8) = packed;
            if (aq) {
                float r[8];
                vmx_unpack16x8<F16>(packed, r);
                vmx_q8_1_emit8(r, aq + c * 8, ad + (c >> 2), (c & 3) == 0);
            }
        }
    }
}

// y[i] = bf16(sum of split-K slices): the local output of a row-parallel linear before the all-reduce
template <bool F16>
__global__ __launch_bounds__(256) void slices_to_bf16_kernel(const float *__restrict__ partial, int splits, int64_t n,
                                                              uint16_t *__restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = vmx_st16<F16>(sum_partials(partial, splits, n, i));
}

__global__ __launch_bounds__(256) void f16_to_bf16_into_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, int64_t n,
                                                                int cols, int ldo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[(i / cols) * ldo + (i % cols)] = vmx_f32_to_bf16(vmx_f16_to_f32(in[i]));
}

__global__ __launch_bounds__(256) void embed_kernel(uint16_t *__restrict__ out, const uint16_t *__restrict__ table,
                                                     const int32_t *__restrict__ ids, int hidden, int vocab) {
    const int t = blockIdx.x;
    const int id = ids[t];
    const bool ok = id >= 0 && id < vocab;
    const uint4 *src = reinterpret_cast<const uint4 *>(table + (int64_t)(ok ? id : 0) * hidden);
    uint4 *dst = reinterpret_cast<uint4 *>(out + (int64_t)t * hidden);
    for (int i = threadIdx.x; i < (hidden >> 3); i += 256) dst[i] = ok ? src[i] : make_uint4(0, 0, 0, 0);
}

// gateup_act_kernel for an EXL3 gate|up launch at decode sizes: a wave owns one 128-column block of the intermediate row and
// requests the split-K slices of its gate and up columns in one go (two columns per lane), applies both output transforms
// (stage order and arithmetic of exl3_out_transform8) and writes act = bf16(silu(bf16 gate) * bf16 up).
// partial: [splits][tokens][2 * inter], gate columns first; svh: the merged gate|up scale vector.  grid (inter / 512, tokens).
template <bool F16>
__global__ __launch_bounds__(256) void exl3_gateup_act_kernel(const float *__restrict__ partial, int splits, int tokens,
                                                               uint16_t *__restrict__ act, int inter, const uint16_t *__restrict__ svh) {
    const int t = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = blockIdx.x * 4 + wave;
    if (blk >= (inter >> 7)) return;
    const int col = blk * 128 + 2 * lane;
    const int64_t slice = (int64_t)tokens * 2 * inter, row = (int64_t)t * 2 * inter;
    const uint32_t svg = *reinterpret_cast<const uint32_t *>(svh + col), svu = *reinterpret_cast<const uint32_t *>(svh + inter + col);
    float g0 = 0.f, g1 = 0.f, u0 = 0.f, u1 = 0.f;
    for (int s0 = 0; s0 < splits; s0 += 8) {
        float2 gv[8], uv[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float *src = partial + min(s0 + q, splits - 1) * slice + row + col;
            gv[q] = *reinterpret_cast<const float2 *>(src);
            uv[q] = *reinterpret_cast<const float2 *>(src + inter);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const bool in = s0 + q < splits;
            g0 += in ? gv[q].x : 0.f; g1 += in ? gv[q].y : 0.f;
            u0 += in ? uv[q].x : 0.f; u1 += in ? uv[q].y : 0.f;
        }
    }
    float e[4] = {g0 + g1, g0 - g1, u0 + u1, u0 - u1};
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const bool neg = (lane & m) != 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float p = vmx_lane_xor_m(e[j], m);
            e[j] = (neg ? -e[j] : e[j]) + p;
        }
    }
    const uint32_t sc[4] = {svg & 0xffffu, svg >> 16, svu & 0xffffu, svu >> 16};
#pragma unroll
    for (int j = 0; j < 4; j++) e[j] = vmx_rnd16<F16>((float)((_Float16)(e[j] * 0.088388347648f) * __builtin_bit_cast(_Float16, (uint16_t)sc[j])));
    const float a0 = (e[0] / (1.0f + expf(-e[0]))) * e[2], a1 = (e[1] / (1.0f + expf(-e[1]))) * e[3];
    *reinterpret_cast<uint32_t *>(act + (int64_t)t * inter + col) = (uint32_t)vmx_st16<F16>(a0) | ((uint32_t)vmx_st16<F16>(a1) << 16);
}

// EXL3 decode rows: h += transform(partial) and the block sums the next linear's prologue normalises with (no x row)
bool exl3_resid_stats_ok(int tokens, int splits, int hidden) {
    static const bool off = vmx_tune("VMX_EXL3_NO_NORM_FUSION", 0) != 0;
    return !off && tokens <= 4 && splits > 0 && hidden % 128 == 0 && hidden <= 8192;
}
void launch_exl3_resid_stats(hipStream_t st, uint16_t *h, const float *partial, int splits, int tokens, int hidden, const uint16_t *had_svh,
                             float *blk_ssq, bool f16 = false) {
    const dim3 grid((hidden / 128 + 3) / 4, tokens);
    if (f16) hipLaunchKernelGGL(exl3_resid_stats_kernel<true>, grid, dim3(256), 0, st, h, partial, splits, tokens, hidden, had_svh, blk_ssq);
    else hipLaunchKernelGGL(exl3_resid_stats_kernel<false>, grid, dim3(256), 0, st, h, partial, splits, tokens, hidden, had_svh, blk_ssq);
}

template <bool F16>
void launch_resid_norm_t(hipStream_t st, uint16_t *h, uint16_t *x, const float *partial, int splits, int tokens, const uint16_t *w,
                         float eps, int hidden, int8_t *aq, float *ad, const uint16_t *had_svh, const uint16_t *emb_table,
                         const int32_t *emb_ids, int vocab) {
    if (hidden <= 2048) hipLaunchKernelGGL((resid_norm_kernel<1, F16>), dim3(tokens), dim3(256), 0, st, h, x, partial, splits, tokens, w, eps, hidden, aq, ad, had_svh, emb_table, emb_ids, vocab);
    else if (hidden <= 4096) hipLaunchKernelGGL((resid_norm_kernel<2, F16>), dim3(tokens), dim3(256), 0, st, h, x, partial, splits, tokens, w, eps, hidden, aq, ad, had_svh, emb_table, emb_ids, vocab);
    else if (hidden <= 8192) hipLaunchKernelGGL((resid_norm_kernel<4, F16>), dim3(tokens), dim3(256), 0, st, h, x, partial, splits, tokens, w, eps, hidden, aq, ad, had_svh, emb_table, emb_ids, vocab);
    else hipLaunchKernelGGL((resid_norm_kernel<8, F16>), dim3(tokens), dim3(256), 0, st, h, x, partial, splits, tokens, w, eps, hidden, aq, ad, had_svh, emb_table, emb_ids, vocab);
}

void launch_resid_norm(hipStream_t st, uint16_t *h, uint16_t *x, const float *partial, int splits, int tokens, const uint16_t *w,
                       float eps, int hidden, int8_t *aq = nullptr, float *ad = nullptr, const uint16_t *had_svh = nullptr,
                       const uint16_t *emb_table = nullptr, const int32_t *emb_ids = nullptr, int vocab = 0, bool f16 = false) {
    if (f16) launch_resid_norm_t<true>(st, h, x, partial, splits, tokens, w, eps, hidden, aq, ad, had_svh, emb_table, emb_ids, vocab);
    else launch_resid_norm_t<false>(st, h, x, partial, splits, tokens, w, eps, hidden, aq, ad, had_svh, emb_table, emb_ids, vocab);
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n) {
        bytes = n;
        hipError_t e = hipMalloc(&p, n ? n : 16);
        if (e != hipSuccess) {
            vmx_set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
            return VMX_ENOMEM;
        }
        return VMX_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
};

// Mirrors `trait QuantizedLinear` (crates/core/src/quantization/config.rs:147-190) for the AWQ/GPTQ
// int4 tile stream: load_weights (one-time repack) + forward (here: split-K partial sums).
struct Int4Linear {
    int in_features = 0, out_features = 0, group = 128;
    DevBuf packed;
    bool loaded = false;
    bool glu = false;  // merged gate|up packed for the fused activation epilogue (forward_glu instead of forward_partial)
    int splits() const { return vmx_int4_splits(out_features, in_features); }
    int forward_partial(const uint16_t *x, float *partial, int M, hipStream_t st) const {
        return vmx_int4_gemm_partial(packed.p, x, partial, M, out_features, in_features, group, st);
    }
    int forward_glu(const uint16_t *x, uint16_t *act, int M, hipStream_t st) const {
        return vmx_int4_gemm_glu(packed.p, x, act, M, out_features, in_features, group, st);
    }

};

// Mirrors IqLinear (crates/core/src/quantization/gguf/mod.rs:262-377): the raw GGUF block bytes of one I-quant
// weight [out, in]; forward = q8_1-quantise the activations, then MMVQ (decode path, M <= 16).
struct IqLinear {
    int ggml_type = 0, in_features = 0, out_features = 0;
    DevBuf raw;
    bool loaded = false;
};

// Mirrors Exl3Linear (crates/core/src/quantization/exl3.rs:204-286): trellis i16 [K/16][N/16][16*bits] as the checkpoint
// stores it, fp16 sign/scale vectors suh [K] and svh [N], codebook 0 default / 1 mcg / 2 mul1.
struct Exl3Lin {
    int in_features = 0, out_features = 0, bits = 0, codebook = 0;
    DevBuf trellis, suh, svh;
    bool loaded = false;
};

// Merged GPTQ-sequential staging of several AWQ linears that share the input (q|k|v, gate|up).
struct MergeStage {
    DevBuf qweight, qzeros, scales;
    int filled = 0;
};

struct Layer {
    Int4Linear qkv, o, gate_up, down;
    IqLinear iq[7];  // q, k, v, o, gate, up, down as GGUF I-quants (alternative to the int4 linears)
    bool iq_ready() const { for (const IqLinear &l : iq) if (!l.loaded) return false; return true; }
    Exl3Lin ex[7];   // ... or as EXL3 trellis-coded linears
    DevBuf ex_svh_qkv, ex_svh_gu;   // svh of q|k|v and of gate|up back to back, when the group can go out as one launch
    bool exl3_ready() const { for (const Exl3Lin &l : ex) if (!l.loaded) return false; return true; }
    // ... or as plain bf16 matrices [out, in] (an unquantized checkpoint): q|k|v, o, gate|up, down, rows concatenated
    DevBuf dn[4];
    int dn_filled = 0;   // bit per slot VMX_L_Q .. VMX_L_DOWN
    bool dense_ready() const { return dn_filled == 0x7f; }
    bool int4_ready() const { return qkv.loaded && o.loaded && gate_up.loaded && down.loaded; }
    bool int4_attn_ready() const { return qkv.loaded && o.loaded; }
    MergeStage qkv_stage, gu_stage;
    DevBuf input_ln, post_ln, q_norm, k_norm;
    DevBuf k_cache, v_cache, kv_scales;  // kv_scales: f32[2] = {k_scale, v_scale}
    // mixture-of-experts MLP (num_experts > 0): dense router [E, hidden] and per-expert int4 tile streams
    DevBuf router, w13_table, w2_table;   // tables: device arrays of E pointers (what vmx_moe_int4_forward takes)
    std::vector<DevBuf> w13, w2;
    int experts_loaded = 0;
    bool moe_ready(int E) const { return E > 0 && experts_loaded == E; }
};

}  // namespace

struct vmx_qwen3 {
    vmx_qwen3_config cfg;
    std::vector<Layer> layers;
    DevBuf embed, lm_head, final_norm, cos_sin;
    bool lm_head_tied = false;
    DevBuf h, x, q, attn, act, partial, partial_b, logits, attn_ws, argmax_ws, red_bf16, keys, q8, positions, ssq;
    DevBuf router_logits, topk_w, topk_ids, moe_ws, moe_out;   // MoE models only
    DevBuf dense_tmp;                                          // unquantized layers only: bf16 [16][widest projection], grown by the loader
    DevBuf exl3_ws;                                            // EXL3 models only: grown by the loader
    vmx_comm *comm = nullptr;  // tensor-parallel group (not owned); NULL = single GPU
    int lin_mask = 15;  // measurement aid: which linears vmx_qwen3_linears_only launches (1 q|k|v, 2 o, 4 gate|up, 8 down)
    bool rows_share_cache = false;  // rows of a decode batch attend to K/V rows written by other rows of the same call
    bool act_f16 = false;   // activations, norm weights, embedding / lm_head tables and the KV cache are fp16 (EXL3 checkpoints); else bf16
    int vocab_local() const { return cfg.tp_size > 1 ? cfg.vocab_size / cfg.tp_size : cfg.vocab_size; }
    int qkv_n() const { return (cfg.num_heads + 2 * cfg.num_kv_heads) * cfg.head_dim; }
};

namespace {

// Algorithmic bytes of one int4 linear launch: packed nibbles + bf16 scales + 4-bit zeros + bf16 in/out.
double int4_alg_bytes(const Int4Linear &l, int M) {
    const double K = l.in_features, N = l.out_features, G = K / l.group;
    return K * N / 2 + G * N * 2 + G * N / 2 + (double)M * K * 2 + (double)M * N * 2;
}

int run_linear(vmx_qwen3 *, const Int4Linear &l, const uint16_t *xin, float *partial, int M, hipStream_t st) {
    return l.forward_partial(xin, partial, M, st);
}

int8_t *q8_aq(vmx_qwen3 *m) { return (int8_t *)m->q8.p; }
float *q8_ad(vmx_qwen3 *m, int tokens, int k) { return (float *)((char *)m->q8.p + (((size_t)tokens * k + 255) & ~(size_t)255)); }

// IqLinear::forward_inner on the decode path (gguf/mod.rs:329-377): activations bf16 -> (exact) f32 -> q8_1 ...
int iq_quantize(vmx_qwen3 *m, const uint16_t *x, int tokens, int k, hipStream_t st) {
    int8_t *aq = (int8_t *)m->q8.p;
    float *ad = (float *)((char *)m->q8.p + (((size_t)tokens * k + 255) & ~(size_t)255));
    return vmx_quantize_q8_1_bf16(x, aq, ad, (int64_t)tokens * (k / 32), st);
}

// ... then MMVQ into row-stride ldy (f32; the consumer rounds to bf16, the cast back to the model dtype)
int iq_forward(vmx_qwen3 *m, const IqLinear &l, float *y, int ldy, int tokens, hipStream_t st) {
    const int8_t *aq = (const int8_t *)m->q8.p;
    const float *ad = (const float *)((const char *)m->q8.p + (((size_t)tokens * l.in_features + 255) & ~(size_t)255));
    for (int r0 = 0; r0 < tokens; r0 += 16) {   // the MMVQ takes at most 16 activation rows (IQ_GEMV_MAX_M)
        const int mr = tokens - r0 < 16 ? tokens - r0 : 16;
        if (int rc = vmx_iq_mmvq_ld(l.ggml_type, l.raw.p, aq + (size_t)r0 * l.in_features, ad + (size_t)r0 * (l.in_features / 32),
                                    y + (size_t)r0 * ldy, l.out_features, l.in_features, mr, ldy, st))
            return rc;
    }
    return VMX_OK;
}

// Exl3Linear::forward on the rows of a bf16 activation tensor; the result (fp16 cast back to bf16, exl3.rs:272-277) lands
// as f32 in y[row * ldy + col], the form the vector kernels take with splits == 1.
int exl3_forward(vmx_qwen3 *m, const Exl3Lin &l, const uint16_t *x, float *y, int ldy, int tokens, hipStream_t st) {
    return vmx_exl3_forward_f32(x, m->act_f16 ? 0 : 1, l.trellis.p, (const uint16_t *)l.suh.p, (const uint16_t *)l.svh.p, y, ldy, tokens, l.in_features,
                                l.out_features, l.bits, l.codebook, m->exl3_ws.p, m->exl3_ws.bytes, st);
}

// Several EXL3 projections of one layer that share x, as one launch pair (M <= 4)
int exl3_forward_group(vmx_qwen3 *m, const Layer &L, std::initializer_list<int> slots, const DevBuf &svh, const uint16_t *x, float *y, int ldy,
                       int tokens, hipStream_t st, const float **partial_out = nullptr, int *splits_out = nullptr,
                       const uint16_t *norm_w = nullptr, const float *blk_ssq = nullptr, float eps = 0.f, const VmxAttnParts *attn_parts = nullptr) {
    const void *tr[3];
    const uint16_t *suh[3];
    int n_out[3], n = 0;
    for (int sl : slots) {
        tr[n] = L.ex[sl].trellis.p;
        suh[n] = (const uint16_t *)L.ex[sl].suh.p;
        n_out[n++] = L.ex[sl].out_features;
    }
    const Exl3Lin &f = L.ex[*slots.begin()];
    return vmx_exl3_forward_multi_f32(x, m->act_f16 ? 0 : 1, n, tr, suh, n_out, (const uint16_t *)svh.p, y, ldy, tokens, f.in_features, f.bits, f.codebook,
                                      m->exl3_ws.p, m->exl3_ws.bytes, st, partial_out, splits_out, norm_w, blk_ssq, eps, attn_parts);
}

__global__ __launch_bounds__(256) void bf16_rows_to_f32_kernel(const uint16_t *__restrict__ in, float *__restrict__ out, int rows, int cols,
                                                                 int ldo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (int64_t)rows * cols) out[(i / cols) * ldo + (i % cols)] = vmx_bf16_to_f32(in[i]);
}

// Unquantized Linear (no bias) on decode / prompt rows: y = bf16(x W^T), 16 rows per launch, widened to the f32 form the
// vector kernels take with one slice (they round to bf16 again, which changes nothing).
int dense_forward(vmx_qwen3 *m, const void *W, const uint16_t *x, float *y, int ldy, int N, int K, int tokens, hipStream_t st) {
    uint16_t *tmp = (uint16_t *)m->dense_tmp.p;
    for (int r0 = 0; r0 < tokens; r0 += 16) {
        const int mr = tokens - r0 < 16 ? tokens - r0 : 16;
        if (int rc = vmx_dense_gemv_launch(tmp, x + (size_t)r0 * K, (const uint16_t *)W, nullptr, mr, N, K, st)) return rc;
        const int64_t n = (int64_t)mr * N;
        hipLaunchKernelGGL(bf16_rows_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tmp, y + (size_t)r0 * ldy, mr, N, ldy);
    }
    VMX_CHECK_LAUNCH("dense_forward");
    return VMX_OK;
}

// Row-parallel epilogue: red_bf16 = all_reduce_sum(bf16(sum of this rank's split-K slices)) — what composing the
// reference's RowParallelLinear (distributed/parallel_layers.rs:275-305) with a quantized linear computes.
int tp_reduce(vmx_qwen3 *m, const float *partial, int splits, int tokens, int hidden, hipStream_t st) {
    const int64_t n = (int64_t)tokens * hidden;
    if (m->act_f16) {
        hipLaunchKernelGGL(slices_to_bf16_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, splits, n,
                           (uint16_t *)m->red_bf16.p);
        VMX_CHECK_LAUNCH("slices_to_f16");
        return vmx_comm_all_reduce_sum_f16(m->comm, (uint16_t *)m->red_bf16.p, (size_t)n, st);
    }
    hipLaunchKernelGGL(slices_to_bf16_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, splits, n,
                       (uint16_t *)m->red_bf16.p);
    VMX_CHECK_LAUNCH("slices_to_bf16");
    return vmx_comm_all_reduce_sum_bf16(m->comm, (uint16_t *)m->red_bf16.p, (size_t)n, st);
}

int make_linear(Int4Linear &l, int K, int N, int group) {
    l.in_features = K;
    l.out_features = N;
    l.group = group;
    return VMX_OK;  // the tile stream is allocated when the weights arrive (stage_pack)
}

int stage_alloc(MergeStage &s, int K, int Ntot, int group) {
    if (s.qweight.p) return VMX_OK;
    if (int rc = s.qweight.alloc((size_t)(K / 8) * Ntot * 4)) return rc;
    if (int rc = s.qzeros.alloc((size_t)(K / group) * (Ntot / 8) * 4)) return rc;
    return s.scales.alloc((size_t)(K / group) * Ntot * 2);
}

void stage_free(MergeStage &s) {
    s.filled = 0;   // a later reload of the group starts from an empty stage
    s.qweight.release();
    s.qzeros.release();
    s.scales.release();
}

// AWQ checkpoint tensors of one projection -> columns [col0, col0+N) of the staging tensors
// (the reference's AwqMarlinLinear::load_weights steps: awq_marlin.rs:621-656).
int stage_put(MergeStage &s, const uint32_t *qweight, const uint32_t *qzeros, const uint16_t *scales_f16, int K, int N, int group,
              int Ntot, int col0, hipStream_t st) {
    if (int rc = vmx_awq_convert_into(qweight, qzeros, (uint32_t *)s.qweight.p, (uint32_t *)s.qzeros.p, K, N, group, Ntot, col0, st))
        return rc;
    const int64_t n = (int64_t)(K / group) * N;
    hipLaunchKernelGGL(f16_to_bf16_into_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, scales_f16,
                       (uint16_t *)s.scales.p + col0, n, N, Ntot);
    VMX_CHECK_LAUNCH("scales f16->bf16");
    return VMX_OK;
}

__global__ __launch_bounds__(256) void fill_u32_cols_kernel(uint32_t *__restrict__ dst, uint32_t value, int64_t n, int cols, int ld) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[(i / cols) * ld + (i % cols)] = value;
}

// GPTQ checkpoint tensors of one projection (qweight u32 [K/8, N] with eight k per word, qzeros u32 [K/group, N/8] with eight n
// per word or NULL for the symmetric zero point 8, scales f16 [K/group, N]) -> columns [col0, col0+N) of the staging tensors, whose
// layout is that same GPTQ-sequential form (what awq_to_gptq_qweight / repack_awq_nibbles produce for AWQ input).
int stage_put_gptq(MergeStage &s, const uint32_t *qweight, const uint32_t *qzeros, const uint16_t *scales_f16, int K, int N, int group,
                   int Ntot, int col0, hipStream_t st) {
    VMX_REQUIRE(N % 8 == 0 && col0 % 8 == 0, "load_gptq_linear: column range [%d, %d) not aligned to the 8-wide zero words", col0, col0 + N);
    VMX_CHECK_HIP(hipMemcpy2DAsync((uint32_t *)s.qweight.p + col0, (size_t)Ntot * 4, qweight, (size_t)N * 4, (size_t)N * 4, K / 8,
                                   hipMemcpyDefault, st));
    if (qzeros) {
        VMX_CHECK_HIP(hipMemcpy2DAsync((uint32_t *)s.qzeros.p + col0 / 8, (size_t)(Ntot / 8) * 4, qzeros, (size_t)(N / 8) * 4,
                                       (size_t)(N / 8) * 4, K / group, hipMemcpyDefault, st));
    } else {
        const int64_t n = (int64_t)(K / group) * (N / 8);
        hipLaunchKernelGGL(fill_u32_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint32_t *)s.qzeros.p + col0 / 8,
                           0x88888888u, n, N / 8, Ntot / 8);
    }
    const int64_t n = (int64_t)(K / group) * N;
    hipLaunchKernelGGL(f16_to_bf16_into_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, scales_f16,
                       (uint16_t *)s.scales.p + col0, n, N, Ntot);
    VMX_CHECK_LAUNCH("gptq staging");
    return VMX_OK;
}

int stage_pack(MergeStage &s, Int4Linear &l, hipStream_t st, bool gate_up = false) {
    if (!l.packed.p)
        if (int rc0 = l.packed.alloc(vmx_int4_packed_bytes(l.in_features, l.out_features, l.group))) return rc0;
    l.glu = gate_up && vmx_int4_glu_fusable(l.out_features, l.in_features);
    int rc = vmx_int4_pack_cols((const uint32_t *)s.qweight.p, (const uint16_t *)s.scales.p, (const uint32_t *)s.qzeros.p, l.packed.p,
                                l.in_features, l.out_features, l.group, l.glu ? 1 : 0, st);
    if (rc) return rc;
    VMX_CHECK_HIP(hipStreamSynchronize(st));  // load time: staging is freed right after
    stage_free(s);
    l.loaded = true;
    return VMX_OK;
}

}  // namespace

extern "C" int vmx_qwen3_create(const vmx_qwen3_config *cfg, vmx_qwen3 **out) {
    VMX_REQUIRE(cfg && out, "qwen3_create: null pointer");
    const vmx_qwen3_config &c = *cfg;
    VMX_REQUIRE(c.hidden_size > 0 && c.hidden_size % 128 == 0 && c.hidden_size <= 16384,
                "qwen3_create: hidden_size (%d) must be a multiple of 128, at most 16384", c.hidden_size);
    VMX_REQUIRE(c.intermediate_size > 0 && c.intermediate_size % 128 == 0, "qwen3_create: intermediate_size (%d) must be a multiple of 128",
                c.intermediate_size);
    VMX_REQUIRE(c.num_layers > 0 && c.num_heads > 0 && c.num_kv_heads > 0 && c.num_heads % c.num_kv_heads == 0,
                "qwen3_create: bad head counts");
    VMX_REQUIRE(c.num_heads / c.num_kv_heads <= 16, "qwen3_create: %d query heads per KV head (the attention kernels take at most 16)",
                c.num_heads / c.num_kv_heads);
    VMX_REQUIRE(c.max_prefill_tokens >= 0 && c.max_prefill_tokens <= 8192, "qwen3_create: max_prefill_tokens (%d) must be in 0..8192",
                c.max_prefill_tokens);
    VMX_REQUIRE(c.head_dim == 64 || c.head_dim == 128 || c.head_dim == 256 || c.head_dim == 512, "qwen3_create: head_dim %d unsupported",
                c.head_dim);
    VMX_REQUIRE(c.max_batch >= 1 && c.max_batch <= 16, "qwen3_create: max_batch (%d) must be in 1..16", c.max_batch);
    VMX_REQUIRE(c.group_size > 0 && c.group_size % 128 == 0, "qwen3_create: group_size (%d) must be a multiple of 128", c.group_size);
    VMX_REQUIRE(c.block_size > 0 && c.num_blocks > 0 && c.max_seq_len > 0 && c.vocab_size > 0 && c.max_position > 0,
                "qwen3_create: bad cache/vocab geometry");
    VMX_REQUIRE(c.kv_cache_dtype == VMX_KV_AUTO || c.kv_cache_dtype == VMX_KV_FP8_E4M3, "qwen3_create: unsupported kv dtype");
    VMX_REQUIRE(c.tp_size >= 0 && c.tp_rank >= 0 && (c.tp_size <= 1 || (c.tp_rank < c.tp_size && c.vocab_size % c.tp_size == 0)),
                "qwen3_create: tp_rank %d / tp_size %d invalid (vocab_size must divide by tp_size)", c.tp_rank, c.tp_size);
    VMX_REQUIRE(c.num_experts >= 0 && c.num_experts <= 256 && (c.num_experts == 0 || (c.experts_per_tok >= 1 && c.experts_per_tok <= c.num_experts)),
                "qwen3_create: num_experts %d / experts_per_tok %d invalid", c.num_experts, c.experts_per_tok);
    // tensor-parallel attention + expert-parallel MLP: rank r owns experts [r * E / tp, (r + 1) * E / tp) (ExpertPlacement::Linear,
    // crates/core/src/moe/expert_map.rs:100-109); the decode batch is replicated, so the combine is one all-reduce, no all-to-all
    VMX_REQUIRE(c.num_experts == 0 || c.tp_size <= 1 || c.num_experts % c.tp_size == 0,
                "qwen3_create: num_experts (%d) must divide by tp_size (%d)", c.num_experts, c.tp_size);
    vmx_qwen3 *m = new (std::nothrow) vmx_qwen3();
    VMX_REQUIRE(m, "qwen3_create: out of host memory");
    m->cfg = c;
    m->layers.resize(c.num_layers);
    // activation buffers hold max(decode batch, prefill chunk) token rows
    const int H = c.hidden_size, I = c.intermediate_size, hd = c.head_dim, B = c.max_batch > c.max_prefill_tokens ? c.max_batch : c.max_prefill_tokens;
    const int NQKV = m->qkv_n(), NO = c.num_heads * hd;
    int rc = VMX_OK;
#define TRY(e) do { if (!rc) rc = (e); } while (0)
    TRY(m->embed.alloc((size_t)c.vocab_size * H * 2));
    m->lm_head_tied = c.tie_word_embeddings != 0;
    const int VL = m->vocab_local();  // lm_head rows held by this rank (vocab-parallel under TP)
    if (!m->lm_head_tied) TRY(m->lm_head.alloc((size_t)VL * H * 2));
    TRY(m->final_norm.alloc((size_t)H * 2));
    TRY(m->cos_sin.alloc((size_t)c.max_position * hd * 4));
    const size_t kv_elem = c.kv_cache_dtype == VMX_KV_FP8_E4M3 ? 1 : 2;
    const size_t kv_bytes = (size_t)c.num_blocks * c.block_size * c.num_kv_heads * hd * kv_elem;
    size_t max_partial = 0;
    for (auto &L : m->layers) {
        TRY(make_linear(L.qkv, H, NQKV, c.group_size));
        TRY(make_linear(L.o, NO, H, c.group_size));
        TRY(make_linear(L.gate_up, H, 2 * I, c.group_size));
        TRY(make_linear(L.down, I, H, c.group_size));
        TRY(L.input_ln.alloc((size_t)H * 2));
        TRY(L.post_ln.alloc((size_t)H * 2));
        TRY(L.q_norm.alloc((size_t)hd * 2));
        TRY(L.k_norm.alloc((size_t)hd * 2));
        TRY(L.k_cache.alloc(kv_bytes));
        TRY(L.v_cache.alloc(kv_bytes));
        TRY(L.kv_scales.alloc(8));
        if (c.num_experts > 0) {
            TRY(L.router.alloc((size_t)c.num_experts * H * 2));
            TRY(L.w13_table.alloc((size_t)c.num_experts * sizeof(void *)));
            TRY(L.w2_table.alloc((size_t)c.num_experts * sizeof(void *)));
            // a null entry = an expert another rank owns: its (token, slot) pairs are skipped here
            if (!rc && (hipMemset(L.w13_table.p, 0, (size_t)c.num_experts * sizeof(void *)) != hipSuccess ||
                        hipMemset(L.w2_table.p, 0, (size_t)c.num_experts * sizeof(void *)) != hipSuccess))
                rc = VMX_ELAUNCH;
            L.w13.resize(c.num_experts);
            L.w2.resize(c.num_experts);
        }
        if (!rc) {
            const float ones[2] = {1.0f, 1.0f};
            if (hipMemcpy(L.kv_scales.p, ones, 8, hipMemcpyHostToDevice) != hipSuccess) rc = VMX_ELAUNCH;
        }
    }
    if (!rc) {
        const Layer &L = m->layers[0];
        const int qkv_wide = vmx_int4_norm_splits(1, NQKV, H, 0);   // decode rows may take the wide kernel's k slicing
        const size_t cands[4] = {(size_t)(L.qkv.splits() > qkv_wide ? L.qkv.splits() : qkv_wide) * NQKV, (size_t)L.o.splits() * H, (size_t)L.gate_up.splits() * 2 * I,
                                 (size_t)L.down.splits() * H};
        for (size_t v : cands) max_partial = v > max_partial ? v : max_partial;
        if (L.qkv.splits() > SMAX || L.o.splits() > SMAX || L.gate_up.splits() > SMAX || L.down.splits() > SMAX) {
            vmx_set_error("qwen3_create: a linear needs more than %d split-K slices (K too large)", SMAX);
            rc = VMX_EUNSUPPORTED;
        }
    }
    TRY(m->partial.alloc(max_partial * B * 4));
    TRY(m->partial_b.alloc(max_partial * B * 4));
    TRY(m->h.alloc((size_t)B * H * 2));
    TRY(m->x.alloc((size_t)B * H * 2));
    TRY(m->q.alloc((size_t)B * NO * 2));
    TRY(m->attn.alloc((size_t)B * NO * 2));
    TRY(m->act.alloc((size_t)B * I * 2));
    TRY(m->logits.alloc((size_t)c.max_batch * VL * 2));
    TRY(m->positions.alloc((size_t)B * 4));
    TRY(m->ssq.alloc((size_t)4 * (H / 16 + 4) * 4));   // per-tile sums of squares of up to four decode rows
    TRY(m->red_bf16.alloc((size_t)B * H * 2));
    {   // q8_1 staging for I-quant linears: int8 [B][Kmax] + f32 [B][Kmax/32]
        const size_t kmax = (size_t)(H > I ? (H > NO ? H : NO) : (I > NO ? I : NO));
        TRY(m->q8.alloc(B * kmax + 256 + B * (kmax / 32) * 4));
    }
    if (c.num_experts > 0) {
        TRY(m->router_logits.alloc((size_t)B * c.num_experts * 2));
        TRY(m->topk_w.alloc((size_t)B * c.experts_per_tok * 4));
        TRY(m->topk_ids.alloc((size_t)B * c.experts_per_tok * 4));
        TRY(m->moe_ws.alloc(vmx_moe_int4_workspace_bytes(B, c.experts_per_tok, H, I)));
        TRY(m->moe_out.alloc((size_t)B * H * 2));
    }
    TRY(m->keys.alloc((size_t)c.max_batch * 8));
    {
        const size_t a = vmx_paged_attention_workspace_bytes(c.max_batch, c.num_heads, hd, c.max_seq_len);
        size_t b = vmx_split_decode_attention_workspace_bytes(c.max_batch, c.num_heads, hd);
        const int lp = vmx_split_decode_attention_long_parts(hd, c.num_heads / c.num_kv_heads, hd, c.max_seq_len, 0, c.num_kv_heads, 1);
        const size_t l = lp > 0 ? vmx_split_decode_attention_long_workspace_bytes(c.max_batch, c.num_heads, hd, 64) : 0;   // any count
        if (l > b) b = l;
        TRY(m->attn_ws.alloc(a > b ? a : b));
    }
    TRY(m->argmax_ws.alloc(vmx_argmax_workspace_bytes(c.max_batch, VL)));
#undef TRY
    if (rc) {
        vmx_qwen3_destroy(m);
        return rc;
    }
    *out = m;
    return VMX_OK;
}

extern "C" void vmx_qwen3_destroy(vmx_qwen3 *m) {
    if (!m) return;
    for (auto &L : m->layers) {
        for (IqLinear &l : L.iq) l.raw.release();
        for (DevBuf &b : L.w13) b.release();
        for (DevBuf &b : L.w2) b.release();
        for (DevBuf *b : {&L.router, &L.w13_table, &L.w2_table}) b->release();
        for (Exl3Lin &e : L.ex)
            for (DevBuf *b : {&e.trellis, &e.suh, &e.svh}) b->release();
        L.ex_svh_qkv.release();
        L.ex_svh_gu.release();
        for (DevBuf &b : L.dn) b.release();
        for (DevBuf *b : {&L.qkv.packed, &L.o.packed, &L.gate_up.packed, &L.down.packed, &L.input_ln, &L.post_ln, &L.q_norm, &L.k_norm,
                          &L.k_cache, &L.v_cache, &L.kv_scales})
            b->release();
        stage_free(L.qkv_stage);
        stage_free(L.gu_stage);
    }
    for (DevBuf *b : {&m->embed, &m->lm_head, &m->final_norm, &m->cos_sin, &m->h, &m->x, &m->q, &m->attn, &m->act, &m->partial, &m->partial_b,
                      &m->logits, &m->attn_ws, &m->argmax_ws, &m->red_bf16, &m->keys, &m->q8, &m->positions, &m->router_logits, &m->topk_w, &m->topk_ids,
                      &m->moe_ws, &m->moe_out, &m->exl3_ws, &m->ssq, &m->dense_tmp})
        b->release();
    delete m;
}

extern "C" int vmx_qwen3_load_tensor(vmx_qwen3 *m, int which, int layer, const void *src, size_t bytes, void *stream) {
    VMX_REQUIRE(m && src, "qwen3_load_tensor: null pointer");
    DevBuf *dst = nullptr;
    switch (which) {
        case VMX_T_EMBED: dst = &m->embed; break;
        case VMX_T_LM_HEAD: dst = m->lm_head_tied ? nullptr : &m->lm_head; break;
        case VMX_T_FINAL_NORM: dst = &m->final_norm; break;
        case VMX_T_COS_SIN: dst = &m->cos_sin; break;
        default: break;
    }
    if (!dst && which == VMX_T_ROUTER) {
        VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size() && m->cfg.num_experts > 0, "qwen3_load_tensor: no router on this model / layer %d",
                    layer);
        dst = &m->layers[layer].router;
    }
    if (!dst && which >= VMX_T_INPUT_LN && which <= VMX_T_KV_SCALES) {
        VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size(), "qwen3_load_tensor: layer %d out of range", layer);
        Layer &L = m->layers[layer];
        DevBuf *tab[5] = {&L.input_ln, &L.post_ln, &L.q_norm, &L.k_norm, &L.kv_scales};
        dst = tab[which - VMX_T_INPUT_LN];
    }
    VMX_REQUIRE(dst, "qwen3_load_tensor: tensor id %d not loadable on this model", which);
    VMX_REQUIRE(bytes == dst->bytes, "qwen3_load_tensor: tensor %d expects %zu bytes, got %zu", which, dst->bytes, bytes);
    VMX_CHECK_HIP(hipMemcpyAsync(dst->p, src, bytes, hipMemcpyDefault, vmx_stream(stream)));
    return VMX_OK;
}

namespace {
// format 0: AWQ tensors (stage_put), 1: GPTQ tensors (stage_put_gptq; qzeros may be NULL)
int load_int4_linear(vmx_qwen3 *m, int layer, int slot, const uint32_t *qweight, const uint32_t *qzeros, const uint16_t *scales_f16, int format,
                     void *stream) {
    const vmx_qwen3_config &c = m->cfg;
    Layer &L = m->layers[layer];
    hipStream_t st = vmx_stream(stream);
    const int H = c.hidden_size, I = c.intermediate_size, hd = c.head_dim, g = c.group_size;
    const int NQ = c.num_heads * hd, NKV = c.num_kv_heads * hd;
    auto put = [&](MergeStage &stg, int K, int N, int Ntot, int col0) {
        return format == 0 ? stage_put(stg, qweight, qzeros, scales_f16, K, N, g, Ntot, col0, st)
                           : stage_put_gptq(stg, qweight, qzeros, scales_f16, K, N, g, Ntot, col0, st);
    };
    MergeStage single;
    switch (slot) {
        case VMX_L_Q: case VMX_L_K: case VMX_L_V: {
            const int Ntot = m->qkv_n();
            if (int rc = stage_alloc(L.qkv_stage, H, Ntot, g)) return rc;
            const int N = slot == VMX_L_Q ? NQ : NKV;
            const int col0 = slot == VMX_L_Q ? 0 : (slot == VMX_L_K ? NQ : NQ + NKV);
            if (int rc = put(L.qkv_stage, H, N, Ntot, col0)) return rc;
            L.qkv_stage.filled |= 1 << (slot - VMX_L_Q);
            if (L.qkv_stage.filled == 7) return stage_pack(L.qkv_stage, L.qkv, st);
            return VMX_OK;
        }
        case VMX_L_GATE: case VMX_L_UP: {
            if (int rc = stage_alloc(L.gu_stage, H, 2 * I, g)) return rc;
            if (int rc = put(L.gu_stage, H, I, 2 * I, slot == VMX_L_GATE ? 0 : I)) return rc;
            L.gu_stage.filled |= slot == VMX_L_GATE ? 1 : 2;
            if (L.gu_stage.filled == 3) return stage_pack(L.gu_stage, L.gate_up, st, true);
            return VMX_OK;
        }
        case VMX_L_O: {
            if (int rc = stage_alloc(single, NQ, H, g)) return rc;
            int rc = put(single, NQ, H, H, 0);
            if (!rc) rc = stage_pack(single, L.o, st);
            stage_free(single);
            return rc;
        }
        case VMX_L_DOWN: {
            if (int rc = stage_alloc(single, I, H, g)) return rc;
            int rc = put(single, I, H, H, 0);
            if (!rc) rc = stage_pack(single, L.down, st);
            stage_free(single);
            return rc;
        }
        default:
            vmx_set_error("qwen3_load_int4_linear: unknown slot %d", slot);
            return VMX_EINVAL;
    }
}
}  // namespace

extern "C" int vmx_qwen3_load_awq_linear(vmx_qwen3 *m, int layer, int slot, const uint32_t *qweight, const uint32_t *qzeros,
                                         const uint16_t *scales_f16, void *stream) {
    VMX_REQUIRE(m && qweight && qzeros && scales_f16, "qwen3_load_awq_linear: null pointer");
    VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size(), "qwen3_load_awq_linear: layer %d out of range", layer);
    return load_int4_linear(m, layer, slot, qweight, qzeros, scales_f16, 0, stream);
}

// GptqLinear / MarlinLinear::load_weights (crates/core/src/quantization/gptq.rs:430-460, marlin.rs:473-560) for one projection of a
// 4-bit GPTQ checkpoint: qweight u32 [in/8, out], scales f16 [in/group, out], qzeros u32 [in/group, out/8] or NULL.
// With qzeros the weight is (nib - z) * s with z as stored (gptq_dequant.cu:48-87); without, (nib - 8) * s, the Marlin form that
// ignores the zero tensor (marlin_gemm.cu:43-48).  g_idx is not taken: like the reference's kernels, groups follow k / group_size.
extern "C" int vmx_qwen3_load_gptq_linear(vmx_qwen3 *m, int layer, int slot, const uint32_t *qweight, const uint32_t *qzeros,
                                          const uint16_t *scales_f16, void *stream) {
    VMX_REQUIRE(m && qweight && scales_f16, "qwen3_load_gptq_linear: null pointer");
    VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size(), "qwen3_load_gptq_linear: layer %d out of range", layer);
    return load_int4_linear(m, layer, slot, qweight, qzeros, scales_f16, 1, stream);
}

// One GPTQ expert of a mixture-of-experts layer (MarlinLinear semantics, w = (nib - 8) * scale, qzeros / g_idx ignored:
// crates/core/kernels/marlin_gemm.cu:43-48): w1 (gate) and w3 (up) [hidden -> inter] are merged into one tile stream,
// w2 (down) [inter -> hidden] is the second.  qweight u32 [K/8, N] GPTQ-sequential, scales f16 [K/group, N].
extern "C" int vmx_qwen3_load_gptq_expert(vmx_qwen3 *m, int layer, int expert, const uint32_t *w1_qweight, const uint16_t *w1_scales_f16,
                                          const uint32_t *w3_qweight, const uint16_t *w3_scales_f16, const uint32_t *w2_qweight,
                                          const uint16_t *w2_scales_f16, void *stream) {
    VMX_REQUIRE(m && w1_qweight && w1_scales_f16 && w3_qweight && w3_scales_f16 && w2_qweight && w2_scales_f16,
                "qwen3_load_gptq_expert: null pointer");
    const vmx_qwen3_config &c = m->cfg;
    VMX_REQUIRE(c.num_experts > 0, "qwen3_load_gptq_expert: the model has no experts");
    VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size() && expert >= 0 && expert < c.num_experts,
                "qwen3_load_gptq_expert: layer %d / expert %d out of range", layer, expert);
    if (c.tp_size > 1) {
        const int per = c.num_experts / c.tp_size;
        VMX_REQUIRE(expert / per == c.tp_rank, "qwen3_load_gptq_expert: expert %d belongs to rank %d, this is rank %d of %d", expert,
                    expert / per, c.tp_rank, c.tp_size);
    }
    hipStream_t st = vmx_stream(stream);
    Layer &L = m->layers[layer];
    const int H = c.hidden_size, I = c.intermediate_size, g = c.group_size;
    DevBuf qw13, sc13, sc2;
    int rc = qw13.alloc((size_t)(H / 8) * 2 * I * 4);
    if (!rc) rc = sc13.alloc((size_t)(H / g) * 2 * I * 2);
    if (!rc) rc = sc2.alloc((size_t)(I / g) * H * 2);
    const bool fresh = L.w13[expert].p == nullptr;
    if (!rc && fresh) rc = L.w13[expert].alloc(vmx_int4_packed_bytes(H, 2 * I, g));
    if (!rc && !L.w2[expert].p) rc = L.w2[expert].alloc(vmx_int4_packed_bytes(I, H, g));
    auto fail = [&](int code) { qw13.release(); sc13.release(); sc2.release(); return code; };
    if (rc) return fail(rc);
    // gate | up side by side: columns [0, I) and [I, 2I)
    const size_t row = (size_t)I * 4;
    if (hipMemcpy2DAsync(qw13.p, 2 * row, w1_qweight, row, row, H / 8, hipMemcpyDefault, st) != hipSuccess ||
        hipMemcpy2DAsync((char *)qw13.p + row, 2 * row, w3_qweight, row, row, H / 8, hipMemcpyDefault, st) != hipSuccess) {
        vmx_set_error("qwen3_load_gptq_expert: copying qweight failed");
        return fail(VMX_ELAUNCH);
    }
    const int64_t n13 = (int64_t)(H / g) * I, n2 = (int64_t)(I / g) * H;
    hipLaunchKernelGGL(f16_to_bf16_into_kernel, dim3((unsigned)((n13 + 255) / 256)), dim3(256), 0, st, w1_scales_f16, (uint16_t *)sc13.p, n13, I, 2 * I);
    hipLaunchKernelGGL(f16_to_bf16_into_kernel, dim3((unsigned)((n13 + 255) / 256)), dim3(256), 0, st, w3_scales_f16, (uint16_t *)sc13.p + I, n13, I, 2 * I);
    hipLaunchKernelGGL(f16_to_bf16_into_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, w2_scales_f16, (uint16_t *)sc2.p, n2, H, H);
    if (hipGetLastError() != hipSuccess) { vmx_set_error("qwen3_load_gptq_expert: scale conversion launch failed"); return fail(VMX_ELAUNCH); }
    rc = vmx_int4_pack_cols((const uint32_t *)qw13.p, (const uint16_t *)sc13.p, nullptr, L.w13[expert].p, H, 2 * I, g,
                            vmx_int4_glu_fusable(2 * I, H) ? 1 : 0, st);   // gate|up interleaved per tile for the fused activation
    if (!rc) rc = vmx_int4_pack(w2_qweight, (const uint16_t *)sc2.p, nullptr, L.w2[expert].p, I, H, g, st);
    if (rc) return fail(rc);
    void *p13 = L.w13[expert].p, *p2 = L.w2[expert].p;
    if (hipMemcpyAsync((char *)L.w13_table.p + expert * sizeof(void *), &p13, sizeof(void *), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync((char *)L.w2_table.p + expert * sizeof(void *), &p2, sizeof(void *), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {   // load time: the staging buffers are freed right after
        vmx_set_error("qwen3_load_gptq_expert: table update failed");
        return fail(VMX_ELAUNCH);
    }
    if (fresh) L.experts_loaded++;
    return fail(VMX_OK);
}

extern "C" int vmx_qwen3_load_iq_linear(vmx_qwen3 *m, int layer, int slot, int ggml_type, const void *blocks, size_t bytes,
                                        void *stream) {
    VMX_REQUIRE(m && blocks, "qwen3_load_iq_linear: null pointer");
    VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size() && slot >= VMX_L_Q && slot <= VMX_L_DOWN,
                "qwen3_load_iq_linear: layer %d / slot %d out of range", layer, slot);
    const int ts = vmx_iq_type_size(ggml_type);
    VMX_REQUIRE(ts > 0, "qwen3_load_iq_linear: ggml type %d is not a supported I-quant", ggml_type);
    const vmx_qwen3_config &c = m->cfg;
    const int H = c.hidden_size, I = c.intermediate_size, NQ = c.num_heads * c.head_dim, NKV = c.num_kv_heads * c.head_dim;
    const int in_f[7] = {H, H, H, NQ, H, H, I}, out_f[7] = {NQ, NKV, NKV, H, I, I, H};
    const int K = in_f[slot], N = out_f[slot];
    VMX_REQUIRE(K % 256 == 0, "qwen3_load_iq_linear: in_features %d not a multiple of 256", K);
    const size_t need = (size_t)N * (K / 256) * ts;
    VMX_REQUIRE(bytes == need, "qwen3_load_iq_linear: expected %zu bytes for [%d, %d] of type %d, got %zu", need, N, K, ggml_type, bytes);
    IqLinear &l = m->layers[layer].iq[slot];
    l.raw.release();
    if (int rc = l.raw.alloc(need)) return rc;
    VMX_CHECK_HIP(hipMemcpyAsync(l.raw.p, blocks, need, hipMemcpyDefault, vmx_stream(stream)));
    l.ggml_type = ggml_type;
    l.in_features = K;
    l.out_features = N;
    l.loaded = true;
    return VMX_OK;
}

// One projection of an unquantized checkpoint: bf16 [out, in] (the reference's dense Linear of models/qwen3.rs).  q, k, v and
// gate, up land in merged matrices (rows concatenated), so a layer is four dense GEMVs.
extern "C" int vmx_qwen3_load_dense_linear(vmx_qwen3 *m, int layer, int slot, const uint16_t *weight_bf16, size_t bytes, void *stream) {
    VMX_REQUIRE(m && weight_bf16, "qwen3_load_dense_linear: null pointer");
    VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size() && slot >= VMX_L_Q && slot <= VMX_L_DOWN,
                "qwen3_load_dense_linear: layer %d / slot %d out of range", layer, slot);
    VMX_REQUIRE(m->cfg.num_experts == 0, "qwen3_load_dense_linear: mixture-of-experts models take int4 projections");
    const vmx_qwen3_config &c = m->cfg;
    const int H = c.hidden_size, I = c.intermediate_size, NQ = c.num_heads * c.head_dim, NKV = c.num_kv_heads * c.head_dim;
    const int in_f[7] = {H, H, H, NQ, H, H, I}, out_f[7] = {NQ, NKV, NKV, H, I, I, H};
    const int buf_of[7] = {0, 0, 0, 1, 2, 2, 3}, row0[7] = {0, NQ, NQ + NKV, 0, 0, I, 0}, rows_of[4] = {NQ + 2 * NKV, H, 2 * I, H};
    const int K = in_f[slot], N = out_f[slot];
    VMX_REQUIRE(K % 8 == 0, "qwen3_load_dense_linear: in_features %d not a multiple of 8", K);
    const size_t need = (size_t)N * K * 2;
    VMX_REQUIRE(bytes == need, "qwen3_load_dense_linear: expected %zu bytes for bf16 [%d, %d], got %zu", need, N, K, bytes);
    Layer &L = m->layers[layer];
    DevBuf &b = L.dn[buf_of[slot]];
    if (!b.p)
        if (int rc = b.alloc((size_t)rows_of[buf_of[slot]] * K * 2)) return rc;
    VMX_CHECK_HIP(hipMemcpyAsync((char *)b.p + (size_t)row0[slot] * K * 2, weight_bf16, need, hipMemcpyDefault, vmx_stream(stream)));
    const size_t tmp_need = (size_t)16 * (2 * I > NQ + 2 * NKV ? 2 * I : NQ + 2 * NKV) * 2;
    if (m->dense_tmp.bytes < tmp_need) {
        VMX_CHECK_HIP(hipStreamSynchronize(vmx_stream(stream)));
        m->dense_tmp.release();
        if (int rc = m->dense_tmp.alloc(tmp_need)) return rc;
    }
    L.dn_filled |= 1 << slot;
    return VMX_OK;
}

extern "C" int vmx_qwen3_load_exl3_linear(vmx_qwen3 *m, int layer, int slot, const void *trellis, const uint16_t *suh,
                                          const uint16_t *svh, int in_features, int out_features, int bits, int codebook, void *stream) {
    VMX_REQUIRE(m && trellis && suh && svh, "qwen3_load_exl3_linear: null pointer");
    VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size() && slot >= VMX_L_Q && slot <= VMX_L_DOWN,
                "qwen3_load_exl3_linear: layer %d / slot %d out of range", layer, slot);
    VMX_REQUIRE(bits >= 1 && bits <= 8, "qwen3_load_exl3_linear: bits per weight (%d) must be in 1..8", bits);
    VMX_REQUIRE(codebook >= 0 && codebook <= 2, "qwen3_load_exl3_linear: codebook %d unknown (0 default, 1 mcg, 2 mul1)", codebook);
    const vmx_qwen3_config &c = m->cfg;
    const int H = c.hidden_size, I = c.intermediate_size, NQ = c.num_heads * c.head_dim, NKV = c.num_kv_heads * c.head_dim;
    const int in_f[7] = {H, H, H, NQ, H, H, I}, out_f[7] = {NQ, NKV, NKV, H, I, I, H};
    const int K = in_f[slot], N = out_f[slot];
    VMX_REQUIRE(in_features == K && out_features == N, "qwen3_load_exl3_linear: trellis is [%d -> %d], the model's projection is [%d -> %d]",
                in_features, out_features, K, N);
    VMX_REQUIRE(K % 128 == 0 && N % 128 == 0, "qwen3_load_exl3_linear: in_features (%d) and out_features (%d) must be multiples of 128", K, N);
    hipStream_t st = vmx_stream(stream);
    Exl3Lin &l = m->layers[layer].ex[slot];
    for (DevBuf *b : {&l.trellis, &l.suh, &l.svh}) b->release();
    const size_t tb = (size_t)(K / 16) * (N / 16) * 16 * bits * 2;
    if (int rc = l.trellis.alloc(tb)) return rc;
    if (int rc = l.suh.alloc((size_t)K * 2)) return rc;
    if (int rc = l.svh.alloc((size_t)N * 2)) return rc;
    VMX_CHECK_HIP(hipMemcpyAsync(l.trellis.p, trellis, tb, hipMemcpyDefault, st));
    VMX_CHECK_HIP(hipMemcpyAsync(l.suh.p, suh, (size_t)K * 2, hipMemcpyDefault, st));
    VMX_CHECK_HIP(hipMemcpyAsync(l.svh.p, svh, (size_t)N * 2, hipMemcpyDefault, st));
    const int rows = c.max_batch > c.max_prefill_tokens ? c.max_batch : c.max_prefill_tokens;
    const size_t need = vmx_exl3_gemm_workspace_bytes(rows, K, N);
    if (need > m->exl3_ws.bytes) {   // load time: no step is in flight
        VMX_CHECK_HIP(hipStreamSynchronize(st));
        m->exl3_ws.release();
        if (int rc = m->exl3_ws.alloc(need)) return rc;
    }
    l.in_features = K;
    l.out_features = N;
    l.bits = bits;
    l.codebook = codebook;
    l.loaded = true;
    // projections that share their input and agree in bit width and codebook are launched together: merged svh vectors
    Layer &Lr = m->layers[layer];
    auto merge = [&](DevBuf &dst, std::initializer_list<int> slots) -> int {
        dst.release();
        size_t total = 0;
        const Exl3Lin &first = Lr.ex[*slots.begin()];
        for (int sl : slots) {
            const Exl3Lin &e = Lr.ex[sl];
            if (!e.loaded || e.bits != first.bits || e.codebook != first.codebook) return VMX_OK;
            total += (size_t)e.out_features * 2;
        }
        if (int rc = dst.alloc(total)) return rc;
        size_t off = 0;
        for (int sl : slots) {
            const Exl3Lin &e = Lr.ex[sl];
            VMX_CHECK_HIP(hipMemcpyAsync((char *)dst.p + off, e.svh.p, (size_t)e.out_features * 2, hipMemcpyDeviceToDevice, st));
            off += (size_t)e.out_features * 2;
        }
        const size_t need_m = vmx_exl3_gemm_workspace_bytes(4, first.in_features, (int)(total / 2));
        if (need_m > m->exl3_ws.bytes) {
            VMX_CHECK_HIP(hipStreamSynchronize(st));
            m->exl3_ws.release();
            if (int rc = m->exl3_ws.alloc(need_m)) return rc;
        }
        return VMX_OK;
    };
    if (slot <= VMX_L_V) { if (int rc = merge(Lr.ex_svh_qkv, {VMX_L_Q, VMX_L_K, VMX_L_V})) return rc; }
    else if (slot == VMX_L_GATE || slot == VMX_L_UP) { if (int rc = merge(Lr.ex_svh_gu, {VMX_L_GATE, VMX_L_UP})) return rc; }
    return VMX_OK;
}

namespace {
// A mixture-of-experts model runs int4 attention projections plus the expert tables; a dense one needs all seven
// projections in ONE of the three formats.
bool layer_ready(const vmx_qwen3 *m, const Layer &L) {
    if (m->cfg.num_experts > 0) return L.int4_attn_ready() && L.moe_ready(m->cfg.num_experts / (m->cfg.tp_size > 1 ? m->cfg.tp_size : 1));
    return L.int4_ready() || L.iq_ready() || L.exl3_ready() || L.dense_ready();
}

// VMX_NO_FOLDED_LAYERS=1 keeps the seven-launch layer (norm kernels of their own) for A/B measurements
bool folded_layers_enabled() {
    static const bool on = !(vmx_tune("VMX_NO_FOLDED_LAYERS", 0) != 0);
    return on;
}

__global__ void iota_positions_kernel(int32_t *pos, int offset, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos[i] = offset + i;
}

// The layer stack on `batch` token rows, up to and including the final norm (m->x).  prefill_offset < 0: every row is one
// sequence's new token (paged decode attention over block_tables [batch, max_blocks_per_seq] / seq_lens); >= 0: the rows
// are consecutive tokens of ONE sequence at positions prefill_offset + i (prefill attention over block_tables [max_blocks]).
int run_layers(vmx_qwen3 *m, const int32_t *input_ids, const int32_t *positions, const int32_t *slot_mapping,
               const int32_t *block_tables, const int32_t *seq_lens, int batch, int max_blocks_per_seq, int max_seq_len,
               int prefill_offset, hipStream_t st) {
    const vmx_qwen3_config &c = m->cfg;
    const int H = c.hidden_size, I = c.intermediate_size, hd = c.head_dim, B = batch;
    const int Hq = c.num_heads, Hkv = c.num_kv_heads, NQKV_ = m->qkv_n();
    uint16_t *h = (uint16_t *)m->h.p, *x = (uint16_t *)m->x.p, *q = (uint16_t *)m->q.p;
    float *PA = (float *)m->partial.p, *PB = (float *)m->partial_b.p;  // ping-pong: each linear reads one, writes the other
    const float scale = 1.0f / sqrtf((float)hd);
    const int fp8 = c.kv_cache_dtype == VMX_KV_FP8_E4M3;
    const bool tp = c.tp_size > 1 || m->comm != nullptr;  // a 1-rank communicator still exercises the collective path
    VMX_REQUIRE(!tp || m->comm, "qwen3_decode: tp_size %d but no communicator attached (vmx_qwen3_set_comm)", c.tp_size);
    if (int rc = vmx_comm_check_health(m->comm, "qwen3 forward")) return rc;   // an earlier collective of this group failed
    // fp16 activations (EXL3 checkpoints): every tensor the bf16 form rounds to bf16 is rounded to fp16 instead, the cache holds fp16
    const bool f16 = m->act_f16;
    const int kvd = f16 ? VMX_KV_F16_INTERNAL : c.kv_cache_dtype;
    if (f16)
        for (const Layer &L : m->layers)
            VMX_REQUIRE(L.exl3_ready() && !L.int4_ready() && c.num_experts == 0,
                        "qwen3 forward: fp16 activations are built for models whose projections are all EXL3 linears");

    // the lookup is folded into the first layer's norm kernel (VMX_NO_EMBED_FUSION=1: its own launch)
    static const bool no_embed_fusion = vmx_tune("VMX_NO_EMBED_FUSION", 0) != 0;
    bool embed_pending = !no_embed_fusion;
    if (!embed_pending) hipLaunchKernelGGL(embed_kernel, dim3(B), dim3(256), 0, st, h, (const uint16_t *)m->embed.p, input_ids, H, c.vocab_size);
    const float *pending = nullptr;  // down-projection output (f32 slices in PB) not yet folded into the residual stream
    int pending_splits = 0;
    const uint16_t *pending_svh = nullptr;  // EXL3: `pending` holds raw split-K sums whose output transform the norm kernel applies
    uint16_t *attn = (uint16_t *)m->attn.p, *act = (uint16_t *)m->act.p;
    const int NQ = Hq * hd, NKV = Hkv * hd;
    float *ssq = (float *)m->ssq.p;
    bool have_ssq = false;   // ssq holds the tile sums of squares of the current h (left by a residual-epilogue linear)
    for (const Layer &L : m->layers) {
        const bool moe = c.num_experts > 0;  // int4 attention projections + mixture-of-experts MLP
        const bool ex = !moe && !L.int4_ready() && L.exl3_ready();   // EXL3 layer: every projection an Exl3Linear
        const bool dn = !moe && !L.int4_ready() && !ex && !L.iq_ready() && L.dense_ready();   // unquantized layer: bf16 matrices
        const bool iq = !moe && !L.int4_ready() && !ex && !dn;       // GGUF I-quant layer: every projection is an IqLinear
        const bool one = iq || ex || dn;                             // linear outputs arrive as one f32 slice
        int s_qkv = one ? 1 : L.qkv.splits();
        const int s_o = one ? 1 : L.o.splits(), s_gu = one ? 1 : L.gate_up.splits(),
                  s_down = one ? 1 : L.down.splits();
        // h += pending; x = rms_norm(h);  q|k|v = linear(x)
        // EXL3 decode rows: the consumers of o, gate|up and down apply the output transform themselves (one launch less each)
        const bool ex_fold = ex && B <= 4 && !tp && H % 128 == 0 && I % 128 == 0 && L.ex_svh_gu.p;
        // Dense int4 layer on one or two decode rows: both norms move into the prologue of the linear that reads them and both
        // residual adds into the epilogue of the linear that feeds them -- five launches (q|k|v, attention, o, gate|up, down).
        const bool fold = folded_layers_enabled() && !tp && !moe && !one && L.int4_ready() && L.gate_up.glu && !pending &&
                          vmx_int4_norm_splits(B, NQKV_, H, 0) > 0 && vmx_int4_norm_splits(B, 2 * I, H, 1) == 1 &&
                          vmx_int4_resid_fusable(B, H, NQ) && vmx_int4_resid_fusable(B, H, I);
        // Mixture-of-experts layer on one to four decode rows: the attention half takes the same forms (norm in the q|k|v prologue,
        // attention partitions merged by o_proj, residual add in o_proj's epilogue) and the expert combine adds the layer's
        // output to the residual stream itself, leaving the tile sums for the next layer's q|k|v.
        const bool mfold = moe && folded_layers_enabled() && !tp && !pending && prefill_offset < 0 && !m->rows_share_cache &&
                           L.int4_attn_ready() && H % 256 == 0 && NQ <= 4 * 16 * 128 && vmx_int4_norm_splits(B, NQKV_, H, 0) > 0 &&
                           vmx_int4_resid_fusable(B, H, NQ) && vmx_split_decode_attention_parts(hd, Hq / Hkv, hd, max_seq_len, 0) > 0;
        const bool fold_in = (fold || mfold) && have_ssq;   // the first layer's h comes from the embedding: no tile sums yet
        // EXL3 decode rows: the residual kernel leaves block sums of squares and q|k|v normalises its rows itself
        const bool ex_norm_in = ex && B <= 4 && L.ex_svh_qkv.p && pending_svh && exl3_resid_stats_ok(B, pending_splits, H);
        const uint16_t *ex_in = ex_norm_in ? h : x, *ex_in_w = ex_norm_in ? (const uint16_t *)L.input_ln.p : nullptr;
        if (embed_pending && (ex_norm_in || fold_in || pending)) {   // not reached: the first layer has neither pending output nor tile sums
            hipLaunchKernelGGL(embed_kernel, dim3(B), dim3(256), 0, st, h, (const uint16_t *)m->embed.p, input_ids, H, c.vocab_size);
            embed_pending = false;
        }
        if (ex_norm_in) launch_exl3_resid_stats(st, h, pending, pending_splits, B, H, pending_svh, ssq, f16);
        else if (!fold_in)
            launch_resid_norm(st, h, x, pending, pending_splits, B, (const uint16_t *)L.input_ln.p, c.rms_norm_eps, H,
                              iq ? q8_aq(m) : nullptr, iq ? q8_ad(m, B, H) : nullptr, pending_svh,
                              embed_pending ? (const uint16_t *)m->embed.p : nullptr, input_ids, c.vocab_size, f16);
        embed_pending = false;
        pending_svh = nullptr;
        have_ssq = false;
        if (fold_in) s_qkv = vmx_int4_norm_splits(B, NQKV_, H, 0);   // the wide kernel has its own k slicing
        const float *qkv_src = PA;
        const uint16_t *qkv_svh = nullptr;
        bool ex_split = false;
        if (fold_in) {
            if (int rc = vmx_int4_gemm_partial_norm(L.qkv.packed.p, h, (const uint16_t *)L.input_ln.p, c.rms_norm_eps, ssq, PA, B, NQKV_, H,
                                                    L.qkv.group, st))
                return rc;
        } else if (ex && B <= 4 && L.ex_svh_qkv.p) {
            // a 128-wide head is one Hadamard block: the fused attention kernel applies the output transform of q|k|v itself
            static const bool no_had_fusion = vmx_tune("VMX_EXL3_NO_HAD_FUSION", 0) != 0;
            const bool in_attn = !no_had_fusion && hd == 128 && prefill_offset < 0 && !m->rows_share_cache &&
                                 vmx_fused_decode_attention_ok(hd, Hq / Hkv, hd, max_seq_len, 0);
            // ... and so does the attention spread over several workgroups per KV head, whose partitions o_proj merges in its prologue
            static const bool no_ex_split = vmx_tune("VMX_EXL3_NO_SPLIT_ATTN", 0) != 0;
            const int ex_p = vmx_split_decode_attention_parts(hd, Hq / Hkv, hd, max_seq_len, 0);
            ex_split = in_attn && !no_ex_split && !tp && H % 128 == 0 && I % 128 == 0 && L.ex_svh_gu.p && ex_p >= 1 && ex_p <= 4;
            if (in_attn) {
                if (int rc = exl3_forward_group(m, L, {VMX_L_Q, VMX_L_K, VMX_L_V}, L.ex_svh_qkv, ex_in, nullptr, NQKV_, B, st, &qkv_src, &s_qkv,
                                                ex_in_w, ssq, c.rms_norm_eps))
                    return rc;
                qkv_svh = (const uint16_t *)L.ex_svh_qkv.p;
            } else if (int rc = exl3_forward_group(m, L, {VMX_L_Q, VMX_L_K, VMX_L_V}, L.ex_svh_qkv, ex_in, PA, NQKV_, B, st, nullptr, nullptr,
                                                   ex_in_w, ssq, c.rms_norm_eps))
                return rc;
        } else if (ex) {
            if (int rc = exl3_forward(m, L.ex[VMX_L_Q], x, PA, NQKV_, B, st)) return rc;
            if (int rc = exl3_forward(m, L.ex[VMX_L_K], x, PA + NQ, NQKV_, B, st)) return rc;
            if (int rc = exl3_forward(m, L.ex[VMX_L_V], x, PA + NQ + NKV, NQKV_, B, st)) return rc;
        } else if (dn) {
            if (int rc = dense_forward(m, L.dn[0].p, x, PA, NQKV_, NQKV_, H, B, st)) return rc;
        } else if (iq) {
            if (int rc = iq_forward(m, L.iq[VMX_L_Q], PA, NQKV_, B, st)) return rc;
            if (int rc = iq_forward(m, L.iq[VMX_L_K], PA + NQ, NQKV_, B, st)) return rc;
            if (int rc = iq_forward(m, L.iq[VMX_L_V], PA + NQ + NKV, NQKV_, B, st)) return rc;
        } else if (int rc = run_linear(m, L.qkv, x, PA, B, st)) return rc;
        const float *ks = (const float *)L.kv_scales.p;
        // five-launch layers: attention spread over several workgroups per KV head, merged by o_proj while it stages its activations
        const bool split_attn = ex_split || ((fold || mfold) && prefill_offset < 0 && !m->rows_share_cache && NQ <= 4 * 16 * 128 &&
                                             vmx_split_decode_attention_parts(hd, Hq / Hkv, hd, max_seq_len, 0) > 0);
        VmxAttnParts sparts{};
        // long contexts: the same kernel with many partitions and the separate merge instead of the three-launch path
        int long_p = !split_attn && prefill_offset < 0 && !m->rows_share_cache
                         ? vmx_split_decode_attention_long_parts(hd, Hq / Hkv, hd, max_seq_len, 0, Hkv, B) : 0;
        if (long_p > 0 && vmx_split_decode_attention_long_workspace_bytes(B, Hq, hd, long_p) > m->attn_ws.bytes) long_p = 0;
        const bool fused_attn = !split_attn && long_p == 0 && prefill_offset < 0 && !m->rows_share_cache &&
                                vmx_fused_decode_attention_ok(hd, Hq / Hkv, hd, max_seq_len, 0);
        if (long_p > 0) {
            if (int rc = vmx_split_decode_attention(&sparts, m->attn_ws.p, qkv_src, s_qkv, L.k_cache.p, L.v_cache.p,
                                                    c.qk_norm ? (const uint16_t *)L.q_norm.p : nullptr,
                                                    c.qk_norm ? (const uint16_t *)L.k_norm.p : nullptr, c.rms_norm_eps,
                                                    (const float *)m->cos_sin.p, positions, slot_mapping, block_tables, seq_lens, scale, B, Hq,
                                                    Hkv, hd, hd, max_blocks_per_seq, max_seq_len, c.block_size, 0, kvd, ks, ks + 1, 0,
                                                    st, qkv_svh, long_p))
                return rc;
            if (int rc = vmx_paged_attention_merge(attn, &sparts, seq_lens, B, Hq, hd, 0, st, iq ? q8_aq(m) : nullptr,
                                                   iq ? q8_ad(m, B, NQ) : nullptr, f16 ? 1 : 0))
                return rc;
        } else if (split_attn) {
            if (int rc = vmx_split_decode_attention(&sparts, m->attn_ws.p, qkv_src, s_qkv, L.k_cache.p, L.v_cache.p,
                                                    c.qk_norm ? (const uint16_t *)L.q_norm.p : nullptr,
                                                    c.qk_norm ? (const uint16_t *)L.k_norm.p : nullptr, c.rms_norm_eps,
                                                    (const float *)m->cos_sin.p, positions, slot_mapping, block_tables, seq_lens, scale, B, Hq,
                                                    Hkv, hd, hd, max_blocks_per_seq, max_seq_len, c.block_size, 0, kvd, ks, ks + 1, 0,
                                                    st, qkv_svh))
                return rc;
        } else if (fused_attn) {
            if (int rc = vmx_fused_decode_attention(attn, qkv_src, s_qkv, L.k_cache.p, L.v_cache.p, c.qk_norm ? (const uint16_t *)L.q_norm.p : nullptr,
                                                    c.qk_norm ? (const uint16_t *)L.k_norm.p : nullptr, c.rms_norm_eps,
                                                    (const float *)m->cos_sin.p, positions, slot_mapping, block_tables, seq_lens, scale, B, Hq,
                                                    Hkv, hd, hd, max_blocks_per_seq, c.block_size, 0, kvd, ks, ks + 1,
                                                    iq ? q8_aq(m) : nullptr, iq ? q8_ad(m, B, NQ) : nullptr, 0, st, qkv_svh))
                return rc;
        } else if (f16)
            hipLaunchKernelGGL(qkv_post_kernel<true>, dim3(Hq + 2 * Hkv, B), dim3(64), 0, st, PA, s_qkv, B, q, L.k_cache.p,
                               L.v_cache.p, c.qk_norm ? (const uint16_t *)L.q_norm.p : nullptr,
                               c.qk_norm ? (const uint16_t *)L.k_norm.p : nullptr, c.rms_norm_eps, (const float *)m->cos_sin.p, positions,
                               slot_mapping, Hq, Hkv, hd, hd, 0, ks, ks + 1);
        else
        hipLaunchKernelGGL(qkv_post_kernel<false>, dim3(Hq + 2 * Hkv, B), dim3(64), 0, st, PA, s_qkv, B, q, L.k_cache.p,
                           L.v_cache.p, c.qk_norm ? (const uint16_t *)L.q_norm.p : nullptr,
                           c.qk_norm ? (const uint16_t *)L.k_norm.p : nullptr, c.rms_norm_eps, (const float *)m->cos_sin.p, positions,
                           slot_mapping, Hq, Hkv, hd, hd, fp8, ks, ks + 1);
        if (fused_attn || split_attn || long_p > 0) {
        } else if (prefill_offset >= 0) {
            if (int rc = f16 ? vmx_prefill_attention_f16(attn, q, L.k_cache.p, L.v_cache.p, block_tables, B, prefill_offset, scale, Hq, Hkv,
                                                         hd, c.block_size, 0, st)
                             : vmx_prefill_attention_bf16(attn, q, L.k_cache.p, L.v_cache.p, block_tables, B, prefill_offset, scale, Hq, Hkv,
                                                          hd, c.block_size, 0, c.kv_cache_dtype, ks, ks + 1, st))
                return rc;
            if (iq)
                if (int rc = iq_quantize(m, attn, B, NQ, st)) return rc;
        } else {
            VmxAttnParts ap;
            if (int rc = vmx_paged_attention_parts(q, L.k_cache.p, L.v_cache.p, block_tables, seq_lens, scale, B, Hq, Hkv,
                                                   max_blocks_per_seq, max_seq_len, hd, c.block_size, 0, kvd, ks, ks + 1,
                                                   m->attn_ws.p, &ap, st))
                return rc;
            if (int rc = vmx_paged_attention_merge(attn, &ap, seq_lens, B, Hq, hd, 0, st, iq ? q8_aq(m) : nullptr,
                                                   iq ? q8_ad(m, B, NQ) : nullptr, f16 ? 1 : 0))
                return rc;
        }
        if (fold) {   // h += o_proj(attn);  act = glu(gate|up(rms_norm(h)));  h += down(act)
            if (split_attn) {
                if (int rc = vmx_int4_gemm_resid_attn(L.o.packed.p, &sparts, hd, h, ssq, B, H, NQ, L.o.group, st)) return rc;
            } else if (int rc = vmx_int4_gemm_resid(L.o.packed.p, attn, h, ssq, B, H, NQ, L.o.group, st)) return rc;
            if (int rc = vmx_int4_gemm_glu_norm(L.gate_up.packed.p, h, (const uint16_t *)L.post_ln.p, c.rms_norm_eps, ssq, act, B, 2 * I, H,
                                                L.gate_up.group, st))
                return rc;
            if (int rc = vmx_int4_gemm_resid(L.down.packed.p, act, h, ssq, B, H, I, L.down.group, st)) return rc;
            pending = nullptr;
            have_ssq = true;
            continue;
        }
        if (mfold) {   // h += o_proj(attn);  x = rms_norm(h);  h += experts(x)
            if (split_attn) {
                if (int rc = vmx_int4_gemm_resid_attn(L.o.packed.p, &sparts, hd, h, ssq, B, H, NQ, L.o.group, st)) return rc;
            } else if (int rc = vmx_int4_gemm_resid(L.o.packed.p, attn, h, ssq, B, H, NQ, L.o.group, st)) return rc;
            static const bool router_fused = !(vmx_tune("VMX_NO_ROUTER_FUSION", 0) != 0);
            if (router_fused && c.num_experts <= 8 && c.experts_per_tok <= c.num_experts && H <= 4096) {
                // post_attention_layernorm, the dense router and the top-k softmax in one launch
                if (H <= 2048)
                    hipLaunchKernelGGL((norm_router_topk_kernel<1, 8>), dim3(B), dim3(256), 0, st, h, x, (const uint16_t *)L.post_ln.p, c.rms_norm_eps,
                                       H, (const uint16_t *)L.router.p, c.num_experts, c.experts_per_tok, (float *)m->topk_w.p, (int32_t *)m->topk_ids.p);
                else
                    hipLaunchKernelGGL((norm_router_topk_kernel<2, 8>), dim3(B), dim3(256), 0, st, h, x, (const uint16_t *)L.post_ln.p, c.rms_norm_eps,
                                       H, (const uint16_t *)L.router.p, c.num_experts, c.experts_per_tok, (float *)m->topk_w.p, (int32_t *)m->topk_ids.p);
                VMX_CHECK_LAUNCH("norm + router + top-k");
            } else {
                launch_resid_norm(st, h, x, nullptr, 0, B, (const uint16_t *)L.post_ln.p, c.rms_norm_eps, H, nullptr, nullptr, nullptr);
                uint16_t *rl = (uint16_t *)m->router_logits.p;
                if (int rc = vmx_dense_gemv_launch(rl, x, (const uint16_t *)L.router.p, nullptr, B, c.num_experts, H, st)) return rc;
                if (int rc = vmx_topk_softmax(rl, VMX_BF16, (float *)m->topk_w.p, (int32_t *)m->topk_ids.p, B, c.num_experts, c.experts_per_tok, 1, 1,
                                              st))
                    return rc;
            }
            if (int rc = vmx_moe_int4_forward_glu((uint16_t *)m->moe_out.p, x, (const float *)m->topk_w.p, (const int32_t *)m->topk_ids.p,
                                                  (const void *const *)L.w13_table.p, (const void *const *)L.w2_table.p, B, c.experts_per_tok,
                                                  c.num_experts, H, I, c.group_size, m->moe_ws.p, m->moe_ws.bytes,
                                                  vmx_int4_glu_fusable(2 * I, H) ? 1 : 0, st, h, ssq))
                return rc;
            pending = nullptr;
            have_ssq = true;
            continue;
        }
        const float *o_src = PB;
        int o_splits = s_o;
        const uint16_t *o_svh = nullptr;
        if (ex_fold) {
            if (int rc = exl3_forward_group(m, L, {VMX_L_O}, L.ex[VMX_L_O].svh, attn, nullptr, H, B, st, &o_src, &o_splits, nullptr, nullptr, 0.f,
                                            ex_split ? &sparts : nullptr))
                return rc;
            o_svh = (const uint16_t *)L.ex[VMX_L_O].svh.p;
        } else if (ex) {
            if (int rc = exl3_forward(m, L.ex[VMX_L_O], attn, PB, H, B, st)) return rc;
        } else if (dn) {
            if (int rc = dense_forward(m, L.dn[1].p, attn, PB, H, H, NQ, B, st)) return rc;
        } else if (iq) {
            if (int rc = iq_forward(m, L.iq[VMX_L_O], PB, H, B, st)) return rc;
        } else if (int rc = run_linear(m, L.o, attn, PB, B, st)) return rc;
        // h += o; x = rms_norm(h);  gate|up = linear(x);  act = silu(gate) * up;  down = linear(act)
        if (tp) {  // row-parallel: local bf16 output, summed over the ranks
            if (int rc = tp_reduce(m, PB, o_splits, B, H, st)) return rc;
            o_src = (const float *)m->red_bf16.p;
            o_splits = -1;
        }
        const bool ex_norm_gu = ex_fold && o_svh && exl3_resid_stats_ok(B, o_splits, H);
        if (ex_norm_gu) launch_exl3_resid_stats(st, h, o_src, o_splits, B, H, o_svh, ssq, f16);
        else
        launch_resid_norm(st, h, x, o_src, o_splits, B, (const uint16_t *)L.post_ln.p, c.rms_norm_eps, H, iq ? q8_aq(m) : nullptr,
                          iq ? q8_ad(m, B, H) : nullptr, o_svh, nullptr, nullptr, 0, f16);
        if (moe) {
            // QuantizedMoELayer::forward (models/mixtral_quantized.rs:138-195): dense router, softmax (bf16) -> top-k -> renormalise,
            // out = sum_j bf16(expert_j(x) * w_j), experts = down(silu(gate x) * up x); no host read of the routing
            uint16_t *rl = (uint16_t *)m->router_logits.p;
            const int rstep = (c.num_experts <= 16 && B >= 2) ? B : 16;   // up to 16 experts: one launch, a workgroup per row
            for (int r0 = 0; r0 < B; r0 += rstep) {
                const int mr = B - r0 < rstep ? B - r0 : rstep;
                if (int rc = vmx_dense_gemv_launch(rl + (size_t)r0 * c.num_experts, x + (size_t)r0 * H, (const uint16_t *)L.router.p, nullptr, mr,
                                                   c.num_experts, H, st))
                    return rc;
            }
            if (int rc = vmx_topk_softmax(rl, VMX_BF16, (float *)m->topk_w.p, (int32_t *)m->topk_ids.p, B, c.num_experts, c.experts_per_tok, 1, 1, st))
                return rc;
            if (int rc = vmx_moe_int4_forward_glu((uint16_t *)m->moe_out.p, x, (const float *)m->topk_w.p, (const int32_t *)m->topk_ids.p,
                                                  (const void *const *)L.w13_table.p, (const void *const *)L.w2_table.p, B, c.experts_per_tok,
                                                  c.num_experts, H, I, c.group_size, m->moe_ws.p, m->moe_ws.bytes,
                                                  vmx_int4_glu_fusable(2 * I, H) ? 1 : 0, st))
                return rc;
            if (tp)   // every rank holds the weighted sum over its own experts: the sum over ranks is the layer's output
                if (int rc = vmx_comm_all_reduce_sum_bf16(m->comm, (uint16_t *)m->moe_out.p, (size_t)B * H, st)) return rc;
            pending = (const float *)m->moe_out.p;
            pending_splits = -1;
            continue;
        }
        const float *gu_src = PA;
        int gu_splits = s_gu;
        const uint16_t *gu_svh = nullptr;
        if (ex_fold) {
            if (int rc = exl3_forward_group(m, L, {VMX_L_GATE, VMX_L_UP}, L.ex_svh_gu, ex_norm_gu ? h : x, nullptr, 2 * I, B, st, &gu_src, &gu_splits,
                                            ex_norm_gu ? (const uint16_t *)L.post_ln.p : nullptr, ssq, c.rms_norm_eps))
                return rc;
            gu_svh = (const uint16_t *)L.ex_svh_gu.p;
        } else if (ex && B <= 4 && L.ex_svh_gu.p) {
            if (int rc = exl3_forward_group(m, L, {VMX_L_GATE, VMX_L_UP}, L.ex_svh_gu, x, PA, 2 * I, B, st)) return rc;
        } else if (ex) {
            if (int rc = exl3_forward(m, L.ex[VMX_L_GATE], x, PA, 2 * I, B, st)) return rc;
            if (int rc = exl3_forward(m, L.ex[VMX_L_UP], x, PA + I, 2 * I, B, st)) return rc;
        } else if (dn) {
            if (int rc = dense_forward(m, L.dn[2].p, x, PA, 2 * I, 2 * I, H, B, st)) return rc;
        } else if (iq) {
            if (int rc = iq_forward(m, L.iq[VMX_L_GATE], PA, 2 * I, B, st)) return rc;
            if (int rc = iq_forward(m, L.iq[VMX_L_UP], PA + I, 2 * I, B, st)) return rc;
        } else if (L.gate_up.glu) {  // act comes out of the GEMM epilogue
            if (int rc = L.gate_up.forward_glu(x, act, B, st)) return rc;
        } else if (int rc = run_linear(m, L.gate_up, x, PA, B, st)) return rc;
        static const bool no_ex_act = vmx_tune("VMX_EXL3_NO_WAVE_ACT", 0) != 0;
        if (ex_fold && gu_svh && !no_ex_act) {
            if (f16) hipLaunchKernelGGL(exl3_gateup_act_kernel<true>, dim3((I / 128 + 3) / 4, B), dim3(256), 0, st, gu_src, gu_splits, B, act, I, gu_svh);
            else hipLaunchKernelGGL(exl3_gateup_act_kernel<false>, dim3((I / 128 + 3) / 4, B), dim3(256), 0, st, gu_src, gu_splits, B, act, I, gu_svh);
        } else if (one || !L.gate_up.glu) {
            int64_t blocks = ((int64_t)B * I / 8 + 255) / 256;
            if (f16)
                hipLaunchKernelGGL(gateup_act_kernel<true>, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, st, gu_src, gu_splits, B, act,
                                   I, (int8_t *)nullptr, (float *)nullptr, gu_svh);
            else
            hipLaunchKernelGGL(gateup_act_kernel<false>, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, st, gu_src, gu_splits, B, act, I,
                               iq ? q8_aq(m) : nullptr, iq ? q8_ad(m, B, I) : nullptr, gu_svh);
        }
        if (ex_fold) {
            if (int rc = exl3_forward_group(m, L, {VMX_L_DOWN}, L.ex[VMX_L_DOWN].svh, act, nullptr, H, B, st, &pending, &pending_splits)) return rc;
            pending_svh = (const uint16_t *)L.ex[VMX_L_DOWN].svh.p;
            continue;
        } else if (ex) {
            if (int rc = exl3_forward(m, L.ex[VMX_L_DOWN], act, PB, H, B, st)) return rc;
        } else if (dn) {
            if (int rc = dense_forward(m, L.dn[3].p, act, PB, H, H, I, B, st)) return rc;
        } else if (iq) {
            if (int rc = iq_forward(m, L.iq[VMX_L_DOWN], PB, H, B, st)) return rc;
        } else if (int rc = run_linear(m, L.down, act, PB, B, st)) return rc;
        pending = PB;
        pending_splits = s_down;
        if (tp) {
            if (int rc = tp_reduce(m, PB, pending_splits, B, H, st)) return rc;
            // the all-reduced bf16 vector is parked in the (now free) PA buffer so that red_bf16 can be reused
            VMX_CHECK_HIP(hipMemcpyAsync(PA, m->red_bf16.p, (size_t)B * H * 2, hipMemcpyDeviceToDevice, st));
            pending = PA;
            pending_splits = -1;
        }
    }
    launch_resid_norm(st, h, x, pending, pending_splits, B, (const uint16_t *)m->final_norm.p, c.rms_norm_eps, H, nullptr, nullptr, pending_svh,
                      nullptr, nullptr, 0, f16);
    VMX_CHECK_LAUNCH("qwen3 layers");
    return VMX_OK;
}

// lm_head (+ greedy ids) on `rows` rows of normed activations starting at x
int run_head(vmx_qwen3 *m, const uint16_t *x, int rows, int32_t *out_ids, uint16_t *logits_out, void *stream) {
    const vmx_qwen3_config &c = m->cfg;
    hipStream_t st = vmx_stream(stream);
    const int H = c.hidden_size, B = rows;
    const bool tp = c.tp_size > 1 || m->comm != nullptr;
    uint16_t *logits = logits_out ? logits_out : (uint16_t *)m->logits.p;
    const int VL = m->vocab_local();
    const int64_t v0 = tp ? (int64_t)c.tp_rank * VL : 0;  // first vocabulary row of this rank's lm_head slice
    const uint16_t *head = m->lm_head_tied ? (const uint16_t *)m->embed.p + v0 * H : (const uint16_t *)m->lm_head.p;
    const int f16 = m->act_f16 ? 1 : 0;   // fp16 table, fp16 logits
    if (int rc = vmx_dense_gemv_launch(logits, x, head, nullptr, B, VL, H, st, f16)) return rc;
    if (out_ids) {
        if (!tp) {
            if (int rc = f16 ? vmx_argmax_f16(out_ids, logits, B, VL, m->argmax_ws.p, m->argmax_ws.bytes, stream)
                             : vmx_argmax_bf16(out_ids, logits, B, VL, m->argmax_ws.p, m->argmax_ws.bytes, stream))
                return rc;
        } else {  // vocab-parallel: best (value, global index) key per rank, max over ranks
            unsigned long long *keys = (unsigned long long *)m->keys.p;
            if (int rc = vmx_argmax_keys(keys, logits, B, VL, (uint32_t)v0, m->argmax_ws.p, st, f16)) return rc;
            if (int rc = vmx_comm_all_reduce_max_u64(m->comm, keys, B, stream)) return rc;
            if (int rc = vmx_argmax_keys_to_ids(keys, out_ids, B, st)) return rc;
        }
    }
    return VMX_OK;
}
}  // namespace

// One decode step for `batch` sequences (one new token each).  Pure launch sequence.
extern "C" int vmx_qwen3_decode(vmx_qwen3 *m, const int32_t *input_ids, const int32_t *positions, const int32_t *slot_mapping,
                                const int32_t *block_tables, const int32_t *seq_lens, int batch, int max_blocks_per_seq,
                                int max_seq_len, int32_t *out_ids, uint16_t *logits_out, void *stream) {
    VMX_REQUIRE(m && input_ids && positions && slot_mapping && block_tables && seq_lens, "qwen3_decode: null pointer");
    const vmx_qwen3_config &c = m->cfg;
    VMX_REQUIRE(batch >= 1 && batch <= c.max_batch, "qwen3_decode: batch %d exceeds max_batch %d", batch, c.max_batch);
    VMX_REQUIRE(max_seq_len >= 1 && max_seq_len <= c.max_seq_len, "qwen3_decode: max_seq_len %d exceeds the model's %d", max_seq_len,
                c.max_seq_len);
    for (const Layer &L : m->layers) VMX_REQUIRE(layer_ready(m, L), "qwen3_decode: weights not fully loaded");
    const bool tp = c.tp_size > 1 || m->comm != nullptr;  // a 1-rank communicator still exercises the collective path
    VMX_REQUIRE(!tp || m->comm, "qwen3_decode: tp_size %d but no communicator attached (vmx_qwen3_set_comm)", c.tp_size);
    if (int rc = run_layers(m, input_ids, positions, slot_mapping, block_tables, seq_lens, batch, max_blocks_per_seq, max_seq_len, -1,
                            vmx_stream(stream)))
        return rc;
    return run_head(m, (const uint16_t *)m->x.p, batch, out_ids, logits_out, stream);
}

// ModelForward::forward for one sequence (crates/core/src/engine/model_forward.rs:22-64): q_len prompt tokens at positions
// seqlen_offset .. seqlen_offset + q_len - 1.  The linears run on all rows (16 per MFMA pass), K/V rows are written to the
// paged cache at slot_mapping, attention is the causal prefill kernel over the cache; only the LAST row goes through the
// lm_head (greedy id and, if asked for, its logits) - the only row the engine samples from.  Pure launch sequence.
extern "C" int vmx_qwen3_prefill(vmx_qwen3 *m, const int32_t *input_ids, int seqlen_offset, const int32_t *slot_mapping,
                                 const int32_t *block_table, int q_len, int32_t *out_id_last, uint16_t *logits_last, void *stream) {
    VMX_REQUIRE(m && input_ids && slot_mapping && block_table, "qwen3_prefill: null pointer");
    const vmx_qwen3_config &c = m->cfg;
    VMX_REQUIRE(q_len >= 1 && q_len <= c.max_prefill_tokens, "qwen3_prefill: q_len %d exceeds max_prefill_tokens %d", q_len, c.max_prefill_tokens);
    VMX_REQUIRE(seqlen_offset >= 0 && seqlen_offset + q_len <= c.max_position, "qwen3_prefill: positions %d..%d exceed max_position %d",
                seqlen_offset, seqlen_offset + q_len - 1, c.max_position);
    for (const Layer &L : m->layers) VMX_REQUIRE(layer_ready(m, L), "qwen3_prefill: weights not fully loaded");
    const bool tp = c.tp_size > 1 || m->comm != nullptr;
    VMX_REQUIRE(!tp || m->comm, "qwen3_prefill: tp_size %d but no communicator attached (vmx_qwen3_set_comm)", c.tp_size);
    hipStream_t st = vmx_stream(stream);
    int32_t *pos = (int32_t *)m->positions.p;
    hipLaunchKernelGGL(iota_positions_kernel, dim3((q_len + 255) / 256), dim3(256), 0, st, pos, seqlen_offset, q_len);
    if (int rc = run_layers(m, input_ids, pos, slot_mapping, block_table, nullptr, q_len, 0, 0, seqlen_offset, st)) return rc;
    return run_head(m, (const uint16_t *)m->x.p + (size_t)(q_len - 1) * c.hidden_size, 1, out_id_last, logits_last, stream);
}

// Rows of one decode call normally belong to different sequences with disjoint cache blocks.  A caller that feeds
// consecutive tokens of ONE sequence as rows of a decode call (each with its own seq_len over a shared block table) must
// say so: every row's K/V must then be in the cache before any row's attention starts, which the one-launch short-context
// attention does not provide across rows.
extern "C" int vmx_qwen3_set_rows_share_cache(vmx_qwen3 *m, int on) {
    VMX_REQUIRE(m, "qwen3_set_rows_share_cache: null model");
    m->rows_share_cache = on != 0;
    return VMX_OK;
}

// The 16-bit type of activations, norm weights, the embedding / lm_head tables, logits and an un-quantized KV cache: VMX_BF16
// (default) or VMX_F16.  The reference runs EXL3 checkpoints in fp16 whatever the configured dtype
// (crates/core/src/quantization/weight_loader.rs:453-461; Exl3Linear::forward casts to fp16 and back otherwise,
// exl3.rs:222-277): with VMX_F16 every tensor the bf16 form rounds to bf16 is rounded to fp16 instead and 16-bit tensors are
// loaded as fp16 bits.  Call before loading tensors; fp16 needs every projection to be an EXL3 linear and a 16-bit KV cache.
extern "C" int vmx_qwen3_set_activation_dtype(vmx_qwen3 *m, int dtype) {
    VMX_REQUIRE(m, "qwen3_set_activation_dtype: null model");
    VMX_REQUIRE(dtype == VMX_BF16 || dtype == VMX_F16, "qwen3_set_activation_dtype: dtype %d (VMX_BF16 or VMX_F16)", dtype);
    VMX_REQUIRE(dtype == VMX_BF16 || (m->cfg.kv_cache_dtype == VMX_KV_AUTO && m->cfg.num_experts == 0),
                "qwen3_set_activation_dtype: fp16 activations take a 16-bit KV cache and a dense MLP");
    m->act_f16 = dtype == VMX_F16;
    return VMX_OK;
}

// Attach the tensor-parallel communicator (not owned).  Required before decoding when tp_size > 1.
extern "C" int vmx_qwen3_set_comm(vmx_qwen3 *m, vmx_comm *comm) {
    VMX_REQUIRE(m, "qwen3_set_comm: null model");
    VMX_REQUIRE(!comm || (vmx_comm_world(comm) == (m->cfg.tp_size > 1 ? m->cfg.tp_size : 1) && vmx_comm_rank(comm) == m->cfg.tp_rank),
                "qwen3_set_comm: communicator rank/world do not match the model's tp_rank/tp_size");
    m->comm = comm;
    return VMX_OK;
}

// Parity aid: copies of internal activations after the last vmx_qwen3_decode call.
extern "C" int vmx_qwen3_debug_read(vmx_qwen3 *m, int which, void *dst, size_t bytes, void *stream) {
    VMX_REQUIRE(m && dst, "qwen3_debug_read: null pointer");
    DevBuf *src = which == 0 ? &m->h : which == 1 ? &m->x : which == 2 ? &m->logits : nullptr;
    VMX_REQUIRE(src && bytes <= src->bytes, "qwen3_debug_read: bad request");
    VMX_CHECK_HIP(hipMemcpyAsync(dst, src->p, bytes, hipMemcpyDefault, vmx_stream(stream)));
    return VMX_OK;
}

// Parity aid: copy a layer's K (which = 0) or V (1) cache to `buf` (write = 0) or from it (write = 1); device or host pointer,
// bytes = the whole cache [blocks, block_size, kv_heads, head_dim] in the cache dtype.
extern "C" int vmx_qwen3_debug_kv(vmx_qwen3 *m, int layer, int which, int write, void *buf, size_t bytes, void *stream) {
    VMX_REQUIRE(m && buf, "qwen3_debug_kv: null pointer");
    VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size() && (which == 0 || which == 1), "qwen3_debug_kv: layer %d / which %d", layer, which);
    DevBuf &c = which == 0 ? m->layers[layer].k_cache : m->layers[layer].v_cache;
    VMX_REQUIRE(bytes == c.bytes, "qwen3_debug_kv: the cache holds %zu bytes, got %zu", c.bytes, bytes);
    VMX_CHECK_HIP(write ? hipMemcpyAsync(c.p, buf, bytes, hipMemcpyDefault, vmx_stream(stream))
                        : hipMemcpyAsync(buf, c.p, bytes, hipMemcpyDefault, vmx_stream(stream)));
    return VMX_OK;
}

// ---- measurement aid: the int4 linear launches of one decode step and nothing else (4 per layer, in
// step order, on the step's own activation buffers and weights).  Replayed back to back between two
// HIP events this gives the kernel's average launch duration on its own stream.
extern "C" int vmx_qwen3_linears_only(vmx_qwen3 *m, int batch, void *stream) {
    VMX_REQUIRE(m && batch >= 1 && batch <= m->cfg.max_batch, "qwen3_linears_only: bad arguments");
    for (const Layer &L : m->layers)
        VMX_REQUIRE(L.int4_ready() || L.exl3_ready(), "qwen3_linears_only: the model has neither int4 nor EXL3 linears loaded");
    hipStream_t st = vmx_stream(stream);
    const uint16_t *x = (const uint16_t *)m->x.p, *attn = (const uint16_t *)m->attn.p, *act = (const uint16_t *)m->act.p;
    float *PA = (float *)m->partial.p, *PB = (float *)m->partial_b.p;
    const vmx_qwen3_config &c = m->cfg;
    const int H = c.hidden_size, I = c.intermediate_size, NQ = c.num_heads * c.head_dim, NQKV_ = m->qkv_n();
    uint16_t *h = (uint16_t *)m->h.p;
    for (const Layer &L : m->layers) {
        if (!L.int4_ready()) {   // EXL3 layer: the launches a decode step of up to 4 rows makes for its seven projections
            VMX_REQUIRE(batch <= 4 && !m->comm && H % 128 == 0 && I % 128 == 0 && L.ex_svh_gu.p && L.ex_svh_qkv.p,
                        "qwen3_linears_only: EXL3 layers are covered for 1..4 rows on one GPU");
            const int mk = m->lin_mask;
            const float *src = nullptr;
            int sp = 0;
            if (mk & 1)
                if (int rc = exl3_forward_group(m, L, {VMX_L_Q, VMX_L_K, VMX_L_V}, L.ex_svh_qkv, x, PA, NQKV_, batch, st)) return rc;
            if (mk & 2)
                if (int rc = exl3_forward_group(m, L, {VMX_L_O}, L.ex[VMX_L_O].svh, attn, nullptr, H, batch, st, &src, &sp)) return rc;
            if (mk & 4)
                if (int rc = exl3_forward_group(m, L, {VMX_L_GATE, VMX_L_UP}, L.ex_svh_gu, x, nullptr, 2 * I, batch, st, &src, &sp)) return rc;
            if (mk & 8)
                if (int rc = exl3_forward_group(m, L, {VMX_L_DOWN}, L.ex[VMX_L_DOWN].svh, act, nullptr, H, batch, st, &src, &sp)) return rc;
            continue;
        }
        const bool fold = folded_layers_enabled() && !m->comm && c.tp_size <= 1 && c.num_experts == 0 && L.gate_up.glu &&
                          vmx_int4_norm_splits(batch, NQKV_, H, 0) > 0 && vmx_int4_norm_splits(batch, 2 * I, H, 1) == 1 &&
                          vmx_int4_resid_fusable(batch, H, NQ) && vmx_int4_resid_fusable(batch, H, I);
        const int mk = m->lin_mask;
        if (fold) {   // what the step launches for such a layer
            float *ssq = (float *)m->ssq.p;
            if (mk & 1)
                if (int rc = vmx_int4_gemm_partial_norm(L.qkv.packed.p, h, (const uint16_t *)L.input_ln.p, c.rms_norm_eps, ssq, PA, batch, NQKV_,
                                                        H, L.qkv.group, st))
                    return rc;
            if (mk & 2)
                if (int rc = vmx_int4_gemm_resid(L.o.packed.p, attn, h, ssq, batch, H, NQ, L.o.group, st)) return rc;
            if (mk & 4)
                if (int rc = vmx_int4_gemm_glu_norm(L.gate_up.packed.p, h, (const uint16_t *)L.post_ln.p, c.rms_norm_eps, ssq,
                                                    (uint16_t *)m->act.p, batch, 2 * I, H, L.gate_up.group, st))
                    return rc;
            if (mk & 8)
                if (int rc = vmx_int4_gemm_resid(L.down.packed.p, act, h, ssq, batch, H, I, L.down.group, st)) return rc;
            continue;
        }
        if (mk & 1)
            if (int rc = L.qkv.forward_partial(x, PA, batch, st)) return rc;
        if (mk & 2)
            if (int rc = L.o.forward_partial(attn, PB, batch, st)) return rc;
        if (mk & 4)
            if (int rc = L.gate_up.glu ? L.gate_up.forward_glu(x, (uint16_t *)m->act.p, batch, st) : L.gate_up.forward_partial(x, PA, batch, st)) return rc;
        if (mk & 8)
            if (int rc = L.down.forward_partial(act, PB, batch, st)) return rc;
    }
    return VMX_OK;
}

// Which of a layer's four linears the two measurement calls cover: bit 0 q|k|v, 1 o_proj, 2 gate|up, 3 down_proj.
extern "C" int vmx_qwen3_linears_select(vmx_qwen3 *m, int mask) {
    VMX_REQUIRE(m && mask >= 1 && mask <= 15, "qwen3_linears_select: mask must be 1..15");
    m->lin_mask = mask;
    return VMX_OK;
}

// Launch count and summed algorithmic bytes (packed nibbles + bf16 scales + 4-bit zeros + bf16
// activations in and out) of vmx_qwen3_linears_only.
extern "C" int vmx_qwen3_linears_info(vmx_qwen3 *m, int batch, int *launches, double *alg_bytes) {
    VMX_REQUIRE(m && launches && alg_bytes, "qwen3_linears_info: null pointer");
    double by = 0.0;
    int n = 0;
    for (const Layer &L : m->layers) {
        int bit = 1;
        if (!L.int4_ready() && L.exl3_ready()) {   // trellis words + fp16 sign/scale vectors + activations in (fp16) and out (f32 sums)
            const int kind[7] = {1, 1, 1, 2, 4, 4, 8};
            for (int s = 0; s < 7; s++) {
                if (!(m->lin_mask & kind[s])) continue;
                const Exl3Lin &e = L.ex[s];
                by += (double)e.in_features * e.out_features * e.bits / 8.0 + 2.0 * e.in_features + 2.0 * e.out_features +
                      (double)batch * (2.0 * e.in_features + 2.0 * e.out_features);
            }
            for (int b = 1; b <= 8; b <<= 1) n += (m->lin_mask & b) ? 1 : 0;
            continue;
        }
        for (const Int4Linear *l : {&L.qkv, &L.o, &L.gate_up, &L.down}) {
            if (m->lin_mask & bit) {
                by += int4_alg_bytes(*l, batch);
                n++;
            }
            bit <<= 1;
        }
    }
    *launches = n;
    *alg_bytes = by;
    return VMX_OK;
}

namespace {
// ids <- next token; position, seq_len += 1; slot from the block table (host-side engine bookkeeping of
// the reference, done on the device so that consecutive steps can be replayed from one graph).  A row whose next position
// would fall outside its block-table row (max_blocks_per_seq * block_size tokens) is frozen: slot = -1, position and seq_len
// keep their values, so replaying a graph past the capacity cannot overwrite live cache rows.  The caller keeps that
// capacity within the model's max_position / max_seq_len.
__global__ void decode_advance_kernel(int32_t *ids, const int32_t *next_ids, int32_t *positions, int32_t *slots, int32_t *seq_lens,
                                      const int32_t *block_tables, int batch, int max_blocks_per_seq, int block_size) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int p = positions[b] + 1;
    ids[b] = next_ids[b];
    if (p >= max_blocks_per_seq * block_size) {   // the row's block table is full: it stops advancing and writes no more K/V rows
        slots[b] = -1;
        return;
    }
    positions[b] = p;
    seq_lens[b] = p + 1;
    slots[b] = block_tables[b * max_blocks_per_seq + p / block_size] * block_size + p % block_size;
}
}  // namespace

extern "C" int vmx_decode_advance(int32_t *input_ids, const int32_t *next_ids, int32_t *positions, int32_t *slot_mapping,
                                  int32_t *seq_lens, const int32_t *block_tables, int batch, int max_blocks_per_seq, int block_size,
                                  void *stream) {
    VMX_REQUIRE(input_ids && next_ids && positions && slot_mapping && seq_lens && block_tables, "decode_advance: null pointer");
    VMX_REQUIRE(batch >= 1 && max_blocks_per_seq >= 1 && block_size >= 1, "decode_advance: bad shape");
    hipLaunchKernelGGL(decode_advance_kernel, dim3((batch + 63) / 64), dim3(64), 0, vmx_stream(stream), input_ids, next_ids, positions,
                       slot_mapping, seq_lens, block_tables, batch, max_blocks_per_seq, block_size);
    VMX_CHECK_LAUNCH("decode_advance");
    return VMX_OK;
}

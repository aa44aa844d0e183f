// Whole-token decode executor for the quantized Gemma 4 decoder (GGUF I-quant linears, FP8 or bf16 paged KV cache
// with per-layer geometry), mirroring the paged-decode path of the reference's QuantizedGemma4ForCausalLM
// (crates/core/src/models/gemma4_quantized.rs: attention :1311-1424, MLP :1468-1473, layer :1743-1774,
// model :2217-2279) behind a C ABI.
//
// Operator order and rounding points are the reference's: IqLinear returns f32 cast back to bf16; every norm output,
// residual sum, layer-scalar product, rotated q/k, activation and product is rounded to bf16 where the reference's
// separate tensor ops round it.  The launch structure is this library's own: one fused vector kernel folds
// "bf16(linear out) -> post-norm -> + residual -> * layer_scalar -> next pre-norm -> q8_1" (the sandwich norms of
// two consecutive sub-blocks), one kernel does per-head q/k/v norms + RoPE + KV write, one does GeGLU + q8_1.
// A layer is 10 launches (4 MMVQ: q|k|v and gate|up each go out as one multi-tensor launch); the step is a fixed launch sequence (hipGraph-capturable).
#include "vmx_kernels.h"

#include <math.h>
#include <new>
#include <vector>

namespace {

__device__ __forceinline__ float rbf(float v) { return vmx_bf16_to_f32(vmx_f32_to_bf16(v)); }

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

// h[t, :] = bf16(table[ids[t], :] * scale)   (embed_tokens.forward(ids) * sqrt(hidden), gemma4_quantized.rs:2217-2218)
__global__ __launch_bounds__(256) void g4_embed_kernel(uint16_t *__restrict__ out, const uint16_t *__restrict__ table,
                                                        const int32_t *__restrict__ ids, int hidden, int vocab, float scale) {
    const int t = blockIdx.x;
    const int id = ids[t];
    const bool ok = id >= 0 && id < vocab;
    const uint4 *src = reinterpret_cast<const uint4 *>(table + (int64_t)(ok ? id : 0) * hidden);
    uint4 *dst = reinterpret_cast<uint4 *>(out + (int64_t)t * hidden);
    for (int i = threadIdx.x; i < (hidden >> 3); i += 256) {
        float f[8];
        unpack8(ok ? src[i] : make_uint4(0, 0, 0, 0), f);
#pragma unroll
        for (int j = 0; j < 8; j++) f[j] *= scale;
        dst[i] = pack8(f);
    }
}

// One workgroup per token; thread t owns the 8-element chunks t, t+256, ... (CH of them) in registers.
//   o given:  o_r = bf16(o);  n = bf16(o_r * rsqrt(mean(o_r^2) + eps) * w_post);  h = bf16(n + h_in);
//             scalar given: h = bf16(h * scalar)                       (post-norm, residual add, layer_scalar)
//   always:   x = bf16(h * rsqrt(mean(h^2) + eps) * w_next)  (+ its q8_1 form)   (the next sub-block's pre-norm);
//             w_next == nullptr: x = h (the per-layer-input gate projects the residual stream itself)
template <int CH>
__global__ __launch_bounds__(256) void g4_norm_kernel(uint16_t *__restrict__ h, const uint16_t *__restrict__ h_in, uint16_t *__restrict__ x, const float *__restrict__ o,
                                                       const uint16_t *__restrict__ w_post, const uint16_t *__restrict__ scalar,
                                                       const uint16_t *__restrict__ w_next, float eps, int hidden,
                                                       int8_t *__restrict__ aq, float *__restrict__ ad) {
    __shared__ float red[4];
    const int64_t row = (int64_t)blockIdx.x * hidden;
    const int nchunk = hidden >> 3;
    // every global load of the kernel is requested here, before the first reduction (the kernel is a chain of latencies)
    float4 oa[CH], ob[CH];
    uint4 hraw[CH], wp[CH], wn[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int ci = min(threadIdx.x + 256 * c, nchunk - 1);
        if (o) {
            oa[c] = *reinterpret_cast<const float4 *>(o + row + 8 * ci);
            ob[c] = *reinterpret_cast<const float4 *>(o + row + 8 * ci + 4);
            wp[c] = *reinterpret_cast<const uint4 *>(w_post + 8 * ci);
        }
        hraw[c] = *reinterpret_cast<const uint4 *>(h_in + row + 8 * ci);   // h_in == h, or the buffer the fused gate|up launch left the stream in
        wn[c] = w_next ? *reinterpret_cast<const uint4 *>(w_next + 8 * ci) : make_uint4(0, 0, 0, 0);
    }
    const float sc = scalar ? vmx_bf16_to_f32(scalar[0]) : 1.0f;
    float hv[CH][8];
    if (o) {
        float ov[CH][8];
        float v1 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if ((int)threadIdx.x + 256 * c < nchunk) {
                const float t8[8] = {oa[c].x, oa[c].y, oa[c].z, oa[c].w, ob[c].x, ob[c].y, ob[c].z, ob[c].w};
#pragma unroll
                for (int j = 0; j < 8; j++) { ov[c][j] = rbf(t8[j]); v1 += ov[c][j] * ov[c][j]; }
            }
        }
        v1 = vmx_block_sum_256(v1, red);
        const float inv1 = rsqrtf(v1 / (float)hidden + eps);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int ci = threadIdx.x + 256 * c;
            if (ci < nchunk) {
                float wv[8], hr[8];
                unpack8(wp[c], wv);
                unpack8(hraw[c], hr);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float v = rbf(rbf(ov[c][j] * inv1 * wv[j]) + hr[j]);
                    if (scalar) v = rbf(v * sc);
                    hv[c][j] = v;
                }
                *reinterpret_cast<uint4 *>(h + row + 8 * ci) = pack8(hv[c]);
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CH; c++) unpack8(hraw[c], hv[c]);
    }
    float v2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        if ((int)threadIdx.x + 256 * c < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; j++) v2 += hv[c][j] * hv[c][j];
        }
    }
    v2 = vmx_block_sum_256(v2, red);
    const float inv2 = rsqrtf(v2 / (float)hidden + eps);
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int ci = threadIdx.x + 256 * c;
        if (ci < nchunk) {
            float wv[8], out[8];
            unpack8(wn[c], wv);
#pragma unroll
            for (int j = 0; j < 8; j++) out[j] = w_next ? hv[c][j] * inv2 * wv[j] : hv[c][j];
            const uint4 packed = pack8(out);
            *reinterpret_cast<uint4 *>(x + row + 8 * ci) = packed;
            if (aq) {
                float xr[8];
                unpack8(packed, xr);
                vmx_q8_1_emit8(xr, aq + row + 8 * ci, ad + (row >> 5) + (ci >> 2), (ci & 3) == 0);
            }
        }
    }
}

// grid (Hq + 2*Hkv, tokens), block 64.  proj: f32 [tokens][ntot] = q | k | (v) projections.  Per head: bf16 round,
// per-head RMSNorm (q, k weighted; v unweighted), NeoX RoPE on the first rot_dim dims of q and k, bf16 rounds after
// each; q -> q_out, k and v -> the layer's paged cache (bf16 or FP8 e4m3).  v_from_k: V is the K projection
// (before K's norm and rotation) put through the unweighted norm.  stage != nullptr: K and V go to that bf16 buffer instead of the cache.
__global__ __launch_bounds__(64) void g4_qkv_post_kernel(const float *__restrict__ proj, int ntot, uint16_t *__restrict__ q_out,
                                                          void *__restrict__ k_cache, void *__restrict__ v_cache,
                                                          const uint16_t *__restrict__ q_norm_w, const uint16_t *__restrict__ k_norm_w,
                                                          float eps, const float *__restrict__ cos_sin,
                                                          const int32_t *__restrict__ positions, const int32_t *__restrict__ slots, int Hq,
                                                          int Hkv, int hd, int rot_dim, int v_from_k, int fp8,
                                                          const float *__restrict__ k_scale, const float *__restrict__ v_scale,
                                                          uint16_t *__restrict__ stage) {
    __shared__ float vals[512];
    const int head = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
    const bool is_q = head < Hq, is_k = !is_q && head < Hq + Hkv;
    int src_head = head;
    if (!is_q && !is_k && v_from_k) src_head = head - Hkv;
    const int64_t base = (int64_t)t * ntot + (int64_t)src_head * hd;
    // The kernel is a chain of memory round trips, so everything with a known address is requested at once (hd <= 512: at
    // most 8 elements per lane); only the cos / sin row waits for the position.
    const uint16_t *nw = is_q ? q_norm_w : (is_k ? k_norm_w : nullptr);
    const int pos = positions[t];
    const int slot = (is_q || stage) ? 0 : slots[t];
    const float kv_sc = (!is_q && !stage && fp8) ? (is_k ? *k_scale : *v_scale) : 1.0f;
    float pv[8], nwv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int i = min(lane + 64 * j, hd - 1);
        pv[j] = proj[base + i];
        nwv[j] = nw ? vmx_bf16_to_f32(nw[i]) : 1.0f;
    }
    const int half = rot_dim >> 1;
    const bool rot = (is_q || is_k) && half > 0;
    const float *cs = cos_sin + (int64_t)pos * rot_dim;
    float cv[4], sv[4];   // pair i = lane + 64 j < half (half <= 256)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = max(min(lane + 64 * j, half - 1), 0);
        cv[j] = rot ? cs[i] : 1.0f;
        sv[j] = rot ? cs[half + i] : 0.0f;
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        pv[j] = rbf(pv[j]);
        if (lane + 64 * j < hd) ss += pv[j] * pv[j];
    }
    ss = vmx_wave_sum(ss);
    const float inv = rsqrtf(ss / (float)hd + eps);
#pragma unroll
    for (int j = 0; j < 8; j++)
        if (lane + 64 * j < hd) vals[lane + 64 * j] = rbf(pv[j] * inv * nwv[j]);
    __syncthreads();
    if (rot) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = lane + 64 * j;
            if (i < half) {
                const float xv = vals[i], yv = vals[half + i], c = cv[j], sn = sv[j];
                vals[i] = rbf(xv * c - yv * sn);
                vals[half + i] = rbf(yv * c + xv * sn);
            }
        }
        __syncthreads();
    }
    if (is_q) {
        for (int i = lane; i < hd; i += 64) q_out[((int64_t)t * Hq + head) * hd + i] = vmx_f32_to_bf16(vals[i]);
        return;
    }
    const int kvh = is_k ? head - Hq : head - Hq - Hkv;
    if (stage) {   // first write of an uncalibrated FP8 cache: bf16 K | V rows [tokens][Hkv * hd] each, scaled and scattered afterwards
        uint16_t *dst_row = stage + ((int64_t)(is_k ? 0 : gridDim.y) + t) * Hkv * hd + (int64_t)kvh * hd;
        for (int i = lane; i < hd; i += 64) dst_row[i] = vmx_f32_to_bf16(vals[i]);
        return;
    }
    if (slot < 0) return;
    const int64_t dst = ((int64_t)slot * Hkv + kvh) * hd;
    void *cache = is_k ? k_cache : v_cache;
    if (!fp8) {
        for (int i = lane; i < hd; i += 64) ((uint16_t *)cache)[dst + i] = vmx_f32_to_bf16(vals[i]);
    } else {   // clamp(x / scale, +-448) -> bf16 -> e4m3 (kv_cache/cache_engine.rs:202-223)
        const float sc = kv_sc;
        for (int i = lane; i < hd; i += 64) {
            float a = vals[i] / sc;
            a = rbf(fminf(fmaxf(a, -448.0f), 448.0f));
            ((uint8_t *)cache)[dst + i] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(a, a, 0, false) & 0xFFu);
        }
    }
}

// act[t, i] = bf16(bf16(gelu_erf(bf16(gate))) * bf16(up)) from f32 [tokens][2*inter] (gate columns first), 8 elements per
// thread, plus the q8_1 form of the result for the down projection.
__global__ __launch_bounds__(256) void g4_geglu_kernel(const float *__restrict__ gu, int tokens, uint16_t *__restrict__ act, int inter,
                                                        int8_t *__restrict__ aq, float *__restrict__ ad) {
    const int64_t chunks = (int64_t)tokens * inter / 8;
    const int cpr = inter / 8;
    for (int64_t c0 = (int64_t)blockIdx.x * 256; c0 < chunks; c0 += (int64_t)gridDim.x * 256) {
        const int64_t c = c0 + threadIdx.x;
        if (c < chunks) {   // chunks % 4 == 0: 4-lane q8_1 groups are all in or all out
            const int64_t t = c / cpr, col = (c % cpr) * 8;
            const float *gp = gu + t * 2 * inter + col, *up = gp + inter;
            const float4 g0 = *reinterpret_cast<const float4 *>(gp), g1 = *reinterpret_cast<const float4 *>(gp + 4);
            const float4 u0 = *reinterpret_cast<const float4 *>(up), u1 = *reinterpret_cast<const float4 *>(up + 4);
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, u[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float gg = rbf(g[j]);
                const float ge = rbf(0.5f * gg * (1.0f + erff(gg * 0.70710678118654752f)));
                r[j] = ge * rbf(u[j]);
            }
            const uint4 packed = pack8(r);
            *reinterpret_cast<uint4 *>(act + c * 8) = packed;
            float rr[8];
            unpack8(packed, rr);
            vmx_q8_1_emit8(rr, aq + c * 8, ad + (c >> 2), (c & 3) == 0);
        }
    }
}

// Per-layer inputs (compute_per_layer_inputs, gemma4_quantized.rs:2143-2192), grid (layers, tokens), block 64:
//   a = bf16(bf16(proj) * bf16(hidden^-1/2));  n = bf16(a * rsqrt(mean(a^2) + eps) * w);     (per (token, layer) slice of width P)
//   pli = bf16(bf16(n + e) * bf16(2^-1/2))     with e = bf16(embed_per_layer[id] * bf16(sqrt(P))) gathered beforehand
__global__ __launch_bounds__(64) void g4_ple_inputs_kernel(const float *__restrict__ proj, const uint16_t *__restrict__ emb,
                                                            const uint16_t *__restrict__ w, uint16_t *__restrict__ pli, int P, float eps,
                                                            float proj_scale, float input_scale) {
    __shared__ float vals[1024];
    const int64_t base = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * P;
    float ss = 0.f;
    for (int i = threadIdx.x; i < P; i += 64) {
        const float a = rbf(rbf(proj[base + i]) * proj_scale);
        vals[i] = a;
        ss += a * a;
    }
    ss = vmx_wave_sum(ss);
    const float inv = rsqrtf(ss / (float)P + eps);
    for (int i = threadIdx.x; i < P; i += 64) {
        const float n = rbf(vals[i] * inv * vmx_bf16_to_f32(w[i]));
        pli[base + i] = vmx_f32_to_bf16(rbf(n + vmx_bf16_to_f32(emb[base + i])) * input_scale);
    }
}

// apply_ple's gate (gemma4_quantized.rs:1674-1676): gated = bf16(bf16(gelu_tanh(bf16(gate))) * pli[token][layer]) and its q8_1
// form for per_layer_projection.  One thread per 8 values, P % 32 == 0; pli row stride = pli_stride elements.
__global__ __launch_bounds__(256) void g4_ple_gate_kernel(const float *__restrict__ gate, const uint16_t *__restrict__ pli, int64_t pli_stride,
                                                           int tokens, int P, uint16_t *__restrict__ gated, int8_t *__restrict__ aq,
                                                           float *__restrict__ ad) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x, chunks = (int64_t)tokens * P / 8;
    if (c >= chunks) return;
    const int cpr = P / 8;
    const int64_t t = c / cpr, col = (c % cpr) * 8;
    float r[8], pv[8];
    unpack8(*reinterpret_cast<const uint4 *>(pli + t * pli_stride + col), pv);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float g = rbf(gate[t * P + col + j]);
        const float ge = rbf(0.5f * g * (1.0f + tanhf(0.7978845608028654f * g * (1.0f + 0.044715f * g * g))));
        r[j] = ge * pv[j];
    }
    const uint4 packed = pack8(r);
    *reinterpret_cast<uint4 *>(gated + c * 8) = packed;
    float rr[8];
    unpack8(packed, rr);
    vmx_q8_1_emit8(rr, aq + c * 8, ad + (c >> 2), (c & 3) == 0);
}

__global__ __launch_bounds__(256) void g4_f32_to_bf16_kernel(const float *__restrict__ in, uint16_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = vmx_f32_to_bf16(in[i]);
}

__global__ __launch_bounds__(256) void g4_bf16_to_f32_kernel(const uint16_t *__restrict__ in, float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = vmx_bf16_to_f32(in[i]);
}

// logits (bf16 in place, or from f32) -> optional soft-cap: bf16(bf16(tanh(bf16(l * bf16(1/cap)))) * cap)  (soft_cap, :61-67)
__global__ __launch_bounds__(256) void g4_logits_kernel(uint16_t *__restrict__ logits, const float *__restrict__ src_f32, int64_t n,
                                                         float inv_cap_bf16, float cap) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float l = src_f32 ? rbf(src_f32[i]) : vmx_bf16_to_f32(logits[i]);
    if (cap > 0.f) l = rbf(rbf(tanhf(rbf(l * inv_cap_bf16))) * cap);
    logits[i] = vmx_f32_to_bf16(l);
}

struct Buf {
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

// IqLinear (crates/core/src/quantization/gguf/mod.rs:262-377): raw GGUF block bytes of one [out, in] weight.
struct IqLin {
    int ggml_type = 0, in_features = 0, out_features = 0;
    Buf raw;
    Buf packed;   // load-time repack for the one-row decode kernel (IQ3_XXS, IQ3_S, IQ2_S)
    Buf mma;      // 16-row-tile form of the same types for 2 .. 16 rows on the int8 matrix cores (only when max_batch > 1);
                  // the raw blocks stay for the other types, the dequantising prompt path and the parity aids
    bool loaded = false;
    double alg_bytes(int m) const {
        return (double)out_features * (in_features / 256) * vmx_iq_type_size(ggml_type) + (double)m * in_features * (1.0 + 4.0 / 32.0) +
               (double)m * out_features * 4.0;
    }
};

struct G4Layer {
    bool sliding = false, k_eq_v = false;
    // KV-shared layer (gemma4_quantized.rs:805-826, :1332-1337): no k/v projections or norms, q only; attention reads the
    // cache of layer kv_target, the last earlier non-shared layer of the same attention type (:441-469)
    bool kv_shared = false;
    int kv_target = -1;
    bool kv_calibrated = true;   // false: the next K/V write first pins the FP8 scales from what it writes (KVScales::calibrate_if_needed)
    int hd = 0, kvh = 0, rot = 0, window = 0;
    int inter = 0;   // MLP width: twice the base on a double-wide layer (:514-525)
    IqLin lin[9];   // VMX_L_Q .. VMX_L_DOWN, VMX_L_PLE_GATE, VMX_L_PLE_PROJ
    bool ple = false;
    Buf input_ln, post_attn_ln, pre_ffn_ln, post_ffn_ln, q_norm, k_norm, scalar, k_cache, v_cache, kv_scales, post_ple_norm;
    bool ready() const {
        for (int s = 0; s < 7; s++)
            if (!lin[s].loaded && !(s == VMX_L_V && k_eq_v) && !((s == VMX_L_K || s == VMX_L_V) && kv_shared)) return false;
        if (ple && (!lin[VMX_L_PLE_GATE].loaded || !lin[VMX_L_PLE_PROJ].loaded)) return false;
        return true;
    }
    int nq(int heads) const { return heads * hd; }
    int nkv() const { return kvh * hd; }
};

}  // namespace

struct vmx_gemma4 {
    vmx_gemma4_config cfg;
    std::vector<G4Layer> layers;
    Buf embed, final_norm, cos_sin_s, cos_sin_f;
    Buf h, x, q, attn, act, proj, out, logits, logits_f32, x_f32, attn_ws, argmax_ws, q8, q8b, h2, positions;
    void *q8_alt = nullptr;   // non-null while a linear reads its q8_1 activations from q8b (written by the fused gate|up + GeGLU launch)
    float embed_scale = 1.f;
    // per-layer inputs (hidden_size_per_layer_input > 0): embed_tokens_per_layer, per_layer_model_projection and its norm, and the
    // step's buffers [rows][layers * P]: gathered embedding rows, projection output, combined inputs
    vmx_comm *comm = nullptr;   // tensor-parallel group (not owned)
    Buf red_bf16, keys;         // row-parallel outputs on their way through the all-reduce; argmax keys of the vocabulary-parallel head
    int vocab_local() const { return cfg.tp_size > 1 ? cfg.vocab_size / cfg.tp_size : cfg.vocab_size; }
    float kv_headroom = 1.5f;   // DEFAULT_KV_HEADROOM_FACTOR (kv_cache/quantization.rs:108)
    Buf gemm_ws;                // dequantised weight of one projection (f32 + bf16 halves) for passes of more than 16 rows
    Buf a16;                    // fp16 copy of the rows a prompt-sized linear multiplies (the one-pass integer form)
    Buf kv_stage;               // bf16 K | V rows of one pass, used by an FP8 cache's first (calibrating) write
    int mmvq_mask = 15;         // measurement: which projections vmx_gemma4_mmvq_only launches (1 q|k|v, 2 o, 4 gate|up, 8 down)
    IqLin ple_model_proj;
    Buf ple_embed, ple_proj_norm, ple_e, ple_p, pli;
    float ple_embed_scale = 1.f, ple_proj_scale = 1.f, ple_input_scale = 1.f;
};

namespace {
int8_t *aq_of(vmx_gemma4 *m) { return (int8_t *)(m->q8_alt ? m->q8_alt : m->q8.p); }
float *ad_of(vmx_gemma4 *m, int tokens, int k) { return (float *)((char *)aq_of(m) + (((size_t)tokens * k + 255) & ~(size_t)255)); }

// Projections that share the q8_1 activations (1..3 of them: q|k|v, gate|up, or a single one), outputs at y + off[i]
int g_mma_min_rows = 3;   // rows from which the matrix-core MMVQ takes over from the per-row kernels (Gemma-4-12B: 3 rows 602 -> 628, 4 rows 724 -> 800 tok/s against the threshold 5; 2 rows 477 -> 437; tests: vmx_debug_iq_mma_min_rows)

int mmvq_group(vmx_gemma4 *m, const IqLin *const *lins, const int *offs, int n, float *y, int ldy, int tokens, hipStream_t st) {
    VmxMmvqSeg raw[3], pk[3];
    const int k = lins[0]->in_features;
    for (int i = 0; i < n; i++) {
        raw[i] = VmxMmvqSeg{lins[i]->raw.p, lins[i]->ggml_type, lins[i]->out_features, offs[i]};
        pk[i] = VmxMmvqSeg{lins[i]->packed.p, lins[i]->ggml_type, lins[i]->out_features, offs[i]};
    }
    if (vmx_iq_packed_preferred(pk, n, k, tokens)) return vmx_iq_mmvq_packed_multi(pk, n, aq_of(m), ad_of(m, tokens, k), y, k, st);
    // several rows: one decode of a sub-block for all of them on the int8 matrix cores, where every tensor has that form
    bool all_mma = tokens >= g_mma_min_rows;
    VmxMmvqSeg mm[3];
    for (int i = 0; i < n; i++) {
        mm[i] = VmxMmvqSeg{lins[i]->mma.p, lins[i]->ggml_type, lins[i]->out_features, offs[i]};
        all_mma = all_mma && lins[i]->mma.p != nullptr;
    }
    if (all_mma) return vmx_iq_mmvq_mma_multi(mm, n, aq_of(m), ad_of(m, tokens, k), y, k, tokens, ldy, st);
    return vmx_iq_mmvq_multi(raw, n, aq_of(m), ad_of(m, tokens, k), y, k, tokens, ldy, st);
}

int mmvq(vmx_gemma4 *m, const IqLin &l, float *y, int ldy, int tokens, hipStream_t st) {
    const IqLin *one[1] = {&l};
    const int off[1] = {0};
    if (l.packed.p) {
        const VmxMmvqSeg pk = {l.packed.p, l.ggml_type, l.out_features, 0};
        if (vmx_iq_packed_preferred(&pk, 1, l.in_features, tokens)) return mmvq_group(m, one, off, 1, y, ldy, tokens, st);
    }
    if (l.mma.p && tokens >= g_mma_min_rows) return mmvq_group(m, one, off, 1, y, ldy, tokens, st);
    return vmx_iq_mmvq_ld(l.ggml_type, l.raw.p, aq_of(m), ad_of(m, tokens, l.in_features), y, l.out_features, l.in_features, tokens, ldy, st);
}

bool g_mma_geglu = true;   // gate|up + GeGLU + q8_1 in one matrix-core launch where the shapes allow (tests: vmx_debug_gemma4_mma_geglu)

// the conditions under which mmvq_group would send gate|up to the matrix cores, plus those of the fused epilogue
bool mma_geglu_ok(vmx_gemma4 *m, const G4Layer &L, int H, int tokens) {
    const IqLin &g = L.lin[VMX_L_GATE], &u = L.lin[VMX_L_UP];
    if (!g_mma_geglu || m->q8_alt || tokens < g_mma_min_rows || tokens > 16 || !g.mma.p || !u.mma.p) return false;
    const VmxMmvqSeg pk[2] = {{g.packed.p, g.ggml_type, g.out_features, 0}, {u.packed.p, u.ggml_type, u.out_features, g.out_features}};
    if (vmx_iq_packed_preferred(pk, 2, H, tokens)) return false;
    const VmxMmvqSeg gm = {g.mma.p, g.ggml_type, g.out_features, 0}, um = {u.mma.p, u.ggml_type, u.out_features, 0};
    return vmx_iq_mma_geglu_ok(&gm, &um, H, tokens);
}

bool g_split_always = false;   // measurement: prompt-sized linears always through the two-term bf16 split (vmx_debug_gemma4_split_always)

// A linear on `rows` token rows: the q8_1 MMVQ up to IQ_GEMV_MAX_M = 16 rows (activations already quantised in m->q8), beyond that
// the reference's large-M form (gguf/mod.rs:358-365): dequantise the weight once, multiply the unquantised bf16 rows `a`.
int g4_linears(vmx_gemma4 *m, const IqLin *const *lins, const int *offs, int n, const uint16_t *a, float *y, int ldy, int rows,
               hipStream_t st) {
    if (rows <= 16) {
        if (n == 1 && offs[0] == 0) return mmvq(m, *lins[0], y, ldy, rows, st);
        return mmvq_group(m, lins, offs, n, y, ldy, rows, st);
    }
    // types whose dequantised weight is (f32 scale per super-block) x (integer that fp16 holds exactly) take one matrix-core pass
    // over fp16 activations; the others (IQ4_XS) the exact two-term bf16 split
    bool a16_ready = false;
    for (int i = 0; i < n; i++) {
        const IqLin &l = *lins[i];
        if (vmx_iq_int_form_ok(l.ggml_type) && !g_split_always) {
            if (!a16_ready) {
                if (int rc = vmx_bf16_rows_to_f16(a, (uint16_t *)m->a16.p, (int64_t)rows * l.in_features, st)) return rc;
                a16_ready = true;
            }
            if (int rc = vmx_iq_dequant_int(l.ggml_type, l.raw.p, m->gemm_ws.p, l.out_features, l.in_features, st)) return rc;
            if (int rc = vmx_int_gemm_f32((const uint16_t *)m->a16.p, m->gemm_ws.p, y + offs[i], rows, l.out_features, l.in_features, ldy, st))
                return rc;
            continue;
        }
        if (int rc = vmx_iq_dequant_split(l.ggml_type, l.raw.p, m->gemm_ws.p, l.out_features, l.in_features, st)) return rc;
        if (int rc = vmx_split_gemm_f32(a, m->gemm_ws.p, y + offs[i], rows, l.out_features, l.in_features, ldy, st)) return rc;
    }
    return VMX_OK;
}

int g4_linear(vmx_gemma4 *m, const IqLin &l, const uint16_t *a, float *y, int ldy, int rows, hipStream_t st) {
    const IqLin *one[1] = {&l};
    const int off[1] = {0};
    return g4_linears(m, one, off, 1, a, y, ldy, rows, st);
}

// Row-parallel linear under tensor parallelism: each rank's f32 output is rounded to bf16 (the reference's local matmul result),
// summed over the ranks, and handed on as f32 again for the post-norm that follows.
int g4_tp_reduce(vmx_gemma4 *m, float *out, int tokens, int hidden, hipStream_t st) {
    const int64_t n = (int64_t)tokens * hidden;
    hipLaunchKernelGGL(g4_f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, (uint16_t *)m->red_bf16.p, n);
    if (int rc = vmx_comm_all_reduce_sum_bf16(m->comm, (uint16_t *)m->red_bf16.p, (size_t)n, st)) return rc;
    hipLaunchKernelGGL(g4_bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint16_t *)m->red_bf16.p, out, n);
    return VMX_OK;
}

void launch_norm(hipStream_t st, int tokens, uint16_t *h, uint16_t *x, const float *o, const uint16_t *w_post, const uint16_t *scalar,
                 const uint16_t *w_next, float eps, int hidden, int8_t *aq, float *ad, const uint16_t *h_in = nullptr) {
    if (!h_in) h_in = h;
    if (hidden <= 2048) hipLaunchKernelGGL(g4_norm_kernel<1>, dim3(tokens), dim3(256), 0, st, h, h_in, x, o, w_post, scalar, w_next, eps, hidden, aq, ad);
    else if (hidden <= 4096) hipLaunchKernelGGL(g4_norm_kernel<2>, dim3(tokens), dim3(256), 0, st, h, h_in, x, o, w_post, scalar, w_next, eps, hidden, aq, ad);
    else if (hidden <= 8192) hipLaunchKernelGGL(g4_norm_kernel<4>, dim3(tokens), dim3(256), 0, st, h, h_in, x, o, w_post, scalar, w_next, eps, hidden, aq, ad);
    else hipLaunchKernelGGL(g4_norm_kernel<8>, dim3(tokens), dim3(256), 0, st, h, h_in, x, o, w_post, scalar, w_next, eps, hidden, aq, ad);
}
}  // namespace

extern "C" int vmx_gemma4_create(const vmx_gemma4_config *cfg, const int *layer_flags, vmx_gemma4 **out) {
    VMX_REQUIRE(cfg && layer_flags && out, "gemma4_create: null pointer");
    const vmx_gemma4_config &c = *cfg;
    VMX_REQUIRE(c.hidden_size > 0 && c.hidden_size % 256 == 0 && c.hidden_size <= 16384,
                "gemma4_create: hidden_size (%d) must be a multiple of 256, at most 16384", c.hidden_size);
    VMX_REQUIRE(c.intermediate_size > 0 && c.intermediate_size % 256 == 0, "gemma4_create: intermediate_size (%d) must be a multiple of 256",
                c.intermediate_size);
    VMX_REQUIRE(c.num_layers > 0 && c.num_heads > 0 && c.kv_heads_sliding > 0 && c.kv_heads_full > 0 &&
                    c.num_heads % c.kv_heads_sliding == 0 && c.num_heads % c.kv_heads_full == 0,
                "gemma4_create: bad head counts");
    VMX_REQUIRE(c.num_heads / c.kv_heads_sliding <= 16 && c.num_heads / c.kv_heads_full <= 16,
                "gemma4_create: more than 16 query heads per KV head (%d / %d, %d / %d)", c.num_heads, c.kv_heads_sliding, c.num_heads,
                c.kv_heads_full);
    for (int hd : {c.head_dim_sliding, c.head_dim_full})
        VMX_REQUIRE(hd == 64 || hd == 128 || hd == 256 || hd == 512, "gemma4_create: head_dim %d unsupported", hd);
    VMX_REQUIRE(c.rot_dim_sliding > 0 && c.rot_dim_sliding % 2 == 0 && c.rot_dim_sliding <= c.head_dim_sliding && c.rot_dim_full > 0 &&
                    c.rot_dim_full % 2 == 0 && c.rot_dim_full <= c.head_dim_full,
                "gemma4_create: rotary dims (%d, %d) must be even and at most the head dims", c.rot_dim_sliding, c.rot_dim_full);
    VMX_REQUIRE(c.max_batch >= 1 && c.max_batch <= 16, "gemma4_create: max_batch (%d) must be in 1..16", c.max_batch);
    VMX_REQUIRE(c.block_size > 0 && c.num_blocks > 0 && c.max_seq_len > 0 && c.vocab_size > 0 && c.max_position > 0 && c.sliding_window > 0,
                "gemma4_create: bad cache/vocab geometry");
    VMX_REQUIRE(c.kv_cache_dtype == VMX_KV_AUTO || c.kv_cache_dtype == VMX_KV_FP8_E4M3, "gemma4_create: unsupported kv dtype");
    VMX_REQUIRE(c.embed_ggml_type == 0 || (vmx_ggml_type_size(c.embed_ggml_type) > 0 && vmx_iq_type_size(c.embed_ggml_type) < 0),
                "gemma4_create: token_embd type %d unsupported (bf16 or a GGML legacy/K-quant type)", c.embed_ggml_type);
    VMX_REQUIRE(c.embed_ggml_type == 0 || c.hidden_size % vmx_ggml_block_elems(c.embed_ggml_type) == 0, "gemma4_create: hidden_size vs embed block");
    VMX_REQUIRE(c.max_prefill_tokens >= 0 && c.max_prefill_tokens <= 8192, "gemma4_create: max_prefill_tokens (%d) must be in 0..8192",
                c.max_prefill_tokens);
    VMX_REQUIRE(c.tp_size >= 0 && c.tp_rank >= 0 && (c.tp_size <= 1 || (c.tp_rank < c.tp_size && c.vocab_size % c.tp_size == 0)),
                "gemma4_create: tp_rank %d / tp_size %d invalid (vocab_size must divide by tp_size)", c.tp_rank, c.tp_size);
    VMX_REQUIRE(c.hidden_size_per_layer_input >= 0 && c.hidden_size_per_layer_input % 256 == 0 && c.hidden_size_per_layer_input <= 1024,
                "gemma4_create: hidden_size_per_layer_input (%d) must be 0 or a multiple of 256, at most 1024", c.hidden_size_per_layer_input);
    VMX_REQUIRE(c.hidden_size_per_layer_input == 0 || c.vocab_size_per_layer_input > 0,
                "gemma4_create: vocab_size_per_layer_input must be positive with per-layer inputs");
    VMX_REQUIRE(c.hidden_size_per_layer_input == 0 || c.ple_embed_ggml_type == 0 ||
                    (vmx_ggml_type_size(c.ple_embed_ggml_type) > 0 && vmx_iq_type_size(c.ple_embed_ggml_type) < 0),
                "gemma4_create: per-layer embedding type %d unsupported (bf16 or a GGML legacy/K-quant type)", c.ple_embed_ggml_type);
    vmx_gemma4 *m = new (std::nothrow) vmx_gemma4();
    VMX_REQUIRE(m, "gemma4_create: out of host memory");
    m->cfg = c;
    m->layers.resize(c.num_layers);
    // embed_tokens.forward(ids) * sqrt(hidden) is a bf16 affine: the factor is rounded to bf16 first
    auto to_bf16 = [](double v) {
        const float s = (float)v;
        uint32_t u;
        memcpy(&u, &s, 4);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
        float r;
        memcpy(&r, &u, 4);
        return r;
    };
    m->embed_scale = to_bf16(sqrt((double)c.hidden_size));
    const int P = c.hidden_size_per_layer_input, LP = P * c.num_layers;
    if (P > 0) {   // the three affine factors of compute_per_layer_inputs, each a bf16 scalar (gemma4_quantized.rs:2159-2161)
        m->ple_embed_scale = to_bf16(sqrt((double)P));
        m->ple_proj_scale = to_bf16(pow((double)c.hidden_size, -0.5));
        m->ple_input_scale = to_bf16(pow(2.0, -0.5));
    }
    const int H = c.hidden_size, I = c.intermediate_size, B = c.max_batch;
    const size_t kv_elem = c.kv_cache_dtype == VMX_KV_FP8_E4M3 ? 1 : 2;
    int rc = VMX_OK;
#define TRY(e) do { if (!rc) rc = (e); } while (0)
    if (c.embed_ggml_type == 0) TRY(m->embed.alloc((size_t)c.vocab_size * H * 2));
    else TRY(m->embed.alloc((size_t)c.vocab_size * (H / vmx_ggml_block_elems(c.embed_ggml_type)) * vmx_ggml_type_size(c.embed_ggml_type)));
    TRY(m->final_norm.alloc((size_t)H * 2));
    TRY(m->cos_sin_s.alloc((size_t)c.max_position * c.rot_dim_sliding * 4));
    TRY(m->cos_sin_f.alloc((size_t)c.max_position * c.rot_dim_full * 4));
    int max_nq = 0, max_proj = 2 * I, max_hd = 0, max_inter = I, max_nkv = 0;
    int first_shared = c.num_layers;
    for (int li = c.num_layers - 1; li >= 0 && (layer_flags[li] & VMX_G4_KV_SHARED); li--) first_shared = li;
    for (int li = 0; li < c.num_layers; li++) {
        G4Layer &L = m->layers[li];
        L.sliding = (layer_flags[li] & VMX_G4_SLIDING) != 0;
        L.k_eq_v = (layer_flags[li] & VMX_G4_K_EQ_V) != 0;
        L.kv_shared = (layer_flags[li] & VMX_G4_KV_SHARED) != 0;
        L.inter = (layer_flags[li] & VMX_G4_DOUBLE_WIDE_MLP) ? 2 * I : I;
        max_inter = L.inter > max_inter ? L.inter : max_inter;
        if (L.kv_shared) {
            if (li < first_shared) {
                vmx_set_error("gemma4_create: layer %d is KV-shared but layer %d after it is not (shared layers are the last ones)", li,
                              first_shared - 1);
                rc = rc ? rc : VMX_EINVAL;
            }
            for (int cand = first_shared - 1; cand >= 0 && L.kv_target < 0; cand--)
                if (((layer_flags[cand] & VMX_G4_SLIDING) != 0) == L.sliding) L.kv_target = cand;
            if (L.kv_target < 0) {
                vmx_set_error("gemma4_create: KV-shared layer %d has no earlier %s layer to share a cache with", li,
                              L.sliding ? "sliding" : "full-attention");
                rc = rc ? rc : VMX_EINVAL;
            }
            L.k_eq_v = false;
        }
        L.hd = L.sliding ? c.head_dim_sliding : c.head_dim_full;
        L.kvh = L.sliding ? c.kv_heads_sliding : c.kv_heads_full;
        L.rot = L.sliding ? c.rot_dim_sliding : c.rot_dim_full;
        L.window = L.sliding ? c.sliding_window : 0;
        const int NQ = c.num_heads * L.hd, NKV = L.kvh * L.hd;
        if (NQ % 256 != 0) {
            vmx_set_error("gemma4_create: num_heads * head_dim (%d) must be a multiple of 256 (o_proj input)", NQ);
            rc = rc ? rc : VMX_EINVAL;
        }
        const int in_f[7] = {H, H, H, NQ, H, H, L.inter}, out_f[7] = {NQ, NKV, NKV, H, L.inter, L.inter, H};
        for (int s = 0; s < 7; s++) { L.lin[s].in_features = in_f[s]; L.lin[s].out_features = out_f[s]; }
        if (P > 0) {   // per_layer_input_gate [P, hidden], per_layer_projection [hidden, P], post_per_layer_input_norm [hidden] (:1605-1625)
            L.ple = true;
            L.lin[VMX_L_PLE_GATE].in_features = H;
            L.lin[VMX_L_PLE_GATE].out_features = P;
            L.lin[VMX_L_PLE_PROJ].in_features = P;
            L.lin[VMX_L_PLE_PROJ].out_features = H;
            TRY(L.post_ple_norm.alloc((size_t)H * 2));
        }
        max_nq = NQ > max_nq ? NQ : max_nq;
        max_nkv = NKV > max_nkv ? NKV : max_nkv;
        max_proj = NQ + 2 * NKV > max_proj ? NQ + 2 * NKV : max_proj;
        max_proj = 2 * L.inter > max_proj ? 2 * L.inter : max_proj;
        max_hd = L.hd > max_hd ? L.hd : max_hd;
        TRY(L.input_ln.alloc((size_t)H * 2));
        TRY(L.post_attn_ln.alloc((size_t)H * 2));
        TRY(L.pre_ffn_ln.alloc((size_t)H * 2));
        TRY(L.post_ffn_ln.alloc((size_t)H * 2));
        TRY(L.q_norm.alloc((size_t)L.hd * 2));
        TRY(L.k_norm.alloc((size_t)L.hd * 2));
        TRY(L.scalar.alloc(2));
        const size_t kvb = (size_t)c.num_blocks * c.block_size * L.kvh * L.hd * kv_elem;
        if (!L.kv_shared) {
            TRY(L.k_cache.alloc(kvb));
            TRY(L.v_cache.alloc(kvb));
        }
        TRY(L.kv_scales.alloc(vmx_kv_scales_state_bytes()));
        if (!rc) {
            const uint16_t one_bf16 = 0x3F80;
            if (vmx_kv_scales_init(L.kv_scales.p, 1.0f, 1.0f, 1, nullptr) != VMX_OK ||
                hipMemcpy(L.scalar.p, &one_bf16, 2, hipMemcpyHostToDevice) != hipSuccess)
                rc = VMX_ELAUNCH;
        }
    }
    int R = B > 16 ? B : 16;   // activation rows: a decode batch, or one pass of the prompt path (16 rows, or max_prefill_tokens)
    R = c.max_prefill_tokens > R ? c.max_prefill_tokens : R;
    if (P > 0) {
        if (c.ple_embed_ggml_type != 0 && LP % vmx_ggml_block_elems(c.ple_embed_ggml_type) != 0) {
            vmx_set_error("gemma4_create: layers * hidden_size_per_layer_input (%d) vs the per-layer embedding's block size", LP);
            rc = rc ? rc : VMX_EINVAL;
        } else if (c.ple_embed_ggml_type == 0) {
            TRY(m->ple_embed.alloc((size_t)c.vocab_size_per_layer_input * LP * 2));
        } else {
            TRY(m->ple_embed.alloc((size_t)c.vocab_size_per_layer_input * (LP / vmx_ggml_block_elems(c.ple_embed_ggml_type)) *
                                   vmx_ggml_type_size(c.ple_embed_ggml_type)));
        }
        m->ple_model_proj.in_features = H;
        m->ple_model_proj.out_features = LP;
        TRY(m->ple_proj_norm.alloc((size_t)P * 2));
        TRY(m->ple_e.alloc((size_t)R * LP * 2));
        TRY(m->ple_p.alloc((size_t)R * LP * 4));
        TRY(m->pli.alloc((size_t)R * LP * 2));
        max_proj = P > max_proj ? P : max_proj;
    }
    if (c.max_prefill_tokens > 16) {   // the largest projection's dequantised form
        size_t wmax = 0;
        for (const G4Layer &L : m->layers)
            for (const IqLin &l : L.lin) {
                const size_t w = vmx_iq_gemm_workspace(l.out_features, l.in_features);
                wmax = w > wmax ? w : wmax;
            }
        if (P > 0) {
            const size_t w = vmx_iq_gemm_workspace(LP, H);
            wmax = w > wmax ? w : wmax;
        }
        TRY(m->gemm_ws.alloc(wmax));
        int kmax = H;
        for (const G4Layer &L : m->layers)
            for (const IqLin &l : L.lin) kmax = l.in_features > kmax ? l.in_features : kmax;
        TRY(m->a16.alloc((size_t)c.max_prefill_tokens * kmax * 2));
    }
    TRY(m->red_bf16.alloc((size_t)R * H * 2));
    TRY(m->keys.alloc((size_t)R * 8));
    TRY(m->positions.alloc((size_t)R * 4));
    TRY(m->kv_stage.alloc((size_t)R * 2 * max_nkv * 2));
    TRY(m->h.alloc((size_t)R * H * 2));
    TRY(m->x.alloc((size_t)R * H * 2));
    TRY(m->q.alloc((size_t)R * max_nq * 2));
    TRY(m->attn.alloc((size_t)R * max_nq * 2));
    TRY(m->act.alloc((size_t)R * max_inter * 2));
    TRY(m->proj.alloc((size_t)R * max_proj * 4));
    TRY(m->out.alloc((size_t)R * H * 4));
    TRY(m->logits.alloc((size_t)B * c.vocab_size * 2));
    if (c.embed_ggml_type != 0) {
        TRY(m->logits_f32.alloc((size_t)B * c.vocab_size * 4));
        TRY(m->x_f32.alloc((size_t)R * H * 4));
    }
    {
        size_t kmax = (size_t)(H > max_inter ? H : max_inter);
        kmax = (size_t)max_nq > kmax ? (size_t)max_nq : kmax;
        kmax = (size_t)P > kmax ? (size_t)P : kmax;
        TRY(m->q8.alloc(R * kmax + 256 + R * (kmax / 32) * 4));
        const size_t r16 = (size_t)(R < 16 ? R : 16);       // up to 16 rows: the fused gate|up launches read q8 and write this
        TRY(m->q8b.alloc(r16 * kmax + 256 + r16 * (kmax / 32) * 4));
        TRY(m->h2.alloc((size_t)H * 2));                   // one row of the residual stream, written by the norm-fused gate|up launch
    }
    {
        const size_t a = vmx_paged_attention_workspace_bytes(B, c.num_heads, max_hd, c.max_seq_len);
        const size_t b = vmx_split_decode_attention_workspace_bytes(B, c.num_heads, max_hd);
        TRY(m->attn_ws.alloc(a > b ? a : b));
    }
    TRY(m->argmax_ws.alloc(vmx_argmax_workspace_bytes(B, c.vocab_size)));
#undef TRY
    if (rc) {
        vmx_gemma4_destroy(m);
        return rc;
    }
    *out = m;
    return VMX_OK;
}

extern "C" void vmx_gemma4_destroy(vmx_gemma4 *m) {
    if (!m) return;
    for (G4Layer &L : m->layers) {
        for (IqLin &l : L.lin) { l.raw.release(); l.packed.release(); }
        for (Buf *b : {&L.input_ln, &L.post_attn_ln, &L.pre_ffn_ln, &L.post_ffn_ln, &L.q_norm, &L.k_norm, &L.scalar, &L.k_cache, &L.v_cache,
                       &L.kv_scales, &L.post_ple_norm})
            b->release();
    }
    m->ple_model_proj.raw.release();
    m->ple_model_proj.packed.release();
    for (Buf *b : {&m->ple_embed, &m->ple_proj_norm, &m->ple_e, &m->ple_p, &m->pli, &m->kv_stage, &m->red_bf16, &m->keys, &m->gemm_ws, &m->a16}) b->release();
    for (Buf *b : {&m->embed, &m->final_norm, &m->cos_sin_s, &m->cos_sin_f, &m->positions, &m->h, &m->x, &m->q, &m->attn, &m->act, &m->proj, &m->out,
                   &m->logits, &m->logits_f32, &m->x_f32, &m->attn_ws, &m->argmax_ws, &m->q8, &m->q8b, &m->h2})
        b->release();
    delete m;
}

extern "C" int vmx_gemma4_load_tensor(vmx_gemma4 *m, int which, int layer, const void *src, size_t bytes, void *stream) {
    VMX_REQUIRE(m && src, "gemma4_load_tensor: null pointer");
    Buf *dst = nullptr;
    switch (which) {
        case VMX_G4T_EMBED: dst = &m->embed; break;
        case VMX_G4T_FINAL_NORM: dst = &m->final_norm; break;
        case VMX_G4T_COS_SIN_SLIDING: dst = &m->cos_sin_s; break;
        case VMX_G4T_COS_SIN_FULL: dst = &m->cos_sin_f; break;
        case VMX_G4T_PLE_EMBED: dst = m->ple_embed.p ? &m->ple_embed : nullptr; break;
        case VMX_G4T_PLE_PROJ_NORM: dst = m->ple_proj_norm.p ? &m->ple_proj_norm : nullptr; break;
        default: break;
    }
    if (which == VMX_G4T_POST_PLE_NORM) {
        VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size(), "gemma4_load_tensor: layer %d out of range", layer);
        dst = m->layers[layer].ple ? &m->layers[layer].post_ple_norm : nullptr;
    }
    if (!dst && which >= VMX_G4T_INPUT_LN && which <= VMX_G4T_KV_SCALES) {
        VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size(), "gemma4_load_tensor: layer %d out of range", layer);
        G4Layer &L = m->layers[layer];
        Buf *tab[8] = {&L.input_ln, &L.post_attn_ln, &L.pre_ffn_ln, &L.post_ffn_ln, &L.q_norm, &L.k_norm, &L.scalar, &L.kv_scales};
        dst = tab[which - VMX_G4T_INPUT_LN];
    }
    VMX_REQUIRE(dst, "gemma4_load_tensor: unknown tensor id %d", which);
    if (which == VMX_G4T_KV_SCALES) {   // explicit scales are calibrated scales (KVScales::set, kv_cache/quantization.rs:218-224)
        VMX_REQUIRE(bytes == 8, "gemma4_load_tensor: KV scales are two f32, got %zu bytes", bytes);
        float ks[2];
        VMX_CHECK_HIP(hipMemcpy(ks, src, 8, hipMemcpyDefault));
        m->layers[layer].kv_calibrated = true;
        return vmx_kv_scales_init(dst->p, ks[0], ks[1], 1, stream);
    }
    VMX_REQUIRE(bytes == dst->bytes, "gemma4_load_tensor: tensor %d expects %zu bytes, got %zu", which, dst->bytes, bytes);
    VMX_CHECK_HIP(hipMemcpyAsync(dst->p, src, bytes, hipMemcpyDefault, vmx_stream(stream)));
    return VMX_OK;
}

extern "C" int vmx_gemma4_load_iq_linear(vmx_gemma4 *m, int layer, int slot, int ggml_type, const void *blocks, size_t bytes,
                                         void *stream) {
    VMX_REQUIRE(m && blocks, "gemma4_load_iq_linear: null pointer");
    const bool model_level = slot == VMX_L_PLE_MODEL_PROJ;
    VMX_REQUIRE(model_level || (layer >= 0 && layer < (int)m->layers.size() && slot >= VMX_L_Q && slot <= VMX_L_PLE_PROJ),
                "gemma4_load_iq_linear: layer %d / slot %d out of range", layer, slot);
    const int ts = vmx_iq_type_size(ggml_type);
    VMX_REQUIRE(ts > 0, "gemma4_load_iq_linear: ggml type %d is not a supported I-quant", ggml_type);
    VMX_REQUIRE(slot < VMX_L_PLE_GATE || m->cfg.hidden_size_per_layer_input > 0,
                "gemma4_load_iq_linear: slot %d needs hidden_size_per_layer_input > 0", slot);
    G4Layer &L = m->layers[model_level ? 0 : layer];
    VMX_REQUIRE(!(slot == VMX_L_V && L.k_eq_v), "gemma4_load_iq_linear: layer %d reuses K as V and has no v_proj", layer);
    VMX_REQUIRE(!((slot == VMX_L_K || slot == VMX_L_V) && L.kv_shared),
                "gemma4_load_iq_linear: layer %d shares layer %d's KV cache and has no k_proj / v_proj", layer, L.kv_target);
    IqLin &l = model_level ? m->ple_model_proj : L.lin[slot];
    VMX_REQUIRE(l.in_features % 256 == 0, "gemma4_load_iq_linear: in_features %d not a multiple of 256", l.in_features);
    const size_t need = (size_t)l.out_features * (l.in_features / 256) * ts;
    VMX_REQUIRE(bytes == need, "gemma4_load_iq_linear: expected %zu bytes for [%d, %d] of type %d, got %zu", need, l.out_features,
                l.in_features, ggml_type, bytes);
    l.raw.release();
    if (int rc = l.raw.alloc(need)) return rc;
    VMX_CHECK_HIP(hipMemcpyAsync(l.raw.p, blocks, need, hipMemcpyDefault, vmx_stream(stream)));
    l.ggml_type = ggml_type;
    l.packed.release();
    if (const size_t pb = vmx_iq_packed_bytes(ggml_type, l.out_features, l.in_features)) {   // types with a packed form
        if (int rc = l.packed.alloc(pb)) return rc;
        if (int rc = vmx_iq_pack(ggml_type, l.raw.p, l.packed.p, l.out_features, l.in_features, stream)) return rc;
    }
    l.mma.release();
    if (m->cfg.max_batch > 1)
        if (const size_t mb = vmx_iq_mma_packed_bytes(ggml_type, l.out_features, l.in_features)) {
            if (int rc = l.mma.alloc(mb)) return rc;
            if (int rc = vmx_iq_mma_pack(ggml_type, l.raw.p, l.mma.p, l.out_features, l.in_features, stream)) return rc;
        }
    l.loaded = true;
    return VMX_OK;
}

namespace {
__global__ void g4_iota_kernel(int32_t *pos, int offset, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos[i] = offset + i;
}

// The layer stack on `rows` token rows, up to and including the final norm (m->x).  prefill_offset < 0: every row is one
// sequence's new token (paged decode attention over block_tables [rows, max_blocks_per_seq] / seq_lens); >= 0: the rows
// are consecutive tokens of ONE sequence at positions prefill_offset + i (causal attention over the cache, block_tables =
// that sequence's block table).
int g4_run_layers(vmx_gemma4 *m, const int32_t *input_ids, const int32_t *positions, const int32_t *slot_mapping,
                  const int32_t *block_tables, const int32_t *seq_lens, int rows, int max_blocks_per_seq, int max_seq_len, int prefill_offset,
                  hipStream_t st) {
    const vmx_gemma4_config &c = m->cfg;
    const int H = c.hidden_size, B = rows, Hq = c.num_heads;
    uint16_t *h = (uint16_t *)m->h.p, *x = (uint16_t *)m->x.p, *q = (uint16_t *)m->q.p, *attn = (uint16_t *)m->attn.p,
             *act = (uint16_t *)m->act.p;
    float *proj = (float *)m->proj.p, *out = (float *)m->out.p;
    const int fp8 = c.kv_cache_dtype == VMX_KV_FP8_E4M3;
    const bool tp = c.tp_size > 1 || m->comm != nullptr;   // a 1-rank communicator still exercises the collective path
    VMX_REQUIRE(!tp || m->comm, "gemma4: tp_size %d but no communicator attached (vmx_gemma4_set_comm)", c.tp_size);
    if (int rc = vmx_comm_check_health(m->comm, "gemma4 forward")) return rc;   // an earlier collective of this group failed

    if (c.embed_ggml_type == 0) {
        hipLaunchKernelGGL(g4_embed_kernel, dim3(B), dim3(256), 0, st, h, (const uint16_t *)m->embed.p, input_ids, H, c.vocab_size,
                           m->embed_scale);
    } else if (int rc = vmx_kquant_gather_scaled_bf16(c.embed_ggml_type, m->embed.p, input_ids, h, B, H, c.vocab_size, m->embed_scale, st)) {
        return rc;
    }
    const int P = c.hidden_size_per_layer_input, LP = P * c.num_layers;
    if (P > 0) {   // per-layer inputs of this step, [rows][layers][P] (compute_per_layer_inputs, :2143-2192)
        uint16_t *pe = (uint16_t *)m->ple_e.p;
        if (c.ple_embed_ggml_type == 0) {
            hipLaunchKernelGGL(g4_embed_kernel, dim3(B), dim3(256), 0, st, pe, (const uint16_t *)m->ple_embed.p, input_ids, LP,
                               c.vocab_size_per_layer_input, m->ple_embed_scale);
        } else if (int rc = vmx_kquant_gather_scaled_bf16(c.ple_embed_ggml_type, m->ple_embed.p, input_ids, pe, B, LP,
                                                          c.vocab_size_per_layer_input, m->ple_embed_scale, st)) {
            return rc;
        }
        if (int rc = vmx_quantize_q8_1_bf16(h, aq_of(m), ad_of(m, B, H), (int64_t)B * (H / 32), st)) return rc;
        if (int rc = g4_linear(m, m->ple_model_proj, h, (float *)m->ple_p.p, LP, B, st)) return rc;
        hipLaunchKernelGGL(g4_ple_inputs_kernel, dim3(c.num_layers, B), dim3(64), 0, st, (const float *)m->ple_p.p, pe,
                           (const uint16_t *)m->ple_proj_norm.p, (uint16_t *)m->pli.p, P, c.rms_norm_eps, m->ple_proj_scale,
                           m->ple_input_scale);
    }
    // pending: the previous layer's last linear output still to be post-normed, added, scaled
    const float *pending = nullptr;
    const uint16_t *pending_w = nullptr, *pending_scalar = nullptr;
    const uint16_t *h_src = nullptr;   // non-null: the stream was left in m->h2 by a norm-fused gate|up launch; the next norm moves it back to h
    int li = -1;
    for (const G4Layer &L : m->layers) {
        li++;
        const G4Layer &KV = L.kv_shared ? m->layers[L.kv_target] : L;   // whose paged cache this layer's attention reads
        const int hd = L.hd, Hkv = L.kvh, NQ = Hq * hd, NKV = Hkv * hd, I = L.inter;
        const int nseg = L.kv_shared ? 1 : (L.k_eq_v ? 2 : 3);
        const int ntot = NQ + (nseg - 1) * NKV;
        // (previous layer's tail) + input_layernorm -> x, q8_1(x)
        // One row whose stream the previous layer's norm-fused gate|up launch left in h2: post_feedforward_layernorm + residual +
        // layer scalar + input_layernorm + q8_1 happen in the prologue of the q|k|v launch (every workgroup for itself; the
        // stream goes back to h), so the layer boundary costs no launch of its own.
        bool norm_in_qkv = false;
        if (B == 1 && pending && h_src && h_src != h) {
            const int offs[3] = {0, NQ, NQ + NKV};
            VmxMmvqSeg pk[3];
            for (int i = 0; i < nseg; i++)
                pk[i] = VmxMmvqSeg{L.lin[VMX_L_Q + i].packed.p, L.lin[VMX_L_Q + i].ggml_type, L.lin[VMX_L_Q + i].out_features, offs[i]};
            if (vmx_iq_packed_norm_multi_ok(pk, nseg, H, B)) {
                if (int rc = vmx_iq_mmvq_packed_norm_multi(pk, nseg, pending, h_src, h, pending_w, pending_scalar, (const uint16_t *)L.input_ln.p,
                                                           c.rms_norm_eps, proj, H, st))
                    return rc;
                norm_in_qkv = true;
                h_src = nullptr;
            }
        }
        if (!norm_in_qkv) {
            launch_norm(st, B, h, x, pending, pending_w, pending_scalar, (const uint16_t *)L.input_ln.p, c.rms_norm_eps, H, aq_of(m),
                        ad_of(m, B, H), h_src);
            h_src = nullptr;
            const IqLin *qkv[3] = {&L.lin[VMX_L_Q], &L.lin[VMX_L_K], &L.lin[VMX_L_V]};
            const int offs[3] = {0, NQ, NQ + NKV};
            if (int rc = g4_linears(m, qkv, offs, nseg, x, proj, ntot, B, st)) return rc;
        }
        const float *ks = (const float *)KV.kv_scales.p;
        // short contexts: per-head norms + RoPE + KV write + attention + merge in one launch (not for K == V or KV-shared layers)
        // an FP8 cache whose scales are not pinned yet: this write goes through a bf16 staging buffer, calibrates, then scatters
        const bool first_write = fp8 && !L.kv_shared && !L.kv_calibrated;
        uint16_t *stage = (uint16_t *)m->kv_stage.p;
        const bool fused_attn = prefill_offset < 0 && !L.k_eq_v && !L.kv_shared && !first_write &&
                                vmx_fused_decode_attention_ok(hd, Hq / Hkv, L.rot, max_seq_len, L.window);
        // sliding layers with 256-wide heads: per-head norms + RoPE + KV write inside the attention workgroups (4 per KV head),
        // then the merge that also emits q8_1 for o_proj: two launches instead of three
        const bool split_attn = !fused_attn && prefill_offset < 0 && !L.k_eq_v && !L.kv_shared && !first_write && hd == 256 &&
                                vmx_split_decode_attention_parts_wide(hd, Hq / Hkv, L.rot, max_seq_len, L.window) > 0;
        if (split_attn) {
            VmxAttnParts ap;
            // workgroups per (sequence, KV head): 8 for a few rows; with many rows every compute unit has work at fewer, and each
            // partition repeats the per-head norms and rotation of its q heads (0 = the kernel's own choice)
            static const int p_many = vmx_tune("VMX_G4_SPLIT_P_MANY", 2), p_some = vmx_tune("VMX_G4_SPLIT_P_SOME", 4);
            const int split_p = B >= 12 ? p_many : (B >= 5 ? p_some : 0);
            if (int rc = vmx_split_decode_attention(&ap, m->attn_ws.p, proj, 1, L.k_cache.p, L.v_cache.p, (const uint16_t *)L.q_norm.p,
                                                    (const uint16_t *)L.k_norm.p, c.rms_norm_eps,
                                                    (const float *)(L.sliding ? m->cos_sin_s.p : m->cos_sin_f.p), positions, slot_mapping,
                                                    block_tables, seq_lens, 1.0f, B, Hq, Hkv, hd, L.rot, max_blocks_per_seq, max_seq_len,
                                                    c.block_size, L.window, c.kv_cache_dtype, ks, ks + 1, 1, st, nullptr, split_p))
                return rc;
            if (int rc = vmx_paged_attention_merge(attn, &ap, seq_lens, B, Hq, hd, L.window, st, aq_of(m), ad_of(m, B, NQ))) return rc;
        } else if (fused_attn) {
            if (int rc = vmx_fused_decode_attention(attn, proj, 1, L.k_cache.p, L.v_cache.p, (const uint16_t *)L.q_norm.p,
                                                    (const uint16_t *)L.k_norm.p, c.rms_norm_eps,
                                                    (const float *)(L.sliding ? m->cos_sin_s.p : m->cos_sin_f.p), positions, slot_mapping,
                                                    block_tables, seq_lens, 1.0f, B, Hq, Hkv, hd, L.rot, max_blocks_per_seq, c.block_size,
                                                    L.window, c.kv_cache_dtype, ks, ks + 1, aq_of(m), ad_of(m, B, NQ), 1, st))
                return rc;
        } else {
            // a KV-shared layer launches the query heads only: its K/V for these tokens were written by the target layer
            hipLaunchKernelGGL(g4_qkv_post_kernel, dim3(L.kv_shared ? Hq : Hq + 2 * Hkv, B), dim3(64), 0, st, proj, ntot, q, KV.k_cache.p,
                               KV.v_cache.p, (const uint16_t *)L.q_norm.p, (const uint16_t *)L.k_norm.p, c.rms_norm_eps,
                               (const float *)(L.sliding ? m->cos_sin_s.p : m->cos_sin_f.p), positions, slot_mapping, Hq, Hkv, hd, L.rot,
                               L.k_eq_v ? 1 : 0, fp8, ks, ks + 1, first_write ? stage : nullptr);
            if (first_write) {
                const int64_t n = (int64_t)B * NKV;
                if (int rc = vmx_kv_scales_calibrate_if_needed(L.kv_scales.p, stage, n, stage + n, n, m->kv_headroom, st)) return rc;
                if (int rc = vmx_reshape_and_cache(stage, stage + n, L.k_cache.p, L.v_cache.p, slot_mapping, B, Hkv, hd, c.block_size,
                                                   c.kv_cache_dtype, ks, ks + 1, st))
                    return rc;
            }
            if (prefill_offset >= 0) {   // the rows' K/V are in the cache: causal (windowed) attention over it, then q8_1 for o_proj
                if (int rc = vmx_prefill_attention_bf16(attn, q, KV.k_cache.p, KV.v_cache.p, block_tables, B, prefill_offset, 1.0f, Hq, Hkv,
                                                        hd, c.block_size, L.window, c.kv_cache_dtype, ks, ks + 1, st))
                    return rc;
                if (int rc = vmx_quantize_q8_1_bf16(attn, aq_of(m), ad_of(m, B, NQ), (int64_t)B * (NQ / 32), st)) return rc;
            } else {
                VmxAttnParts ap;
                if (int rc = vmx_paged_attention_parts(q, KV.k_cache.p, KV.v_cache.p, block_tables, seq_lens, 1.0f, B, Hq, Hkv,
                                                       max_blocks_per_seq, max_seq_len, hd, c.block_size, L.window, c.kv_cache_dtype, ks,
                                                       ks + 1, m->attn_ws.p, &ap, st))
                    return rc;
                if (int rc = vmx_paged_attention_merge(attn, &ap, seq_lens, B, Hq, hd, L.window, st, aq_of(m), ad_of(m, B, NQ))) return rc;
            }
        }
        if (int rc = g4_linear(m, L.lin[VMX_L_O], attn, out, H, B, st)) return rc;
        if (tp)
            if (int rc = g4_tp_reduce(m, out, B, H, st)) return rc;
        const VmxMmvqSeg gate_pk = {L.lin[VMX_L_GATE].packed.p, L.lin[VMX_L_GATE].ggml_type, L.lin[VMX_L_GATE].out_features, 0};
        const VmxMmvqSeg up_pk = {L.lin[VMX_L_UP].packed.p, L.lin[VMX_L_UP].ggml_type, L.lin[VMX_L_UP].out_features, 0};
        // one row, no per-layer-input block behind the MLP (it reads x and h of this norm): post_attention_layernorm + residual +
        // pre_feedforward_layernorm + q8_1 happen in the prologue of the gate|up launch, every workgroup for itself
        const bool norm_in_mlp = B == 1 && !L.ple && vmx_iq_packed_norm_geglu_ok(&gate_pk, &up_pk, H, B);
        if (norm_in_mlp) {
            m->q8_alt = m->q8b.p;
            const int rc_glu = vmx_iq_mmvq_packed_norm_geglu(&gate_pk, &up_pk, out, h, (uint16_t *)m->h2.p, (const uint16_t *)L.post_attn_ln.p,
                                                             (const uint16_t *)L.pre_ffn_ln.p, c.rms_norm_eps, act, aq_of(m), ad_of(m, B, I), H, st);
            if (rc_glu) {
                m->q8_alt = nullptr;
                return rc_glu;
            }
            h_src = (const uint16_t *)m->h2.p;
        } else {
        // post_attention_layernorm + residual, pre_feedforward_layernorm -> x, q8_1(x)
        launch_norm(st, B, h, x, out, (const uint16_t *)L.post_attn_ln.p, nullptr, (const uint16_t *)L.pre_ffn_ln.p, c.rms_norm_eps, H,
                    aq_of(m), ad_of(m, B, H));
        }
        if (norm_in_mlp) {
        } else if (B == 1 && vmx_iq_packed_geglu_ok(&gate_pk, &up_pk, H, B)) {
            // one launch: both projections, the GeGLU and q8_1 of the result.  It reads q8_1(x) from q8 while other workgroups
            // already write q8_1(act), so the result goes to q8b and the down projection reads it from there.
            m->q8_alt = m->q8b.p;
            const int rc_glu = vmx_iq_mmvq_packed_geglu(&gate_pk, &up_pk, (const int8_t *)m->q8.p,
                                                        (const float *)((const char *)m->q8.p + (((size_t)H + 255) & ~(size_t)255)), act,
                                                        aq_of(m), ad_of(m, B, I), H, st);
            if (rc_glu) {
                m->q8_alt = nullptr;
                return rc_glu;
            }
        } else if (mma_geglu_ok(m, L, H, B)) {
            // several rows on the matrix cores: both projections, the GeGLU and q8_1 of the result in one launch; it reads q8_1(x)
            // from q8 while other workgroups already write q8_1(act), so the result goes to q8b (read from there by down)
            const VmxMmvqSeg gate_mm = {L.lin[VMX_L_GATE].mma.p, L.lin[VMX_L_GATE].ggml_type, L.lin[VMX_L_GATE].out_features, 0};
            const VmxMmvqSeg up_mm = {L.lin[VMX_L_UP].mma.p, L.lin[VMX_L_UP].ggml_type, L.lin[VMX_L_UP].out_features, 0};
            const int8_t *aq_in = (const int8_t *)m->q8.p;
            const float *ad_in = (const float *)((const char *)m->q8.p + (((size_t)B * H + 255) & ~(size_t)255));
            m->q8_alt = m->q8b.p;
            const int rc_glu = vmx_iq_mmvq_mma_geglu(&gate_mm, &up_mm, aq_in, ad_in, act, aq_of(m), ad_of(m, B, I), H, B, st);
            if (rc_glu) {
                m->q8_alt = nullptr;
                return rc_glu;
            }
        } else {
            const IqLin *gu[2] = {&L.lin[VMX_L_GATE], &L.lin[VMX_L_UP]};
            const int offs[2] = {0, I};
            if (int rc = g4_linears(m, gu, offs, 2, x, proj, 2 * I, B, st)) return rc;
            int64_t blocks = ((int64_t)B * I / 8 + 255) / 256;
            hipLaunchKernelGGL(g4_geglu_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, st, proj, B, act, I, aq_of(m),
                               ad_of(m, B, I));
        }
        const int rc_down = g4_linear(m, L.lin[VMX_L_DOWN], act, out, H, B, st);
        m->q8_alt = nullptr;
        if (rc_down) return rc_down;
        if (tp)
            if (int rc = g4_tp_reduce(m, out, B, H, st)) return rc;
        pending = out;
        pending_w = (const uint16_t *)L.post_ffn_ln.p;
        pending_scalar = (const uint16_t *)L.scalar.p;
        if (L.ple) {   // apply_ple (:1663-1683): h += norm(per_layer_projection(gelu_tanh(gate(h)) * pli[layer])), before the scalar
            launch_norm(st, B, h, x, pending, pending_w, nullptr, nullptr, c.rms_norm_eps, H, aq_of(m), ad_of(m, B, H), h_src);
            h_src = nullptr;
            if (int rc = g4_linear(m, L.lin[VMX_L_PLE_GATE], x, proj, P, B, st)) return rc;
            hipLaunchKernelGGL(g4_ple_gate_kernel, dim3((unsigned)(((int64_t)B * P / 8 + 255) / 256)), dim3(256), 0, st, proj,
                               (const uint16_t *)m->pli.p + (size_t)li * P, (int64_t)LP, B, P, act, aq_of(m), ad_of(m, B, P));
            if (int rc = g4_linear(m, L.lin[VMX_L_PLE_PROJ], act, out, H, B, st)) return rc;
            pending_w = (const uint16_t *)L.post_ple_norm.p;
        }
    }
    launch_norm(st, B, h, x, pending, pending_w, pending_scalar, (const uint16_t *)m->final_norm.p, c.rms_norm_eps, H, nullptr, nullptr, h_src);
    VMX_CHECK_LAUNCH("gemma4 layers");
    for (G4Layer &L : m->layers) L.kv_calibrated = true;   // every cache has had its first write
    return VMX_OK;
}

// tied lm_head (+ soft-cap, greedy ids) on `rows` rows of normed activations starting at x
int g4_run_head(vmx_gemma4 *m, const uint16_t *x, int rows, int32_t *out_ids, uint16_t *logits_out, void *stream) {
    const vmx_gemma4_config &c = m->cfg;
    hipStream_t st = vmx_stream(stream);
    const int H = c.hidden_size, B = rows;
    uint16_t *logits = logits_out ? logits_out : (uint16_t *)m->logits.p;
    const bool tp = c.tp_size > 1 || m->comm != nullptr;
    const int VL = m->vocab_local();                                // this rank's rows of the tied table
    const int64_t v0 = tp ? (int64_t)c.tp_rank * VL : 0;
    const int64_t nlog = (int64_t)B * VL;
    const float cap = c.final_logit_softcap;
    float inv_cap = 0.f;
    if (cap > 0.f) {   // xs / cap is a bf16 multiply by bf16(1 / cap)
        const float ic = 1.0f / cap;
        uint32_t u;
        memcpy(&u, &ic, 4);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
        memcpy(&inv_cap, &u, 4);
    }
    if (c.embed_ggml_type == 0) {
        if (int rc = vmx_dense_gemv_launch(logits, x, (const uint16_t *)m->embed.p + v0 * H, nullptr, B, VL, H, st)) return rc;
        if (cap > 0.f)
            hipLaunchKernelGGL(g4_logits_kernel, dim3((unsigned)((nlog + 255) / 256)), dim3(256), 0, st, logits, (const float *)nullptr, nlog,
                               inv_cap, cap);
    } else {   // tied lm_head over the quantized table (QMatMul, :1849-1865): f32 activations, f32 result cast back to bf16
        float *xf = (float *)m->x_f32.p, *lf = (float *)m->logits_f32.p;
        static const int q3_mma = vmx_tune("VMX_G4_Q3K_MMA_HEAD", 1);
        if (q3_mma && c.embed_ggml_type == 11 && vmx_q3k_gemm16_ok(VL, H, B < 16 ? B : 16)) {
            // several rows over a Q3_K table: one pass of the table on the matrix cores per 16 rows (exact products of the bf16
            // activations with the integer weights) instead of one pass per row
            const size_t row_bytes = (size_t)(H / 256) * 110;
            for (int b0 = 0; b0 < B; b0 += 16) {
                const int mb = B - b0 < 16 ? B - b0 : 16;
                if (mb < 2) {
                    hipLaunchKernelGGL(g4_bf16_to_f32_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, st, x + (int64_t)b0 * H, xf, (int64_t)H);
                    if (int rc = vmx_kquant_gemv_f32(11, (const char *)m->embed.p + (size_t)v0 * row_bytes, xf, lf + (int64_t)b0 * VL, VL, H, 1, stream))
                        return rc;
                } else if (int rc = vmx_q3k_gemm16_bf16((const char *)m->embed.p + (size_t)v0 * row_bytes, x + (int64_t)b0 * H, lf + (int64_t)b0 * VL, VL,
                                                        H, mb, st))
                    return rc;
            }
            hipLaunchKernelGGL(g4_logits_kernel, dim3((unsigned)((nlog + 255) / 256)), dim3(256), 0, st, logits, (const float *)lf, nlog, inv_cap, cap);
            VMX_CHECK_LAUNCH("gemma4(lm_head)");
            goto head_done;
        }
        hipLaunchKernelGGL(g4_bf16_to_f32_kernel, dim3((unsigned)(((int64_t)B * H + 255) / 256)), dim3(256), 0, st, x, xf, (int64_t)B * H);
        for (int b0 = 0; b0 < B; b0 += 8) {
            const int mb = B - b0 < 8 ? B - b0 : 8;
            const size_t row_bytes = (size_t)(H / vmx_ggml_block_elems(c.embed_ggml_type)) * vmx_ggml_type_size(c.embed_ggml_type);
            if (int rc = vmx_kquant_gemv_f32(c.embed_ggml_type, (const char *)m->embed.p + (size_t)v0 * row_bytes, xf + (int64_t)b0 * H,
                                             lf + (int64_t)b0 * VL, VL, H, mb, stream))
                return rc;
        }
        hipLaunchKernelGGL(g4_logits_kernel, dim3((unsigned)((nlog + 255) / 256)), dim3(256), 0, st, logits, (const float *)lf, nlog, inv_cap,
                           cap);
    }
    VMX_CHECK_LAUNCH("gemma4(lm_head)");
head_done:
    if (out_ids) {
        if (!tp) {
            if (int rc = vmx_argmax_bf16(out_ids, logits, B, VL, m->argmax_ws.p, m->argmax_ws.bytes, stream)) return rc;
        } else {   // vocabulary-parallel: best (value, global index) key per rank, max over the ranks
            unsigned long long *keys = (unsigned long long *)m->keys.p;
            if (int rc = vmx_argmax_keys(keys, logits, B, VL, (uint32_t)v0, m->argmax_ws.p, st)) return rc;
            if (int rc = vmx_comm_all_reduce_max_u64(m->comm, keys, B, stream)) return rc;
            if (int rc = vmx_argmax_keys_to_ids(keys, out_ids, B, st)) return rc;
        }
    }
    return VMX_OK;
}
}  // namespace

extern "C" int vmx_gemma4_decode(vmx_gemma4 *m, const int32_t *input_ids, const int32_t *positions, const int32_t *slot_mapping,
                                 const int32_t *block_tables, const int32_t *seq_lens, int batch, int max_blocks_per_seq, int max_seq_len,
                                 int32_t *out_ids, uint16_t *logits_out, void *stream) {
    VMX_REQUIRE(m && input_ids && positions && slot_mapping && block_tables && seq_lens, "gemma4_decode: null pointer");
    const vmx_gemma4_config &c = m->cfg;
    VMX_REQUIRE(batch >= 1 && batch <= c.max_batch, "gemma4_decode: batch %d exceeds max_batch %d", batch, c.max_batch);
    VMX_REQUIRE(max_seq_len >= 1 && max_seq_len <= c.max_seq_len, "gemma4_decode: max_seq_len %d exceeds the model's %d", max_seq_len,
                c.max_seq_len);
    for (const G4Layer &L : m->layers) VMX_REQUIRE(L.ready(), "gemma4_decode: weights not fully loaded");
    VMX_REQUIRE(m->cfg.hidden_size_per_layer_input == 0 || m->ple_model_proj.loaded, "gemma4_decode: per_layer_model_projection not loaded");
    if (int rc = g4_run_layers(m, input_ids, positions, slot_mapping, block_tables, seq_lens, batch, max_blocks_per_seq, max_seq_len, -1,
                               vmx_stream(stream)))
        return rc;
    return g4_run_head(m, (const uint16_t *)m->x.p, batch, out_ids, logits_out, stream);
}

// ModelForward::forward for one sequence (crates/core/src/engine/model_forward.rs:22-64; QuantizedGemma4ForCausalLM::forward,
// models/gemma4_quantized.rs): q_len prompt tokens at positions seqlen_offset .. seqlen_offset + q_len - 1, in passes of 16
// rows (the row limit of the q8_1 MMVQ, IQ_GEMV_MAX_M: the weights are streamed once per pass).  Every pass writes its K/V
// rows to the paged cache and attends causally over the cache (sliding layers inside their window); only the last row of the
// last pass goes through the lm_head.  Pure launch sequence.
extern "C" int vmx_gemma4_prefill(vmx_gemma4 *m, const int32_t *input_ids, int seqlen_offset, const int32_t *slot_mapping,
                                  const int32_t *block_table, int q_len, int32_t *out_id_last, uint16_t *logits_last, void *stream) {
    VMX_REQUIRE(m && input_ids && slot_mapping && block_table, "gemma4_prefill: null pointer");
    const vmx_gemma4_config &c = m->cfg;
    VMX_REQUIRE(q_len >= 1, "gemma4_prefill: q_len %d", q_len);
    VMX_REQUIRE(seqlen_offset >= 0 && seqlen_offset + q_len <= c.max_position && seqlen_offset + q_len <= c.max_seq_len,
                "gemma4_prefill: positions %d..%d exceed max_position %d / max_seq_len %d", seqlen_offset, seqlen_offset + q_len - 1,
                c.max_position, c.max_seq_len);
    for (const G4Layer &L : m->layers) VMX_REQUIRE(L.ready(), "gemma4_prefill: weights not fully loaded");
    VMX_REQUIRE(m->cfg.hidden_size_per_layer_input == 0 || m->ple_model_proj.loaded, "gemma4_prefill: per_layer_model_projection not loaded");
    // The prompt path walks one pass of rows at a time through all layers (16 rows, or max_prefill_tokens when that is larger), so
    // a calibrating first write sees one pass; the reference's first write sees the whole chunk.  Identical scales need the chunk
    // to fit one pass.
    const int pass = c.max_prefill_tokens > 16 ? c.max_prefill_tokens : 16;
    if (c.kv_cache_dtype == VMX_KV_FP8_E4M3 && q_len > pass)
        for (const G4Layer &L : m->layers)
            VMX_REQUIRE(L.kv_shared || L.kv_calibrated,
                        "gemma4_prefill: FP8 KV scales are not calibrated yet; first-write calibration takes a chunk of at most %d tokens "
                        "(one pass; got %d): prefill the first pass separately or load explicit scales", pass, q_len);
    hipStream_t st = vmx_stream(stream);
    int32_t *pos = (int32_t *)m->positions.p;
    int last_rows = 0;
    for (int r0 = 0; r0 < q_len; r0 += pass) {
        const int rows = q_len - r0 < pass ? q_len - r0 : pass;
        hipLaunchKernelGGL(g4_iota_kernel, dim3((rows + 63) / 64), dim3(64), 0, st, pos, seqlen_offset + r0, rows);
        if (int rc = g4_run_layers(m, input_ids + r0, pos, slot_mapping + r0, block_table, nullptr, rows, 0, 0, seqlen_offset + r0, st)) return rc;
        last_rows = rows;
    }
    return g4_run_head(m, (const uint16_t *)m->x.p + (size_t)(last_rows - 1) * c.hidden_size, 1, out_id_last, logits_last, stream);
}

extern "C" int vmx_gemma4_set_comm(vmx_gemma4 *m, vmx_comm *comm) {
    VMX_REQUIRE(m, "gemma4_set_comm: null model");
    VMX_REQUIRE(!comm || (vmx_comm_world(comm) == (m->cfg.tp_size > 1 ? m->cfg.tp_size : 1) && vmx_comm_rank(comm) == m->cfg.tp_rank),
                "gemma4_set_comm: communicator rank/world do not match the model's tp_rank/tp_size");
    m->comm = comm;
    return VMX_OK;
}

// KVScales::reset for every layer (kv_cache/quantization.rs:231-239) with the headroom of the next first-write calibration
extern "C" int vmx_gemma4_kv_scales_reset(vmx_gemma4 *m, float headroom, void *stream) {
    VMX_REQUIRE(m, "gemma4_kv_scales_reset: null pointer");
    VMX_REQUIRE(isfinite(headroom) && headroom > 0.f, "KV scale headroom factor must be a positive finite value, got %g", (double)headroom);
    m->kv_headroom = headroom;
    for (G4Layer &L : m->layers) {
        if (int rc = vmx_kv_scales_init(L.kv_scales.p, 1.0f, 1.0f, 0, stream)) return rc;
        L.kv_calibrated = false;
    }
    return VMX_OK;
}

extern "C" int vmx_gemma4_kv_scales_get(vmx_gemma4 *m, int layer, float *k_scale, float *v_scale, int *calibrated, void *stream) {
    VMX_REQUIRE(m && k_scale && v_scale && calibrated, "gemma4_kv_scales_get: null pointer");
    VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size(), "gemma4_kv_scales_get: layer %d out of range", layer);
    const G4Layer &L = m->layers[layer].kv_shared ? m->layers[m->layers[layer].kv_target] : m->layers[layer];
    uint32_t w[3];
    VMX_CHECK_HIP(hipMemcpyAsync(w, L.kv_scales.p, 12, hipMemcpyDeviceToHost, vmx_stream(stream)));
    VMX_CHECK_HIP(hipStreamSynchronize(vmx_stream(stream)));
    memcpy(k_scale, &w[0], 4);
    memcpy(v_scale, &w[1], 4);
    *calibrated = (int)w[2];
    return VMX_OK;
}

extern "C" int vmx_debug_gemma4_split_always(int on) {
    g_split_always = on != 0;
    return VMX_OK;
}

extern "C" int vmx_gemma4_mmvq_select(vmx_gemma4 *m, int mask) {
    VMX_REQUIRE(m && mask >= 1 && mask <= 15, "gemma4_mmvq_select: mask must be 1..15");
    m->mmvq_mask = mask;
    return VMX_OK;
}

extern "C" int vmx_gemma4_mmvq_info(vmx_gemma4 *m, int batch, int *launches, double *alg_bytes) {
    VMX_REQUIRE(m && launches && alg_bytes, "gemma4_mmvq_info: null pointer");
    double by = 0.0;
    int n = 0;
    const bool multi = batch == 1 && m->cfg.hidden_size <= 4096;   // q|k|v and gate|up go out as one launch each
    const int mask = m->mmvq_mask;
    for (const G4Layer &L : m->layers) {
        for (int s = 0; s < 7; s++) {
            if ((s == VMX_L_V && L.k_eq_v) || ((s == VMX_L_K || s == VMX_L_V) && L.kv_shared)) continue;
            const int kind = s <= VMX_L_V ? 1 : (s == VMX_L_O ? 2 : (s == VMX_L_DOWN ? 8 : 4));
            if (mask & kind) by += L.lin[s].alg_bytes(batch);
        }
        const int nqkv = L.kv_shared ? 1 : (L.k_eq_v ? 2 : 3);
        if (mask & 1) n += multi ? 1 : nqkv;
        if (mask & 2) n += 1;
        if (mask & 4) n += multi ? 1 : 2;
        if (mask & 8) n += 1;
    }
    *launches = n;
    *alg_bytes = by;
    return VMX_OK;
}

extern "C" int vmx_gemma4_mmvq_only(vmx_gemma4 *m, int batch, void *stream) {
    VMX_REQUIRE(m && batch >= 1 && batch <= m->cfg.max_batch, "gemma4_mmvq_only: bad arguments");
    for (const G4Layer &L : m->layers) VMX_REQUIRE(L.ready(), "gemma4_mmvq_only: weights not fully loaded");
    VMX_REQUIRE(m->cfg.hidden_size_per_layer_input == 0 || m->ple_model_proj.loaded, "gemma4_mmvq_only: per_layer_model_projection not loaded");
    hipStream_t st = vmx_stream(stream);
    const vmx_gemma4_config &c = m->cfg;
    float *proj = (float *)m->proj.p, *out = (float *)m->out.p;
    const int H = c.hidden_size, mask = m->mmvq_mask;
    uint16_t *h = (uint16_t *)m->h.p, *h2 = (uint16_t *)m->h2.p;
    // The launches are the ones a decode step makes, in its variants: with one row the q|k|v launch of a layer behind a
    // norm-fused gate|up carries the layer-boundary norms in its prologue, and gate|up carries the post-attention /
    // pre-feedforward norms, the GeGLU and q8_1 (run_layers above).  Buffers hold whatever the last step left: only time is read.
    const G4Layer *prev = nullptr;
    bool prev_norm_in_mlp = false;
    for (const G4Layer &L : m->layers) {
        const int nseg = L.kv_shared ? 1 : (L.k_eq_v ? 2 : 3), I = L.inter;
        const int NQ = c.num_heads * L.hd, NKV = L.kvh * L.hd, ntot = NQ + (nseg - 1) * NKV;
        const IqLin *qkv[3] = {&L.lin[VMX_L_Q], &L.lin[VMX_L_K], &L.lin[VMX_L_V]};
        const int qoffs[3] = {0, NQ, NQ + NKV};
        if (mask & 1) {
            bool done = false;
            if (batch == 1 && prev && prev_norm_in_mlp && !prev->ple) {
                VmxMmvqSeg pk[3];
                for (int i = 0; i < nseg; i++)
                    pk[i] = VmxMmvqSeg{L.lin[VMX_L_Q + i].packed.p, L.lin[VMX_L_Q + i].ggml_type, L.lin[VMX_L_Q + i].out_features, qoffs[i]};
                if (vmx_iq_packed_norm_multi_ok(pk, nseg, H, batch)) {
                    if (int rc = vmx_iq_mmvq_packed_norm_multi(pk, nseg, out, h2, h, (const uint16_t *)prev->post_ffn_ln.p,
                                                               (const uint16_t *)prev->scalar.p, (const uint16_t *)L.input_ln.p, c.rms_norm_eps,
                                                               proj, H, st))
                        return rc;
                    done = true;
                }
            }
            if (!done)
                if (int rc = mmvq_group(m, qkv, qoffs, nseg, proj, ntot, batch, st)) return rc;
        }
        if (mask & 2)
            if (int rc = mmvq(m, L.lin[VMX_L_O], out, H, batch, st)) return rc;
        const IqLin *gu[2] = {&L.lin[VMX_L_GATE], &L.lin[VMX_L_UP]};
        const int goffs[2] = {0, I};
        const VmxMmvqSeg gate_pk = {L.lin[VMX_L_GATE].packed.p, L.lin[VMX_L_GATE].ggml_type, L.lin[VMX_L_GATE].out_features, 0};
        const VmxMmvqSeg up_pk = {L.lin[VMX_L_UP].packed.p, L.lin[VMX_L_UP].ggml_type, L.lin[VMX_L_UP].out_features, 0};
        const bool norm_in_mlp = batch == 1 && !L.ple && vmx_iq_packed_norm_geglu_ok(&gate_pk, &up_pk, H, batch);
        if (mask & 4) {
            int8_t *aq_b = (int8_t *)m->q8b.p;
            float *ad_b = (float *)((char *)m->q8b.p + (((size_t)I + 255) & ~(size_t)255));
            if (norm_in_mlp) {
                if (int rc = vmx_iq_mmvq_packed_norm_geglu(&gate_pk, &up_pk, out, h, h2, (const uint16_t *)L.post_attn_ln.p,
                                                           (const uint16_t *)L.pre_ffn_ln.p, c.rms_norm_eps, (uint16_t *)m->act.p, aq_b, ad_b, H, st))
                    return rc;
            } else if (batch == 1 && vmx_iq_packed_geglu_ok(&gate_pk, &up_pk, H, batch)) {
                if (int rc = vmx_iq_mmvq_packed_geglu(&gate_pk, &up_pk, (const int8_t *)m->q8.p,
                                                      (const float *)((const char *)m->q8.p + (((size_t)H + 255) & ~(size_t)255)),
                                                      (uint16_t *)m->act.p, aq_b, ad_b, H, st))
                    return rc;
            } else if (int rc = mmvq_group(m, gu, goffs, 2, proj, 2 * I, batch, st)) return rc;
        }
        if (mask & 8)
            if (int rc = mmvq(m, L.lin[VMX_L_DOWN], out, H, batch, st)) return rc;
        prev = &L;
        prev_norm_in_mlp = norm_in_mlp;
    }
    return VMX_OK;
}

// Parity aid: copy a layer's K (which = 0) or V (1) cache to `buf` (write = 0) or from it (write = 1)
extern "C" int vmx_gemma4_debug_kv(vmx_gemma4 *m, int layer, int which, int write, void *buf, size_t bytes, void *stream) {
    VMX_REQUIRE(m && buf, "gemma4_debug_kv: null pointer");
    VMX_REQUIRE(layer >= 0 && layer < (int)m->layers.size() && (which == 0 || which == 1), "gemma4_debug_kv: layer %d / which %d", layer, which);
    const G4Layer &L = m->layers[layer];
    VMX_REQUIRE(!L.kv_shared, "gemma4_debug_kv: layer %d owns no cache (it reads layer %d's)", layer, L.kv_target);
    Buf &c = which == 0 ? m->layers[layer].k_cache : m->layers[layer].v_cache;
    VMX_REQUIRE(bytes == c.bytes, "gemma4_debug_kv: the cache holds %zu bytes, got %zu", c.bytes, bytes);
    VMX_CHECK_HIP(write ? hipMemcpyAsync(c.p, buf, bytes, hipMemcpyDefault, vmx_stream(stream))
                        : hipMemcpyAsync(buf, c.p, bytes, hipMemcpyDefault, vmx_stream(stream)));
    return VMX_OK;
}

extern "C" int vmx_gemma4_debug_read(vmx_gemma4 *m, int which, void *dst, size_t bytes, void *stream) {
    VMX_REQUIRE(m && dst, "gemma4_debug_read: null pointer");
    Buf *src = which == 0 ? &m->h : which == 1 ? &m->x : nullptr;
    VMX_REQUIRE(src && bytes <= src->bytes, "gemma4_debug_read: bad request");
    VMX_CHECK_HIP(hipMemcpyAsync(dst, src->p, bytes, hipMemcpyDefault, vmx_stream(stream)));
    return VMX_OK;
}

// Parity aid: 0 = gate|up on the matrix cores writes its f32 projections and the stand-alone GeGLU kernel follows (the fused
// epilogue must give the same bits), 1 = the default.
extern "C" int vmx_debug_gemma4_mma_geglu(int on) {
    g_mma_geglu = on != 0;
    return VMX_OK;
}

// Parity / measurement aid: rows from which the Gemma executor's I-quant linears run on the int8 matrix cores (default 3;
// 17 = never, i.e. the per-row kernels for every batch size).
extern "C" int vmx_debug_iq_mma_min_rows(int rows) {
    VMX_REQUIRE(rows >= 2 && rows <= 17, "debug_iq_mma_min_rows: %d (2 .. 17)", rows);
    g_mma_min_rows = rows;
    return VMX_OK;
}

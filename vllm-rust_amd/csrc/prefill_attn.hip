// Prompt attention over the paged cache on 32 x 32 matrix-core tiles (64- and 128-wide heads).
//
// Same operator as prefill_attn_kernel (attention_kernels.hip): causal, optionally windowed attention of q_len new tokens at
// positions offset .. offset + q_len - 1 over the cache rows of the sequence (AttentionBackend::prefill_attention,
// crates/core/src/layers/attention/backend.rs:60-71; semantics of naive.rs:38-83).  That kernel walks 16-token KV tiles with a
// workgroup barrier and scalar LDS stores per tile and keeps the matrix cores ~1.5 % busy; this one is built for them:
//
//   workgroup = 4 or 8 waves; a wave owns 32 query rows of ONE query head (HPW heads of one KV head x NW / HPW blocks of
//   32 query rows per workgroup: every wave of the workgroup shares every KV tile it fetches); 236 registers, two waves per SIMD;
//   KV tiles of 64 tokens are fetched one tile ahead into registers (16-byte pieces, whole cache rows per wave access) and
//   written to a two-stage LDS ring: one barrier per tile;
//   S^T = K Q^T   v_mfma_f32_32x32x16: A = K rows from LDS (ds_read_b128; 16-byte chunk c of row r sits at chunk
//                 c ^ f(r), so the 16 lanes of an LDS access group hit 16 different bank quads), B = the wave's Q rows, held
//                 in registers for the whole kernel.  The accumulator gives lane (q = lane & 31, hi = lane >> 5) the
//                 scores of ITS query row for tokens (r & 3) + 8 (r >> 2) + 4 hi: the softmax needs one exchange with
//                 lane ^ 32 per tile and no other cross-lane traffic;
//   O^T += V^T P^T   the same instruction: B = P^T straight from the score registers (the k slots of a step are assigned to
//                 the tokens a lane already holds, so P never moves between lanes), A = V^T read from the row-major V tile
//                 with the transposing LDS read (ds_read_b64_tr_b16; V rows are padded by 64 bytes so the four rows of a
//                 block fall into different bank groups).  O^T keeps a query row in one accumulator column: the
//                 online-softmax rescale is a per-lane multiply.
// Rounding points are those of the 16-row kernel: P is rounded to the activation dtype before the PV product, the output
// once at the end; sums are f32.  Summation order differs (64-token tiles), within the attention tolerance of the tests.
#include "vmx_kernels.h"

#include <float.h>
#include <type_traits>

namespace {

typedef __bf16 pa_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pa_f16x8 __attribute__((ext_vector_type(8)));
typedef float pa_f32x16 __attribute__((ext_vector_type(16)));
typedef short pa_s16x4 __attribute__((ext_vector_type(4)));
typedef short pa_s16x8 __attribute__((ext_vector_type(8)));

constexpr int PA_KT = 64;   // tokens per KV tile

template <int HD> constexpr int pa_krow() { return HD * 2; }         // bytes per K row in LDS (swizzled chunks, no padding)
template <int HD> constexpr int pa_vrow() { return HD * 2 + 64; }    // bytes per V row in LDS: +64 B puts 4 consecutive rows on 4 bank groups
template <int HD> constexpr int pa_stage() { return PA_KT * (pa_krow<HD>() + pa_vrow<HD>()); }

// chunk swizzle of K row r: distinct over the 16 rows of every ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31})
template <int HD> __device__ __forceinline__ int pa_swz(int r) { return HD == 128 ? (r & 15) : ((r >> 1) & 7); }

// KV: 0 = bf16 cache and queries, 1 = FP8 e4m3 cache bytes (widened to bf16 on the way into LDS: exact) with bf16 queries,
// 3 = fp16 cache and queries (fp16 forms of the matrix instructions, P rounded to fp16)
template <int HD, int HPW, int KV, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2))) void prefill_attn_tiled_kernel(uint16_t *__restrict__ out, const uint16_t *__restrict__ q,
                                                                  const void *__restrict__ k_cache, const void *__restrict__ v_cache,
                                                                  const int32_t *__restrict__ block_table, int q_len, int offset, float scale,
                                                                  int num_heads, int num_kv_heads, int block_size, int window,
                                                                  const float *__restrict__ k_scale_ptr, const float *__restrict__ v_scale_ptr,
                                                                  int head_groups) {
    constexpr bool F16 = KV == 3;
    constexpr int TT = NW / HPW;                      // 32-row query blocks per workgroup
    constexpr int CH = HD / 8;                        // 16-byte chunks per row
    constexpr int NT = 64 * NW;                       // threads
    constexpr int NLD = PA_KT * CH / NT;              // chunks per thread per tile and tensor
    constexpr int KS = HD / 16;                       // k-steps of S^T
    constexpr int DT = HD / 32;                       // 32-dim tiles of O^T
    constexpr int KROW = pa_krow<HD>(), VROW = pa_vrow<HD>(), STAGE = pa_stage<HD>();
    extern __shared__ __align__(16) unsigned char pa_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    const int hh = wave % HPW, tt = wave / HPW;
    const int nq = num_heads / num_kv_heads;
    // one-dimensional grid, (KV head, head group) fastest: workgroups are dealt to the 8 XCDs round-robin, so with 8 (or a
    // multiple of 8) KV heads every XCD walks the K/V rows of its own heads and its L2 holds them
    const int ngrp = num_kv_heads * head_groups;
    const int by = blockIdx.x % ngrp, bx = blockIdx.x / ngrp;
    const int kvh = by / head_groups, hgrp = by % head_groups;
    const bool head_ok = hgrp * HPW + hh < nq;        // a wave past the last head of the KV head only helps staging
    const int head = kvh * nq + min(hgrp * HPW + hh, nq - 1);
    const int nblk = (q_len + 32 * TT - 1) / (32 * TT);
    const int qb = nblk - 1 - bx;                     // longest contexts first
    const int wg_q0 = qb * 32 * TT, wg_q1 = min(q_len, wg_q0 + 32 * TT);
    const int q0 = wg_q0 + tt * 32;                   // first query row of this wave
    const int qi = q0 + ql;                           // this lane's query row
    const bool wave_live = head_ok && q0 < q_len;
    const int qpos = offset + min(qi, q_len - 1);
    const int qpos_min = offset + min(q0, q_len - 1), qpos_max = offset + min(q0 + 31, q_len - 1);
    const float ksc = (KV == 1 && k_scale_ptr) ? *k_scale_ptr : 1.0f, vsc = (KV == 1 && v_scale_ptr) ? *v_scale_ptr : 1.0f;
    const float c2 = scale * ksc * 1.4426950408889634f;   // scores go through exp2

    // Q^T fragments: lane (column = query row, hi) holds dims 16 c + 8 hi .. + 8 of k-step c
    uint4 qf[KS];
    {
        const uint16_t *qrow = q + ((int64_t)min(qi, q_len - 1) * num_heads + head) * HD + 8 * hi;
#pragma unroll
        for (int c = 0; c < KS; c++) qf[c] = *reinterpret_cast<const uint4 *>(qrow + 16 * c);
    }

    const int hi_tok = offset + wg_q1;                                       // tokens [lo, hi_tok) are needed by the workgroup
    int lo = window > 0 ? max(0, offset + wg_q0 - window + 1) : 0;
    lo &= ~(PA_KT - 1);
    const int ntiles = (hi_tok - lo + PA_KT - 1) / PA_KT;
    const int64_t tok_stride = (int64_t)num_kv_heads * HD;
    const int bs_shift = (block_size & (block_size - 1)) == 0 ? __ffs(block_size) - 1 : -1;   // power-of-two blocks: no division per row

    // staging registers: chunk id = tid + 256 j -> (row, chunk); a wave access covers whole cache rows
    typedef typename std::conditional<KV == 1, uint2, uint4>::type raw_t;
    raw_t kst[NLD], vst[NLD];
    auto issue = [&](int tile) {
        const int t0 = lo + tile * PA_KT;
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int id = tid + NT * j, row = id / CH, ch = id % CH;
            const int t = min(t0 + row, hi_tok - 1);   // rows past the end repeat the last valid row (their weights are 0)
            const int bi = bs_shift >= 0 ? t >> bs_shift : t / block_size;
            const int blk = block_table[bi];
            const int64_t base = ((int64_t)blk * block_size + (t - bi * block_size)) * tok_stride + (int64_t)kvh * HD + ch * 8;
            if constexpr (KV == 1) {
                kst[j] = *reinterpret_cast<const uint2 *>((const uint8_t *)k_cache + base);
                vst[j] = *reinterpret_cast<const uint2 *>((const uint8_t *)v_cache + base);
            } else {
                kst[j] = *reinterpret_cast<const uint4 *>((const uint16_t *)k_cache + base);
                vst[j] = *reinterpret_cast<const uint4 *>((const uint16_t *)v_cache + base);
            }
        }
    };
    auto widen = [](const raw_t &r) -> uint4 {
        if constexpr (KV == 1) {   // e4m3 -> f32 -> bf16 by truncation (exact)
            uint32_t w[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t src = p < 2 ? r.x : r.y;
                const float a = p & 1 ? __builtin_amdgcn_cvt_f32_fp8(src, 2) : __builtin_amdgcn_cvt_f32_fp8(src, 0);
                const float b = p & 1 ? __builtin_amdgcn_cvt_f32_fp8(src, 3) : __builtin_amdgcn_cvt_f32_fp8(src, 1);
                w[p] = (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xFFFF0000u);
            }
            return make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            return r;
        }
    };
    auto commit = [&](int buf) {
        unsigned char *kb = pa_smem + buf * STAGE, *vb = kb + PA_KT * KROW;
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int id = tid + NT * j, row = id / CH, ch = id % CH;
            *reinterpret_cast<uint4 *>(kb + row * KROW + ((ch ^ pa_swz<HD>(row)) << 4)) = widen(kst[j]);
            *reinterpret_cast<uint4 *>(vb + row * VROW + (ch << 4)) = widen(vst[j]);
        }
    };

    float m_run = -1e30f, l_run = 0.f;
    pa_f32x16 acc[DT];
#pragma unroll
    for (int d = 0; d < DT; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[d][r] = 0.f;

    // per-lane LDS offsets that do not depend on the tile
    const int krow_off = ql * KROW;                                              // + 32 u KROW; chunk (2 c + hi) ^ swz(row)
    const int li = lane & 15, g16 = (lane >> 4) & 1;
    const int vtr_off = (4 * hi + (li >> 2)) * VROW + (16 * g16 + 4 * (li & 3)) * 2;   // + (32 u + 16 ks + 8 h) VROW + 64 dt

    issue(0);
    commit(0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; tile++) {
        const int t0 = lo + tile * PA_KT;
        const bool more = tile + 1 < ntiles;
        if (more) issue(tile + 1);
        // does this tile intersect the wave's causal / window range at all (wave-uniform)
        const bool need = wave_live && t0 <= qpos_max && (window <= 0 || t0 + PA_KT - 1 > qpos_min - window);
        if (__builtin_amdgcn_readfirstlane(need ? 1 : 0)) {
            const unsigned char *kb = pa_smem + (tile & 1) * STAGE, *vb = kb + PA_KT * KROW;
            pa_f32x16 s[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
#pragma unroll
                for (int r = 0; r < 16; r++) s[u][r] = 0.f;
                const int row = 32 * u + ql;
                const unsigned char *kr = kb + 32 * u * KROW + krow_off;
                const int sw = pa_swz<HD>(row);
#pragma unroll
                for (int c = 0; c < KS; c++) {
                    const uint4 ka = *reinterpret_cast<const uint4 *>(kr + (((2 * c + hi) ^ sw) << 4));
                    if constexpr (F16)
                        s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pa_f16x8, ka), __builtin_bit_cast(pa_f16x8, qf[c]), s[u], 0, 0, 0);
                    else
                        s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pa_bf16x8, ka), __builtin_bit_cast(pa_bf16x8, qf[c]), s[u], 0, 0, 0);
                }
            }
            // online softmax of this lane's query row over its 32 tokens of the tile (the other 32 are in lane ^ 32).  Scores
            // stay unscaled in the accumulators: p = exp2(s * c2 - m) is one multiply-add and one v_exp_f32 per score.
            const bool full = t0 + PA_KT - 1 <= qpos_min && (window <= 0 || t0 > qpos_max - window);   // wave-uniform: no mask needed
            float tmax = -1e30f;
            if (!full) {
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int tpos = t0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const bool ok = tpos <= qpos && (window <= 0 || tpos > qpos - window);
                        s[u][r] = ok ? s[u][r] : -FLT_MAX;   // exp2(-FLT_MAX * c2 - m) = 0 for any finite m >= -1e30
                    }
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int r = 0; r < 16; r++) tmax = fmaxf(tmax, s[u][r]);
            tmax = fmaxf(tmax * c2, -1e30f);   // scaled-log2 domain; a row without a visible token stays at the floor
            tmax = fmaxf(tmax, vmx_lane_xor<32>(tmax));
            // The running maximum follows the tile maxima lazily: it moves (and the accumulators are rescaled) only when a row's
            // maximum has grown by more than 2^8 since the last move -- weights stay below 2^8, sums and products are f32, and
            // the rounding of P is relative, so nothing is lost; in the steady state of a long context no tile rescales.  The
            // decision is taken for the whole wave so that the accumulators (one query row per lane pair) move together.
            if (__builtin_amdgcn_readfirstlane(__any(tmax > m_run + 8.0f) ? 1 : 0)) {
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < DT; d++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[d][r] *= alpha;
            }
            float psum = 0.f;
            pa_s16x8 pf[2][2];   // P^T fragments: [half u][k-step ks], k slot j of the step = score register 8 ks + j
            const float nm = -m_run;
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[u][r], c2, nm));
                    psum += p;
                    pf[u][r >> 3][r & 7] = (short)vmx_st16<F16>(p);
                }
            l_run += psum;
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const unsigned char *vr = vb + (32 * u + 16 * ks) * VROW + vtr_off;
#pragma unroll
                    for (int d = 0; d < DT; d++) {
                        const pa_s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (pa_s16x4 __attribute__((address_space(3))) *)(vr + 64 * d));
                        const pa_s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (pa_s16x4 __attribute__((address_space(3))) *)(vr + 8 * VROW + 64 * d));
                        const pa_s16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        if constexpr (F16)
                            acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pa_f16x8, va), __builtin_bit_cast(pa_f16x8, pf[u][ks]), acc[d], 0, 0, 0);
                        else
                            acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pa_bf16x8, va), __builtin_bit_cast(pa_bf16x8, pf[u][ks]), acc[d], 0, 0, 0);
                    }
                }
        }
        if (more) commit((tile + 1) & 1);
        __syncthreads();
    }
    const float l = l_run + vmx_lane_xor<32>(l_run);
    if (wave_live && qi < q_len) {
        const float inv = vsc / l;
        uint16_t *o = out + ((int64_t)qi * num_heads + head) * HD + 4 * hi;
#pragma unroll
        for (int d = 0; d < DT; d++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {   // acc[d][4 rg + i] = O[query row][dim 32 d + 8 rg + 4 hi + i]
                const uint32_t w0 = (uint32_t)vmx_st16<F16>(acc[d][4 * rg] * inv) | ((uint32_t)vmx_st16<F16>(acc[d][4 * rg + 1] * inv) << 16);
                const uint32_t w1 = (uint32_t)vmx_st16<F16>(acc[d][4 * rg + 2] * inv) | ((uint32_t)vmx_st16<F16>(acc[d][4 * rg + 3] * inv) << 16);
                *reinterpret_cast<uint2 *>(o + 32 * d + 8 * rg) = make_uint2(w0, w1);
            }
    }
}

template <int HD, int HPW, int KV, int NW>
int launch_tiled(hipStream_t st, uint16_t *out, const uint16_t *q, const void *kc, const void *vc, const int32_t *bt, int q_len, int offset,
                 float scale, int H, int KVH, int bs, int window, const float *ks, const float *vs) {
    constexpr int TT = NW / HPW;
    const int nq = H / KVH, groups = (nq + HPW - 1) / HPW;
    const dim3 grid(((q_len + 32 * TT - 1) / (32 * TT)) * KVH * groups);
    const size_t lds = 2 * (size_t)pa_stage<HD>();
    if (lds > 64 * 1024)
        VMX_CHECK_HIP(hipFuncSetAttribute((const void *)prefill_attn_tiled_kernel<HD, HPW, KV, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((prefill_attn_tiled_kernel<HD, HPW, KV, NW>), grid, dim3(64 * NW), lds, st, out, q, kc, vc, bt, q_len, offset, scale, H, KVH,
                       bs, window, ks, vs, groups);
    VMX_CHECK_LAUNCH("prefill_attention(32x32 tiles)");
    return VMX_OK;
}

}  // namespace

// kv: 0 bf16, 1 FP8 e4m3, 3 fp16 (queries and output fp16).  VMX_EUNSUPPORTED without an error message when the geometry is
// not covered here (the caller then takes the 16-row kernel).
int vmx_prefill_attention_tiled(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache, const int32_t *block_table,
                                int q_len, int seqlen_offset, float scale, int num_heads, int num_kv_heads, int head_dim, int block_size,
                                int window, int kv, const float *k_scale, const float *v_scale, hipStream_t st) {
    static const bool off = vmx_tune("VMX_NO_TILED_PREFILL", 0) != 0;
    if (off || (head_dim != 64 && head_dim != 128) || (kv != 0 && kv != 1 && kv != 3)) return VMX_EUNSUPPORTED;
    const int nq = num_heads / num_kv_heads;
    const int hpw = nq >= 4 ? 4 : (nq >= 2 ? 2 : 1);
#define PA_ARGS st, out, q, k_cache, v_cache, block_table, q_len, seqlen_offset, scale, num_heads, num_kv_heads, block_size, window, k_scale, v_scale
    // eight waves (twice the query rows per KV tile fetched) once the prompt gives every compute unit such a workgroup
    static const int nw_env = vmx_tune("VMX_PREFILL_NW", 0);
    const int64_t wg8 = (int64_t)((q_len + 32 * (8 / hpw) - 1) / (32 * (8 / hpw))) * num_kv_heads * ((nq + hpw - 1) / hpw);
    const bool eight = nw_env ? nw_env == 8 : wg8 >= 2 * vmx_num_cus();
#define PA_NW(HD, HPW, KVT) return eight ? launch_tiled<HD, HPW, KVT, 8>(PA_ARGS) : launch_tiled<HD, HPW, KVT, 4>(PA_ARGS)
#define PA_KV(HD, HPW)                    \
    if (kv == 3) { PA_NW(HD, HPW, 3); }   \
    if (kv == 1) { PA_NW(HD, HPW, 1); }   \
    PA_NW(HD, HPW, 0)
#define PA_HPW(HD)               \
    if (hpw == 4) { PA_KV(HD, 4); } \
    if (hpw == 2) { PA_KV(HD, 2); } \
    PA_KV(HD, 1)
    if (head_dim == 64) { PA_HPW(64); }
    PA_HPW(128);
#undef PA_HPW
#undef PA_KV
#undef PA_NW
#undef PA_ARGS
}

// Mixture-of-experts operators for gfx950: router top-k softmax, token->expert alignment, and a
// device-side quantized MoE decode layer over int4 (AWQ/GPTQ) experts.
// Semantics: crates/core/kernels/topk_softmax.cu:63-311 (softmax with expf, max-subtracted; k rounds
// of argmax, ties -> smaller index; optional renormalisation by the selected sum),
// crates/core/kernels/fused_moe_align.cu:34-278 (moe_align_block_size), and the reference's
// quantized Mixtral layer, which has no fused kernel: a host loop over tokens and their top-k experts
// (crates/core/src/models/mixtral_quantized.rs:92-99,138-195).  Here the whole layer stays on the
// device: expert ids are read by a grouped int4 GEMV (blockIdx.y = (token, slot) pair), so there is no
// host read of the router output per token.
#include "vmx_kernels.h"
#include <float.h>

namespace {

__device__ __forceinline__ float rb(float v) { return vmx_bf16_to_f32(vmx_f32_to_bf16(v)); }

// One wave per token; lane l holds experts l, l+64, ... (up to 4 per lane: E <= 256).
template <bool BF16_IN>
__global__ __launch_bounds__(256) void topk_softmax_kernel(const void *__restrict__ logits, float *__restrict__ w_out,
                                                            int32_t *__restrict__ id_out, int tokens, int E, int k, int renorm,
                                                            int round_probs_bf16) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= tokens) return;
    float v[4];
    float mx = -FLT_MAX;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int e = lane + 64 * j;
        v[j] = -FLT_MAX;
        if (e < E) v[j] = BF16_IN ? vmx_bf16_to_f32(((const uint16_t *)logits)[(int64_t)t * E + e]) : ((const float *)logits)[(int64_t)t * E + e];
        mx = fmaxf(mx, v[j]);
    }
    mx = vmx_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        v[j] = (lane + 64 * j < E) ? expf(v[j] - mx) : 0.f;
        sum += v[j];
    }
    sum = vmx_wave_sum(sum);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        v[j] = (lane + 64 * j < E) ? v[j] / sum : -FLT_MAX;
        if (round_probs_bf16 && lane + 64 * j < E) v[j] = rb(v[j]);
    }
    float selected = 0.f;
    for (int ki = 0; ki < k; ki++) {
        // key: larger value first, then smaller index
        float bv = -FLT_MAX;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int e = lane + 64 * j;
            if (e < E && (v[j] > bv)) { bv = v[j]; bi = e; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = vmx_lane_xor_m(bv, off);
            const int oi = vmx_lane_xor_m(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { w_out[(int64_t)t * k + ki] = bv; id_out[(int64_t)t * k + ki] = bi; }
        selected += bv;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (lane + 64 * j == bi) v[j] = -FLT_MAX;
    }
    if (renorm && lane == 0) {
        const float f = selected > 0.f ? 1.0f / selected : 1.0f;
        for (int ki = 0; ki < k; ki++) w_out[(int64_t)t * k + ki] *= f;
    }
}

// Single workgroup.  sorted_token_ids: per expert, the flattened (token*k + slot) indices routed to it in
// ascending order, each expert's run padded to a multiple of block_size with `numel`; expert_ids: the
// expert of each block; total_tokens_post_pad[0] = padded length.
__global__ __launch_bounds__(256) void moe_align_kernel(const int32_t *__restrict__ topk_ids, int32_t *__restrict__ sorted,
                                                         int32_t *__restrict__ expert_ids, int32_t *__restrict__ total_post_pad,
                                                         int numel, int E, int block_size, int sorted_capacity) {
    extern __shared__ int32_t sh[];  // counts[E] | offsets[E+1]
    int32_t *counts = sh, *offs = sh + E;
    for (int e = threadIdx.x; e < E; e += 256) counts[e] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < numel; i += 256) {
        const int e = topk_ids[i];
        if (e >= 0 && e < E) atomicAdd(&counts[e], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int e = 0; e < E; e++) {
            offs[e] = acc;
            acc += (counts[e] + block_size - 1) / block_size * block_size;
        }
        offs[E] = acc;
        total_post_pad[0] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < sorted_capacity; i += 256) sorted[i] = numel;
    for (int b = threadIdx.x; b < offs[E] / block_size; b += 256) {
        int e = 0;
        while (e + 1 < E && offs[e + 1] <= b * block_size) e++;
        expert_ids[b] = e;
    }
    __syncthreads();
    // stable placement: thread e walks the ids in order (numel is small on the decode path)
    for (int e = threadIdx.x; e < E; e += 256) {
        int pos = offs[e];
        for (int i = 0; i < numel; i++)
            if (topk_ids[i] == e) sorted[pos++] = i;
    }
}

// The same result for up to 256 experts in O(numel / 1024) steps of one 1024-thread workgroup (the form above walks all ids once
// per expert): histogram with LDS atomics, serial prefix over the experts, then the ids in rounds of 1024 -- inside a wave the
// rank of an id among the earlier lanes with the same expert comes from ballots (one per distinct expert present in the wave),
// across the 16 waves from per-(wave, expert) counts in LDS -- so that an expert's indices still ascend.  One workgroup on
// purpose: the sort is a few microseconds in front of GEMMs that stream the experts, and a multi-workgroup form would cost
// two more dependent launches.
__global__ __launch_bounds__(1024) void moe_align_fast_kernel(const int32_t *__restrict__ topk_ids, int32_t *__restrict__ sorted,
                                                               int32_t *__restrict__ expert_ids, int32_t *__restrict__ total_post_pad,
                                                               int numel, int E, int block_size, int sorted_capacity) {
    extern __shared__ int32_t sh[];   // counts[E] | offs[E + 1] | cursor[E] | wave_cnt[16][E]
    int32_t *counts = sh, *offs = sh + E, *cursor = sh + 2 * E + 1, *wave_cnt = sh + 3 * E + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < E; e += 1024) counts[e] = 0;
    for (int i = tid; i < 16 * E; i += 1024) wave_cnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < numel; i += 1024) {
        const int e = topk_ids[i];
        if (e >= 0 && e < E) atomicAdd(&counts[e], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int e = 0; e < E; e++) {
            offs[e] = acc;
            cursor[e] = acc;
            acc += (counts[e] + block_size - 1) / block_size * block_size;
        }
        offs[E] = acc;
        total_post_pad[0] = acc;
    }
    __syncthreads();
    for (int i = tid; i < sorted_capacity; i += 1024) sorted[i] = numel;
    for (int b = tid; b < offs[E] / block_size; b += 1024) {
        int lo = 0, hi = E - 1;   // the last expert whose run starts at or before this block (it is the one with rows there:
        while (lo < hi) {         // an empty expert shares its start with its successor, which is then the later one)
            const int mid = (lo + hi + 1) >> 1;
            if (offs[mid] <= b * block_size) lo = mid;
            else hi = mid - 1;
        }
        expert_ids[b] = lo;
    }
    __syncthreads();   // the padding fill is ordered before the placement below
    for (int base = 0; base < numel; base += 1024) {
        const int i = base + tid;
        const int e = i < numel ? topk_ids[i] : -1;
        const bool valid = e >= 0 && e < E;
        int rank = 0;
        unsigned long long remaining = __ballot(valid);
        while (remaining) {   // wave-uniform: one pass per distinct expert among this wave's ids
            const int leader = __ffsll((long long)remaining) - 1;
            const int le = __shfl(e, leader, 64);
            const unsigned long long mask = __ballot(valid && e == le);
            if (valid && e == le) rank = __popcll(mask & ((1ull << lane) - 1ull));
            if (lane == leader) wave_cnt[wave * E + le] = __popcll(mask);
            remaining &= ~mask;
        }
        __syncthreads();
        if (valid) {
            int before = 0;
            for (int w = 0; w < wave; w++) before += wave_cnt[w * E + e];
            sorted[cursor[e] + before + rank] = i;
        }
        __syncthreads();
        for (int ee = tid; ee < E; ee += 1024) {
            int tot = 0;
            for (int w = 0; w < 16; w++) {
                tot += wave_cnt[w * E + ee];
                wave_cnt[w * E + ee] = 0;
            }
            cursor[ee] += tot;
        }
        __syncthreads();
    }
}

// out[t, :] = sum over slots j (in slot order) of bf16( bf16(expert_out[t*k+j, :]) * bf16(w[t,j]) ), each partial
// sum rounded to bf16: the reference's `token_output + expert_out * weight` chain on bf16 tensors.
// Pairs whose expert has a null table entry (held by another rank) contribute nothing: with expert parallelism every rank
// sums its own experts' terms and the ranks' sums are all-reduced.
__global__ __launch_bounds__(256) void moe_combine_kernel(uint16_t *__restrict__ out, const float *__restrict__ partial, int splits,
                                                           int pairs, const float *__restrict__ w, int k, int hidden,
                                                           const void *const *__restrict__ w2_table, const int32_t *__restrict__ ids,
                                                           int num_experts, uint16_t *__restrict__ h_resid, float *__restrict__ ssq) {
    // h_resid != NULL: the layer's residual add happens here as well, h = bf16(h + out), and ssq[t][hidden / 16] receives the
    // sums of squares of the 16-column tiles of the new h (what the norm prologue of the next layer's q|k|v GEMM reads);
    // the launch then has one thread per column (hidden % 256 == 0), so a tile is 16 consecutive lanes.
    const int t = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hidden; i += gridDim.x * 256) {
        // The launch is a chain of memory round trips, so the loads of two experts and eight split-K slices go out together
        // (branch-free: a slot past the end re-reads the last valid one and is not added); sums in slice order, experts in j order.
        const uint16_t hold = h_resid ? h_resid[(int64_t)t * hidden + i] : (uint16_t)0;
        float acc = 0.f;
        for (int j0 = 0; j0 < k; j0 += 2) {
            int p[2];
            bool live[2];
            float wv[2], o[2] = {0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; u++) {
                p[u] = t * k + min(j0 + u, k - 1);
                int e = ids[p[u]];
                e = e < 0 ? 0 : (e >= num_experts ? num_experts - 1 : e);
                live[u] = j0 + u < k && w2_table[e] != nullptr;
                wv[u] = w[p[u]];
            }
            for (int s0 = 0; s0 < splits; s0 += 8) {
                float v[2][8];
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int q = 0; q < 8; q++) v[u][q] = partial[((int64_t)min(s0 + q, splits - 1) * pairs + p[u]) * hidden + i];
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int q = 0; q < 8; q++) o[u] += s0 + q < splits ? v[u][q] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (live[u]) acc = rb(acc + rb(rb(o[u]) * rb(wv[u])));
        }
        out[(int64_t)t * hidden + i] = vmx_f32_to_bf16(acc);
        if (h_resid) {
            const uint16_t hn = vmx_f32_to_bf16(vmx_bf16_to_f32(hold) + acc);
            h_resid[(int64_t)t * hidden + i] = hn;
            float sq = vmx_bf16_to_f32(hn);
            sq *= sq;
            sq = vmx_row16_sum(sq);
            if ((i & 15) == 0) ssq[(int64_t)t * (hidden >> 4) + (i >> 4)] = sq;
        }
    }
}

// act[p, i] = bf16(silu(bf16(gate)) * bf16(up)) from partial[splits][pairs][2*inter]
__global__ __launch_bounds__(256) void moe_act_kernel(const float *__restrict__ partial, int splits, int pairs, uint16_t *__restrict__ act,
                                                       int inter) {
    const int64_t total = (int64_t)pairs * inter, slice = (int64_t)pairs * 2 * inter;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i / inter, c = i % inter;
        float g = 0.f, u = 0.f;
        for (int s = 0; s < splits; s++) {
            g += partial[s * slice + p * 2 * inter + c];
            u += partial[s * slice + p * 2 * inter + inter + c];
        }
        g = rb(g);
        act[i] = vmx_f32_to_bf16((g / (1.0f + expf(-g))) * rb(u));
    }
}

}  // namespace

extern "C" int vmx_topk_softmax(const void *logits, int logits_dtype, float *topk_weights, int32_t *topk_ids, int tokens,
                                int num_experts, int k, int renormalize, int round_probs_bf16, void *stream) {
    VMX_REQUIRE(logits && topk_weights && topk_ids, "topk_softmax: null pointer");
    VMX_REQUIRE(logits_dtype == VMX_F32 || logits_dtype == VMX_BF16, "topk_softmax: logits dtype must be f32 or bf16");
    VMX_REQUIRE(num_experts >= 1 && num_experts <= 256, "topk_softmax: num_experts (%d) must be in 1..256", num_experts);
    VMX_REQUIRE(k >= 1 && k <= num_experts, "topk_softmax: k (%d) must be in 1..num_experts", k);
    if (tokens <= 0) return VMX_OK;
    const dim3 grid((tokens + 3) / 4), block(256);
    if (logits_dtype == VMX_BF16)
        hipLaunchKernelGGL(topk_softmax_kernel<true>, grid, block, 0, vmx_stream(stream), logits, topk_weights, topk_ids, tokens, num_experts,
                           k, renormalize, round_probs_bf16);
    else
        hipLaunchKernelGGL(topk_softmax_kernel<false>, grid, block, 0, vmx_stream(stream), logits, topk_weights, topk_ids, tokens, num_experts,
                           k, renormalize, round_probs_bf16);
    VMX_CHECK_LAUNCH("topk_softmax");
    return VMX_OK;
}

extern "C" int vmx_moe_align_block_size(const int32_t *topk_ids, int32_t *sorted_token_ids, int32_t *expert_ids,
                                        int32_t *total_tokens_post_pad, int numel, int num_experts, int block_size,
                                        int sorted_capacity, void *stream) {
    VMX_REQUIRE(topk_ids && sorted_token_ids && expert_ids && total_tokens_post_pad, "moe_align_block_size: null pointer");
    VMX_REQUIRE(numel >= 0 && num_experts >= 1 && num_experts <= 1024 && block_size >= 1, "moe_align_block_size: bad arguments");
    VMX_REQUIRE(sorted_capacity >= numel + num_experts * (block_size - 1), "moe_align_block_size: sorted_token_ids capacity %d too small",
                sorted_capacity);
    if (num_experts <= 256)
        hipLaunchKernelGGL(moe_align_fast_kernel, dim3(1), dim3(1024), (19 * num_experts + 1) * sizeof(int32_t), vmx_stream(stream), topk_ids,
                           sorted_token_ids, expert_ids, total_tokens_post_pad, numel, num_experts, block_size, sorted_capacity);
    else
        hipLaunchKernelGGL(moe_align_kernel, dim3(1), dim3(256), (2 * num_experts + 1) * sizeof(int32_t), vmx_stream(stream), topk_ids,
                           sorted_token_ids, expert_ids, total_tokens_post_pad, numel, num_experts, block_size, sorted_capacity);
    VMX_CHECK_LAUNCH("moe_align_block_size");
    return VMX_OK;
}

extern "C" size_t vmx_moe_int4_workspace_bytes(int tokens, int k, int hidden, int inter) {
    if (tokens <= 0 || k <= 0 || hidden <= 0 || inter <= 0) return 0;
    const size_t pairs = (size_t)tokens * k;
    const size_t s1 = (size_t)vmx_int4_splits(2 * inter, hidden), s2 = (size_t)vmx_int4_splits(hidden, inter);
    size_t a = s1 * pairs * 2 * inter * 4, b = s2 * pairs * hidden * 4;
    a = (a + 255) & ~(size_t)255;
    b = (b + 255) & ~(size_t)255;
    // + the sorted (token, slot) indices and the blocks' experts of the token-grouped forms (up to 1024 experts)
    const size_t sort_words = pairs + (size_t)1024 * 63 + pairs / 16 + 1024 + 64;
    return a + b + ((pairs * inter * 2 + 255) & ~(size_t)255) + ((sort_words * 4 + 255) & ~(size_t)255);
}

namespace {
// Pairs from which an expert's weights are streamed once per block of its rows instead of once per pair: 16-row blocks through
// the small-M kernel for decode batches, 64-row blocks through the 64 x 128-tile kernel for prompts.  The thresholds are where
// the sort launch and the slower per-byte rate of the block kernels are paid back (measured on Mixtral-8x7B shapes).
int moe_block_rows(int pairs, int num_experts, int N1, int K1, int N2, int K2) {
    if (pairs >= 96 && N1 % 128 == 0 && N2 % 128 == 0) return 64;
    if (pairs >= 12 && vmx_int4_token_grouped16_ok(N1, K1) && vmx_int4_token_grouped16_ok(N2, K2)) return 16;
    return 0;
}
}  // namespace

namespace {
int g_moe_force_rows = -1;   // tests: -1 = choose by size, 0 = one pass per pair, 16 / 64 = that block form
}
// Test hook: force the form vmx_moe_int4_forward takes (-1 restores the choice by size).
extern "C" int vmx_debug_moe_block_rows(int rows) {
    VMX_REQUIRE(rows == -1 || rows == 0 || rows == 16 || rows == 64, "debug_moe_block_rows: %d", rows);
    g_moe_force_rows = rows;
    return VMX_OK;
}

extern "C" int vmx_moe_int4_forward(uint16_t *out, const uint16_t *x, const float *topk_weights, const int32_t *topk_ids,
                                    const void *const *w13_packed, const void *const *w2_packed, int tokens, int k, int num_experts,
                                    int hidden, int inter, int group, void *workspace, size_t workspace_bytes, void *stream) {
    return vmx_moe_int4_forward_glu(out, x, topk_weights, topk_ids, w13_packed, w2_packed, tokens, k, num_experts, hidden, inter, group,
                                    workspace, workspace_bytes, 0, vmx_stream(stream), nullptr, nullptr);
}

int vmx_moe_int4_forward_glu(uint16_t *out, const uint16_t *x, const float *topk_weights, const int32_t *topk_ids,
                             const void *const *w13_packed, const void *const *w2_packed, int tokens, int k, int num_experts, int hidden,
                             int inter, int group, void *workspace, size_t workspace_bytes, int w13_glu, hipStream_t st, uint16_t *h_resid,
                             float *ssq) {
    VMX_REQUIRE(!h_resid || (ssq && hidden % 256 == 0), "moe_int4_forward: the residual epilogue needs ssq and hidden %% 256 == 0");
    VMX_REQUIRE(out && x && topk_weights && topk_ids && w13_packed && w2_packed && workspace, "moe_int4_forward: null pointer");
    VMX_REQUIRE(tokens >= 1 && k >= 1 && num_experts >= 1, "moe_int4_forward: bad counts");
    VMX_REQUIRE(hidden % 128 == 0 && inter % 128 == 0 && group > 0 && group % 128 == 0 && hidden % group == 0 && inter % group == 0,
                "moe_int4_forward: hidden (%d), inter (%d) and group (%d) must be multiples of 128", hidden, inter, group);
    VMX_REQUIRE(workspace_bytes >= vmx_moe_int4_workspace_bytes(tokens, k, hidden, inter), "moe_int4_forward: workspace too small");
    const int pairs = tokens * k;
    const int s1 = vmx_int4_splits(2 * inter, hidden), s2 = vmx_int4_splits(hidden, inter);
    char *ws = (char *)workspace;
    float *p1 = (float *)ws;
    ws += ((size_t)s1 * pairs * 2 * inter * 4 + 255) & ~(size_t)255;
    float *p2 = (float *)ws;
    ws += ((size_t)s2 * pairs * hidden * 4 + 255) & ~(size_t)255;
    uint16_t *act = (uint16_t *)ws;
    ws += ((size_t)pairs * inter * 2 + 255) & ~(size_t)255;
    const int brows = g_moe_force_rows >= 0 ? g_moe_force_rows : moe_block_rows(pairs, num_experts, 2 * inter, hidden, hidden, inter);
    if (brows) {
        // token-grouped: sort the pairs by expert into blocks of `brows` rows, then one pass per block that streams the expert's
        // tile stream once for all its rows (fused_moe_align.cu:34-278 + the grouped GEMM of fused_moe_gemm.cu:72-545)
        VMX_REQUIRE(num_experts <= 1024, "moe_int4_forward: at most 1024 experts");
        const int cap = pairs + num_experts * (brows - 1), max_blocks = (cap + brows - 1) / brows;
        int32_t *sorted = (int32_t *)ws, *block_expert = sorted + ((cap + 63) & ~63), *total = block_expert + ((max_blocks + 63) & ~63);
        if (int rc = vmx_moe_align_block_size(topk_ids, sorted, block_expert, total, pairs, num_experts, brows, cap, st)) return rc;
        if (w13_glu) {
            if (int rc = vmx_int4_gemm_token_grouped(w13_packed, sorted, block_expert, total, max_blocks, brows, num_experts, x, k, nullptr, act,
                                                     pairs, 2 * inter, hidden, group, st))
                return rc;
        } else {
            if (int rc = vmx_int4_gemm_token_grouped(w13_packed, sorted, block_expert, total, max_blocks, brows, num_experts, x, k, p1, nullptr,
                                                     pairs, 2 * inter, hidden, group, st))
                return rc;
            int64_t blocks = ((int64_t)pairs * inter + 255) / 256;
            hipLaunchKernelGGL(moe_act_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st, p1, s1, pairs, act, inter);
            VMX_CHECK_LAUNCH("moe_act");
        }
        if (int rc = vmx_int4_gemm_token_grouped(w2_packed, sorted, block_expert, total, max_blocks, brows, num_experts, act, 1, p2, nullptr,
                                                 pairs, hidden, inter, group, st))
            return rc;
        hipLaunchKernelGGL(moe_combine_kernel, dim3((hidden + 255) / 256, tokens), dim3(256), 0, st, out, p2, s2, pairs, topk_weights, k, hidden,
                           w2_packed, topk_ids, num_experts, h_resid, ssq);
        VMX_CHECK_LAUNCH("moe_combine");
        return VMX_OK;
    }
    // gate|up of every (token, slot) pair with its expert's weights; x row = token
    if (w13_glu) {   // act straight from the GEMM epilogue
        if (int rc = vmx_int4_gemm_grouped_glu(w13_packed, topk_ids, num_experts, x, k, act, pairs, 2 * inter, hidden, group, st)) return rc;
    } else {
        if (int rc = vmx_int4_gemm_grouped(w13_packed, topk_ids, num_experts, x, k, p1, pairs, 2 * inter, hidden, group, st)) return rc;
        int64_t blocks = ((int64_t)pairs * inter + 255) / 256;
        hipLaunchKernelGGL(moe_act_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st, p1, s1, pairs, act, inter);
        VMX_CHECK_LAUNCH("moe_act");
    }
    // down projection; x row = pair
    if (int rc = vmx_int4_gemm_grouped(w2_packed, topk_ids, num_experts, act, 1, p2, pairs, hidden, inter, group, st)) return rc;
    hipLaunchKernelGGL(moe_combine_kernel, dim3((hidden + 255) / 256, tokens), dim3(256), 0, st, out, p2, s2, pairs, topk_weights, k, hidden, w2_packed, topk_ids,
                       num_experts, h_resid, ssq);
    VMX_CHECK_LAUNCH("moe_combine");
    return VMX_OK;
}

// ---------------------------------------------------------------- expert parallelism with token dispatch (data-parallel attention)
// EPMoELayer::forward (crates/core/src/moe/ep_layer.rs:180-210) with TokenDispatcher::dispatch / combine
// (moe/token_dispatch.rs:113-231): every rank holds its OWN tokens and E / world whole experts; a (token, slot) pair travels to
// the rank that owns its expert, runs there, and its output row travels back, where the routing weights are applied.  The
// reference reads the routing on the host, exchanges counts with all_to_all and rows with three all_to_all_v; here the routing
// never leaves the device: the rows are grouped by destination rank by one workgroup, the counts ride in the header of the
// exchange (vmx_comm_all_to_all_v_dev), and the receiver runs the token-grouped expert GEMMs over however many rows arrived.
// The result equals vmx_moe_int4_forward on the rank's tokens with all experts resident, bit for bit: the value a row carries
// back is the expert's bf16 output, the weighted sum over the slots is taken at the source in slot order.
namespace {

__device__ __forceinline__ int ep_owner(int e, int per, int world, int round_robin) { return round_robin ? e % world : e / per; }
__device__ __forceinline__ int ep_local(int e, int per, int world, int round_robin) { return round_robin ? e / world : e % per; }

// One workgroup of 1024 threads: send_pairs [pairs] = the (token, slot) pair indices grouped by destination rank (ascending
// inside a group), pos_of_pair [pairs] = where each pair went, send_rows [world] = rows per destination.
__global__ __launch_bounds__(1024) void ep_route_kernel(const int32_t *__restrict__ topk_ids, int pairs, int num_experts, int world,
                                                         int round_robin, int32_t *__restrict__ send_pairs, int32_t *__restrict__ pos_of_pair,
                                                         int32_t *__restrict__ send_rows) {
    __shared__ int32_t counts[8], cursor[8], wave_cnt[16][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = num_experts / world;
    if (tid < 8) counts[tid] = 0;
    if (tid < 128) wave_cnt[tid >> 3][tid & 7] = 0;
    __syncthreads();
    for (int i = tid; i < pairs; i += 1024) {
        const int e = topk_ids[i];
        if (e >= 0 && e < num_experts) atomicAdd(&counts[ep_owner(e, per, world, round_robin)], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int r = 0; r < world; r++) {
            cursor[r] = acc;
            send_rows[r] = counts[r];
            acc += counts[r];
        }
    }
    __syncthreads();
    for (int base = 0; base < pairs; base += 1024) {
        const int i = base + tid;
        const int e = i < pairs ? topk_ids[i] : -1;
        const bool valid = e >= 0 && e < num_experts;
        const int dst = valid ? ep_owner(e, per, world, round_robin) : -1;
        int rank = 0;
        for (int r = 0; r < world; r++) {
            const unsigned long long mask = __ballot(dst == r);
            if (dst == r) rank = __popcll(mask & ((1ull << lane) - 1ull));
            if (lane == 0) wave_cnt[wave][r] = __popcll(mask);
        }
        __syncthreads();
        if (valid) {
            int before = 0;
            for (int w = 0; w < wave; w++) before += wave_cnt[w][dst];
            const int pos = cursor[dst] + before + rank;
            send_pairs[pos] = i;
            pos_of_pair[i] = pos;
        } else if (i < pairs) {
            pos_of_pair[i] = -1;
        }
        __syncthreads();
        if (tid < world) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += wave_cnt[w][tid];
            cursor[tid] += tot;
        }
        __syncthreads();
    }
}

// send_x [i] = x [token of send_pairs[i]], send_e [i] = the pair's expert as the destination rank numbers its experts
__global__ __launch_bounds__(256) void ep_gather_kernel(const uint16_t *__restrict__ x, const int32_t *__restrict__ topk_ids,
                                                         const int32_t *__restrict__ send_pairs, const int32_t *__restrict__ send_rows, int world,
                                                         int k, int hidden, int num_experts, int round_robin, uint16_t *__restrict__ send_x,
                                                         int32_t *__restrict__ send_e) {
    int total = 0;
    for (int r = 0; r < world; r++) total += send_rows[r];
    const int i = blockIdx.x;
    if (i >= total) return;
    const int p = send_pairs[i];
    const uint4 *src = reinterpret_cast<const uint4 *>(x + (int64_t)(p / k) * hidden);
    uint4 *dst = reinterpret_cast<uint4 *>(send_x + (int64_t)i * hidden);
    for (int c = threadIdx.x; c < hidden / 8; c += 256) dst[c] = src[c];
    if (threadIdx.x == 0) send_e[i] = ep_local(topk_ids[p], num_experts / world, world, round_robin);
}

// y [i] = bf16(sum of the split-K slices of row i) for the rows that arrived; ids beyond them are -1
__global__ __launch_bounds__(256) void ep_rows_finish_kernel(const float *__restrict__ partial, int splits, int cap_rows, int hidden,
                                                              const int32_t *__restrict__ recv_e, uint16_t *__restrict__ y) {
    const int i = blockIdx.y;
    if (recv_e[i] < 0) return;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < hidden; c += gridDim.x * 256) {
        float o = 0.f;
        for (int sidx = 0; sidx < splits; sidx++) o += partial[((int64_t)sidx * cap_rows + i) * hidden + c];
        y[(int64_t)i * hidden + c] = vmx_f32_to_bf16(o);
    }
}

// out [t] = sum over the slots (in slot order) of bf16(bf16 expert output * bf16 weight), each partial sum rounded to bf16:
// the arithmetic of moe_combine_kernel, with the expert outputs read from the rows that came back
__global__ __launch_bounds__(256) void ep_combine_kernel(uint16_t *__restrict__ out, const uint16_t *__restrict__ ret_y,
                                                          const int32_t *__restrict__ pos_of_pair, const float *__restrict__ w, int k,
                                                          int hidden) {
    const int t = blockIdx.y;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < hidden; c += gridDim.x * 256) {
        float acc = 0.f;
        for (int j = 0; j < k; j++) {
            const int pos = pos_of_pair[t * k + j];
            if (pos < 0) continue;
            const float o = vmx_bf16_to_f32(ret_y[(int64_t)pos * hidden + c]);
            acc = rb(acc + rb(o * rb(w[t * k + j])));
        }
        out[(int64_t)t * hidden + c] = vmx_f32_to_bf16(acc);
    }
}

size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

// cap_rows: the most rows this rank may receive in one call (at most world * tokens * k of the largest rank)
extern "C" size_t vmx_moe_ep_workspace_bytes(int tokens, int k, int hidden, int inter, int cap_rows) {
    if (tokens <= 0 || k <= 0 || hidden <= 0 || inter <= 0 || cap_rows <= 0) return 0;
    const size_t pairs = (size_t)tokens * k;
    return 3 * al256(pairs * 4) + 4 * 256 + al256((size_t)cap_rows * 4) + 2 * al256(pairs * hidden * 2) + 2 * al256((size_t)cap_rows * hidden * 2) +
           vmx_moe_int4_workspace_bytes(cap_rows, 1, hidden, inter);
}

extern "C" int vmx_moe_ep_int4_forward(uint16_t *out, const uint16_t *x, const float *topk_weights, const int32_t *topk_ids,
                                       const void *const *w13_local, const void *const *w2_local, int tokens, int k, int num_experts,
                                       int hidden, int inter, int group, int w13_glu, int round_robin, vmx_comm *comm, int cap_rows,
                                       void *workspace, size_t workspace_bytes, void *stream) {
    VMX_REQUIRE(out && x && topk_weights && topk_ids && w13_local && w2_local && comm && workspace, "moe_ep_int4_forward: null pointer");
    const int world = vmx_comm_world(comm);
    VMX_REQUIRE(world >= 1 && world <= 8 && num_experts % world == 0, "moe_ep_int4_forward: %d experts over %d ranks (at most 8, whole experts per rank)",
                num_experts, world);
    VMX_REQUIRE(tokens >= 1 && k >= 1 && k <= num_experts && cap_rows >= tokens * k, "moe_ep_int4_forward: bad counts (cap_rows %d < %d)", cap_rows, tokens * k);
    VMX_REQUIRE(hidden % 128 == 0 && inter % 128 == 0 && group > 0 && group % 128 == 0 && hidden % group == 0 && inter % group == 0,
                "moe_ep_int4_forward: hidden (%d), inter (%d) and group (%d) must be multiples of 128", hidden, inter, group);
    VMX_REQUIRE(!w13_glu || vmx_int4_glu_fusable(2 * inter, hidden), "moe_ep_int4_forward: gate|up [%d -> %d] cannot take the gated epilogue", hidden, 2 * inter);
    VMX_REQUIRE(workspace_bytes >= vmx_moe_ep_workspace_bytes(tokens, k, hidden, inter, cap_rows), "moe_ep_int4_forward: workspace too small");
    hipStream_t st = vmx_stream(stream);
    const int pairs = tokens * k, per = num_experts / world;
    char *ws = (char *)workspace;
    auto take = [&](size_t bytes) { char *q = ws; ws += al256(bytes); return q; };
    int32_t *send_pairs = (int32_t *)take((size_t)pairs * 4), *pos_of_pair = (int32_t *)take((size_t)pairs * 4), *send_e = (int32_t *)take((size_t)pairs * 4);
    int32_t *send_rows = (int32_t *)take(64), *recv_rows = (int32_t *)take(64), *recv_rows2 = (int32_t *)take(64), *back_rows = (int32_t *)take(64);
    int32_t *recv_e = (int32_t *)take((size_t)cap_rows * 4);
    uint16_t *send_x = (uint16_t *)take((size_t)pairs * hidden * 2), *ret_y = (uint16_t *)take((size_t)pairs * hidden * 2);
    uint16_t *recv_x = (uint16_t *)take((size_t)cap_rows * hidden * 2), *y = (uint16_t *)take((size_t)cap_rows * hidden * 2);
    // 1. group the pairs by destination rank, gather their rows
    hipLaunchKernelGGL(ep_route_kernel, dim3(1), dim3(1024), 0, st, topk_ids, pairs, num_experts, world, round_robin, send_pairs, pos_of_pair, send_rows);
    VMX_CHECK_LAUNCH("ep_route");
    hipLaunchKernelGGL(ep_gather_kernel, dim3(pairs), dim3(256), 0, st, x, topk_ids, send_pairs, send_rows, world, k, hidden, num_experts, round_robin,
                       send_x, send_e);
    VMX_CHECK_LAUNCH("ep_gather");
    // 2. dispatch: expert ids and rows to the owners (counts travel with them)
    VMX_CHECK_HIP(hipMemsetAsync(recv_e, 0xFF, (size_t)cap_rows * 4, st));
    if (int rc = vmx_comm_all_to_all_v_dev(comm, send_e, send_rows, recv_e, recv_rows, 4, (size_t)cap_rows, st)) return rc;
    if (int rc = vmx_comm_all_to_all_v_dev(comm, send_x, send_rows, recv_x, recv_rows2, (size_t)hidden * 2, (size_t)cap_rows, st)) return rc;
    // 3. the local experts over the rows that arrived (one slot per row, ids < 0 = no row): token-grouped GEMMs
    {
        char *mws = ws;
        const int s1 = vmx_int4_splits(2 * inter, hidden), s2 = vmx_int4_splits(hidden, inter);
        float *p1 = (float *)mws;
        mws += al256((size_t)s1 * cap_rows * 2 * inter * 4);
        float *p2 = (float *)mws;
        mws += al256((size_t)s2 * cap_rows * hidden * 4);
        uint16_t *act = (uint16_t *)mws;
        mws += al256((size_t)cap_rows * inter * 2);
        const int brows = (cap_rows >= 96 && (2 * inter) % 128 == 0 && hidden % 128 == 0) ? 64
                          : (vmx_int4_token_grouped16_ok(2 * inter, hidden) && vmx_int4_token_grouped16_ok(hidden, inter)) ? 16 : 64;
        VMX_REQUIRE(brows == 16 || ((2 * inter) % 128 == 0 && hidden % 128 == 0), "moe_ep_int4_forward: shapes fit neither block form");
        const int cap = cap_rows + per * (brows - 1), max_blocks = (cap + brows - 1) / brows;
        int32_t *sorted = (int32_t *)mws, *block_expert = sorted + ((cap + 63) & ~63), *total = block_expert + ((max_blocks + 63) & ~63);
        if (int rc = vmx_moe_align_block_size(recv_e, sorted, block_expert, total, cap_rows, per, brows, cap, st)) return rc;
        if (w13_glu) {
            if (int rc = vmx_int4_gemm_token_grouped(w13_local, sorted, block_expert, total, max_blocks, brows, per, recv_x, 1, nullptr, act, cap_rows,
                                                     2 * inter, hidden, group, st))
                return rc;
        } else {
            if (int rc = vmx_int4_gemm_token_grouped(w13_local, sorted, block_expert, total, max_blocks, brows, per, recv_x, 1, p1, nullptr, cap_rows,
                                                     2 * inter, hidden, group, st))
                return rc;
            int64_t blocks = ((int64_t)cap_rows * inter + 255) / 256;
            hipLaunchKernelGGL(moe_act_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st, p1, s1, cap_rows, act, inter);
            VMX_CHECK_LAUNCH("moe_act");
        }
        if (int rc = vmx_int4_gemm_token_grouped(w2_local, sorted, block_expert, total, max_blocks, brows, per, act, 1, p2, nullptr, cap_rows, hidden,
                                                 inter, group, st))
            return rc;
        hipLaunchKernelGGL(ep_rows_finish_kernel, dim3((hidden + 255) / 256, cap_rows), dim3(256), 0, st, p2, s2, cap_rows, hidden, recv_e, y);
        VMX_CHECK_LAUNCH("ep_rows_finish");
    }
    // 4. combine: the rows go back the way they came, the source applies the routing weights in slot order
    if (int rc = vmx_comm_all_to_all_v_dev(comm, y, recv_rows, ret_y, back_rows, (size_t)hidden * 2, (size_t)pairs, st)) return rc;
    hipLaunchKernelGGL(ep_combine_kernel, dim3((hidden + 255) / 256, tokens), dim3(256), 0, st, out, ret_y, pos_of_pair, topk_weights, k, hidden);
    VMX_CHECK_LAUNCH("ep_combine");
    return VMX_OK;
}

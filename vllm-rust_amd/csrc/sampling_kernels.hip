// Non-greedy sampling on the device: temperature scaling, softmax to probabilities and top-k / top-p
// filtered multinomial sampling (crates/core/kernels/sampling.cu: temperature_scale_bf16 :102-117,
// softmax_to_probs(_f32) :123-190,351-419, top_k_top_p_sample(_per_seq) :197-293,425-520).
//
// The reference's sampler is ONE thread per sequence walking the vocabulary many times.  The same
// decisions are taken here by a 256-thread workgroup per sequence:
//   top-k threshold  = k-th largest DISTINCT probability      (k rounds of a block-wide "max below prev")
//   top-p threshold  = distinct values added from the top, once per occurrence, until the running sum
//                      reaches top_p * filtered_sum
//   draw             = first index (in vocabulary order) whose running filtered sum exceeds rand * filtered_sum
// Thresholds are exact (comparisons only).  Sums are formed per contiguous 1/256 slice of the vocabulary and
// then combined in slice order, so they can differ from the reference's single running f32 sum in the last
// bits; a draw can only change when rand * sum falls within that rounding distance of a boundary.
#include "vmx_internal.h"
#include <float.h>

namespace {

__device__ __forceinline__ float block_max(float v, float *sh) {
    v = vmx_wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return r;
}

template <bool BF16>
__global__ __launch_bounds__(256) void softmax_probs_kernel(float *__restrict__ probs, const void *__restrict__ logits, int vocab) {
    __shared__ float sh[4];
    const int64_t row = (int64_t)blockIdx.x * vocab;
    auto ld = [&](int i) -> float {
        return BF16 ? vmx_bf16_to_f32(((const uint16_t *)logits)[row + i]) : ((const float *)logits)[row + i];
    };
    float mx = -FLT_MAX;
    for (int i = threadIdx.x; i < vocab; i += 256) mx = fmaxf(mx, ld(i));
    mx = block_max(mx, sh);
    float sum = 0.f;
    for (int i = threadIdx.x; i < vocab; i += 256) {
        const float e = __expf(ld(i) - mx);
        probs[row + i] = e;
        sum += e;
    }
    sum = vmx_block_sum_256(sum, sh);
    const float inv = 1.0f / (sum + 1e-6f);
    for (int i = threadIdx.x; i < vocab; i += 256) probs[row + i] *= inv;
}

__global__ __launch_bounds__(256) void temperature_scale_kernel(uint16_t *__restrict__ logits, float inv_t, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        logits[i] = vmx_f32_to_bf16(vmx_bf16_to_f32(logits[i]) * inv_t);
}

__global__ __launch_bounds__(256) void top_k_top_p_kernel(int32_t *__restrict__ out, const float *__restrict__ probs,
                                                           const float *__restrict__ rand_vals, const int32_t *__restrict__ top_k_arr,
                                                           const float *__restrict__ top_p_arr, int top_k_all, float top_p_all,
                                                           int vocab) {
    __shared__ float sh[4];
    __shared__ float part[256];
    __shared__ int cnt_sh[4];
    __shared__ float bcast[2];
    const int seq = blockIdx.x, tid = threadIdx.x;
    const float *p = probs + (int64_t)seq * vocab;
    const int top_k = top_k_arr ? top_k_arr[seq] : top_k_all;
    const float top_p = top_p_arr ? top_p_arr[seq] : top_p_all;
    const float rnd = rand_vals[seq];
    // contiguous slice of the vocabulary per thread (vocabulary order matters for the draw)
    const int per = (vocab + 255) / 256, lo = min(tid * per, vocab), hi = min(lo + per, vocab);

    float threshold = 0.0f;
    if (top_k > 0 && top_k < vocab) {
        float prev = FLT_MAX;
        for (int rank = 0; rank < top_k; rank++) {
            float best = -FLT_MAX;
            for (int i = tid; i < vocab; i += 256) {
                const float v = p[i];
                if (v < prev && v > best) best = v;
            }
            best = block_max(best, sh);
            if (best <= -FLT_MAX) break;
            prev = best;
            threshold = best;
        }
    }
    auto filtered_sum = [&](float thr) -> float {
        float s = 0.f;
        for (int i = lo; i < hi; i++)
            if (p[i] >= thr) s += p[i];
        part[tid] = s;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int j = 0; j < 256; j++) t += part[j];
            bcast[0] = t;
        }
        __syncthreads();
        const float r = bcast[0];
        __syncthreads();
        return r;
    };
    float fsum = filtered_sum(threshold);

    if (top_p < 1.0f && top_p > 0.0f) {
        float cumsum = 0.f, top_p_thresh = 0.f, prev = FLT_MAX;
        const float target = top_p * fsum;
        while (cumsum < target) {
            float best = -FLT_MAX;
            for (int i = tid; i < vocab; i += 256) {
                const float v = p[i];
                if (v >= threshold && v < prev && v > best) best = v;
            }
            best = block_max(best, sh);
            if (best <= -FLT_MAX) break;
            prev = best;
            top_p_thresh = best;
            int c = 0;
            for (int i = tid; i < vocab; i += 256) c += (p[i] == best) ? 1 : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
            if ((tid & 63) == 0) cnt_sh[tid >> 6] = c;
            __syncthreads();
            const int occ = cnt_sh[0] + cnt_sh[1] + cnt_sh[2] + cnt_sh[3];
            __syncthreads();
            for (int j = 0; j < occ; j++) {  // one addition per occurrence, stop once the target is reached
                cumsum += best;
                if (cumsum >= target) break;
            }
        }
        if (top_p_thresh > threshold) threshold = top_p_thresh;
        fsum = filtered_sum(threshold);
    }

    // multinomial draw in vocabulary order
    const float target = rnd * fsum;
    float s = 0.f;
    for (int i = lo; i < hi; i++)
        if (p[i] >= threshold) s += p[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        float run = 0.f;
        int owner = -1;
        for (int j = 0; j < 256; j++) {
            if (run + part[j] > target) { owner = j; break; }
            run += part[j];
        }
        bcast[0] = run;
        bcast[1] = __int_as_float(owner);
    }
    __syncthreads();
    const int owner = __float_as_int(bcast[1]);
    if (owner < 0) {
        if (tid == 0) out[seq] = vocab - 1;
        return;
    }
    if (tid == owner) {
        // The slice was chosen from its separately rounded sum; re-adding its elements one by one from `run` can stay at or
        // below the target by a rounding error.  The sample then is the slice's last kept token, never a filtered one.
        float run = bcast[0];
        int sel = -1, last_kept = -1;
        for (int i = lo; i < hi; i++)
            if (p[i] >= threshold) {
                last_kept = i;
                run += p[i];
                if (run > target) { sel = i; break; }
            }
        out[seq] = sel >= 0 ? sel : (last_kept >= 0 ? last_kept : vocab - 1);
    }
}

// temperature_scale_f32_per_seq (crates/core/kernels/sampling.cu:529-543): row r of the f32 logits times inv_temps[r], in place.
// grid (column tiles, rows): a whole vocabulary row per workgroup would leave most of the chip idle at small batches.
__global__ __launch_bounds__(256) void temperature_per_seq_kernel(float *__restrict__ logits, const float *__restrict__ inv_temps, int vocab) {
    const float inv_t = inv_temps[blockIdx.y];
    float *row = logits + (int64_t)blockIdx.y * vocab;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vocab; i += gridDim.x * 256) row[i] *= inv_t;
}

// multinomial_sample_no_filter_parallel (sampling.cu:566-631): inverse-CDF sample of an unfiltered distribution.  The token
// depends on the order of the f32 additions, so the order is the reference's: 256 contiguous chunks of ceil(vocab / 256)
// probabilities, each summed front to back; the chunk sums added front to back for the total and again for the prefix
// walk; the chosen chunk re-added element by element on top of the prefix.  One workgroup of 256 per row.  Thread t reads
// its chunk as a stream of its own (a cache line serves 16 consecutive steps of one lane); the serial part is one lane.
__global__ __launch_bounds__(256) void multinomial_no_filter_kernel(int32_t *__restrict__ out, const float *__restrict__ probs,
                                                                    const float *__restrict__ rand_vals, int vocab) {
    __shared__ float chunk_sums[256];
    const int seq = blockIdx.x, tid = threadIdx.x;
    const float *p = probs + (int64_t)seq * vocab;
    const int chunk = (vocab + 255) / 256;
    const int lo = min(tid * chunk, vocab), hi = min(lo + chunk, vocab);
    float s = 0.0f;
    for (int i = lo; i < hi; i++) s += p[i];
    chunk_sums[tid] = s;
    __syncthreads();
    if (tid == 0) {
        float total = 0.0f;
        for (int c = 0; c < 256; c++) total += chunk_sums[c];
        const float target = rand_vals[seq] * total;
        float cdf = 0.0f;
        int tc = 255;
        for (int c = 0; c < 256; c++) {
            const float next = cdf + chunk_sums[c];
            if (next > target) { tc = c; break; }
            cdf = next;
        }
        const int c0 = min(tc * chunk, vocab), c1 = min(c0 + chunk, vocab);
        int sel = vocab - 1;
        for (int i = c0; i < c1; i++) {
            cdf += p[i];
            if (cdf > target) { sel = i; break; }
        }
        out[seq] = sel;
    }
}

}  // namespace

extern "C" int vmx_temperature_scale_f32_per_seq(float *logits, const float *inv_temps, int rows, int vocab, void *stream) {
    VMX_REQUIRE(logits && inv_temps && rows >= 0 && vocab > 0, "temperature_scale_per_seq: bad arguments");
    if (rows == 0) return VMX_OK;
    int tiles = (vocab + 2047) / 2048;   // eight elements per lane
    hipLaunchKernelGGL(temperature_per_seq_kernel, dim3(tiles, rows), dim3(256), 0, vmx_stream(stream), logits, inv_temps, vocab);
    VMX_CHECK_LAUNCH("temperature_scale_per_seq");
    return VMX_OK;
}

extern "C" int vmx_multinomial_sample_no_filter(int32_t *out_ids, const float *probs, const float *rand_vals, int rows, int vocab,
                                                void *stream) {
    VMX_REQUIRE(out_ids && probs && rand_vals && rows >= 0 && vocab > 0, "multinomial_sample_no_filter: bad arguments");
    if (rows == 0) return VMX_OK;
    hipLaunchKernelGGL(multinomial_no_filter_kernel, dim3(rows), dim3(256), 0, vmx_stream(stream), out_ids, probs, rand_vals, vocab);
    VMX_CHECK_LAUNCH("multinomial_sample_no_filter");
    return VMX_OK;
}

extern "C" int vmx_temperature_scale_bf16(uint16_t *logits, float inv_temperature, int rows, int vocab, void *stream) {
    VMX_REQUIRE(logits && rows >= 0 && vocab > 0, "temperature_scale: bad arguments");
    const int64_t n = (int64_t)rows * vocab;
    if (n == 0) return VMX_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(temperature_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, vmx_stream(stream), logits, inv_temperature, n);
    VMX_CHECK_LAUNCH("temperature_scale");
    return VMX_OK;
}

extern "C" int vmx_softmax_to_probs(float *probs, const void *logits, int logits_dtype, int rows, int vocab, void *stream) {
    VMX_REQUIRE(probs && logits && rows >= 0 && vocab > 0, "softmax_to_probs: bad arguments");
    VMX_REQUIRE(logits_dtype == VMX_F32 || logits_dtype == VMX_BF16, "softmax_to_probs: logits dtype must be f32 or bf16");
    if (rows == 0) return VMX_OK;
    if (logits_dtype == VMX_BF16) hipLaunchKernelGGL(softmax_probs_kernel<true>, dim3(rows), dim3(256), 0, vmx_stream(stream), probs, logits, vocab);
    else hipLaunchKernelGGL(softmax_probs_kernel<false>, dim3(rows), dim3(256), 0, vmx_stream(stream), probs, logits, vocab);
    VMX_CHECK_LAUNCH("softmax_to_probs");
    return VMX_OK;
}

extern "C" int vmx_top_k_top_p_sample(int32_t *out_ids, const float *probs, const float *rand_vals, const int32_t *top_k_per_seq,
                                      const float *top_p_per_seq, int top_k, float top_p, int rows, int vocab, void *stream) {
    VMX_REQUIRE(out_ids && probs && rand_vals && rows >= 0 && vocab > 0, "top_k_top_p_sample: bad arguments");
    if (rows == 0) return VMX_OK;
    hipLaunchKernelGGL(top_k_top_p_kernel, dim3(rows), dim3(256), 0, vmx_stream(stream), out_ids, probs, rand_vals, top_k_per_seq,
                       top_p_per_seq, top_k, top_p, vocab);
    VMX_CHECK_LAUNCH("top_k_top_p_sample");
    return VMX_OK;
}

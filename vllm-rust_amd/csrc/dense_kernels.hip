// Dense bf16 GEMV for small M (lm_head / tied embedding head on the decode path).  The reference
// sends this product to cuBLAS through Candle (crates/core/src/models/qwen3_quantized.rs:1258-1302,
// cuda_kernels.rs half_matmul_pooled): y[m, n] = bf16(sum_k x[m, k] * W[n, k]), f32 accumulation.
// gfx950 form: one wave per output row, the row streamed with 16-byte loads (1 KiB per
// wave-instruction), v_dot2c_f32_bf16 on packed pairs, x rows held in LDS, wave64 butterfly.
#include "vmx_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// F16: the operands are fp16 pairs (v_dot2_f32_f16), else bf16 pairs (v_dot2_f32_bf16); f32 accumulation either way
template <bool F16> __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    if constexpr (F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
    else return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// grid-stride over rows: wave w of workgroup b takes rows (b*4 + w) * R + r.  K % 512 == 0 is not
// required: the tail is handled with a bounds check on the 8-element chunk index.
template <int MT, int R, bool F16>
__global__ __launch_bounds__(256) void dense_gemv_kernel(uint16_t *__restrict__ y, const uint16_t *__restrict__ x,
                                                          const uint16_t *__restrict__ W, const uint16_t *__restrict__ bias, int M,
                                                          int N, int K) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint4 *xs = reinterpret_cast<uint4 *>(smem);  // [MT][K/8]
    const int chunks = K >> 3;
    for (int i = threadIdx.x; i < M * chunks; i += 256) xs[i] = reinterpret_cast<const uint4 *>(x)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves_total = gridDim.x * 4;
    for (int row0 = (blockIdx.x * 4 + wave) * R; row0 < N; row0 += waves_total * R) {
        float acc[R][MT];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int m = 0; m < MT; m++) acc[r][m] = 0.f;
        for (int c = lane; c < chunks; c += 64) {
            uint4 wv[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int row = min(row0 + r, N - 1);
                wv[r] = reinterpret_cast<const uint4 *>(W + (int64_t)row * K)[c];
            }
#pragma unroll
            for (int m = 0; m < MT; m++) {
                if (m < M) {
                    const uint4 xv = xs[m * chunks + c];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        float a = acc[r][m];
                        a = dot2<F16>(wv[r].x, xv.x, a);
                        a = dot2<F16>(wv[r].y, xv.y, a);
                        a = dot2<F16>(wv[r].z, xv.z, a);
                        a = dot2<F16>(wv[r].w, xv.w, a);
                        acc[r][m] = a;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int m = 0; m < MT; m++) {
                float v = vmx_wave_sum(acc[r][m]);
                if (lane == 0 && m < M && row0 + r < N) {
                    if (bias) v += vmx_ld16<F16>(bias[row0 + r]);
                    y[(int64_t)m * N + row0 + r] = vmx_st16<F16>(v);
                }
            }
    }
}

// 5..16 activation rows: GEMM-shaped, on the matrix cores.  The kernel above stages M x K activations (128 KB at 16 rows of
// 4096: one workgroup per CU) and spends M*R*4 dot2 per 16 weight bytes; measured 218 / 454 / 1428 us at 1 / 8 / 16 rows on
// the 151936 x 4096 lm_head.  Here a wave owns 16 consecutive rows of W (= 16 output columns of one v_mfma_f32_16x16x32_bf16,
// A = the activation rows, so 1..16 rows cost the same matrix instructions).  Lane (n = lane & 15, g = lane >> 4) reads the 64
// contiguous bytes W[row0 + n][128 b + 32 g .. + 32) of every 128-deep k-block b; its j-th 16 bytes serve MFMA step j, and the
// A fragment of step j is read from LDS at the matching k (x[m][128 b + 32 g + 8 j .. + 8]): the matrix instruction only needs
// A and B to agree on which k sits where.  Activations are staged PK = 1024 k at a time (33 KB at 16 rows).  f32 accumulation
// in k order within a lane group, groups summed by the MFMA: re-associated against the kernel above, same products.
template <bool F16>
__global__ __launch_bounds__(256) void dense_gemm16_kernel(uint16_t *__restrict__ y, const uint16_t *__restrict__ x,
                                                            const uint16_t *__restrict__ W, const uint16_t *__restrict__ bias, int M,
                                                            int N, int K) {
    constexpr int PK = 1024, XROW = PK + 8;   // k per pass; staged row stride (+16 B against bank conflicts)
    __shared__ __align__(16) uint16_t xs[16 * XROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & 15, g = lane >> 4;
    const int tiles = (N + 15) >> 4;
    const int waves_total = gridDim.x * 4;
    // every wave of the workgroup walks the same number of tiles (barriers inside): tiles past the end are clamped, not stored
    const int rounds = (tiles + waves_total - 1) / waves_total;
    for (int rd = 0; rd < rounds; rd++) {
        const int tile = rd * waves_total + blockIdx.x * 4 + wave;
        const int row = min(tile * 16 + nl, N - 1);
        const uint16_t *wrow = W + (int64_t)row * K;
        vmx_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // the weights of 128-deep block i + 1 are requested before block i is multiplied: with load and use in the same iteration
        // a wave has one block (4 KB) in flight and the launch runs at the memory latency, not the memory rate
        const int nblk = K >> 7;
        uint4 wn[4];
#pragma unroll
        for (int j = 0; j < 4; j++) wn[j] = *reinterpret_cast<const uint4 *>(wrow + 32 * g + 8 * j);
        for (int blk = 0; blk < nblk; blk++) {
            const int k0 = (blk >> 3) * PK, b = (blk & 7) * 128;
            if ((blk & 7) == 0) {
                const int pk = min(PK, K - k0);   // K % 128 == 0 (checked on the host)
                if (blk || rd) __syncthreads();   // the previous pass has been read by every wave
                for (int c = tid; c < 16 * (PK / 8); c += 256) {
                    const int m = c / (PK / 8), cc = c % (PK / 8);
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (m < M && cc * 8 < pk) v = *reinterpret_cast<const uint4 *>(x + (int64_t)m * K + k0 + cc * 8);
                    *reinterpret_cast<uint4 *>(xs + m * XROW + cc * 8) = v;
                }
                __syncthreads();
            }
            uint4 wv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) wv[j] = wn[j];
            if (blk + 1 < nblk) {
#pragma unroll
                for (int j = 0; j < 4; j++) wn[j] = *reinterpret_cast<const uint4 *>(wrow + (blk + 1) * 128 + 32 * g + 8 * j);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint4 au = *reinterpret_cast<const uint4 *>(xs + nl * XROW + b + 32 * g + 8 * j);
                if constexpr (F16)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, au), __builtin_bit_cast(f16x8_t, wv[j]), acc, 0, 0, 0);
                else
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, au), __builtin_bit_cast(vmx_bf16x8, wv[j]), acc,
                                                                  0, 0, 0);
            }
        }
        // lane holds activation rows m = 4 g + r of output column (W row) tile * 16 + nl
        const int col = tile * 16 + nl;
        if (tile < tiles && col < N) {
            const float bv = bias ? vmx_ld16<F16>(bias[col]) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = 4 * g + r;
                if (m < M) y[(int64_t)m * N + col] = vmx_st16<F16>(acc[r] + bv);
            }
        }
    }
}

// A handful of output columns (a mixture-of-experts router: N = 8) for several rows: the matrix-core kernel above would run
// as ONE workgroup walking all of K (19 us at 16 rows of 4096).  Here a workgroup owns one activation row and all N <= 16
// weight rows; a thread multiplies its 8-element chunks of the row with the same chunks of every weight row, then the
// workgroup adds up.  f32 accumulation, chunk order per thread, threads summed by wave then across the four waves.
template <bool F16>
__global__ __launch_bounds__(256) void dense_fewcols_kernel(uint16_t *__restrict__ y, const uint16_t *__restrict__ x,
                                                             const uint16_t *__restrict__ W, const uint16_t *__restrict__ bias, int N,
                                                             int K) {
    __shared__ float red[4][16];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, chunks = K >> 3;
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    for (int c = tid; c < chunks; c += 256) {
        const uint4 xv = reinterpret_cast<const uint4 *>(x + (int64_t)m * K)[c];
        uint4 wv[16];
#pragma unroll
        for (int e = 0; e < 16; e++) wv[e] = reinterpret_cast<const uint4 *>(W + (int64_t)min(e, N - 1) * K)[c];
#pragma unroll
        for (int e = 0; e < 16; e++) {
            float a = acc[e];
            a = dot2<F16>(wv[e].x, xv.x, a);
            a = dot2<F16>(wv[e].y, xv.y, a);
            a = dot2<F16>(wv[e].z, xv.z, a);
            a = dot2<F16>(wv[e].w, xv.w, a);
            acc[e] = a;
        }
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const float v = vmx_wave_sum(acc[e]);
        if (lane == 0) red[wave][e] = v;
    }
    __syncthreads();
    if (tid < N) {
        float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        if (bias) v += vmx_ld16<F16>(bias[tid]);
        y[(int64_t)m * N + tid] = vmx_st16<F16>(v);
    }
}

__global__ __launch_bounds__(256) void embedding_lookup_kernel(uint16_t *__restrict__ out, const uint16_t *__restrict__ table,
                                                                const int32_t *__restrict__ ids, int hidden, int vocab) {
    const int t = blockIdx.x;
    int id = ids[t];
    const bool ok = id >= 0 && id < vocab;
    const uint4 *src = reinterpret_cast<const uint4 *>(table + (int64_t)(ok ? id : 0) * hidden);
    uint4 *dst = reinterpret_cast<uint4 *>(out + (int64_t)t * hidden);
    for (int i = threadIdx.x; i < (hidden >> 3); i += 256) dst[i] = ok ? src[i] : make_uint4(0, 0, 0, 0);
}

}  // namespace

template <bool F16>
static int dense_gemv_launch_t(uint16_t *y, const uint16_t *x, const uint16_t *W, const uint16_t *bias, int M, int N, int K, hipStream_t st) {
    if (M >= 2 && N <= 16) {   // a few output columns: one workgroup per row
        hipLaunchKernelGGL((dense_fewcols_kernel<F16>), dim3(M), dim3(256), 0, st, y, x, W, bias, N, K);
        VMX_CHECK_LAUNCH("dense_fewcols");
        return VMX_OK;
    }
    static const int mfma_min_m = vmx_tune("VMX_DENSE_MFMA_MIN_M", 5);   // 0 disables
    if (mfma_min_m > 0 && M >= mfma_min_m && K % 128 == 0) {   // GEMM-shaped: matrix cores, activations staged in k-passes
        int g16 = ((N + 15) / 16 + 3) / 4;   // 4 waves x one 16-row tile of W per round
        if (g16 > 4096) g16 = 4096;   // one round of 16-row tiles for vocabularies up to 262144 (151936 rows at 16 activation rows: 286 -> 276 us)
        hipLaunchKernelGGL((dense_gemm16_kernel<F16>), dim3(g16), dim3(256), 0, st, y, x, W, bias, M, N, K);
        VMX_CHECK_LAUNCH("dense_gemm16");
        return VMX_OK;
    }
    const size_t lds = (size_t)M * K * 2;
    int grid = (N + 7) / 8;  // 4 waves x R=2 rows per workgroup
    if (grid > 4096) grid = 4096;
#define DL(MT, R)                                                                                                        \
    do {                                                                                                                 \
        if (lds > 64 * 1024)                                                                                             \
            VMX_CHECK_HIP(hipFuncSetAttribute((const void *)dense_gemv_kernel<MT, R, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              (int)lds));                                                                \
        hipLaunchKernelGGL((dense_gemv_kernel<MT, R, F16>), dim3(grid), dim3(256), lds, st, y, x, W, bias, M, N, K);     \
    } while (0)
    if (M == 1) DL(1, 2);
    else if (M == 2) DL(2, 2);
    else if (M <= 4) DL(4, 2);
    else if (M <= 8) DL(8, 2);
    else DL(16, 2);
#undef DL
    VMX_CHECK_LAUNCH("dense_gemv");
    return VMX_OK;
}

int vmx_dense_gemv_launch(uint16_t *y, const uint16_t *x, const uint16_t *W, const uint16_t *bias, int M, int N, int K,
                          hipStream_t st, int f16) {
    return f16 ? dense_gemv_launch_t<true>(y, x, W, bias, M, N, K, st) : dense_gemv_launch_t<false>(y, x, W, bias, M, N, K, st);
}

extern "C" int vmx_dense_gemv_bf16(uint16_t *y, const uint16_t *x, const uint16_t *W, const uint16_t *bias, int M, int N, int K,
                                   void *stream) {
    VMX_REQUIRE(y && x && W, "dense_gemv: null pointer");
    VMX_REQUIRE(M >= 1 && M <= 16, "dense_gemv: M (%d) must be in 1..16", M);
    VMX_REQUIRE(N > 0 && K > 0 && K % 8 == 0, "dense_gemv: K (%d) must be a positive multiple of 8", K);
    VMX_REQUIRE((size_t)M * K * 2 <= 160 * 1024, "dense_gemv: M*K too large for LDS staging");
    return vmx_dense_gemv_launch(y, x, W, bias, M, N, K, vmx_stream(stream), 0);
}

// The same product over fp16 operands (fp16 models: the lm_head of an EXL3 checkpoint), fp16 out
extern "C" int vmx_dense_gemv_f16(uint16_t *y, const uint16_t *x, const uint16_t *W, const uint16_t *bias, int M, int N, int K,
                                  void *stream) {
    VMX_REQUIRE(y && x && W, "dense_gemv: null pointer");
    VMX_REQUIRE(M >= 1 && M <= 16, "dense_gemv: M (%d) must be in 1..16", M);
    VMX_REQUIRE(N > 0 && K > 0 && K % 8 == 0, "dense_gemv: K (%d) must be a positive multiple of 8", K);
    VMX_REQUIRE((size_t)M * K * 2 <= 160 * 1024, "dense_gemv: M*K too large for LDS staging");
    return vmx_dense_gemv_launch(y, x, W, bias, M, N, K, vmx_stream(stream), 1);
}

extern "C" int vmx_embedding_lookup_bf16(uint16_t *out, const uint16_t *table, const int32_t *ids, int tokens, int hidden, int vocab,
                                         void *stream) {
    VMX_REQUIRE(out && table && ids, "embedding_lookup: null pointer");
    VMX_REQUIRE(hidden > 0 && hidden % 8 == 0 && vocab > 0, "embedding_lookup: hidden (%d) must be a multiple of 8", hidden);
    if (tokens <= 0) return VMX_OK;
    hipLaunchKernelGGL(embedding_lookup_kernel, dim3(tokens), dim3(256), 0, vmx_stream(stream), out, table, ids, hidden, vocab);
    VMX_CHECK_LAUNCH("embedding_lookup");
    return VMX_OK;
}

// Dense bf16 GEMV for small M (lm_head / tied embedding head on the decode path).  The reference
// sends this product to cuBLAS through Candle (crates/core/src/models/qwen3_quantized.rs:1258-1302,
// cuda_kernels.rs half_matmul_pooled): y[m, n] = bf16(sum_k x[m, k] * W[n, k]), f32 accumulation.
// gfx950 form: one wave per output row, the row streamed with 16-byte loads (1 KiB per
// wave-instruction), v_dot2c_f32_bf16 on packed pairs, x rows held in LDS, wave64 butterfly.
#include "vmx_internal.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// grid-stride over rows: wave w of workgroup b takes rows (b*4 + w) * R + r.  K % 512 == 0 is not
// required: the tail is handled with a bounds check on the 8-element chunk index.
template <int MT, int R>
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
                        a = dot2(wv[r].x, xv.x, a);
                        a = dot2(wv[r].y, xv.y, a);
                        a = dot2(wv[r].z, xv.z, a);
                        a = dot2(wv[r].w, xv.w, a);
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
                    if (bias) v += vmx_bf16_to_f32(bias[row0 + r]);
                    y[(int64_t)m * N + row0 + r] = vmx_f32_to_bf16(v);
                }
            }
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

int vmx_dense_gemv_launch(uint16_t *y, const uint16_t *x, const uint16_t *W, const uint16_t *bias, int M, int N, int K,
                          hipStream_t st) {
    const size_t lds = (size_t)M * K * 2;
    int grid = (N + 7) / 8;  // 4 waves x R=2 rows per workgroup
    if (grid > 4096) grid = 4096;
#define DL(MT, R)                                                                                                        \
    do {                                                                                                                 \
        if (lds > 64 * 1024)                                                                                             \
            VMX_CHECK_HIP(hipFuncSetAttribute((const void *)dense_gemv_kernel<MT, R>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              (int)lds));                                                                \
        hipLaunchKernelGGL((dense_gemv_kernel<MT, R>), dim3(grid), dim3(256), lds, st, y, x, W, bias, M, N, K);          \
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

extern "C" int vmx_dense_gemv_bf16(uint16_t *y, const uint16_t *x, const uint16_t *W, const uint16_t *bias, int M, int N, int K,
                                   void *stream) {
    VMX_REQUIRE(y && x && W, "dense_gemv: null pointer");
    VMX_REQUIRE(M >= 1 && M <= 16, "dense_gemv: M (%d) must be in 1..16", M);
    VMX_REQUIRE(N > 0 && K > 0 && K % 8 == 0, "dense_gemv: K (%d) must be a positive multiple of 8", K);
    VMX_REQUIRE((size_t)M * K * 2 <= 160 * 1024, "dense_gemv: M*K too large for LDS staging");
    return vmx_dense_gemv_launch(y, x, W, bias, M, N, K, vmx_stream(stream));
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

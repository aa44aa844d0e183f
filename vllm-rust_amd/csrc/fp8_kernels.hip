// FP8 (OCP e4m3fn) weight path for gfx950: W8A16 small-M GEMM, weight dequant, static / dynamic
// per-token activation quantisation and dequantisation.
// Semantics: crates/core/kernels/fp8_gemm.cu (fp8_gemm_bf16_* :38-218, fp8_dequant_weights_bf16 :222-243,
// fp8_quantize_input_bf16 :246-330, fp8_scaled_mm_* :332-466) and crates/core/kernels/fp8_quant.cu
// (fp8_static_quant_bf16 :89-112, fp8_dynamic_per_token_quant_bf16 :190-237, fp8_dequant_bf16 :239-262).
// gfx950 converts OCP e4m3 in hardware (v_cvt_pk_f32_fp8 / v_cvt_pk_fp8_f32, round-to-nearest-even);
// saturation to +-448 (CUDA's __nv_fp8_e4m3(float) is satfinite) is applied explicitly before the convert.
#include "vmx_internal.h"
#include <float.h>

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ __forceinline__ uint8_t to_e4m3_sat(float v) {
    v = fminf(fmaxf(v, -448.0f), 448.0f);  // NaN passes through both (fminf/fmaxf return the other operand: keep NaN below)
    const uint32_t p = __builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false);
    return (uint8_t)(p & 0xFFu);
}

__device__ __forceinline__ void e4m3x4_to_f32(uint32_t w, float f[4]) {
    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
    f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}

// y[m, n] = bf16(scale_n * sum_k x[m,k] * e4m3(W[n,k]) + bias[n]).  One wave per R output rows, 16 weights
// (16 B) per lane per step, x rows in LDS as f32.  (The reference multiplies every weight by the scale
// before the product; factoring the per-column scale out only re-associates the f32 sum.)
template <int MT, int R>
__global__ __launch_bounds__(256) void fp8_gemv_kernel(uint16_t *__restrict__ y, const uint16_t *__restrict__ x,
                                                        const uint8_t *__restrict__ W, const float *__restrict__ scale,
                                                        const uint16_t *__restrict__ bias, int M, int N, int K, int per_channel) {
    extern __shared__ __align__(16) unsigned char smem[];
    float *xs = reinterpret_cast<float *>(smem);  // [M][K]
    for (int i = threadIdx.x; i < M * K; i += 256) xs[i] = vmx_bf16_to_f32(x[i]);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = K >> 4;
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
            for (int r = 0; r < R; r++) wv[r] = reinterpret_cast<const uint4 *>(W + (int64_t)min(row0 + r, N - 1) * K)[c];
#pragma unroll
            for (int r = 0; r < R; r++) {
                float wf[16];
                e4m3x4_to_f32(wv[r].x, wf); e4m3x4_to_f32(wv[r].y, wf + 4); e4m3x4_to_f32(wv[r].z, wf + 8); e4m3x4_to_f32(wv[r].w, wf + 12);
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    if (m < M) {
                        const float4 *xp = reinterpret_cast<const float4 *>(xs + m * K + 16 * c);
                        float a = acc[r][m];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float4 xv = xp[j];
                            a = fmaf(wf[4 * j], xv.x, a); a = fmaf(wf[4 * j + 1], xv.y, a);
                            a = fmaf(wf[4 * j + 2], xv.z, a); a = fmaf(wf[4 * j + 3], xv.w, a);
                        }
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
                    v *= per_channel ? scale[row0 + r] : scale[0];
                    if (bias) v += vmx_bf16_to_f32(bias[row0 + r]);
                    y[(int64_t)m * N + row0 + r] = vmx_f32_to_bf16(v);
                }
            }
    }
}

// out[n, k] = bf16(e4m3(W[n,k]) * scale)
__global__ __launch_bounds__(256) void fp8_dequant_kernel(uint16_t *__restrict__ out, const uint8_t *__restrict__ in,
                                                           const float *__restrict__ scale, int64_t rows, int64_t cols, int per_row) {
    const int64_t total4 = rows * cols / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const float s = per_row ? scale[(i * 4) / cols] : scale[0];
        float f[4];
        e4m3x4_to_f32(reinterpret_cast<const uint32_t *>(in)[i], f);
        const uint32_t lo = (uint32_t)vmx_f32_to_bf16(f[0] * s) | ((uint32_t)vmx_f32_to_bf16(f[1] * s) << 16);
        const uint32_t hi = (uint32_t)vmx_f32_to_bf16(f[2] * s) | ((uint32_t)vmx_f32_to_bf16(f[3] * s) << 16);
        reinterpret_cast<uint2 *>(out)[i] = make_uint2(lo, hi);
    }
}

// One workgroup per token.  dynamic: scale = max(absmax/448, 1e-12) written to scales[token];
// static: scale read from scale[token] or scale[0].  out = e4m3(x * (1/scale)), saturating.
__global__ __launch_bounds__(256) void fp8_quant_kernel(uint8_t *__restrict__ out, float *__restrict__ scales_out,
                                                         const uint16_t *__restrict__ in, const float *__restrict__ scale_in,
                                                         int hidden, int dynamic, int per_token) {
    __shared__ float red[4];
    const int t = blockIdx.x;
    const uint16_t *row = in + (int64_t)t * hidden;
    float sc;
    if (dynamic) {
        float mx = 0.f;
        for (int i = threadIdx.x; i < hidden; i += 256) mx = fmaxf(mx, fabsf(vmx_bf16_to_f32(row[i])));
        mx = vmx_wave_max(mx);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        sc = fmaxf(mx / 448.0f, 1e-12f);
        if (threadIdx.x == 0) scales_out[t] = sc;
    } else {
        sc = per_token ? scale_in[t] : scale_in[0];
    }
    const float inv = 1.0f / sc;
    for (int i = threadIdx.x; i < hidden; i += 256) out[(int64_t)t * hidden + i] = to_e4m3_sat(vmx_bf16_to_f32(row[i]) * inv);
}

}  // namespace

extern "C" int vmx_fp8_gemm_bf16(uint16_t *y, const uint16_t *x, const void *weight, const float *scale, const uint16_t *bias, int M,
                                 int N, int K, int per_channel_scale, void *stream) {
    VMX_REQUIRE(y && x && weight && scale, "fp8_gemm: null pointer");
    VMX_REQUIRE(M >= 1 && M <= 16, "fp8_gemm: M (%d) must be in 1..16 (larger M takes the dequant + GEMM path)", M);
    VMX_REQUIRE(N > 0 && K > 0 && K % 16 == 0, "fp8_gemm: K (%d) must be a positive multiple of 16", K);
    const size_t lds = (size_t)M * K * 4;
    VMX_REQUIRE(lds <= 160 * 1024, "fp8_gemm: M*K too large for LDS staging");
    int grid = (N + 7) / 8;
    if (grid > 4096) grid = 4096;
    hipStream_t st = vmx_stream(stream);
#define FL(MT)                                                                                                          \
    do {                                                                                                                \
        if (lds > 64 * 1024)                                                                                            \
            VMX_CHECK_HIP(hipFuncSetAttribute((const void *)fp8_gemv_kernel<MT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              (int)lds));                                                               \
        hipLaunchKernelGGL((fp8_gemv_kernel<MT, 2>), dim3(grid), dim3(256), lds, st, y, x, (const uint8_t *)weight, scale, bias, M, N, \
                           K, per_channel_scale);                                                                       \
    } while (0)
    if (M == 1) FL(1);
    else if (M == 2) FL(2);
    else if (M <= 4) FL(4);
    else if (M <= 8) FL(8);
    else FL(16);
#undef FL
    VMX_CHECK_LAUNCH("fp8_gemm");
    return VMX_OK;
}

extern "C" int vmx_fp8_dequant_bf16(uint16_t *out, const void *in, const float *scale, int64_t rows, int64_t cols, int per_row_scale,
                                    void *stream) {
    VMX_REQUIRE(out && in && scale, "fp8_dequant: null pointer");
    VMX_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0, "fp8_dequant: cols (%lld) must be a positive multiple of 4", (long long)cols);
    if (rows == 0) return VMX_OK;
    int64_t blocks = (rows * cols / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fp8_dequant_kernel, dim3((unsigned)blocks), dim3(256), 0, vmx_stream(stream), out, (const uint8_t *)in, scale, rows,
                       cols, per_row_scale);
    VMX_CHECK_LAUNCH("fp8_dequant");
    return VMX_OK;
}

extern "C" int vmx_fp8_quant_bf16(void *out, float *scales_out, const uint16_t *in, const float *scale_in, int tokens, int hidden,
                                  int dynamic_per_token, int per_token_scale, void *stream) {
    VMX_REQUIRE(out && in, "fp8_quant: null pointer");
    VMX_REQUIRE(dynamic_per_token ? scales_out != nullptr : scale_in != nullptr, "fp8_quant: scale pointer missing");
    VMX_REQUIRE(tokens >= 0 && hidden > 0, "fp8_quant: bad shape");
    if (tokens == 0) return VMX_OK;
    hipLaunchKernelGGL(fp8_quant_kernel, dim3(tokens), dim3(256), 0, vmx_stream(stream), (uint8_t *)out, scales_out, in, scale_in, hidden,
                       dynamic_per_token, per_token_scale);
    VMX_CHECK_LAUNCH("fp8_quant");
    return VMX_OK;
}

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

// ---- matrix-core GEMM over fp8 / bf16 operands: W8A8 scaled_mm, W8A16 for any M, and the dense GEMM behind block-scaled weights ----
// out[m, n] = epilogue(sum_k A[m,k] * B[n,k]);  A [M, K] and B [N, K] row-major, each fp8 e4m3 or bf16.
// Workgroup = 8 waves on one 16-column tile of B and MB 16-row blocks of A; the waves split K in interleaved 64-deep chunks and
// merge through LDS.  Lane (i = lane & 15, g = lane >> 4) reads 16 consecutive k of row i at k = chunk * 64 + 16 g: one 16-byte
// load for fp8, two for bf16; the two halves are the operands of two 16x16x32 MFMAs (both operands carry the same k, which is all
// the instruction needs).  fp8 x fp8 runs on v_mfma_f32_16x16x32_fp8_fp8; a bf16 operand makes the other one bf16 too (fp8 -> bf16
// is exact) and runs v_mfma_f32_16x16x32_bf16.  Products are exact in f32 in both forms; only the f32 summation order differs
// from a scalar loop.
typedef float fp8mm_f32x4 __attribute__((ext_vector_type(4)));

struct Fp8MmEpilogue {
    const float *a_scale, *b_scale;   // null, [1] or per row of the operand
    int a_per_row, b_per_row;
    const uint16_t *bias;             // bf16 [N] or null
    int round_before_bias;            // bf16(bf16(acc) + bias): a bf16 matmul followed by a bf16 broadcast add
    int out_f32;
};

__device__ __forceinline__ uint4 fp8x8_to_bf16x8(uint32_t lo, uint32_t hi) {
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    auto pk = [](float x, float y) { return (__float_as_uint(x) >> 16) | (__float_as_uint(y) & 0xFFFF0000u); };
    return make_uint4(pk(a[0], a[1]), pk(b[0], b[1]), pk(c[0], c[1]), pk(d[0], d[1]));
}

template <bool A_FP8, bool B_FP8, int MB>
__global__ __launch_bounds__(512) void fp8_mm_kernel(void *__restrict__ out, const void *__restrict__ A, const void *__restrict__ B, int M,
                                                      int N, int K, Fp8MmEpilogue ep) {
    constexpr int NW = 8;
    __shared__ fp8mm_f32x4 red[NW][MB][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MB);
    const int64_t brow = (int64_t)min(n0 + i, N - 1) * K;
    int64_t arow[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) arow[mb] = (int64_t)min(m0 + mb * 16 + i, M - 1) * K;
    fp8mm_f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) acc[mb] = fp8mm_f32x4{0.f, 0.f, 0.f, 0.f};
    const int chunks = (K + 63) >> 6;
    for (int ch = wave; ch < chunks; ch += NW) {
        const int k = ch * 64 + 16 * g;
        const bool live = k < K;                 // K % 16 == 0: a lane's 16 k are all inside or all outside
        const int kc = live ? k : 0;
        uint4 b0, b1;                            // bf16 x 8 each, or (fp8 x fp8) b0 = the 16 fp8 bytes
        if (B_FP8) {
            const uint4 raw = *reinterpret_cast<const uint4 *>((const uint8_t *)B + brow + kc);
            if (A_FP8) b0 = raw;
            else { b0 = fp8x8_to_bf16x8(raw.x, raw.y); b1 = fp8x8_to_bf16x8(raw.z, raw.w); }
        } else {
            const uint4 *p = reinterpret_cast<const uint4 *>((const uint16_t *)B + brow + kc);
            b0 = p[0]; b1 = p[1];
        }
        if (!live) { b0 = make_uint4(0, 0, 0, 0); b1 = b0; }
#pragma unroll
        for (int mb = 0; mb < MB; mb++) {
            if (A_FP8 && B_FP8) {
                uint4 a = *reinterpret_cast<const uint4 *>((const uint8_t *)A + arow[mb] + kc);
                if (!live) a = make_uint4(0, 0, 0, 0);   // a NaN byte times the zeroed B operand would still be NaN
                const long a_lo = (long)(((unsigned long)a.y << 32) | a.x), a_hi = (long)(((unsigned long)a.w << 32) | a.z);
                const long b_lo = (long)(((unsigned long)b0.y << 32) | b0.x), b_hi = (long)(((unsigned long)b0.w << 32) | b0.z);
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a_lo, b_lo, acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a_hi, b_hi, acc[mb], 0, 0, 0);
            } else {
                uint4 a0, a1;
                if (A_FP8) {
                    const uint4 raw = *reinterpret_cast<const uint4 *>((const uint8_t *)A + arow[mb] + kc);
                    a0 = fp8x8_to_bf16x8(raw.x, raw.y); a1 = fp8x8_to_bf16x8(raw.z, raw.w);
                } else {
                    const uint4 *p = reinterpret_cast<const uint4 *>((const uint16_t *)A + arow[mb] + kc);
                    a0 = p[0]; a1 = p[1];
                }
                if (!live) { a0 = make_uint4(0, 0, 0, 0); a1 = a0; }
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, a0), __builtin_bit_cast(vmx_bf16x8, b0), acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, a1), __builtin_bit_cast(vmx_bf16x8, b1), acc[mb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; mb++) red[wave][mb][lane] = acc[mb];
    __syncthreads();
    for (int idx = threadIdx.x; idx < MB * 64; idx += 512) {
        const int mb = idx >> 6, l = idx & 63;
        fp8mm_f32x4 v = red[0][mb][l];
#pragma unroll
        for (int w = 1; w < NW; w++) v += red[w][mb][l];
        const int col = n0 + (l & 15);
        if (col >= N) continue;
        const float sb = ep.b_scale ? (ep.b_per_row ? ep.b_scale[col] : ep.b_scale[0]) : 1.0f;
        const float bias = ep.bias ? vmx_bf16_to_f32(ep.bias[col]) : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = m0 + mb * 16 + 4 * (l >> 4) + j;
            if (row >= M) continue;
            const float sa = ep.a_scale ? (ep.a_per_row ? ep.a_scale[row] : ep.a_scale[0]) : 1.0f;
            float r = v[j];
            if (ep.a_scale) r *= sa;
            if (ep.b_scale) r *= sb;
            if (ep.round_before_bias) r = vmx_bf16_to_f32(vmx_f32_to_bf16(r));
            if (ep.bias) r += bias;
            if (ep.out_f32) ((float *)out)[(int64_t)row * N + col] = r;
            else ((uint16_t *)out)[(int64_t)row * N + col] = vmx_f32_to_bf16(r);
        }
    }
}

template <bool A_FP8, bool B_FP8>
void fp8_mm_launch(void *out, const void *A, const void *B, int M, int N, int K, const Fp8MmEpilogue &ep, hipStream_t st) {
    const int nt = (N + 15) / 16;
    if (M <= 16) hipLaunchKernelGGL((fp8_mm_kernel<A_FP8, B_FP8, 1>), dim3(nt, 1), dim3(512), 0, st, out, A, B, M, N, K, ep);
    else if (M <= 32) hipLaunchKernelGGL((fp8_mm_kernel<A_FP8, B_FP8, 2>), dim3(nt, 1), dim3(512), 0, st, out, A, B, M, N, K, ep);
    else hipLaunchKernelGGL((fp8_mm_kernel<A_FP8, B_FP8, 4>), dim3(nt, (M + 63) / 64), dim3(512), 0, st, out, A, B, M, N, K, ep);
}

// out[n, k] = bf16(e4m3(W[n,k]) * scale[n / block_n][k / block_k])
__global__ __launch_bounds__(256) void fp8_dequant_block_kernel(uint16_t *__restrict__ out, const uint8_t *__restrict__ in,
                                                                 const float *__restrict__ scale, int64_t rows, int64_t cols, int block_n,
                                                                 int block_k, int scale_cols) {
    const int64_t total4 = rows * cols / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t r = (i * 4) / cols, c = (i * 4) % cols;
        float f[4];
        e4m3x4_to_f32(reinterpret_cast<const uint32_t *>(in)[i], f);
        uint16_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = vmx_f32_to_bf16(f[j] * scale[(r / block_n) * scale_cols + (c + j) / block_k]);
        reinterpret_cast<uint2 *>(out)[i] = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
    }
}

// fused_rmsnorm_fp8 (crates/core/kernels/layernorm_quant.cu:33-134): one workgroup per row; normed = x * rsqrt(mean(x^2) + eps) * w
// kept in f32 (no bf16 rounding), scale = max(absmax(normed) / 448, 1e-12), out = e4m3(normed * (1 / scale)).
__global__ __launch_bounds__(256) void fused_rmsnorm_fp8_kernel(uint8_t *__restrict__ out, float *__restrict__ scales,
                                                                 const uint16_t *__restrict__ in, const uint16_t *__restrict__ w, int hidden,
                                                                 float eps) {
    __shared__ float red[4];
    const int64_t base = (int64_t)blockIdx.x * hidden;
    float ss = 0.f;
    for (int i = threadIdx.x; i < hidden; i += 256) {
        const float v = vmx_bf16_to_f32(in[base + i]);
        ss += v * v;
    }
    ss = vmx_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rms = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)hidden + eps);
    __syncthreads();
    float mx = 0.f;
    for (int i = threadIdx.x; i < hidden; i += 256) mx = fmaxf(mx, fabsf(vmx_bf16_to_f32(in[base + i]) * rms * vmx_bf16_to_f32(w[i])));
    mx = vmx_wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    const float sc = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) / 448.0f, 1e-12f);
    const float inv = 1.0f / sc;
    for (int i = threadIdx.x; i < hidden; i += 256)
        out[base + i] = to_e4m3_sat(vmx_bf16_to_f32(in[base + i]) * rms * vmx_bf16_to_f32(w[i]) * inv);
    if (threadIdx.x == 0) scales[blockIdx.x] = sc;
}

}  // namespace

// y [M, N] = bf16(bf16(x [M, K] @ W [N, K]^T) + bias) on the bf16 matrix cores (the GEMM behind every "dequantise, then multiply"
// linear of this library); K % 16 == 0.
int vmx_bf16_gemm_nt(uint16_t *y, const uint16_t *x, const uint16_t *w, const uint16_t *bias, int M, int N, int K, hipStream_t st) {
    Fp8MmEpilogue ep = {nullptr, nullptr, 0, 0, bias, 1, 0};
    fp8_mm_launch<false, false>(y, x, w, M, N, K, ep, st);
    VMX_CHECK_LAUNCH("bf16_gemm_nt");
    return VMX_OK;
}

extern "C" int vmx_fp8_gemm_bf16(uint16_t *y, const uint16_t *x, const void *weight, const float *scale, const uint16_t *bias, int M,
                                 int N, int K, int per_channel_scale, void *stream) {
    VMX_REQUIRE(y && x && weight && scale, "fp8_gemm: null pointer");
    VMX_REQUIRE(M >= 1, "fp8_gemm: M (%d) must be positive", M);
    VMX_REQUIRE(N > 0 && K > 0 && K % 16 == 0, "fp8_gemm: K (%d) must be a positive multiple of 16", K);
    const size_t lds = (size_t)M * K * 4;
    if (M > 16 || lds > 160 * 1024) {   // prompt-sized M (or rows too long for the LDS-staged kernel): bf16 matrix cores over fp8 -> bf16 weights
        Fp8MmEpilogue ep = {nullptr, scale, 0, per_channel_scale, bias, 0, 0};
        fp8_mm_launch<false, true>(y, x, weight, M, N, K, ep, vmx_stream(stream));
        VMX_CHECK_LAUNCH("fp8_gemm (matrix-core path)");
        return VMX_OK;
    }
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

extern "C" int vmx_fp8_scaled_mm(void *out, int out_f32, const void *a, const void *b, const float *a_scale, const float *b_scale,
                                 const uint16_t *bias, int M, int N, int K, int a_scale_mode, int b_scale_mode, void *stream) {
    VMX_REQUIRE(out && a && b && a_scale && b_scale, "fp8_scaled_mm: null pointer");
    VMX_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 16 == 0, "fp8_scaled_mm: K (%d) must be a positive multiple of 16 (M %d, N %d)", K, M, N);
    VMX_REQUIRE((a_scale_mode == 0 || a_scale_mode == 1) && (b_scale_mode == 0 || b_scale_mode == 1),
                "fp8_scaled_mm: scale modes are 0 (per tensor) or 1 (per row / per column), got %d / %d", a_scale_mode, b_scale_mode);
    VMX_REQUIRE(!(out_f32 && bias), "fp8_scaled_mm: the f32-output form takes no bias");
    if (M == 0) return VMX_OK;
    Fp8MmEpilogue ep = {a_scale, b_scale, a_scale_mode, b_scale_mode, bias, 0, out_f32};
    fp8_mm_launch<true, true>(out, a, b, M, N, K, ep, vmx_stream(stream));
    VMX_CHECK_LAUNCH("fp8_scaled_mm");
    return VMX_OK;
}

extern "C" int vmx_fp8_dequant_block_bf16(uint16_t *out, const void *in, const float *scale, int64_t rows, int64_t cols, int block_n,
                                          int block_k, void *stream) {
    VMX_REQUIRE(out && in && scale, "fp8_dequant_block: null pointer");
    VMX_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0, "fp8_dequant_block: cols (%lld) must be a positive multiple of 4", (long long)cols);
    VMX_REQUIRE(block_n > 0 && block_k > 0, "fp8_dequant_block: block sizes must be non-zero, got [%d, %d]", block_n, block_k);
    if (rows == 0) return VMX_OK;
    int64_t blocks = (rows * cols / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fp8_dequant_block_kernel, dim3((unsigned)blocks), dim3(256), 0, vmx_stream(stream), out, (const uint8_t *)in, scale,
                       rows, cols, block_n, block_k, (int)((cols + block_k - 1) / block_k));
    VMX_CHECK_LAUNCH("fp8_dequant_block");
    return VMX_OK;
}

extern "C" int vmx_fp8_block_gemm_bf16(uint16_t *y, const uint16_t *x, const void *weight, const float *block_scale, const uint16_t *bias,
                                       int M, int N, int K, int block_n, int block_k, uint16_t *workspace, void *stream) {
    VMX_REQUIRE(y && x && weight && block_scale && workspace, "fp8_block_gemm: null pointer");
    VMX_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 16 == 0, "fp8_block_gemm: K (%d) must be a positive multiple of 16", K);
    if (int rc = vmx_fp8_dequant_block_bf16(workspace, weight, block_scale, N, K, block_n, block_k, stream)) return rc;
    if (M == 0) return VMX_OK;
    Fp8MmEpilogue ep = {nullptr, nullptr, 0, 0, bias, 1, 0};
    fp8_mm_launch<false, false>(y, x, workspace, M, N, K, ep, vmx_stream(stream));
    VMX_CHECK_LAUNCH("fp8_block_gemm");
    return VMX_OK;
}

extern "C" int vmx_fused_rmsnorm_fp8(void *out, float *scales, const uint16_t *input, const uint16_t *weight, int rows, int hidden,
                                     float eps, void *stream) {
    VMX_REQUIRE(out && scales && input && weight, "fused_rmsnorm_fp8: null pointer");
    VMX_REQUIRE(rows >= 0 && hidden > 0, "fused_rmsnorm_fp8: bad shape [%d, %d]", rows, hidden);
    if (rows == 0) return VMX_OK;
    hipLaunchKernelGGL(fused_rmsnorm_fp8_kernel, dim3(rows), dim3(256), 0, vmx_stream(stream), (uint8_t *)out, scales, input, weight, hidden,
                       eps);
    VMX_CHECK_LAUNCH("fused_rmsnorm_fp8");
    return VMX_OK;
}

// Prompt-sized linears over GGUF I-quant weights.  Beyond IQ_GEMV_MAX_M = 16 rows the reference dequantises the whole weight to
// dense f32 and runs one f32 GEMM on the f32 activations (IqLinear::forward_inner, crates/core/src/quantization/gguf/mod.rs:346-365):
// no q8_1 rounding of the activations on that path.  gfx950 has no fast f32 matrix instruction, but the activations are bf16 values
// and a dequantised weight w = d * scale * grid needs at most ~16 mantissa bits, so w is split as hi = bf16(w), lo = bf16(w - hi) and
// the product is two bf16 MFMA passes into one f32 accumulator: x * hi and x * lo are exact in f32, and hi + lo reproduces w to
// 2^-17 relative -- below the f32 summation noise of a dot over thousands of terms.
#include <algorithm>
#include "vmx_internal.h"
#include "vmx_kernels.h"

namespace {

typedef float sg_f32x4 __attribute__((ext_vector_type(4)));

// C[m, n] (f32, row stride ldc) = sum_k A[m,k] * (Bhi[n,k] + Blo[n,k]);  A [M, K] bf16, Bhi / Blo [N, K] bf16, K % 32 == 0.
// Workgroup = 4 waves side by side on a 64 x 256 tile of C; a wave owns 64 rows x 64 columns = 4 x 4 accumulators of
// v_mfma_f32_16x16x32_bf16.  Lane (i = lane & 15, g = lane >> 4) feeds row i of a 16-row block with the 8 values k0 + 8 g .. + 7,
// read straight from global memory as one 16-byte load per operand block (the four waves of a workgroup re-read the A rows
// through the vector L1).  No LDS, no barriers: out-of-range rows and columns read a clamped address and are not stored.
// KSPLIT: few output tiles (down_proj, o_proj: N = hidden) would leave most compute units idle, so the four waves of a workgroup
// share ONE 64 x 64 tile and take a quarter of K each; the quarters are added through LDS in wave order (deterministic).
template <bool LO, bool KSPLIT>
__global__ __launch_bounds__(256) void split_gemm_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ Bhi,
                                                          const uint16_t *__restrict__ Blo, float *__restrict__ C, int M, int N, int K,
                                                          int ldc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = KSPLIT ? blockIdx.x * 64 : blockIdx.x * 256 + wave * 64;
    if (!KSPLIT && n0 >= N) return;
    const int kq = K / 4;   // KSPLIT: K % 128 == 0
    const int k_begin = KSPLIT ? wave * kq : 0, k_end = KSPLIT ? k_begin + kq : K;
    int64_t arow[4], brow[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        arow[b] = (int64_t)min(m0 + b * 16 + i, M - 1) * K + 8 * g;
        brow[b] = (int64_t)min(n0 + b * 16 + i, N - 1) * K + 8 * g;
    }
    sg_f32x4 acc[4][4];
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int cb = 0; cb < 4; cb++) acc[rb][cb] = sg_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
        uint4 a[4], bh[4], bl[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            a[b] = *reinterpret_cast<const uint4 *>(A + arow[b] + k0);
            bh[b] = *reinterpret_cast<const uint4 *>(Bhi + brow[b] + k0);
            if (LO) bl[b] = *reinterpret_cast<const uint4 *>(Blo + brow[b] + k0);
        }
#pragma unroll
        for (int cb = 0; cb < 4; cb++)
#pragma unroll
            for (int rb = 0; rb < 4; rb++) {
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, a[rb]), __builtin_bit_cast(vmx_bf16x8, bh[cb]),
                                                                      acc[rb][cb], 0, 0, 0);
                if (LO)
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, a[rb]),
                                                                          __builtin_bit_cast(vmx_bf16x8, bl[cb]), acc[rb][cb], 0, 0, 0);
            }
    }
    if constexpr (KSPLIT) {
        // wave w keeps row block w: it adds the four waves' sums of that block in wave order
        __shared__ sg_f32x4 red[4][4][4][64];   // [source wave][row block][column block][lane]: 64 KiB
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int cb = 0; cb < 4; cb++) red[wave][rb][cb][lane] = acc[rb][cb];
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            sg_f32x4 t = red[0][wave][cb][lane];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                const sg_f32x4 o = red[w][wave][cb][lane];
                t[0] += o[0]; t[1] += o[1]; t[2] += o[2]; t[3] += o[3];
            }
            const int col = n0 + cb * 16 + i;
            if (col >= N) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = m0 + wave * 16 + 4 * g + j;
                if (row < M) C[(int64_t)row * ldc + col] = t[j];
            }
        }
        return;
    }
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            const int col = n0 + cb * 16 + i;
            if (col >= N) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = m0 + rb * 16 + 4 * g + j;
                if (row < M) C[(int64_t)row * ldc + col] = acc[rb][cb][j];
            }
        }
}

// The one-pass form for weights that are (f32 scale per super-block) x (small integer), see iq_dequant_int_kernel:
// C[m, n] = sum_sb Dt[sb][n] * ( sum_{k in sb} A16[m, k] * Q[n, k] ), A16 = the activations as fp16 (bf16 values convert exactly
// inside fp16's range), Q fp16 integers: every product is exact, the 256 products of a super-block are added in f32 by the matrix
// core, and the scale is applied once per super-block to the finished sum -- the reference's f32 GEMM on the dequantised
// weight up to summation order, at half the matrix instructions of the hi / lo split.  Same tiling as split_gemm_kernel;
// KSPLIT: the four waves of a workgroup share one 64 x 64 tile and take the super-block ranges [nsb w / 4, nsb (w + 1) / 4).
template <bool KSPLIT>
__global__ __launch_bounds__(256) void int_gemm_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ Q,
                                                        const float *__restrict__ Dt, float *__restrict__ C, int M, int N, int K, int ldc) {
    typedef _Float16 sg_f16x8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = KSPLIT ? blockIdx.x * 64 : blockIdx.x * 256 + wave * 64;
    if (!KSPLIT && n0 >= N) return;
    const int nsb = K >> 8;
    const int sb_begin = KSPLIT ? (nsb * wave) / 4 : 0, sb_end = KSPLIT ? (nsb * (wave + 1)) / 4 : nsb;
    int64_t arow[4], brow[4];
    int dcol[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        arow[b] = (int64_t)min(m0 + b * 16 + i, M - 1) * K + 8 * g;
        brow[b] = (int64_t)min(n0 + b * 16 + i, N - 1) * K + 8 * g;
        dcol[b] = min(n0 + b * 16 + i, N - 1);
    }
    sg_f32x4 acc[4][4];
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int cb = 0; cb < 4; cb++) acc[rb][cb] = sg_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int sb = sb_begin; sb < sb_end; sb++) {
        float dsc[4];
#pragma unroll
        for (int cb = 0; cb < 4; cb++) dsc[cb] = Dt[(int64_t)sb * N + dcol[cb]];
        sg_f32x4 tmp[4][4];
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int cb = 0; cb < 4; cb++) tmp[rb][cb] = sg_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int k0 = sb * 256; k0 < sb * 256 + 256; k0 += 32) {
            uint4 a[4], q[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                a[b] = *reinterpret_cast<const uint4 *>(A + arow[b] + k0);
                q[b] = *reinterpret_cast<const uint4 *>(Q + brow[b] + k0);
            }
#pragma unroll
            for (int cb = 0; cb < 4; cb++)
#pragma unroll
                for (int rb = 0; rb < 4; rb++)
                    tmp[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sg_f16x8, a[rb]), __builtin_bit_cast(sg_f16x8, q[cb]),
                                                                         tmp[rb][cb], 0, 0, 0);
        }
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int cb = 0; cb < 4; cb++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[rb][cb][j] = fmaf(dsc[cb], tmp[rb][cb][j], acc[rb][cb][j]);
    }
    if constexpr (KSPLIT) {
        __shared__ sg_f32x4 red[4][4][4][64];   // [source wave][row block][column block][lane]: 64 KiB
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int cb = 0; cb < 4; cb++) red[wave][rb][cb][lane] = acc[rb][cb];
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            sg_f32x4 t = red[0][wave][cb][lane];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                const sg_f32x4 o = red[w][wave][cb][lane];
                t[0] += o[0]; t[1] += o[1]; t[2] += o[2]; t[3] += o[3];
            }
            const int col = n0 + cb * 16 + i;
            if (col >= N) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = m0 + wave * 16 + 4 * g + j;
                if (row < M) C[(int64_t)row * ldc + col] = t[j];
            }
        }
        return;
    }
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            const int col = n0 + cb * 16 + i;
            if (col >= N) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = m0 + rb * 16 + 4 * g + j;
                if (row < M) C[(int64_t)row * ldc + col] = acc[rb][cb][j];
            }
        }
}

// bf16 -> fp16, saturating at the largest finite half (a bf16 value inside fp16's normal range converts exactly; smaller ones
// lose bits that are below 2^-24 in absolute terms)
__global__ __launch_bounds__(256) void bf16_to_f16_kernel(uint16_t *__restrict__ out, const uint16_t *__restrict__ in, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const uint4 v = reinterpret_cast<const uint4 *>(in)[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float a = fminf(fmaxf(__uint_as_float(w[j] << 16), -65504.0f), 65504.0f);
            const float b = fminf(fmaxf(__uint_as_float(w[j] & 0xFFFF0000u), -65504.0f), 65504.0f);
            o[j] = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)b) << 16);
        }
        reinterpret_cast<uint4 *>(out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// f32 rows [rows, K] -> bf16 rows [rows, 3K]: v = hi + lo with hi = bf16(v), lo = bf16(v - hi).  WEIGHT rows are laid out
// hi | hi | lo and ACTIVATION rows hi | lo | hi, so that one bf16 GEMM over 3K computes xhi.whi + xlo.whi + xhi.wlo: the
// f32 product short of the lo.lo term (2^-16 relative per factor pair, below the f32 summation noise of a long dot).
template <bool WEIGHT>
__global__ __launch_bounds__(256) void split3_rows_kernel(uint16_t *__restrict__ out, const float *__restrict__ in, int64_t rows, int K) {
    const int64_t n = rows * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / K;
        const int k = (int)(i - r * K);
        const float v = in[i];
        const uint16_t hi = vmx_f32_to_bf16(v), lo = vmx_f32_to_bf16(v - vmx_bf16_to_f32(hi));
        uint16_t *o = out + r * 3 * K;
        o[k] = hi;
        o[K + k] = WEIGHT ? hi : lo;
        o[2 * K + k] = WEIGHT ? lo : hi;
    }
}

}  // namespace

// QMatMul::forward beyond the 8-row vector kernels (candle's dequantize-and-matmul path behind QuantizedLinear,
// crates/core/src/quantization/gguf/mod.rs): the K-quant / legacy-block weight dequantised to dense f32, times f32 activations.
// workspace: f32 weight [N*K] | bf16 weight thirds [N, 3K] | bf16 activation thirds [M, 3K]
extern "C" size_t vmx_kquant_gemm_workspace_bytes(int m, int n_out, int n_in) {
    if (m <= 0 || n_out <= 0 || n_in <= 0) return 0;
    return (size_t)n_out * n_in * 4 + (size_t)n_out * n_in * 6 + (size_t)m * n_in * 6;
}

extern "C" int vmx_kquant_gemm_f32(int ggml_type, const void *w, const float *x, float *y, int m, int n_out, int n_in, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    VMX_REQUIRE(w && x && y && workspace, "kquant_gemm: null pointer");
    VMX_REQUIRE(vmx_ggml_type_size(ggml_type) > 0 && vmx_iq_type_size(ggml_type) <= 0, "kquant_gemm: ggml type %d is not a K-quant / legacy block type",
                ggml_type);
    VMX_REQUIRE(m >= 0 && n_out > 0 && n_in > 0 && n_in % vmx_ggml_block_elems(ggml_type) == 0,
                "kquant_gemm: n_in (%d) must be a positive multiple of the block size", n_in);
    if (m == 0) return VMX_OK;
    VMX_REQUIRE(workspace_bytes >= vmx_kquant_gemm_workspace_bytes(m, n_out, n_in), "kquant_gemm: workspace of %zu bytes, need %zu",
                workspace_bytes, vmx_kquant_gemm_workspace_bytes(m, n_out, n_in));
    hipStream_t st = vmx_stream(stream);
    const int64_t nk = (int64_t)n_out * n_in;
    float *wf = (float *)workspace;
    uint16_t *w3 = (uint16_t *)(wf + nk), *x3 = w3 + 3 * nk;
    if (int rc = vmx_kquant_dequantize_f32(ggml_type, w, wf, nk / vmx_ggml_block_elems(ggml_type), stream)) return rc;
    const int gw = (int)std::min<int64_t>((nk + 255) / 256, 16384), gx = (int)std::min<int64_t>(((int64_t)m * n_in + 255) / 256, 16384);
    hipLaunchKernelGGL((split3_rows_kernel<true>), dim3(gw), dim3(256), 0, st, w3, wf, (int64_t)n_out, n_in);
    hipLaunchKernelGGL((split3_rows_kernel<false>), dim3(gx), dim3(256), 0, st, x3, x, (int64_t)m, n_in);
    const int K3 = 3 * n_in;
    const int wide_wgs = ((n_out + 255) / 256) * ((m + 63) / 64);
    if (wide_wgs < 2 * vmx_num_cus() && K3 % 128 == 0)
        hipLaunchKernelGGL((split_gemm_kernel<false, true>), dim3((n_out + 63) / 64, (m + 63) / 64), dim3(256), 0, st, x3, w3, w3, y, m, n_out, K3, n_out);
    else
        hipLaunchKernelGGL((split_gemm_kernel<false, false>), dim3((n_out + 255) / 256, (m + 63) / 64), dim3(256), 0, st, x3, w3, w3, y, m, n_out, K3, n_out);
    VMX_CHECK_LAUNCH("kquant_gemm");
    return VMX_OK;
}

size_t vmx_iq_gemm_workspace(int N, int K) { return (size_t)N * K * 4; }   // the two bf16 halves

// workspace -> (hi bf16 [N*K], lo bf16 [N*K])
int vmx_iq_dequant_split(int ggml_type, const void *blocks, void *workspace, int N, int K, hipStream_t st) {
    const int64_t n = (int64_t)N * K;
    uint16_t *hi = (uint16_t *)workspace, *lo = hi + n;
    return vmx_iq_dequantize_split_bf16(ggml_type, blocks, hi, lo, n / 256, st);
}

// C[:, 0..N) (row stride ldc) = A @ (hi + lo)^T with the halves a previous vmx_iq_dequant_split left in `workspace`
int vmx_split_gemm_f32(const uint16_t *A, const void *workspace, float *C, int M, int N, int K, int ldc, hipStream_t st) {
    const int64_t n = (int64_t)N * K;
    const uint16_t *hi = (const uint16_t *)workspace, *lo = hi + n;
    // under about two workgroups per compute unit the four waves of a workgroup split K over one 64 x 64 tile instead
    const int wide_wgs = ((N + 255) / 256) * ((M + 63) / 64);
    if (wide_wgs < 2 * vmx_num_cus() && K % 128 == 0 && K >= 1024)
        hipLaunchKernelGGL((split_gemm_kernel<true, true>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, A, hi, lo, C, M, N, K, ldc);
    else
        hipLaunchKernelGGL((split_gemm_kernel<true, false>), dim3((N + 255) / 256, (M + 63) / 64), dim3(256), 0, st, A, hi, lo, C, M, N, K, ldc);
    VMX_CHECK_LAUNCH("split_gemm");
    return VMX_OK;
}

// ---- the integer form: workspace = Q fp16 [N*K] | Dt f32 [K/256][N] | A16 fp16 [M*K]
size_t vmx_iq_gemm_int_workspace(int M, int N, int K) {
    return (size_t)N * K * 2 + (((size_t)N * (K / 256) * 4 + 255) & ~(size_t)255) + (size_t)M * K * 2;
}

int vmx_iq_dequant_int(int ggml_type, const void *blocks, void *workspace, int N, int K, hipStream_t st) {
    uint16_t *Q = (uint16_t *)workspace;
    float *Dt = (float *)(Q + (int64_t)N * K);
    return vmx_iq_dequantize_int_f16(ggml_type, blocks, Q, Dt, N, K / 256, st);
}

// a16: fp16 copy of the activations (vmx_bf16_rows_to_f16), shared by the linears that read the same rows
int vmx_bf16_rows_to_f16(const uint16_t *A, uint16_t *a16, int64_t n, hipStream_t st) {
    const int64_t n8 = n / 8;
    if (n8 <= 0) return VMX_OK;
    hipLaunchKernelGGL(bf16_to_f16_kernel, dim3((unsigned)std::min<int64_t>((n8 + 255) / 256, 4096)), dim3(256), 0, st, a16, A, n8);
    VMX_CHECK_LAUNCH("bf16_to_f16");
    return VMX_OK;
}

int vmx_int_gemm_f32(const uint16_t *a16, const void *workspace, float *C, int M, int N, int K, int ldc, hipStream_t st) {
    const uint16_t *Q = (const uint16_t *)workspace;
    const float *Dt = (const float *)(Q + (int64_t)N * K);
    const int wide_wgs = ((N + 255) / 256) * ((M + 63) / 64);
    if (wide_wgs < 2 * vmx_num_cus() && K >= 1024)
        hipLaunchKernelGGL((int_gemm_kernel<true>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, a16, Q, Dt, C, M, N, K, ldc);
    else
        hipLaunchKernelGGL((int_gemm_kernel<false>), dim3((N + 255) / 256, (M + 63) / 64), dim3(256), 0, st, a16, Q, Dt, C, M, N, K, ldc);
    VMX_CHECK_LAUNCH("int_gemm");
    return VMX_OK;
}

extern "C" size_t vmx_iq_gemm_workspace_bytes(int n_out, int n_in) {
    return n_out > 0 && n_in > 0 ? vmx_iq_gemm_workspace(n_out, n_in) : 0;
}

extern "C" int vmx_iq_gemm_f32(int ggml_type, const void *blocks, const uint16_t *x, float *y, int m, int n_out, int n_in, void *workspace,
                               size_t workspace_bytes, void *stream) {
    VMX_REQUIRE(blocks && x && y && workspace, "iq_gemm: null pointer");
    VMX_REQUIRE(vmx_iq_type_size(ggml_type) > 0, "iq_gemm: ggml type %d is not a supported I-quant", ggml_type);
    VMX_REQUIRE(m >= 0 && n_out > 0 && n_in > 0 && n_in % 256 == 0, "iq_gemm: n_in (%d) must be a positive multiple of 256", n_in);
    VMX_REQUIRE(workspace_bytes >= vmx_iq_gemm_workspace(n_out, n_in), "iq_gemm: workspace of %zu bytes, need %zu", workspace_bytes,
                vmx_iq_gemm_workspace(n_out, n_in));
    if (m == 0) return VMX_OK;
    hipStream_t st = vmx_stream(stream);
    if (vmx_iq_int_form_ok(ggml_type) && workspace_bytes >= vmx_iq_gemm_int_workspace(m, n_out, n_in)) {
        // exact one-pass form: integer weights in fp16, one f32 scale per super-block
        uint16_t *a16 = (uint16_t *)((char *)workspace + vmx_iq_gemm_int_workspace(0, n_out, n_in));
        if (int rc = vmx_iq_dequant_int(ggml_type, blocks, workspace, n_out, n_in, st)) return rc;
        if (int rc = vmx_bf16_rows_to_f16(x, a16, (int64_t)m * n_in, st)) return rc;
        return vmx_int_gemm_f32(a16, workspace, y, m, n_out, n_in, n_out, st);
    }
    if (int rc = vmx_iq_dequant_split(ggml_type, blocks, workspace, n_out, n_in, st)) return rc;
    return vmx_split_gemm_f32(x, workspace, y, m, n_out, n_in, n_out, st);
}

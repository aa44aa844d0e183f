// GPTQ 8-bit linears: w[k][n] = (q - z) * s with q, z bytes packed four to a u32 (crates/core/kernels/gptq_dequant.cu:128-170),
// y = x @ W (+ bias) as gptq_gemm does for 8-bit checkpoints (crates/core/src/quantization/gptq_cuda.rs:322-360).
// An 8-bit weight is a byte per element, so the decode path is: dequantise once into a bf16 [N, K] workspace, then the bf16
// matrix-core GEMM.  (4-bit checkpoints take the fused tile-stream kernels of int4_kernels.hip.)
#include "vmx_internal.h"
#include "vmx_kernels.h"

namespace {

// One thread per (k quad, n): reads are coalesced over n.  TRANSPOSED: out[n][k] (GEMM operand), else out[k][n] (the reference's layout).
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void gptq_dequant_int8_kernel(uint16_t *__restrict__ out, const uint32_t *__restrict__ qweight,
                                                                 const uint16_t *__restrict__ scales, const uint32_t *__restrict__ qzeros,
                                                                 int K, int N, int group_size) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)(K / 4) * N;
    if (idx >= total) return;
    const int n = (int)(idx % N), kq = (int)(idx / N), k0 = kq * 4;
    const uint32_t packed = qweight[(int64_t)kq * N + n];
    const int zeros_per_row = (N + 3) / 4;
    uint16_t o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int g = group_size > 0 ? (k0 + j) / group_size : 0;
        const float s = vmx_bf16_to_f32(scales[(int64_t)g * N + n]);
        const int z = (int)((qzeros[(int64_t)g * zeros_per_row + (n >> 2)] >> (8 * (n & 3))) & 0xFFu);
        const int q = (int)((packed >> (8 * j)) & 0xFFu);
        o[j] = vmx_f32_to_bf16((float)(q - z) * s);
    }
    if constexpr (TRANSPOSED) {
        *reinterpret_cast<uint2 *>(out + (int64_t)n * K + k0) = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) out[(int64_t)(k0 + j) * N + n] = o[j];
    }
}

int check_shapes(const char *what, int K, int N, int group_size) {
    VMX_REQUIRE(K > 0 && N > 0 && K % 4 == 0, "%s: in_features (%d) must be a positive multiple of 4", what, K);
    VMX_REQUIRE(group_size <= 0 || K % group_size == 0, "%s: in_features %d not a multiple of group_size %d", what, K, group_size);
    return VMX_OK;
}

}  // namespace

extern "C" int vmx_gptq_dequant_int8_bf16(uint16_t *out, const uint32_t *qweight, const uint16_t *scales, const uint32_t *qzeros, int K, int N,
                                          int group_size, int transposed, void *stream) {
    VMX_REQUIRE(out && qweight && scales && qzeros, "gptq_dequant_int8: null pointer");
    if (int rc = check_shapes("gptq_dequant_int8", K, N, group_size)) return rc;
    const int64_t total = (int64_t)(K / 4) * N;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (transposed) hipLaunchKernelGGL(gptq_dequant_int8_kernel<true>, grid, dim3(256), 0, vmx_stream(stream), out, qweight, scales, qzeros, K, N, group_size);
    else hipLaunchKernelGGL(gptq_dequant_int8_kernel<false>, grid, dim3(256), 0, vmx_stream(stream), out, qweight, scales, qzeros, K, N, group_size);
    VMX_CHECK_LAUNCH("gptq_dequant_int8");
    return VMX_OK;
}

extern "C" int vmx_gptq_int8_gemm_bf16(uint16_t *y, const uint16_t *x, const uint32_t *qweight, const uint16_t *scales, const uint32_t *qzeros,
                                       const uint16_t *bias, int M, int N, int K, int group_size, uint16_t *workspace, void *stream) {
    VMX_REQUIRE(y && x && qweight && scales && qzeros && workspace, "gptq_int8_gemm: null pointer");
    if (int rc = check_shapes("gptq_int8_gemm", K, N, group_size)) return rc;
    VMX_REQUIRE(K % 16 == 0, "gptq_int8_gemm: in_features (%d) must be a multiple of 16", K);
    VMX_REQUIRE(M >= 0, "gptq_int8_gemm: M %d", M);
    if (int rc = vmx_gptq_dequant_int8_bf16(workspace, qweight, scales, qzeros, K, N, group_size, 1, stream)) return rc;
    if (M == 0) return VMX_OK;
    return vmx_bf16_gemm_nt(y, x, workspace, bias, M, N, K, vmx_stream(stream));
}

// Measurement aids (not part of the product path): raw HBM streaming probes with the same access
// shape as the int4 linear (one 1 KiB unit per wave-instruction, UPW units per wave, 4 waves per workgroup).
#include "vmx_internal.h"

namespace {
template <int UPW>
__global__ __launch_bounds__(256) void stream_read_kernel(const uint4 *__restrict__ p, unsigned *__restrict__ sink, long long units) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long u0 = ((long long)blockIdx.x * 4 + wave) * UPW;
    uint4 r[UPW];
#pragma unroll
    for (int u = 0; u < UPW; u++) r[u] = (u0 + u < units) ? p[(u0 + u) * 64 + lane] : make_uint4(0, 0, 0, 0);
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < UPW; u++) acc ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
    if (acc == 0x12345678u) sink[0] = acc;  // practically never taken; keeps the loads alive
}
// out[blockIdx.x] = id of the XCD (accelerator complex die) the workgroup runs on
__global__ void xcc_id_kernel(int *__restrict__ out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(v & 0xF);
}
}  // namespace

extern "C" int vmx_debug_xcc_ids(int *out, int n_wgs, int wg_threads, void *stream) {
    VMX_REQUIRE(out && n_wgs > 0 && wg_threads > 0 && wg_threads <= 1024, "debug_xcc_ids: bad arguments");
    hipLaunchKernelGGL(xcc_id_kernel, dim3(n_wgs), dim3(wg_threads), 0, vmx_stream(stream), out);
    VMX_CHECK_LAUNCH("debug_xcc_ids");
    return VMX_OK;
}

extern "C" int vmx_debug_stream_read(const void *p, size_t bytes, int upw, void *sink, void *stream) {
    VMX_REQUIRE(p && sink && bytes >= 1024, "debug_stream_read: bad arguments");
    const long long units = (long long)(bytes / 1024);
    hipStream_t st = vmx_stream(stream);
#define L(U) hipLaunchKernelGGL((stream_read_kernel<U>), dim3((unsigned)((units + 4 * U - 1) / (4 * U))), dim3(256), 0, st, (const uint4 *)p, (unsigned *)sink, units)
    switch (upw) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 4: L(4); break;
        case 8: L(8); break;
        case 16: L(16); break;
        default: vmx_set_error("debug_stream_read: upw must be 1,2,4,8,16"); return VMX_EINVAL;
    }
#undef L
    VMX_CHECK_LAUNCH("debug_stream_read");
    return VMX_OK;
}

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
// Every register-only cross-lane helper of vmx_internal.h against its ds_bpermute form (__shfl_xor), on values whose sums
// round differently in a different order: out[lane] = number of helpers whose bits differ on that lane.
__global__ void lane_ops_check_kernel(int *__restrict__ out, unsigned seed) {
    const int l = threadIdx.x & 63;
    int bad = 0;
    bad += vmx_lane_xor<1>(l + 100) != (l ^ 1) + 100;
    bad += vmx_lane_xor<2>(l + 100) != (l ^ 2) + 100;
    bad += vmx_lane_xor<4>(l + 100) != (l ^ 4) + 100;
    bad += vmx_lane_xor<8>(l + 100) != (l ^ 8) + 100;
    bad += vmx_lane_xor<16>(l + 100) != (l ^ 16) + 100;
    bad += vmx_lane_xor<32>(l + 100) != (l ^ 32) + 100;
    for (int m = 1; m < 64; m <<= 1) bad += vmx_lane_xor_m(l + 7, m) != (l ^ m) + 7;
    for (unsigned round = 0; round < 32; round++) {
        unsigned h = (seed + round * 64u + (unsigned)l) * 2654435761u;
        h ^= h >> 15;
        const float v = (float)(h & 0xFFFFu) * ((h >> 16) % 5u == 0u ? 37.25f : 0.001953125f) - 11.0f;
        float ws = v, wm = v, rs = v, rm = v;
        for (int off = 32; off > 0; off >>= 1) { ws += __shfl_xor(ws, off, 64); wm = fmaxf(wm, __shfl_xor(wm, off, 64)); }
        for (int off = 1; off < 16; off <<= 1) { rs += __shfl_xor(rs, off, 64); rm = fmaxf(rm, __shfl_xor(rm, off, 64)); }
        const float rm32 = fmaxf(rm, __shfl_xor(rm, 16, 64));
        bad += __float_as_uint(vmx_wave_sum(v)) != __float_as_uint(ws);
        bad += __float_as_uint(vmx_wave_max(v)) != __float_as_uint(wm);
        bad += __float_as_uint(vmx_row16_sum(v)) != __float_as_uint(rs);
        bad += __float_as_uint(vmx_row16_max(v)) != __float_as_uint(rm);
        bad += __float_as_uint(vmx_row32_max(v)) != __float_as_uint(rm32);
    }
    out[l] = bad;
}
}  // namespace

extern "C" int vmx_debug_lane_ops_check(int *out64, unsigned seed, void *stream) {
    VMX_REQUIRE(out64, "debug_lane_ops_check: null pointer");
    hipLaunchKernelGGL(lane_ops_check_kernel, dim3(1), dim3(64), 0, vmx_stream(stream), out64, seed);
    VMX_CHECK_LAUNCH("debug_lane_ops_check");
    return VMX_OK;
}

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

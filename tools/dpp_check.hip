// Checks on the GPU that the DPP / permlane forms of the wave butterfly give the bits of the __shfl_xor form.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "../vllm-rust_amd/csrc/vmx_internal.h"
__global__ void stages(int *o) {
    const int l = threadIdx.x;
    float a, b;
    vmx_swap_pair<32>((float)l, a, b); o[l] = (int)a; o[64 + l] = (int)b;
    vmx_swap_pair<16>((float)l, a, b); o[128 + l] = (int)a; o[192 + l] = (int)b;
    o[256 + l] = (int)vmx_dpp<VMX_DPP_ROR8>((float)l);
    o[320 + l] = (int)vmx_dpp<VMX_DPP_ROR4>((float)l);
    o[384 + l] = (int)vmx_dpp<VMX_DPP_XOR2>((float)l);
    o[448 + l] = (int)vmx_dpp<VMX_DPP_XOR1>((float)l);
    o[512 + l] = (int)vmx_dpp<VMX_DPP_HALF_MIRROR>((float)l);
    int bad = 0;
    bad += vmx_lane_xor<1>(l + 100) != (l ^ 1) + 100;
    bad += vmx_lane_xor<2>(l + 100) != (l ^ 2) + 100;
    bad += vmx_lane_xor<4>(l + 100) != (l ^ 4) + 100;
    bad += vmx_lane_xor<8>(l + 100) != (l ^ 8) + 100;
    bad += vmx_lane_xor<16>(l + 100) != (l ^ 16) + 100;
    bad += vmx_lane_xor<32>(l + 100) != (l ^ 32) + 100;
    for (int m = 1; m < 64; m <<= 1) bad += vmx_lane_xor_m(l + 7, m) != (l ^ m) + 7;
    float r = (float)(l * l % 37), rs = r, rm = r;
    for (int m = 1; m < 16; m <<= 1) { rs += __shfl_xor(rs, m, 64); rm = fmaxf(rm, __shfl_xor(rm, m, 64)); }
    bad += vmx_row16_sum(r) != rs;
    bad += vmx_row16_max(r) != rm;
    rm = fmaxf(rm, __shfl_xor(rm, 16, 64));
    bad += vmx_row32_max(r) != rm;
    o[576 + l] = bad;
}
__global__ void k(const float *x, float *y, float *z, float *ym, float *zm) {
    float v = x[threadIdx.x];
    y[threadIdx.x] = vmx_wave_sum(v);
    ym[threadIdx.x] = vmx_wave_max(v);
    float w = v, m = v;
    for (int off = 32; off > 0; off >>= 1) { w += __shfl_xor(w, off, 64); m = fmaxf(m, __shfl_xor(m, off, 64)); }
    z[threadIdx.x] = w; zm[threadIdx.x] = m;
}
int main() {
    int *o; (void)hipMalloc(&o, 640 * 4);
    stages<<<1, 64>>>(o);
    int ho[640]; (void)hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    const char *names[9] = {"swap32.a", "swap32.b", "swap16.a", "swap16.b", "ror8", "ror4", "xor2", "xor1", "half_mirror"};
    for (int s = 0; s < 9; s++) { printf("%-12s", names[s]); for (int i = 0; i < 64; i++) printf(" %d", ho[64 * s + i]); printf("\n"); }
    int lane_bad = 0;
    for (int i = 0; i < 64; i++) lane_bad += ho[576 + i];
    printf("lane_xor / row mismatches %d\n", lane_bad);
    float *x, *y, *z, *ym, *zm; (void)hipMalloc(&x, 256); (void)hipMalloc(&y, 256); (void)hipMalloc(&z, 256); (void)hipMalloc(&ym, 256); (void)hipMalloc(&zm, 256);
    float h[64], a[64], b[64], am[64], bm[64];
    int bad = 0;
    for (int t = 0; t < 200; t++) {
        for (int i = 0; i < 64; i++) h[i] = (float)(rand() % 100000) / 100000.f * ((i * 7 + t) % 5 == 0 ? 1000.f : 1.f) - 0.3f;
        (void)hipMemcpy(x, h, 256, hipMemcpyHostToDevice);
        k<<<1, 64>>>(x, y, z, ym, zm);
        (void)hipMemcpy(a, y, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(b, z, 256, hipMemcpyDeviceToHost);
        (void)hipMemcpy(am, ym, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(bm, zm, 256, hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; i++) if (memcmp(&a[i], &b[i], 4) || memcmp(&am[i], &bm[i], 4)) bad++;
    }
    printf("mismatches %d (last %g %g)\n", bad, a[5], b[5]);
    return bad != 0 || lane_bad != 0;
}

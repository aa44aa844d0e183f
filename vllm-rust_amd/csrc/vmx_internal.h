// Internal helpers shared by the libvmx translation units (host + device).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include "../../include/vmx.h"

// ---------------------------------------------------------------- host side
void vmx_set_error(const char *fmt, ...);
int vmx_num_cus(void);

#define VMX_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            vmx_set_error(__VA_ARGS__);   \
            return VMX_EINVAL;            \
        }                                 \
    } while (0)

#define VMX_CHECK_LAUNCH(what)                                                     \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            vmx_set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
            return VMX_ELAUNCH;                                                    \
        }                                                                          \
    } while (0)

#define VMX_CHECK_HIP(expr)                                                        \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess) {                                                   \
            vmx_set_error("%s: %s", #expr, hipGetErrorString(e__));               \
            return VMX_ELAUNCH;                                                    \
        }                                                                          \
    } while (0)

static inline hipStream_t vmx_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Measurement switches (kernel variants kept for A/B runs, forced shapes, ablations).  The shipped library never looks at the
// environment: every switch is its default.  `make TUNING=1` builds lib/libvmx_tuning.so, in which the same names are read
// from the environment; nothing but the profiling scripts under tools/ loads that build.
#ifdef VMX_TUNING
#include <stdlib.h>
static inline int vmx_tune(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}
static inline const char *vmx_tune_str(const char *name) { return getenv(name); }
#else
static inline int vmx_tune(const char *, int dflt) { return dflt; }
static inline const char *vmx_tune_str(const char *) { return nullptr; }
#endif

// ---------------------------------------------------------------- device side
#define VMX_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 vmx_bf16x8;
typedef __attribute__((ext_vector_type(4))) float vmx_f32x4;

__device__ __forceinline__ float vmx_bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// Round-to-nearest-even; a plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32).
__device__ __forceinline__ uint16_t vmx_f32_to_bf16(float f) {
    __bf16 b = static_cast<__bf16>(f);
    return __builtin_bit_cast(uint16_t, b);
}

__device__ __forceinline__ float vmx_f16_to_f32(uint16_t h) {
    return static_cast<float>(__builtin_bit_cast(_Float16, h));
}

// One 16-bit activation element in the model's activation dtype: bf16 (F16 = false) or IEEE fp16 (EXL3 checkpoints run in
// fp16 end to end, crates/core/src/quantization/weight_loader.rs:453-461).  Both stores round to nearest even.
template <bool F16> __device__ __forceinline__ float vmx_ld16(uint16_t b) {
    if constexpr (F16) return vmx_f16_to_f32(b);
    else return vmx_bf16_to_f32(b);
}
template <bool F16> __device__ __forceinline__ uint16_t vmx_st16(float v) {
    if constexpr (F16) return __builtin_bit_cast(uint16_t, static_cast<_Float16>(v));
    else return vmx_f32_to_bf16(v);
}
template <bool F16> __device__ __forceinline__ float vmx_rnd16(float v) { return vmx_ld16<F16>(vmx_st16<F16>(v)); }
// eight consecutive elements <-> f32
template <bool F16> __device__ __forceinline__ void vmx_unpack16x8(const uint4 v, float f[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f[2 * j] = vmx_ld16<F16>((uint16_t)(w[j] & 0xFFFFu));
        f[2 * j + 1] = vmx_ld16<F16>((uint16_t)(w[j] >> 16));
    }
}
template <bool F16> __device__ __forceinline__ uint4 vmx_pack16x8(const float f[8]) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = (uint32_t)vmx_st16<F16>(f[2 * j]) | ((uint32_t)vmx_st16<F16>(f[2 * j + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Cross-lane exchanges without the LDS pipe (ds_bpermute costs a round trip of ~100 clocks per step): DPP row operations inside
// a row of 16 lanes, v_permlane16_swap / v_permlane32_swap across rows.
template <int CTRL> __device__ __forceinline__ float vmx_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int VMX_DPP_XOR1 = 0xB1, VMX_DPP_XOR2 = 0x4E;        // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int VMX_DPP_ROR4 = 0x124, VMX_DPP_ROR8 = 0x128;      // row_ror:4, row_ror:8 (= lane ^ 8)
constexpr int VMX_DPP_HALF_MIRROR = 0x141;                     // lane -> 7 - lane inside each 8 lanes
// The values of lane and lane ^ 32 (MASK 32) or lane ^ 16 (MASK 16) as an unordered pair {a, b}: combine them with a
// commutative operation.
template <int MASK> __device__ __forceinline__ void vmx_swap_pair(float v, float &a, float &b) {
    static_assert(MASK == 16 || MASK == 32, "row exchanges only");
    // written as assembly: the compiler's builtin hands back the first result register for both halves of its result pair.
    // The swap needs its operands written two wait states earlier (s_nop 1).
    a = v;
    b = v;
    if constexpr (MASK == 32) asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a = low half everywhere, b = high half
    else asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));                          // a = even rows, b = odd rows
}
// Butterfly over the whole wave in the order lane ^ 32, 16, 8, 4, 2, 1; every lane ends with the same bits.  The step for
// lane ^ 4 is a rotation by 4: after the lane ^ 8 step lanes i and i ^ 8 hold equal values, so (i + 4) mod 16 serves as i ^ 4.
__device__ __forceinline__ float vmx_wave_sum(float v) {
    float a, b;
    vmx_swap_pair<32>(v, a, b); v = a + b;
    vmx_swap_pair<16>(v, a, b); v = a + b;
    v += vmx_dpp<VMX_DPP_ROR8>(v);
    v += vmx_dpp<VMX_DPP_ROR4>(v);
    v += vmx_dpp<VMX_DPP_XOR2>(v);
    v += vmx_dpp<VMX_DPP_XOR1>(v);
    return v;
}

__device__ __forceinline__ float vmx_wave_max(float v) {
    float a, b;
    vmx_swap_pair<32>(v, a, b); v = fmaxf(a, b);
    vmx_swap_pair<16>(v, a, b); v = fmaxf(a, b);
    v = fmaxf(v, vmx_dpp<VMX_DPP_ROR8>(v));
    v = fmaxf(v, vmx_dpp<VMX_DPP_ROR4>(v));
    v = fmaxf(v, vmx_dpp<VMX_DPP_XOR2>(v));
    v = fmaxf(v, vmx_dpp<VMX_DPP_XOR1>(v));
    return v;
}

// The value of lane ^ M for a compile-time M, without the LDS crossbar.  M = 4: two masked row shifts (lanes with bit 2 clear
// take lane + 4, the others lane - 4).
template <int M> __device__ __forceinline__ float vmx_lane_xor(float v) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "one lane bit");
    if constexpr (M == 1) return vmx_dpp<VMX_DPP_XOR1>(v);
    else if constexpr (M == 2) return vmx_dpp<VMX_DPP_XOR2>(v);
    else if constexpr (M == 4) {
        const int x = __builtin_bit_cast(int, v);
        int t = __builtin_amdgcn_update_dpp(x, x, 0x104 /* row_shl:4 */, 0xF, 0x5, false);
        t = __builtin_amdgcn_update_dpp(t, x, 0x114 /* row_shr:4 */, 0xF, 0xA, false);
        return __builtin_bit_cast(float, t);
    } else if constexpr (M == 8) return vmx_dpp<VMX_DPP_ROR8>(v);
    else {
        float a, b;
        vmx_swap_pair<M>(v, a, b);
        const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        return (lane & M) ? a : b;
    }
}
template <int M> __device__ __forceinline__ int vmx_lane_xor(int v) { return __builtin_bit_cast(int, vmx_lane_xor<M>(__builtin_bit_cast(float, v))); }
// ... for an M that is a constant only after loop unrolling
__device__ __forceinline__ float vmx_lane_xor_m(float v, int m) {
    switch (m) {
        case 1: return vmx_lane_xor<1>(v);
        case 2: return vmx_lane_xor<2>(v);
        case 4: return vmx_lane_xor<4>(v);
        case 8: return vmx_lane_xor<8>(v);
        case 16: return vmx_lane_xor<16>(v);
        default: return vmx_lane_xor<32>(v);
    }
}
__device__ __forceinline__ int vmx_lane_xor_m(int v, int m) { return __builtin_bit_cast(int, vmx_lane_xor_m(__builtin_bit_cast(float, v), m)); }
// Butterflies over each row of 16 lanes in the order lane ^ 1, 2, 4, 8 (every lane of a row ends with the same bits).  The
// lane ^ 4 step is the mirror inside 8 lanes: after the first two steps the four lanes of a quad hold equal values.
__device__ __forceinline__ float vmx_row16_sum(float v) {
    v += vmx_dpp<VMX_DPP_XOR1>(v);
    v += vmx_dpp<VMX_DPP_XOR2>(v);
    v += vmx_dpp<VMX_DPP_HALF_MIRROR>(v);
    v += vmx_dpp<VMX_DPP_ROR8>(v);
    return v;
}
__device__ __forceinline__ float vmx_row16_max(float v) {
    v = fmaxf(v, vmx_dpp<VMX_DPP_XOR1>(v));
    v = fmaxf(v, vmx_dpp<VMX_DPP_XOR2>(v));
    v = fmaxf(v, vmx_dpp<VMX_DPP_HALF_MIRROR>(v));
    v = fmaxf(v, vmx_dpp<VMX_DPP_ROR8>(v));
    return v;
}
// ... and over 32 lanes (lane ^ 1 .. 16)
__device__ __forceinline__ float vmx_row32_max(float v) {
    v = vmx_row16_max(v);
    float a, b;
    vmx_swap_pair<16>(v, a, b);
    return fmaxf(a, b);
}

// Sum over a 256-thread workgroup (4 waves); `red` is 4 floats of LDS.  Ends synchronised.
__device__ __forceinline__ float vmx_block_sum_256(float v, float *red) {
    v = vmx_wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

// q8_1 of 8 consecutive values held by one lane, where 4 consecutive lanes (aligned to 4) hold one 32-value
// block: d = amax/127 (f32 division), q = roundf(v/d) (half away from zero), all-zero block -> d = 0, q = 0
// (crates/core/kernels/iq_dequant.cu:253-267).  Every lane of the 4-lane group must call it.
__device__ __forceinline__ void vmx_q8_1_emit8(const float v[8], int8_t *aq8, float *ad_block, bool first_lane_of_block) {
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) amax = fmaxf(amax, fabsf(v[j]));
    amax = fmaxf(amax, vmx_dpp<VMX_DPP_XOR1>(amax));
    amax = fmaxf(amax, vmx_dpp<VMX_DPP_XOR2>(amax));
    const float d = amax / 127.0f;
    uint32_t w[2] = {0u, 0u};
    if (amax != 0.0f) {
#pragma unroll
        for (int j = 0; j < 8; j++) w[j >> 2] |= ((uint32_t)((int)roundf(v[j] / d) & 0xFF)) << (8 * (j & 3));
    }
    *reinterpret_cast<uint2 *>(aq8) = make_uint2(w[0], w[1]);
    if (first_lane_of_block) *ad_block = d;
}

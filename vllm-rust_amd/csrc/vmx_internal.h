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

__device__ __forceinline__ float vmx_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float vmx_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
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
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    const float d = amax / 127.0f;
    uint32_t w[2] = {0u, 0u};
    if (amax != 0.0f) {
#pragma unroll
        for (int j = 0; j < 8; j++) w[j >> 2] |= ((uint32_t)((int)roundf(v[j] / d) & 0xFF)) << (8 * (j & 3));
    }
    *reinterpret_cast<uint2 *>(aq8) = make_uint2(w[0], w[1]);
    if (first_lane_of_block) *ad_block = d;
}

/*
 * CPU oracle -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's CPU algorithms for the quantized
 * inference hot path.  It is the checker for the HIP kernels: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product library (libvmx.so) never links or calls anything in this directory.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA contraction, so
 * f32 expressions round exactly as the reference's Rust scalar code does).
 */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H

#include <stdint.h>
#include <string.h>
#include <math.h>

#ifdef __cplusplus
extern "C" {
#endif

/* IEEE binary16 -> binary32 (exact). */
static inline float orc_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* binary32 -> binary16, round-to-nearest-even. */
static inline uint16_t orc_f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t man = x & 0x7fffffu;
    int32_t exp = (int32_t)((x >> 23) & 0xffu);
    if (exp == 255) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
    exp = exp - 127 + 15;
    if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        uint32_t shift = (uint32_t)(14 - exp);
        uint32_t half = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1u))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)exp << 10) | (man >> 13);
    uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;
    return (uint16_t)(sign | half);
}

/* bfloat16 <-> binary32; f32 -> bf16 is round-to-nearest-even, NaN kept quiet. */
static inline float orc_bf16_to_f32(uint16_t h) {
    uint32_t bits = (uint32_t)h << 16;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline uint16_t orc_f32_to_bf16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

static inline float orc_round_bf16(float f) { return orc_bf16_to_f32(orc_f32_to_bf16(f)); }

#ifdef __cplusplus
}
#endif
#endif

// Shared device helpers of the GGUF I-quant kernels: codebook tables, unaligned field loads, sub-block decode, the integer
// scale step of vec_dot_iq*_q8_1 (crates/core/kernels/iq_dequant.cu:274-386; crates/core/src/quantization/gguf/iq/mod.rs:556-658).
#pragma once
#include "vmx_kernels.h"

#define IQ_TABLE_QUAL static __device__ const
#include "_gen/iq_tables_gen.h"

namespace {

constexpr int QK_K = 256;

// ---- unaligned little-endian loads (block sizes 74/82/98/110 are only 2-byte aligned)
struct __attribute__((packed, aligned(2))) u32_a2 { uint32_t v; };
struct __attribute__((packed, aligned(2))) u16_a2 { uint16_t v; };
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return reinterpret_cast<const u32_a2 *>(p)->v; }
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return reinterpret_cast<const u16_a2 *>(p)->v; }

template <int T> struct IqTraits;
template <> struct IqTraits<17> { static constexpr int TS = 74;  static constexpr int LDS_WORDS = 1024; };
template <> struct IqTraits<22> { static constexpr int TS = 82;  static constexpr int LDS_WORDS = 2048; };
template <> struct IqTraits<18> { static constexpr int TS = 98;  static constexpr int LDS_WORDS = 256; };
template <> struct IqTraits<21> { static constexpr int TS = 110; static constexpr int LDS_WORDS = 512; };
template <> struct IqTraits<23> { static constexpr int TS = 136; static constexpr int LDS_WORDS = 4; };

template <int T> __device__ __forceinline__ const uint32_t *codebook_src() {
    if constexpr (T == 17) return reinterpret_cast<const uint32_t *>(iq2xs_grid);
    else if constexpr (T == 22) return reinterpret_cast<const uint32_t *>(iq2s_grid);
    else if constexpr (T == 18) return iq3xxs_grid;
    else if constexpr (T == 21) return iq3s_grid;
    else return reinterpret_cast<const uint32_t *>(kvalues_iq4nl);
}

// Sign-mask table entries a type needs: 128 for the 7-bit "ksigns" types (IQ2_XS, IQ3_XXS), 256 for the types with explicit
// sign bytes (IQ2_S, IQ3_S), none for IQ4_XS.
template <int T> struct SignLut { static constexpr int N = (T == 17 || T == 18) ? 128 : (T == 23 ? 1 : 256); };

// Stage the type's codebook in LDS (as u32 words) and build the sign-mask table: sg[i] = 0xFF in byte j of the 8-byte entry
// when weight j of the group of 8 is negative.  For the 7-bit types entry i is the mask of ggml's ksigns_iq2xs[i]
// (= i | parity(i) << 7).  Two halves so that a kernel can request its weights between them: tables_fetch issues the
// global loads of the codebook words (256 threads), tables_commit writes LDS and ends with a barrier.
template <int T> struct TableRegs { uint32_t w[(IqTraits<T>::LDS_WORDS + 255) / 256]; };

template <int T> __device__ __forceinline__ TableRegs<T> tables_fetch() {
    constexpr int W = IqTraits<T>::LDS_WORDS;
    const uint32_t *src = codebook_src<T>();
    TableRegs<T> t;
#pragma unroll
    for (int j = 0; j < (W + 255) / 256; j++) {
        const int i = threadIdx.x + 256 * j;
        t.w[j] = i < W ? src[i] : 0u;
    }
    return t;
}

template <int T> __device__ __forceinline__ void tables_commit(const TableRegs<T> &t, uint32_t *lds, uint2 *sg) {
    constexpr int W = IqTraits<T>::LDS_WORDS;
#pragma unroll
    for (int j = 0; j < (W + 255) / 256; j++) {
        const int i = threadIdx.x + 256 * j;
        if (i < W) lds[i] = t.w[j];
    }
    if constexpr (T != 23) {
        const int i = threadIdx.x;
        if (i < SignLut<T>::N) {
            uint32_t s8 = (uint32_t)i;
            if constexpr (SignLut<T>::N == 128) s8 |= (__popc(s8) & 1u) << 7;
            const uint32_t l0 = __umul24(s8 & 0xFu, 0x00204081u) & 0x01010101u, l1 = __umul24(s8 >> 4, 0x00204081u) & 0x01010101u;
            sg[i] = make_uint2((l0 << 8) - l0, (l1 << 8) - l1);
        }
    }
    __syncthreads();
}

template <int T> __device__ __forceinline__ void load_tables(uint32_t *lds, uint2 *sg) {   // blockDim.x == 256
    const TableRegs<T> t = tables_fetch<T>();
    tables_commit<T>(t, lds, sg);
}

// Negate the bytes of g selected by mask (0xFF per selected byte).  g's bytes are in 1..127, so the two's-complement
// negation (g ^ 0xFF) + 1 of a byte never carries into its neighbour.
__device__ __forceinline__ int sign4(uint32_t g, uint32_t mask) { return (int)((g ^ mask) + (mask & 0x01010101u)); }

// Eight 4-bit indices of `q4` through the 16-entry int8 table T[0..3] (as 4 words):
// lo = entries of the low nibbles, hi = of the high nibbles.
__device__ __forceinline__ void table16(uint32_t q4, const uint32_t T[4], int &lo, int &hi) {
    auto look = [&](uint32_t idx) -> int {  // idx: 4 bytes, each 0..15
        const uint32_t sel = idx & 0x07070707u;
        const uint32_t a = __builtin_amdgcn_perm(T[1], T[0], sel);   // entries 0..7
        const uint32_t b = __builtin_amdgcn_perm(T[3], T[2], sel);   // entries 8..15
        const uint32_t pick = ((idx & 0x08080808u) >> 1) | 0x03020100u;   // byte j: j (from a) or 4 + j (from b)
        return (int)__builtin_amdgcn_perm(b, a, pick);
    };
    lo = look(q4 & 0x0F0F0F0Fu);
    hi = look((q4 >> 4) & 0x0F0F0F0Fu);
}

struct SubScale { int ls0, ls1; };

// Raw bytes of one 32-weight sub-block, fetched before any of them is interpreted so that the loads
// of several sub-blocks (rows, k positions) can be in flight together.
struct RawSub {
    uint32_t a, b, c, e;  // type-specific words (see load_raw)
    uint32_t x, y;        // small fields (scales, high bits)
    uint32_t d;           // f16 super-block scale bits
};

template <int T> __device__ __forceinline__ RawSub load_raw(const uint8_t *b, int ib) {
    RawSub r;
    r.a = r.b = r.c = r.e = r.x = r.y = 0;
    r.d = ld16(b);
    if constexpr (T == 17) {         // IQ2_XS: qs u16[32]@2, scales u8[8]@66
        r.a = ld32(b + 2 + 8 * ib); r.b = ld32(b + 2 + 8 * ib + 4); r.x = b[66 + ib];
    } else if constexpr (T == 22) {  // IQ2_S: qs u8[32]@2, signs u8[32]@34, qh u8[8]@66, scales u8[8]@74
        r.a = ld32(b + 2 + 4 * ib); r.c = ld32(b + 34 + 4 * ib); r.y = b[66 + ib]; r.x = b[74 + ib];
    } else if constexpr (T == 18) {  // IQ3_XXS: qs u8[64]@2, scales_and_signs u32[8]@66
        r.a = ld32(b + 2 + 8 * ib); r.b = ld32(b + 2 + 8 * ib + 4); r.c = ld32(b + 66 + 4 * ib);
    } else if constexpr (T == 21) {  // IQ3_S: qs u8[64]@2, qh u8[8]@66, signs u8[32]@74, scales u8[4]@106
        r.a = ld32(b + 2 + 8 * ib); r.b = ld32(b + 2 + 8 * ib + 4); r.c = ld32(b + 74 + 4 * ib);
        r.y = b[66 + ib]; r.x = b[106 + (ib >> 1)];
    } else {                         // IQ4_XS: scales_h u16@2, scales_l u8[4]@4, qs u8[128]@8 (8-byte aligned)
        const uint2 *qp = reinterpret_cast<const uint2 *>(b + 8 + 16 * ib);
        const uint2 p0 = qp[0], p1 = qp[1];
        r.a = p0.x; r.b = p0.y; r.c = p1.x; r.e = p1.y;
        r.y = ld16(b + 2); r.x = b[4 + (ib >> 1)];
    }
    return r;
}

// Decode a fetched sub-block into 8 words of 4 signed int8 (word k = weights 4k..4k+3) plus the
// sub-block's integer scale fields.  cb = the type's codebook in LDS, sg = its sign-mask table.
template <int T>
__device__ __forceinline__ void decode_raw(const RawSub &r, int ib, const uint32_t *cb, const uint2 *sg, int q[8], SubScale &sc) {
    sc.ls0 = sc.ls1 = 0;
    if constexpr (T == 17) {
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const uint32_t v = ((l < 2 ? r.a : r.b) >> (16 * (l & 1))) & 0xFFFFu;
            const uint2 g = reinterpret_cast<const uint2 *>(cb)[v & 511u];
            const uint2 mk = sg[v >> 9];
            q[2 * l] = sign4(g.x, mk.x);
            q[2 * l + 1] = sign4(g.y, mk.y);
        }
        sc.ls0 = r.x & 0xF; sc.ls1 = r.x >> 4;
    } else if constexpr (T == 22) {
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const uint32_t idx = ((r.a >> (8 * l)) & 0xFFu) | ((r.y << (8 - 2 * l)) & 0x300u);
            const uint2 g = reinterpret_cast<const uint2 *>(cb)[idx];
            const uint2 mk = sg[(r.c >> (8 * l)) & 0xFFu];
            q[2 * l] = sign4(g.x, mk.x);
            q[2 * l + 1] = sign4(g.y, mk.y);
        }
        sc.ls0 = r.x & 0xF; sc.ls1 = r.x >> 4;
    } else if constexpr (T == 18) {
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const uint32_t pair = ((l < 2 ? r.a : r.b) >> (16 * (l & 1))) & 0xFFFFu;
            const uint2 mk = sg[(r.c >> (7 * l)) & 127u];
            q[2 * l] = sign4(cb[pair & 0xFFu], mk.x);
            q[2 * l + 1] = sign4(cb[pair >> 8], mk.y);
        }
        sc.ls0 = (int)(r.c >> 28);
    } else if constexpr (T == 21) {
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const uint32_t pair = ((l < 2 ? r.a : r.b) >> (16 * (l & 1))) & 0xFFFFu;
            const uint32_t i1 = (pair & 0xFFu) | ((r.y << (8 - 2 * l)) & 256u);
            const uint32_t i2 = (pair >> 8) | ((r.y << (7 - 2 * l)) & 256u);
            const uint2 mk = sg[(r.c >> (8 * l)) & 0xFFu];
            q[2 * l] = sign4(cb[i1], mk.x);
            q[2 * l + 1] = sign4(cb[i2], mk.y);
        }
        sc.ls0 = (r.x >> (4 * (ib & 1))) & 0xF;
    } else {
        uint32_t Tb[4] = {cb[0], cb[1], cb[2], cb[3]};
        const uint32_t w[4] = {r.a, r.b, r.c, r.e};
#pragma unroll
        for (int j = 0; j < 4; j++) table16(w[j], Tb, q[j], q[j + 4]);
        sc.ls0 = (int)(((r.x >> (4 * (ib & 1))) & 0xFu) | (((r.y >> (2 * ib)) & 3u) << 4));
    }
}

template <int T>
__device__ __forceinline__ void decode_sub(const uint8_t *b, int ib, const uint32_t *cb, const uint2 *sg, int q[8], SubScale &sc) {
    const RawSub r = load_raw<T>(b, ib);
    decode_raw<T>(r, ib, cb, sg, q, sc);
}

// Integer scale step of vec_dot_iq*_q8_1 (C truncating division).
template <int T> __device__ __forceinline__ int scale_int(int s0, int s1, const SubScale &sc) {
    if constexpr (T == 17 || T == 22) return (s0 * sc.ls0 + s1 * sc.ls1 + (s0 + s1) / 2) / 4;
    else if constexpr (T == 18) { int s = s0 + s1; return (sc.ls0 * s + s / 2) / 2; }
    else if constexpr (T == 21) return (s0 + s1) * (1 + 2 * sc.ls0);
    else return (s0 + s1) * (sc.ls0 - 32);
}

__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

__device__ __forceinline__ int subdot_halves(const int q[8], const int u[8], int &s1) {
    int s0 = 0;
    s1 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { s0 = dot4(q[k], u[k], s0); s1 = dot4(q[k + 4], u[k + 4], s1); }
    return s0;
}

}  // namespace

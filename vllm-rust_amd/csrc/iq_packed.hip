// q8_1 MMVQ over a load-time REPACKED form of GGUF I-quant weights (IQ3_XXS, IQ3_S, IQ2_S), one activation row.
//
// Semantics are those of mmvq_iq*_f32 / vec_dot_iq*_q8_1 (crates/core/kernels/iq_dequant.cu:274-423; CPU form
// crates/core/src/quantization/gguf/iq/mod.rs:556-725): y[o] = sum_g w_d(o, g/8) * a_d(g) * int_dot(block, g%8, u), the
// integer sub-dots bit for bit (vmx_iq_subdots_packed_i32 exposes them), only the f32 summation order differs.
//
// Why a repack.  A GGUF row is a string of 74..136-byte super-blocks, 2-byte aligned, fields of one sub-block scattered over
// the block: a lane that owns a sub-block needs 3-4 narrow unaligned loads for 12 bytes, and the stream runs at well under
// half of what 16-byte coalesced loads reach.  The packed form holds exactly the same bytes (a permutation, no padding
// beyond rounding the row count up to 8) ordered the way a wave consumes them:
//
//   unit   = 8 consecutive rows x one K-slice of nb <= 16 super-blocks (a row of n_in / 256 super-blocks is cut into
//            S = ceil(n_sb / 16) slices of near-equal length); units of a row group are stored slice after slice.
//   lane l of the wave that owns a unit (l < L = 4 nb) owns sub-blocks 2l and 2l + 1 of every row of the slice; inside the
//   unit each field is a region of 16-byte lane entries, so that every load is one coalesced dwordx4 per lane:
//     IQ3_XXS  A [row][lane]      16 B  qs of both sub-blocks         B [row pair][lane] 2 x 8 B  scales_and_signs words
//              C [super-block]    8 x 2 B  f16 d of the 8 rows (shared by the 4 lanes of the super-block)
//     IQ3_S    A qs as above      B [row pair][lane] 2 x 8 B signs    Q [lane] 8 x 2 B qh     Sc [lane] 8 x 1 B scales    C d
//     IQ2_S    A [row][lane]      qs (8 B) | signs (8 B)              Q [row quad][lane] 4 x (qh 2 B | scales 2 B)        C d
//   13 / 15 / 11 loads per lane move a unit's 196 / 220 / 164 bytes per lane, all requested before anything is decoded.
//
// The decode is bound by vector-ALU issue (a wave64 integer instruction occupies its SIMD for 4 cycles), so the kernel is
// built to spend as few instructions per sub-block as the format allows (~45 for IQ3_XXS against ~77 for the first packed
// version and for the raw-layout kernel):
//   * the activations of a lane's two sub-blocks never change: 64 int8 + their bytewise negation + 2 scales in registers;
//   * codebook and sign tables sit in LDS in 256-byte slots: slot e = 32 copies of grid entry e (lane l reads copy l % 32 =
//     its own bank: conflict-free) followed by 16 copies of sign entry e.  With a 256-byte stride the address of a gather is
//     (index byte << 8) | lane offset, which ONE v_perm_b32 assembles from the code word and the lane's offset register;
//   * a sign entry holds two v_perm selectors: signed activations = perm(-u, u, selector), then one v_dot4 per 4 weights
//     (instead of xor / and / add on the codebook word, or two dot products).
#include "iq_common.h"
#include <stdlib.h>

#ifndef VMX_IQ_AHEAD
#define VMX_IQ_AHEAD 3
#endif

namespace {

constexpr int PK_ROWS = 8;

__host__ __device__ inline int pk_num_slices(int nsb) { return (nsb + 15) / 16; }
__host__ __device__ inline void pk_slice(int nsb, int S, int s, int &sb0, int &nb) {
    const int base = nsb / S, rem = nsb % S;
    nb = base + (s < rem ? 1 : 0);
    sb0 = s * base + (s < rem ? s : rem);
}

template <int T> struct Pk;
// byte offsets of the regions inside a unit of L = 4 nb lanes, LDS table bytes
template <> struct Pk<18> { static constexpr int TS = 98; };
template <> struct Pk<21> { static constexpr int TS = 110; };
template <> struct Pk<22> { static constexpr int TS = 82; };

__host__ __device__ inline bool pk_supported(int t) { return t == 18 || t == 21 || t == 22; }
__host__ __device__ inline int pk_type_size(int t) { return t == 18 ? 98 : t == 21 ? 110 : t == 22 ? 82 : 0; }
inline int pk_lds_bytes(int t) { return t == 18 ? 65536 : t == 21 ? 131072 : 98304; }

// ------------------------------------------------------------------ repack (load time)
// One thread per (padded row, super-block): scatter the raw block's fields to their places in the unit.
__device__ __forceinline__ void copy_bytes(uint8_t *dst, const uint8_t *src, int n) {
    for (int i = 0; i < n; i++) dst[i] = src[i];
}

template <int T>
__global__ __launch_bounds__(256) void iq_pack_kernel(const uint8_t *__restrict__ raw, uint8_t *__restrict__ packed, int n_out, int nsb) {
    constexpr int TS = Pk<T>::TS;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n_out * nsb;
    if (idx >= total) return;
    const int row = (int)(idx / nsb), bg = (int)(idx % nsb);
    const int S = pk_num_slices(nsb);
    // slice of super-block bg
    const int base = nsb / S, rem = nsb % S;
    int s = bg < rem * (base + 1) ? bg / (base + 1) : rem + (bg - rem * (base + 1)) / base;
    int sb0, nb;
    pk_slice(nsb, S, s, sb0, nb);
    const int b = bg - sb0, L = 4 * nb, r = row & 7;
    uint8_t *U = packed + (int64_t)(row >> 3) * PK_ROWS * nsb * TS + (int64_t)PK_ROWS * TS * sb0;
    const uint8_t *blk = raw + ((int64_t)row * nsb + bg) * TS;
    copy_bytes(U + (T == 18 ? 192 * L : T == 21 ? 216 * L : 160 * L) + 16 * b + 2 * r, blk, 2);   // C: d
#pragma unroll
    for (int h = 0; h < 4; h++) {   // the four lanes of the super-block: sub-blocks ib0 = 2h, 2h + 1
        const int l = 4 * b + h, ib0 = 2 * h;
        if constexpr (T == 18) {
            copy_bytes(U + r * 16 * L + 16 * l, blk + 2 + 8 * ib0, 16);
            copy_bytes(U + 128 * L + (r >> 1) * 16 * L + 16 * l + 8 * (r & 1), blk + 66 + 4 * ib0, 8);
        } else if constexpr (T == 21) {
            copy_bytes(U + r * 16 * L + 16 * l, blk + 2 + 8 * ib0, 16);
            copy_bytes(U + 128 * L + (r >> 1) * 16 * L + 16 * l + 8 * (r & 1), blk + 74 + 4 * ib0, 8);
            copy_bytes(U + 192 * L + 16 * l + 2 * r, blk + 66 + ib0, 2);
            copy_bytes(U + 208 * L + 8 * l + r, blk + 106 + (ib0 >> 1), 1);
        } else {
            copy_bytes(U + r * 16 * L + 16 * l, blk + 2 + 4 * ib0, 8);
            copy_bytes(U + r * 16 * L + 16 * l + 8, blk + 34 + 4 * ib0, 8);
            copy_bytes(U + 128 * L + (r >> 2) * 16 * L + 16 * l + 4 * (r & 3), blk + 66 + ib0, 2);
            copy_bytes(U + 128 * L + (r >> 2) * 16 * L + 16 * l + 4 * (r & 3) + 2, blk + 74 + ib0, 2);
        }
    }
}

// ------------------------------------------------------------------ tables
// Sign entry of sign byte s8 as two v_perm_b32 selectors: byte j = j (keep u_j) or 4 + j (take -u_j) for weights j and 4 + j.
__device__ __forceinline__ uint2 sign_sel_entry(uint32_t s8) {
    const uint32_t l0 = __umul24(s8 & 0xFu, 0x00204081u) & 0x01010101u, l1 = __umul24(s8 >> 4, 0x00204081u) & 0x01010101u;
    return make_uint2(0x03020100u | (l0 << 2), 0x03020100u | (l1 << 2));
}

// The LDS image of each type's tables, byte for byte, kept in device memory (built once per GPU by vmx_init): a workgroup
// fills its LDS with plain LDS-DMA copies of it (no registers, no vector-ALU work, served from L2 after the first workgroup).
//   IQ3_XXS / IQ3_S: 256-byte slots, slot e = grid entry e x 32 copies (128 B) | 16 copies of a sign entry (128 B; IQ3_XXS: the
//   entry of the 7-bit index e & 127 with its parity bit, IQ3_S: of sign byte e & 255).  Every sign entry thus exists in two
//   slots, N / 2 apart: lanes 0..15 of a half-wave read the first, lanes 16..31 the second.
//   IQ2_S: grid (8-byte entries) x 8 copies = 64 B at e * 64; sign entries x 16 copies at 65536 + e * 128.
static __device__ __align__(16) unsigned char g_pk_tab18[65536];
static __device__ __align__(16) unsigned char g_pk_tab21[131072];
static __device__ __align__(16) unsigned char g_pk_tab22[98304];

template <int T> __device__ __forceinline__ unsigned char *pk_table_image() {
    if constexpr (T == 18) return g_pk_tab18;
    else if constexpr (T == 21) return g_pk_tab21;
    else return g_pk_tab22;
}
template <int T> struct PkTab { static constexpr int BYTES = T == 18 ? 65536 : T == 21 ? 131072 : 98304; };

// one thread per 16-byte chunk of the image
template <int T> __global__ __launch_bounds__(256) void pk_build_tables_kernel() {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= PkTab<T>::BYTES / 16) return;
    uint4 v;
    if constexpr (T == 22) {
        if (c < 1024 * 4) {
            const uint2 g = reinterpret_cast<const uint2 *>(iq2s_grid)[c >> 2];
            v = make_uint4(g.x, g.y, g.x, g.y);
        } else {
            const uint2 m = sign_sel_entry((uint32_t)((c - 4096) >> 3));
            v = make_uint4(m.x, m.y, m.x, m.y);
        }
    } else {
        const int e = c >> 4;
        if ((c & 15) < 8) {
            const uint32_t g = T == 18 ? iq3xxs_grid[e] : iq3s_grid[e];
            v = make_uint4(g, g, g, g);
        } else {
            uint32_t s8 = (uint32_t)e & (T == 18 ? 127u : 255u);
            if constexpr (T == 18) s8 |= (__popc(s8) & 1u) << 7;
            const uint2 m = sign_sel_entry(s8);
            v = make_uint4(m.x, m.y, m.x, m.y);
        }
    }
    reinterpret_cast<uint4 *>(pk_table_image<T>())[c] = v;
}

// Request the workgroup's copy of the image: wave w takes the 1 KiB pieces w, w + W, ...; lane l of a piece brings 16 bytes.
// Asynchronous (counted in vmcnt like any load); issued before the weight requests so that it completes first.
template <int T> __device__ __forceinline__ void pk_tables_request(unsigned char *smem, int wave, int W, int lane) {
    const unsigned char *img = pk_table_image<T>();
    for (int c0 = wave * 64; c0 < PkTab<T>::BYTES / 16; c0 += W * 64)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(img + (c0 + lane) * 16),
                                         (__attribute__((address_space(3))) void *)(smem + c0 * 16), 16, 0, 0);
}

// ------------------------------------------------------------------ part registers
// A wave works on PPW consecutive "parts" (row pairs) of a unit.  Everything a part needs is requested small fields first,
// then the sign / scale words, then the 16-byte code entries, part after part: vmcnt retires in issue order, so part 0 can
// be decoded while the later parts are still in flight.
template <int T> struct PkPart;
template <> struct PkPart<18> { uint4 A[2], B; uint32_t d2; };
template <> struct PkPart<21> { uint4 A[2], B; uint32_t d2, qh2, sc2; };
template <> struct PkPart<22> { uint4 A[2]; uint2 Q; uint32_t d2; };

template <int T> __device__ __forceinline__ void pk_load_small(PkPart<T> &r, const uint8_t *__restrict__ U, int L, int lc, int rp) {
    if constexpr (T == 18) {
        r.d2 = *reinterpret_cast<const uint32_t *>(U + 192 * L + 16 * (lc >> 2) + 4 * rp);
    } else if constexpr (T == 21) {
        r.d2 = *reinterpret_cast<const uint32_t *>(U + 216 * L + 16 * (lc >> 2) + 4 * rp);
        r.qh2 = *reinterpret_cast<const uint32_t *>(U + 192 * L + 16 * lc + 4 * rp);
        r.sc2 = *reinterpret_cast<const uint16_t *>(U + 208 * L + 8 * lc + 2 * rp);
    } else {
        r.d2 = *reinterpret_cast<const uint32_t *>(U + 160 * L + 16 * (lc >> 2) + 4 * rp);
        r.Q = *reinterpret_cast<const uint2 *>(U + 128 * L + (rp >> 1) * 16 * L + 16 * lc + 8 * (rp & 1));
    }
}

// Row r (0 .. 2 PPW - 1) of an item: its 16-byte code entry, and with an even row the sign / scale words of its pair
template <int T> __device__ __forceinline__ void pk_load_row(PkPart<T> *raw, const uint8_t *__restrict__ U, int L, int lc, int rp0, int r) {
    const int rp = rp0 + (r >> 1);
    PkPart<T> &d = raw[r >> 1];
    if constexpr (T != 22) {
        if ((r & 1) == 0) d.B = *reinterpret_cast<const uint4 *>(U + 128 * L + rp * 16 * L + 16 * lc);
    }
    d.A[r & 1] = *reinterpret_cast<const uint4 *>(U + (2 * rp + (r & 1)) * 16 * L + 16 * lc);
}

__device__ __forceinline__ uint32_t w4(const uint4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// C truncating division by 2 and 4 without a divide
__device__ __forceinline__ int div2(int x) { return (x + (int)((uint32_t)x >> 31)) >> 1; }
__device__ __forceinline__ int div4(int x) { return (x + (int)(((uint32_t)(x >> 31)) >> 30)) >> 2; }
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }   // |a|, |b| < 2^23 here: full-rate multiply

// Address of the table slot of index byte k of `w`: (byte << 8) | off, where off has bits only in its bytes 0 and 2
template <int K> __device__ __forceinline__ uint32_t slot_addr(uint32_t w, uint32_t off) {
    return __builtin_amdgcn_perm(w, off, 0x0c020000u | ((4u + K) << 8));
}

// (a & mask) | c in one instruction (the compiler splits the C expression into v_and + v_add)
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t c) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(c));
    return r;
}

// Table reads by absolute LDS address: the tables are the kernel's only LDS object and start at address 0, so a gather
// address needs no base add (through the array's symbol every address costs one more v_add).
typedef __attribute__((address_space(3))) const uint32_t pk_lds_u32;
typedef uint32_t pk_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const pk_u32x2 pk_lds_u64;
__device__ __forceinline__ uint32_t lds32(uint32_t addr) { return *reinterpret_cast<pk_lds_u32 *>(addr); }
__device__ __forceinline__ uint2 lds64(uint32_t addr) {
    const pk_u32x2 v = *reinterpret_cast<pk_lds_u64 *>(addr);
    return make_uint2(v.x, v.y);
}

// Integer sub-dot (already scaled: what vec_dot_iq*_q8_1 multiplies by d * a_d) of sub-block j (0 / 1) of row ri (0 / 1) of a
// part, returned as a float (exact: |value| < 2^24).  u / nu = the 8 activation words of that sub-block and their bytewise
// negation; lane_g / lane_s = the lane's byte offsets of its grid copy and its sign copy inside a slot.
template <int T>
__device__ __forceinline__ float pk_subdot(const PkPart<T> &raw, int ri, int j, const int u[8], const int nu[8], uint32_t lane_g,
                                           uint32_t lane_s) {
    auto signed_u = [&](int k, uint32_t sel) { return (int)__builtin_amdgcn_perm((uint32_t)nu[k], (uint32_t)u[k], sel); };
    if constexpr (T == 18) {
        const uint32_t q0 = j ? raw.A[ri].z : raw.A[ri].x, q1 = j ? raw.A[ri].w : raw.A[ri].y;
        const uint32_t aux = w4(raw.B, ri * 2 + j);
        const uint32_t ga[8] = {slot_addr<0>(q0, lane_g), slot_addr<1>(q0, lane_g), slot_addr<2>(q0, lane_g), slot_addr<3>(q0, lane_g),
                                slot_addr<0>(q1, lane_g), slot_addr<1>(q1, lane_g), slot_addr<2>(q1, lane_g), slot_addr<3>(q1, lane_g)};
        const uint32_t sa[4] = {and_or(aux << 8, 0x7F00u, lane_s), and_or(aux << 1, 0x7F00u, lane_s), and_or(aux >> 6, 0x7F00u, lane_s),
                                and_or(aux >> 13, 0x7F00u, lane_s)};
        int a = 0;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const uint32_t g0 = lds32(ga[2 * l]), g1 = lds32(ga[2 * l + 1]);
            const uint2 mk = lds64(sa[l]);
            a = dot4((int)g0, signed_u(2 * l, mk.x), a);
            a = dot4((int)g1, signed_u(2 * l + 1, mk.y), a);
        }
        // (ls * sumi + sumi / 2) / 2 with C truncating divisions (iq_dequant.cu vec_dot_iq3_xxs_q8_1) == trunc((2 ls + 1) sumi / 4):
        // 2 ls + 1 is odd, so y = (2 ls + 1) sumi has the parity of sumi and dropping that parity bit towards zero before the
        // division by 4 cannot change the truncated quotient.  |y| < 2^23: exact in the 24-bit multiply and in f32.
        const int y = mul24(a, (int)((aux >> 27) | 1u));
        return truncf(0.25f * (float)y);
    } else if constexpr (T == 21) {
        const uint32_t q0 = j ? raw.A[ri].z : raw.A[ri].x, q1 = j ? raw.A[ri].w : raw.A[ri].y;
        const uint32_t sgn = w4(raw.B, ri * 2 + j);
        const uint32_t qh = raw.qh2 >> (16 * ri + 8 * j);   // bit k = ninth index bit of lookup k (higher bits are masked off below)
        const uint32_t gb[8] = {slot_addr<0>(q0, lane_g), slot_addr<1>(q0, lane_g), slot_addr<2>(q0, lane_g), slot_addr<3>(q0, lane_g),
                                slot_addr<0>(q1, lane_g), slot_addr<1>(q1, lane_g), slot_addr<2>(q1, lane_g), slot_addr<3>(q1, lane_g)};
        const uint32_t sa[4] = {slot_addr<0>(sgn, lane_s), slot_addr<1>(sgn, lane_s), slot_addr<2>(sgn, lane_s), slot_addr<3>(sgn, lane_s)};
        int a = 0;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            // iq3s index = byte | bit << 8: lookups 2l, 2l + 1 take qh bits 2l, 2l + 1 (iq/mod.rs:223-271) -> address bit 16
            const uint32_t a0 = and_or(qh << (16 - 2 * l), 0x10000u, gb[2 * l]);
            const uint32_t a1 = and_or(qh << (15 - 2 * l), 0x10000u, gb[2 * l + 1]);
            const uint32_t g0 = lds32(a0), g1 = lds32(a1);
            const uint2 mk = lds64(sa[l]);
            a = dot4((int)g0, signed_u(2 * l, mk.x), a);
            a = dot4((int)g1, signed_u(2 * l + 1, mk.y), a);
        }
        const uint32_t nib = (raw.sc2 >> (8 * ri + 4 * j)) & 0xFu;
        return (float)mul24(a, (int)(2u * nib + 1u));
    } else {
        const uint32_t qs = j ? raw.A[ri].y : raw.A[ri].x, sgn = j ? raw.A[ri].w : raw.A[ri].z;
        const uint32_t qw = ri ? raw.Q.y : raw.Q.x;
        const uint32_t qh = (qw >> (8 * j)) & 0xFFu, sc = (qw >> (16 + 8 * j)) & 0xFFu;
        int a[2] = {0, 0};
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const uint32_t idx = ((qs >> (8 * l)) & 0xFFu) | ((qh << (8 - 2 * l)) & 0x300u);
            const uint2 g = lds64((idx << 6) + lane_g);
            const uint2 mk = lds64(65536u + (((sgn >> (8 * l)) & 0xFFu) << 7) + lane_s);
            a[l >> 1] = dot4((int)g.x, signed_u(2 * l, mk.x), a[l >> 1]);
            a[l >> 1] = dot4((int)g.y, signed_u(2 * l + 1, mk.y), a[l >> 1]);
        }
        const int ls0 = (int)(sc & 0xFu), ls1 = (int)(sc >> 4);
        return (float)div4(mul24(a[0], ls0) + mul24(a[1], ls1) + div2(a[0] + a[1]));
    }
}

// Sum NR (2, 4 or 8) values per lane over the wave: afterwards the lanes with (lane & 7) == 0 hold the total of value
// index built from lane bits 5, 4, 3 (the top log2(NR) of them).  Halving exchanges first, then plain butterflies.
template <int NR> __device__ __forceinline__ float reduce_rows(const float *v, int lane, int &rsel) {
    float w[4], x[2], z;
    const bool h5 = (lane & 32) != 0, h4 = (lane & 16) != 0, h3 = (lane & 8) != 0;
    if constexpr (NR == 8) {
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = (h5 ? v[i + 4] : v[i]) + vmx_lane_xor<32>(h5 ? v[i] : v[i + 4]);
#pragma unroll
        for (int i = 0; i < 2; i++) x[i] = (h4 ? w[i + 2] : w[i]) + vmx_lane_xor<16>(h4 ? w[i] : w[i + 2]);
        z = (h3 ? x[1] : x[0]) + vmx_lane_xor<8>(h3 ? x[0] : x[1]);
        rsel = (h5 ? 4 : 0) + (h4 ? 2 : 0) + (h3 ? 1 : 0);
    } else if constexpr (NR == 4) {
#pragma unroll
        for (int i = 0; i < 2; i++) x[i] = (h5 ? v[i + 2] : v[i]) + vmx_lane_xor<32>(h5 ? v[i] : v[i + 2]);
        z = (h4 ? x[1] : x[0]) + vmx_lane_xor<16>(h4 ? x[0] : x[1]);
        z += vmx_lane_xor<8>(z);
        rsel = (h5 ? 2 : 0) + (h4 ? 1 : 0);
    } else {
        z = (h5 ? v[1] : v[0]) + vmx_lane_xor<32>(h5 ? v[0] : v[1]);
        z += vmx_lane_xor<16>(z);
        z += vmx_lane_xor<8>(z);
        rsel = h5 ? 1 : 0;
    }
    z += vmx_lane_xor<4>(z);
    z += vmx_lane_xor<2>(z);
    z += vmx_lane_xor<1>(z);
    return z;
}

struct PkSeg {
    const uint8_t *w;   // packed weights
    float *y;           // output of this segment
    int type, n_out, wg0;
};
struct PkArgs {
    PkSeg seg[3];
    int nseg;
    // GeGLU epilogue (iq_pk_mmvq_kernel<.., GLU = true>): seg[0] = gate, seg[1] = up; outputs of the fused activation
    uint16_t *glu_act;   // bf16 [inter]
    int8_t *glu_aq;      // its q8_1 form for the down projection
    float *glu_ad;
    // sandwich-norm prologue (NORM = true): the activation row is computed by every workgroup from the previous linear's f32 output
    // o and the residual stream h instead of being read as q8_1 (see pk_norm_prologue)
    const float *n_o;
    const uint16_t *n_h_in, *n_wpost, *n_wnext;
    uint16_t *n_h_out;
    float n_eps;
    const uint16_t *n_scalar;   // optional layer scalar: h' = bf16(h' * scalar) before the second norm
};

__device__ __forceinline__ float pk_rbf(float x) { return vmx_bf16_to_f32(vmx_f32_to_bf16(x)); }
__device__ __forceinline__ void pk_unpack8(const uint4 &v, float f[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) { f[2 * j] = __uint_as_float(w[j] << 16); f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
}
__device__ __forceinline__ uint4 pk_pack8(const float f[8]) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = (uint32_t)vmx_f32_to_bf16(f[2 * j]) | ((uint32_t)vmx_f32_to_bf16(f[2 * j + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Registers of the norm prologue: thread t < hidden / 8 owns the 8-element chunk t of the row
struct PkNormIn {
    float4 oa, ob;
    uint4 h, wp, wn;
    uint16_t sc;
};

__device__ __forceinline__ void pk_norm_request(PkNormIn &r, const PkArgs &a, int n_in, int tid) {
    const int ci = min(tid, (n_in >> 3) - 1);
    r.oa = *reinterpret_cast<const float4 *>(a.n_o + 8 * ci);
    r.ob = *reinterpret_cast<const float4 *>(a.n_o + 8 * ci + 4);
    r.h = *reinterpret_cast<const uint4 *>(a.n_h_in + 8 * ci);
    r.wp = *reinterpret_cast<const uint4 *>(a.n_wpost + 8 * ci);
    r.wn = *reinterpret_cast<const uint4 *>(a.n_wnext + 8 * ci);
    r.sc = a.n_scalar ? a.n_scalar[0] : (uint16_t)0x3f80;
}

// Gemma's sandwich norms between o_proj and the MLP (gemma4_quantized.rs:1372-1392), with the bf16 rounding points of the
// separate norm kernel:  o_r = bf16(o);  n = bf16(o_r * rsqrt(mean(o_r^2) + eps) * w_post);  h' = bf16(n + h);
// x = bf16(h' * rsqrt(mean(h'^2) + eps) * w_next);  q8_1(x) -> LDS (aq_s int8 [hidden], ad_s f32 [hidden / 32]).
// Every workgroup computes the whole row (hidden <= 4096: one chunk per thread at most); workgroup 0 also stores h' to
// n_h_out, a buffer other than n_h_in (the other workgroups may still be reading that one).  Ends with the LDS stores
// issued, not yet waited for: the caller's barrier does that.
__device__ __forceinline__ void pk_norm_prologue(const PkNormIn &r, const PkArgs &a, int n_in, int tid, int8_t *aq_s, float *ad_s, float *red) {
    const int nchunk = n_in >> 3, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nlive = (nchunk + 63) >> 6;   // waves that own chunks; the others only keep the barriers company
    const bool wlive = wave < nlive, live = tid < nchunk;
    // the per-wave sums of the workgroup in wave order (one vector read per four waves, not a loop of dependent reads)
    auto block_total = [&](const float *rd) {
        const float4 *r4 = reinterpret_cast<const float4 *>(rd);
        const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2], q3 = r4[3];
        const float x[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; w++) t += w < nlive ? x[w] : 0.f;
        return t;
    };
    float ov[8] = {r.oa.x, r.oa.y, r.oa.z, r.oa.w, r.ob.x, r.ob.y, r.ob.z, r.ob.w};
    if (wlive) {
        float v1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) { ov[j] = pk_rbf(ov[j]); v1 += live ? ov[j] * ov[j] : 0.f; }
        v1 = vmx_wave_sum(v1);
        if ((tid & 63) == 0) red[wave] = v1;
    }
    __syncthreads();
    float hv[8];
    if (wlive) {
        const float inv1 = rsqrtf(block_total(red) / (float)n_in + a.n_eps);
        float wv[8], hr[8];
        pk_unpack8(r.wp, wv);
        pk_unpack8(r.h, hr);
        float v2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            hv[j] = pk_rbf(pk_rbf(ov[j] * inv1 * wv[j]) + hr[j]);
            if (a.n_scalar) hv[j] = pk_rbf(hv[j] * vmx_bf16_to_f32(r.sc));   // layer_scalar of the layer whose tail this is
            v2 += live ? hv[j] * hv[j] : 0.f;
        }
        if (blockIdx.x == 0 && live) *reinterpret_cast<uint4 *>(a.n_h_out + 8 * tid) = pk_pack8(hv);
        v2 = vmx_wave_sum(v2);
        if ((tid & 63) == 0) red[16 + wave] = v2;
    }
    __syncthreads();
    if (wlive) {
        const float inv2 = rsqrtf(block_total(red + 16) / (float)n_in + a.n_eps);
        float wv[8], xr[8];
        pk_unpack8(r.wn, wv);
#pragma unroll
        for (int j = 0; j < 8; j++) xr[j] = pk_rbf(hv[j] * inv2 * wv[j]);
        // four consecutive threads hold one 32-value block (nchunk % 4 == 0: a group is all live or all not)
        const int ci = live ? tid : nchunk - 1;
        float scratch_d;
        vmx_q8_1_emit8(xr, live ? aq_s + 8 * ci : aq_s + n_in, live ? ad_s + (ci >> 2) : &scratch_d, (ci & 3) == 0);
    }
}

// One wave = one work item = PPW row pairs of one unit (rows 2 PPW c .. 2 PPW (c + 1) - 1 of row group g, K-slice s; item
// index ((g * (4 / PPW) + c) * S + s).  Workgroup = W waves on consecutive items of one segment (W % S == 0, so the S slices
// of a row set sit in one workgroup and are combined through LDS in slice order).
template <int T, int PPW, bool DBG, int ABL = 0, bool GLU = false, bool NORM = false>
__device__ __forceinline__ void pk_body(unsigned char *smem, float *part, const PkSeg &sg, int wg, const int8_t *__restrict__ aq,
                                        const float *__restrict__ ad, int n_in, int S, int32_t *__restrict__ dbg, int issue_barrier,
                                        const PkArgs &na) {
    constexpr int TS = Pk<T>::TS, CPU_ = 4 / PPW, NR = 2 * PPW;   // chunks per unit, rows per item
#ifndef VMX_IQ_AHEAD_NORM
#define VMX_IQ_AHEAD_NORM VMX_IQ_AHEAD
#endif
    // rows requested ahead of the one being decoded (NORM: the requests made before the prologue are all that is in flight during it)
    constexpr int AHEAD = NORM ? VMX_IQ_AHEAD_NORM : VMX_IQ_AHEAD;
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), W = nthreads >> 6;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ABL == 3: shader-clock stamps of this wave
    auto stamp = [&](int i, float dep) {
        if constexpr (ABL == 3) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[i]) : "v"(dep) : "memory");
    };
    stamp(0, 0.f);
    unsigned long long rt0 = 0;
    if constexpr (ABL == 3) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0) :: "memory");
    // NORM: the activation row is computed here; its inputs are requested first (loads retire in issue order)
    PkNormIn nin;
    int8_t *aq_s = reinterpret_cast<int8_t *>(part) + 512;                 // [n_in] + 8 spare bytes
    float *ad_s = reinterpret_cast<float *>(aq_s + 4096 + 16);             // [n_in / 32]
    float *nred = ad_s + 128;                                              // [32]
    if constexpr (NORM) pk_norm_request(nin, na, n_in, tid);
#ifndef VMX_IQ_NOTAB
    pk_tables_request<T>(smem, wave, W, lane);
#endif
    const int nsb = n_in >> 8;
    const int groups = (sg.n_out + PK_ROWS - 1) / PK_ROWS, items = groups * CPU_ * S;
    // GLU: 16 waves = 8 gate units (waves 0..7) and the 8 up units of the same rows (waves 8..15)
    const int item = GLU ? wg * 8 + (wave & 7) : wg * W + wave;
    const bool valid = item < items;
    const int ic = valid ? item : items - 1;
    const int slice = ic % S, gc = ic / S, group = gc / CPU_, chunk = gc % CPU_;
    int sb0, nb;
    pk_slice(nsb, S, slice, sb0, nb);
    const int L = 4 * nb, lc = min(lane, L - 1);
    const bool live = valid && lane < L;
    // activations of the lane's two sub-blocks
    const int g0 = sb0 * 8 + 2 * lc;
#ifdef VMX_IQ_NOACT
    const int4 *up = reinterpret_cast<const int4 *>(aq);   // measurement only: every wave reads the same 64 bytes
#else
    const int4 *up = reinterpret_cast<const int4 *>(aq + (int64_t)g0 * 32);
#endif
    int4 u0, u1, u2, u3;
    float2 dav;
    if constexpr (!NORM) {
        u0 = up[0]; u1 = up[1]; u2 = up[2]; u3 = up[3];
        dav = *reinterpret_cast<const float2 *>(ad + g0);
    }
    // weights: every load of the item, in the order the decode wants them back
    const uint8_t *U = sg.w + (int64_t)group * PK_ROWS * nsb * TS + (int64_t)PK_ROWS * TS * sb0;
    PkPart<T> raw[PPW];
    if constexpr (ABL == 2) {   // measurement: no weight stream, register data that the compiler cannot fold
#pragma unroll
        for (int p = 0; p < PPW; p++) {
            uint32_t *wr = reinterpret_cast<uint32_t *>(&raw[p]);
            for (int k = 0; k < (int)(sizeof(PkPart<T>) / 4); k++) wr[k] = ((uint32_t)(lane * 2654435761u) ^ (uint32_t)(u0.x + k + 16 * p)) & 0x7F7F7F7Fu;
        }
    } else {
        // A compute unit's memory pipeline holds only so many requests: once it is full, issuing a load BLOCKS the wave until
        // earlier data has arrived (measured: a wave that requests its whole unit up front spends as long issuing as the
        // stream takes, and decodes only afterwards).  So only the small fields and the first AHEAD rows are requested here, and
        // row r + AHEAD right before row r is decoded.
#pragma unroll
        for (int p = 0; p < PPW; p++) pk_load_small<T>(raw[p], U, L, lc, chunk * PPW + p);
#pragma unroll
        for (int r = 0; r < (NR < AHEAD ? NR : AHEAD); r++) pk_load_row<T>(raw, U, L, lc, chunk * PPW, r);
    }
    stamp(1, 0.f);
    if constexpr (NORM) pk_norm_prologue(nin, na, n_in, tid, aq_s, ad_s, nred);
    // tables of every wave of the workgroup (and this wave's first row pair) have landed (NORM: and the q8_1 row is in LDS)
    if constexpr (NORM) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (NORM) {
        const int4 *us = reinterpret_cast<const int4 *>(aq_s + g0 * 32);
        u0 = us[0]; u1 = us[1]; u2 = us[2]; u3 = us[3];
        dav = *reinterpret_cast<const float2 *>(ad_s + g0);
    }
    stamp(5, 0.f);
    float da[2] = {live ? dav.x : 0.f, live ? dav.y : 0.f};
    if constexpr (ABL == 1) {   // measurement: the weight stream alone
#pragma unroll
        for (int r = AHEAD; r < NR; r++) pk_load_row<T>(raw, U, L, lc, chunk * PPW, r);
        uint32_t x = 0;
#pragma unroll
        for (int p = 0; p < PPW; p++) {
            x ^= raw[p].A[0].x ^ raw[p].A[0].y ^ raw[p].A[0].z ^ raw[p].A[0].w ^ raw[p].A[1].x ^ raw[p].A[1].y ^ raw[p].A[1].z ^ raw[p].A[1].w ^ raw[p].d2;
            if constexpr (T != 22) x ^= raw[p].B.x ^ raw[p].B.y ^ raw[p].B.z ^ raw[p].B.w;
        }
        if (x == 0x12345678u && valid) sg.y[0] = 1.0f;
        return;
    }

    asm volatile("" : "+v"(da[0]), "+v"(da[1]));   // computed once (otherwise the select is redone per row); after the requests above
    const int ua[2][8] = {{u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w}, {u2.x, u2.y, u2.z, u2.w, u3.x, u3.y, u3.z, u3.w}};
    int nua[2][8];   // bytewise -u (q8_1 values lie in -127..127, so the negation is representable)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t x = (uint32_t)ua[j][k];
            // per byte ~x + 1 without carries between bytes: add inside the low 7 bits, fold the carry into bit 7 with xor
            const uint32_t t = (~x & 0x7F7F7F7Fu) + 0x01010101u;
            nua[j][k] = (int)(t ^ (~x & 0x80808080u));
            asm volatile("" : "+v"(nua[j][k]));   // keep it in a register: recomputing it at every use costs 3 instructions per word
        }
    const uint32_t lane_g = T == 22 ? (uint32_t)(lane & 7) * 8u : (uint32_t)(lane & 31) * 4u;
    const uint32_t lane_s = T == 22 ? (uint32_t)(lane & 15) * 8u
                                    : 128u + (uint32_t)(lane & 15) * 8u + ((lane & 16) ? (T == 18 ? 32768u : 65536u) : 0u);
    float acc[NR];
#pragma unroll
    for (int p = 0; p < PPW; p++) {
#pragma unroll
        for (int ri = 0; ri < 2; ri++) {
            if constexpr (ABL != 2) {
                if (2 * p + ri + AHEAD < NR) {   // rows 0 .. AHEAD - 1 were requested above
                    __builtin_amdgcn_sched_barrier(0);   // keep the request here: hoisted to the top it would block the wave again
                    pk_load_row<T>(raw, U, L, lc, chunk * PPW, 2 * p + ri + AHEAD);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const float wd = vmx_f16_to_f32((uint16_t)(raw[p].d2 >> (16 * ri)));
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float sub = pk_subdot<T>(raw[p], ri, j, ua[j], nua[j], lane_g, lane_s);
                a += (wd * da[j]) * sub;
                if constexpr (DBG) {
                    const int row = group * PK_ROWS + chunk * NR + 2 * p + ri;
                    if (live && row < sg.n_out) dbg[(int64_t)row * (n_in >> 5) + g0 + j] = (int)sub;
                }
            }
            acc[2 * p + ri] = a;
        }
        if (p == 0) stamp(6, acc[1]);
    }
    if constexpr (DBG) return;
    int rsel;
    const float z = reduce_rows<NR>(acc, lane, rsel);
    if constexpr (GLU) {
        if ((lane & 7) == 0) part[wave * 8 + rsel] = valid ? z : 0.f;   // [gate 64 | up 64] of the workgroup's 64 columns
    } else if (S == 1) {
        const int row = group * PK_ROWS + chunk * NR + rsel;
        if ((lane & 7) == 0 && valid && row < sg.n_out) sg.y[row] = z;
    } else {
        if ((lane & 7) == 0) part[wave * 8 + rsel] = valid ? z : 0.f;
        __syncthreads();
        const int ns = W / S;   // row sets of this workgroup
        if (tid < ns * NR) {
            const int si = tid / NR, r = tid % NR;
            float s = 0.f;
            for (int k = 0; k < S; k++) s += part[(si * S + k) * 8 + r];
            const int gcw = (wg * W) / S + si;   // (group, chunk) of this row set
            const int row = (gcw / CPU_) * PK_ROWS + (gcw % CPU_) * NR + r;
            if (gcw < groups * CPU_ && row < sg.n_out) sg.y[row] = s;
        }
    }
    if constexpr (ABL == 3) {
        stamp(7, z);
        unsigned long long rt1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1) :: "memory");
        if (lane == 0) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(dbg) + ((int64_t)blockIdx.x * 16 + wave) * 10;
            for (int i = 0; i < 8; i++) o[i] = ts[i];
            o[8] = rt0;   // 100 MHz wall clock at entry and end
            o[9] = rt1;
        }
    }
}

template <int PPW, bool DBG, int ABL = 0, bool GLU = false, bool NORM = false>
__global__ __launch_bounds__(1024) void iq_pk_mmvq_kernel(PkArgs a, const int8_t *__restrict__ aq, const float *__restrict__ ad, int n_in,
                                                           int S, int32_t *__restrict__ dbg, int lds_tables, int issue_barrier) {
    // the only LDS object of the kernel, so the tables start at LDS address 0 and a gather needs no base add
    extern __shared__ __align__(16) unsigned char pk_smem[];
    const int b = blockIdx.x;
    int si, wg;
    if constexpr (GLU) {
        si = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 9);   // wave-uniform: waves 8..15 work on the up projection
        wg = b;
    } else {
        si = (a.nseg > 2 && b >= a.seg[2].wg0) ? 2 : ((a.nseg > 1 && b >= a.seg[1].wg0) ? 1 : 0);
        wg = b - a.seg[si].wg0;
    }
    const PkSeg &sg = a.seg[si];
    float *part = reinterpret_cast<float *>(pk_smem + lds_tables);   // [waves][8] slice sums (512 B) behind the tables, then the compact table copy
    switch (GLU ? a.seg[0].type : sg.type) {   // workgroup-uniform (GLU: gate and up share a type)
        case 18: pk_body<18, PPW, DBG, ABL, GLU, NORM>(pk_smem, part, sg, wg, aq, ad, n_in, S, dbg, issue_barrier, a); break;
        case 21: pk_body<21, PPW, DBG, ABL, GLU, NORM>(pk_smem, part, sg, wg, aq, ad, n_in, S, dbg, issue_barrier, a); break;
        default: pk_body<22, PPW, DBG, ABL, GLU, NORM>(pk_smem, part, sg, wg, aq, ad, n_in, S, dbg, issue_barrier, a); break;
    }
    if constexpr (GLU) {
        // act = bf16(bf16(gelu_erf(bf16(gate))) * bf16(up)) and its q8_1 form (two 32-wide blocks), as g4_geglu_kernel computes them
        __syncthreads();
        const int t = threadIdx.x;
        if (t < 64) {
            const int col = b * 64 + t;
            const float gg = vmx_bf16_to_f32(vmx_f32_to_bf16(part[t]));
            const float ge = vmx_bf16_to_f32(vmx_f32_to_bf16(0.5f * gg * (1.0f + erff(gg * 0.70710678118654752f))));
            const uint16_t rb = vmx_f32_to_bf16(ge * vmx_bf16_to_f32(vmx_f32_to_bf16(part[64 + t])));
            const float r = vmx_bf16_to_f32(rb);
            float amax = fabsf(r);
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) amax = fmaxf(amax, vmx_lane_xor_m(amax, o));
            const float d = amax / 127.0f;
            const int q = amax != 0.0f ? (int)roundf(r / d) : 0;
            if (col < a.seg[0].n_out) {
                a.glu_act[col] = rb;
                a.glu_aq[col] = (int8_t)q;
                if ((t & 31) == 0) a.glu_ad[col >> 5] = d;
            }
        }
    }
}

template <int PPW, bool DBG, int ABL = 0, bool GLU = false, bool NORM = false> int pk_launch_ppw(const PkArgs &a, int wgs, int W, int lds, const int8_t *aq, const float *ad, int n_in, int S,
                                               int32_t *dbg, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)iq_pk_mmvq_kernel<PPW, DBG, ABL, GLU, NORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 512 + 8192);
        if (e != hipSuccess) {
            vmx_set_error("iq_mmvq(packed): hipFuncSetAttribute: %s", hipGetErrorString(e));
            return VMX_ELAUNCH;
        }
        attr_set = true;
    }
    static const int issue_barrier = vmx_tune("VMX_IQ_ISSUE_BARRIER", 1);
    hipLaunchKernelGGL((iq_pk_mmvq_kernel<PPW, DBG, ABL, GLU, NORM>), dim3(wgs), dim3(64 * W), lds + 512 + 8192, st, a, aq, ad, n_in, S, dbg, lds, issue_barrier);
    VMX_CHECK_LAUNCH("iq_mmvq(packed)");
    return VMX_OK;
}

bool g_pk_tables_ready[64] = {};   // per device: the table images exist (vmx_init)
void *g_pk_stamps = nullptr;   // measurement: per-wave clock stamps of the next launches (vmx_debug_iq_packed_stamps)

int pk_launch(const VmxMmvqSeg *segs, int nseg, const int8_t *aq, const float *ad, float *y, int n_in, int32_t *dbg, hipStream_t st,
              const PkArgs *norm = nullptr) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g_pk_tables_ready[dev]) {
        vmx_set_error("iq_mmvq(packed): vmx_init(%d) has not run on this device (it builds the decode tables)", dev);
        return VMX_EINVAL;
    }
    const int nsb = n_in / 256, S = pk_num_slices(nsb);
    if (S > 16) {
        vmx_set_error("iq_mmvq(packed): n_in %d beyond 65536", n_in);
        return VMX_EUNSUPPORTED;
    }
    int units_total = 0, lds = 0;
    for (int i = 0; i < nseg; i++) {
        units_total += (segs[i].n_out + PK_ROWS - 1) / PK_ROWS * S;
        const int l = pk_lds_bytes(segs[i].ggml_type);
        lds = l > lds ? l : lds;
    }
    // row pairs per wave: whole units where that already gives every SIMD several waves, fewer rows per wave for small tensors
    static const int force_ppw = vmx_tune("VMX_IQ_PPW", 0);
    int ppw = units_total >= 2048 ? 4 : (units_total >= 1024 ? 2 : 1);
    if (force_ppw == 1 || force_ppw == 2 || force_ppw == 4) ppw = force_ppw;
    if (dbg) ppw = 4;
    const int cpu_ = 4 / ppw;
    // waves per workgroup: spread the items over all compute units, the slices of a row set together
    const int cus = vmx_num_cus();
    int W = (units_total * cpu_ + cus - 1) / cus;
    W = W < 1 ? 1 : (W > 16 ? 16 : W);
    static const int force_w = vmx_tune("VMX_IQ_W", 0);   // measurement only
    if (force_w >= 1 && force_w <= 16) W = force_w;
    W = ((W + S - 1) / S) * S;
    if (W > 16) W = (16 / S) * S;
    PkArgs a = {};
    a.nseg = nseg;
    int wg = 0;
    for (int i = 0; i < nseg; i++) {
        const int items = (segs[i].n_out + PK_ROWS - 1) / PK_ROWS * cpu_ * S;
        a.seg[i] = PkSeg{(const uint8_t *)segs[i].w, y + segs[i].y_off, segs[i].ggml_type, segs[i].n_out, wg};
        wg += (items + W - 1) / W;
    }
    if (norm) {   // sandwich-norm prologue: every workgroup computes the row, one 8-element chunk per thread
        if (S != 1 || 64 * W * 8 < n_in) {
            vmx_set_error("iq_mmvq(packed, norm): %d-wave workgroups / %d K-slices cannot hold a row of %d", W, S, n_in);
            return VMX_EUNSUPPORTED;
        }
        a.n_o = norm->n_o; a.n_h_in = norm->n_h_in; a.n_h_out = norm->n_h_out; a.n_wpost = norm->n_wpost; a.n_wnext = norm->n_wnext;
        a.n_eps = norm->n_eps; a.n_scalar = norm->n_scalar;
        if (ppw == 4) return pk_launch_ppw<4, false, 0, false, true>(a, wg, W, lds, nullptr, nullptr, n_in, S, nullptr, st);
        if (ppw == 2) return pk_launch_ppw<2, false, 0, false, true>(a, wg, W, lds, nullptr, nullptr, n_in, S, nullptr, st);
        return pk_launch_ppw<1, false, 0, false, true>(a, wg, W, lds, nullptr, nullptr, n_in, S, nullptr, st);
    }
    if (dbg) return pk_launch_ppw<4, true>(a, wg, W, lds, aq, ad, n_in, S, dbg, st);
    static const int abl = vmx_tune("VMX_IQ_ABL", 0);   // measurement only: 1 = stream alone, 2 = decode alone
    if (abl == 1 && ppw == 4) return pk_launch_ppw<4, false, 1>(a, wg, W, lds, aq, ad, n_in, S, dbg, st);
    if (abl == 2 && ppw == 4) return pk_launch_ppw<4, false, 2>(a, wg, W, lds, aq, ad, n_in, S, dbg, st);
    if (g_pk_stamps && ppw == 4) return pk_launch_ppw<4, false, 3>(a, wg, W, lds, aq, ad, n_in, S, (int32_t *)g_pk_stamps, st);
    if (ppw == 4) return pk_launch_ppw<4, false>(a, wg, W, lds, aq, ad, n_in, S, dbg, st);
    if (ppw == 2) return pk_launch_ppw<2, false>(a, wg, W, lds, aq, ad, n_in, S, dbg, st);
    return pk_launch_ppw<1, false>(a, wg, W, lds, aq, ad, n_in, S, dbg, st);
}

}  // namespace

bool vmx_iq_packed_supported(int ggml_type) { return pk_supported(ggml_type); }

// Called by vmx_init with `device` current: build the three table images in this GPU's memory (once).
int vmx_iq_packed_init(int device) {
    VMX_REQUIRE(device >= 0 && device < 64, "vmx_init: device %d beyond the 64 this library tracks", device);
    if (g_pk_tables_ready[device]) return VMX_OK;
    hipLaunchKernelGGL(pk_build_tables_kernel<18>, dim3(PkTab<18>::BYTES / 16 / 256), dim3(256), 0, nullptr);
    hipLaunchKernelGGL(pk_build_tables_kernel<21>, dim3(PkTab<21>::BYTES / 16 / 256), dim3(256), 0, nullptr);
    hipLaunchKernelGGL(pk_build_tables_kernel<22>, dim3(PkTab<22>::BYTES / 16 / 256), dim3(256), 0, nullptr);
    VMX_CHECK_LAUNCH("iq_packed table build");
    VMX_CHECK_HIP(hipDeviceSynchronize());
    g_pk_tables_ready[device] = true;
    return VMX_OK;
}

// Measurement aid: while `buf` is non-NULL, packed MMVQ launches over whole 8-row units run a variant that records, per wave,
// 8 shader-clock stamps (entry, loads issued, tables built, after each of the four row pairs, end) at
// buf[(workgroup * 16 + wave) * 8 ..] (u64).
extern "C" int vmx_debug_iq_packed_stamps(void *buf) {
    g_pk_stamps = buf;
    return VMX_OK;
}

extern "C" size_t vmx_iq_packed_bytes(int ggml_type, int n_out, int n_in) {
    if (!pk_supported(ggml_type) || n_out <= 0 || n_in <= 0 || n_in % 256) return 0;
    return (size_t)((n_out + PK_ROWS - 1) / PK_ROWS) * PK_ROWS * (size_t)(n_in / 256) * pk_type_size(ggml_type);
}

extern "C" int vmx_iq_pack(int ggml_type, const void *raw, void *packed, int n_out, int n_in, void *stream) {
    VMX_REQUIRE(raw && packed, "iq_pack: null pointer");
    VMX_REQUIRE(pk_supported(ggml_type), "iq_pack: ggml type %d has no packed form (IQ3_XXS 18, IQ3_S 21, IQ2_S 22)", ggml_type);
    VMX_REQUIRE(n_out > 0 && n_in > 0 && n_in % 256 == 0, "iq_pack: n_in %d not a multiple of 256", n_in);
    VMX_REQUIRE(((uintptr_t)packed & 15) == 0, "iq_pack: packed buffer must be 16-byte aligned");
    hipStream_t st = vmx_stream(stream);
    const int nsb = n_in / 256;
    VMX_CHECK_HIP(hipMemsetAsync(packed, 0, vmx_iq_packed_bytes(ggml_type, n_out, n_in), st));   // rows of the last group past n_out: d = 0
    const int64_t total = (int64_t)n_out * nsb;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (ggml_type == 18) hipLaunchKernelGGL(iq_pack_kernel<18>, grid, dim3(256), 0, st, (const uint8_t *)raw, (uint8_t *)packed, n_out, nsb);
    else if (ggml_type == 21) hipLaunchKernelGGL(iq_pack_kernel<21>, grid, dim3(256), 0, st, (const uint8_t *)raw, (uint8_t *)packed, n_out, nsb);
    else hipLaunchKernelGGL(iq_pack_kernel<22>, grid, dim3(256), 0, st, (const uint8_t *)raw, (uint8_t *)packed, n_out, nsb);
    VMX_CHECK_LAUNCH("iq_pack");
    return VMX_OK;
}

// Whether the packed kernel is the faster one for this group of tensors (measured on MI355X: it wins where one launch holds
// at least ~2000 8-row units of one K-slice; smaller launches are bound by the launch prologue either way, and rows cut
// into several K-slices pay an LDS combine).  VMX_IQ_PACKED_MIN_UNITS overrides the threshold (0 = always, -1 = never).
bool vmx_iq_packed_preferred(const VmxMmvqSeg *segs, int nseg, int n_in, int m) {
    static const int min_units = vmx_tune("VMX_IQ_PACKED_MIN_UNITS", 2048);
    if (m != 1 || min_units < 0 || n_in % 256) return false;
    const int S = pk_num_slices(n_in / 256);
    int units = 0;
    for (int i = 0; i < nseg; i++) {
        if (!pk_supported(segs[i].ggml_type) || !segs[i].w) return false;
        units += (segs[i].n_out + PK_ROWS - 1) / PK_ROWS * S;
    }
    if (min_units == 0) return true;
    return S == 1 && units >= min_units;
}

// Up to three packed tensors that share one q8_1 activation row, one launch; outputs at y + y_off[i]
int vmx_iq_mmvq_packed_multi(const VmxMmvqSeg *segs, int nseg, const int8_t *aq, const float *ad, float *y, int n_in, hipStream_t st) {
    if (nseg < 1 || nseg > 3) {
        vmx_set_error("iq_mmvq(packed): %d segments (1..3)", nseg);
        return VMX_EINVAL;
    }
    for (int i = 0; i < nseg; i++)
        if (!pk_supported(segs[i].ggml_type)) {
            vmx_set_error("iq_mmvq(packed): ggml type %d has no packed form", segs[i].ggml_type);
            return VMX_EUNSUPPORTED;
        }
    return pk_launch(segs, nseg, aq, ad, y, n_in, nullptr, st);
}

// Whether the gate|up pair can go through the fused GeGLU launch: one activation row, both tensors packed in the same type,
// whole rows in one K-slice, 64 columns per workgroup, and a launch large enough for the packed kernel to be the faster one.
bool vmx_iq_packed_geglu_ok(const VmxMmvqSeg *gate, const VmxMmvqSeg *up, int n_in, int m) {
    static const int off = vmx_tune("VMX_IQ_NO_GEGLU_FUSION", 0);
    if (off || gate->ggml_type != up->ggml_type || gate->n_out != up->n_out || gate->n_out % 64) return false;
    const VmxMmvqSeg both[2] = {*gate, *up};
    return vmx_iq_packed_preferred(both, 2, n_in, m);
}

// act[i] = bf16(bf16(gelu_erf(bf16(gate_i))) * bf16(up_i)) with gate = W_gate x, up = W_up x (the q8_1 MMVQ of both, then the
// GeGLU of Gemma's MLP, gemma4_quantized.rs:1390-1400) and q8_1(act) for the down projection, in one launch.
int vmx_iq_mmvq_packed_geglu(const VmxMmvqSeg *gate, const VmxMmvqSeg *up, const int8_t *aq, const float *ad, uint16_t *act, int8_t *aq_out,
                             float *ad_out, int n_in, hipStream_t st) {
    if (!vmx_iq_packed_geglu_ok(gate, up, n_in, 1)) {
        vmx_set_error("iq_mmvq(packed, GeGLU): gate/up pair not eligible (types %d/%d, n_out %d/%d, n_in %d)", gate->ggml_type, up->ggml_type,
                      gate->n_out, up->n_out, n_in);
        return VMX_EUNSUPPORTED;
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g_pk_tables_ready[dev]) {
        vmx_set_error("iq_mmvq(packed): vmx_init(%d) has not run on this device (it builds the decode tables)", dev);
        return VMX_EINVAL;
    }
    PkArgs a = {};
    a.nseg = 2;
    a.seg[0] = PkSeg{(const uint8_t *)gate->w, nullptr, gate->ggml_type, gate->n_out, 0};
    a.seg[1] = PkSeg{(const uint8_t *)up->w, nullptr, up->ggml_type, up->n_out, 0};
    a.glu_act = act;
    a.glu_aq = aq_out;
    a.glu_ad = ad_out;
    return pk_launch_ppw<4, false, 0, true>(a, gate->n_out / 64, 16, pk_lds_bytes(gate->ggml_type), aq, ad, n_in, 1, nullptr, st);
}

// The same launch with Gemma's sandwich norms in its prologue: the activation row is bf16(norm(h + bf16(norm(o)))) computed by
// every workgroup from the o_proj output `o` (f32 [n_in]) and the residual stream; h_out (!= h_in) receives the updated stream.
int vmx_iq_mmvq_packed_norm_geglu(const VmxMmvqSeg *gate, const VmxMmvqSeg *up, const float *o, const uint16_t *h_in, uint16_t *h_out,
                                  const uint16_t *w_post, const uint16_t *w_next, float eps, uint16_t *act, int8_t *aq_out, float *ad_out,
                                  int n_in, hipStream_t st) {
    static const int off = vmx_tune("VMX_IQ_NO_NORM_FUSION", 0);
    if (off || !vmx_iq_packed_geglu_ok(gate, up, n_in, 1) || n_in % 32 || n_in > 4096 || h_in == h_out) {
        vmx_set_error("iq_mmvq(packed, norm + GeGLU): not eligible (n_in %d)", n_in);
        return VMX_EUNSUPPORTED;
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g_pk_tables_ready[dev]) {
        vmx_set_error("iq_mmvq(packed): vmx_init(%d) has not run on this device (it builds the decode tables)", dev);
        return VMX_EINVAL;
    }
    PkArgs a = {};
    a.nseg = 2;
    a.seg[0] = PkSeg{(const uint8_t *)gate->w, nullptr, gate->ggml_type, gate->n_out, 0};
    a.seg[1] = PkSeg{(const uint8_t *)up->w, nullptr, up->ggml_type, up->n_out, 0};
    a.glu_act = act;
    a.glu_aq = aq_out;
    a.glu_ad = ad_out;
    a.n_o = o;
    a.n_h_in = h_in;
    a.n_h_out = h_out;
    a.n_wpost = w_post;
    a.n_wnext = w_next;
    a.n_eps = eps;
    return pk_launch_ppw<4, false, 0, true, true>(a, gate->n_out / 64, 16, pk_lds_bytes(gate->ggml_type), nullptr, nullptr, n_in, 1, nullptr, st);
}

bool vmx_iq_packed_norm_geglu_ok(const VmxMmvqSeg *gate, const VmxMmvqSeg *up, int n_in, int m) {
    static const int off = vmx_tune("VMX_IQ_NO_NORM_FUSION", 0);
    return !off && n_in % 32 == 0 && n_in <= 4096 && vmx_iq_packed_geglu_ok(gate, up, n_in, m);
}

// Up to three packed tensors behind Gemma's layer-boundary norms (gemma4_quantized.rs:1400-1424 then :1311-1320): the row is
//   h' = bf16(bf16(norm(bf16(o)) * w_post) + h) [* scalar];  x = bf16(norm(h') * w_next);  q8_1(x)
// computed by every workgroup from the previous layer's down-projection output o and the residual stream; h_out (!= h_in)
// receives h'.  One activation row, whole rows in one K-slice, n_in <= 4096, workgroups of at least n_in / 8 threads.
bool vmx_iq_packed_norm_multi_ok(const VmxMmvqSeg *segs, int nseg, int n_in, int m) {
    static const int off = vmx_tune("VMX_IQ_NO_QKV_NORM_FUSION", 0);
    if (off || m != 1 || nseg < 1 || nseg > 3 || n_in % 256 || n_in > 4096) return false;
    if (pk_num_slices(n_in / 256) != 1) return false;
    int units = 0;
    for (int i = 0; i < nseg; i++) {
        if (!pk_supported(segs[i].ggml_type) || !segs[i].w) return false;
        units += (segs[i].n_out + PK_ROWS - 1) / PK_ROWS;
    }
    // the launch plan of pk_launch: waves per workgroup from the item count
    const int ppw = units >= 2048 ? 4 : (units >= 1024 ? 2 : 1);
    int W = (units * (4 / ppw) + vmx_num_cus() - 1) / vmx_num_cus();
    W = W < 1 ? 1 : (W > 16 ? 16 : W);
    // measured on the Gemma-4-12B shapes: with 8-wave workgroups (sliding layers, 1 024 units) the fused launch takes 9.9 us against
    // 6.9 + 5.5 us for the raw-layout launch behind a norm kernel; the 9-wave plan of the full-attention layers (1 152 units, one
    // workgroup per compute unit) takes 18.1 us against 8.1 + 5.5 us in the kernel trace, so those keep the two launches (on the
    // whole step the two choices measured within noise of each other: 327-330 vs 328 tok/s)
    static const int max_w = vmx_tune("VMX_IQ_QKV_NORM_MAX_W", 8);
    return 64 * W * 8 >= n_in && W <= max_w;
}
int vmx_iq_mmvq_packed_norm_multi(const VmxMmvqSeg *segs, int nseg, const float *o, const uint16_t *h_in, uint16_t *h_out,
                                  const uint16_t *w_post, const uint16_t *scalar, const uint16_t *w_next, float eps, float *y, int n_in,
                                  hipStream_t st) {
    if (!vmx_iq_packed_norm_multi_ok(segs, nseg, n_in, 1) || h_in == h_out) {
        vmx_set_error("iq_mmvq(packed, norm): not eligible (n_in %d, %d tensors)", n_in, nseg);
        return VMX_EUNSUPPORTED;
    }
    PkArgs na = {};
    na.n_o = o; na.n_h_in = h_in; na.n_h_out = h_out; na.n_wpost = w_post; na.n_wnext = w_next; na.n_eps = eps; na.n_scalar = scalar;
    return pk_launch(segs, nseg, nullptr, nullptr, y, n_in, nullptr, st, &na);
}

extern "C" int vmx_iq_mmvq_packed_geglu_bf16(int ggml_type, const void *gate_packed, const void *up_packed, const int8_t *aq, const float *ad,
                                             uint16_t *act, int8_t *act_q, float *act_d, int inter, int n_in, void *stream) {
    VMX_REQUIRE(gate_packed && up_packed && aq && ad && act && act_q && act_d, "iq_mmvq(packed, GeGLU): null pointer");
    VMX_REQUIRE(pk_supported(ggml_type), "iq_mmvq(packed): ggml type %d has no packed form (IQ3_XXS 18, IQ3_S 21, IQ2_S 22)", ggml_type);
    VMX_REQUIRE(n_in > 0 && n_in % 256 == 0, "iq_matmul: n_in %d not a multiple of 256", n_in);
    VMX_REQUIRE(n_in <= 4096, "iq_mmvq(packed, GeGLU): n_in %d beyond one K-slice (4096)", n_in);
    VMX_REQUIRE(inter > 0 && inter % 64 == 0, "iq_mmvq(packed, GeGLU): intermediate size %d not a multiple of 64", inter);
    VMX_REQUIRE(act_q != aq && act_d != ad, "iq_mmvq(packed, GeGLU): the q8_1 output must not alias the q8_1 input");
    VMX_REQUIRE(((uintptr_t)gate_packed & 15) == 0 && ((uintptr_t)up_packed & 15) == 0 && ((uintptr_t)aq & 15) == 0 && ((uintptr_t)ad & 7) == 0,
                "iq_mmvq(packed): misaligned pointer");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g_pk_tables_ready[dev]) {
        vmx_set_error("iq_mmvq(packed): vmx_init(%d) has not run on this device (it builds the decode tables)", dev);
        return VMX_EINVAL;
    }
    PkArgs a = {};
    a.nseg = 2;
    a.seg[0] = PkSeg{(const uint8_t *)gate_packed, nullptr, ggml_type, inter, 0};
    a.seg[1] = PkSeg{(const uint8_t *)up_packed, nullptr, ggml_type, inter, 0};
    a.glu_act = act;
    a.glu_aq = act_q;
    a.glu_ad = act_d;
    return pk_launch_ppw<4, false, 0, true>(a, inter / 64, 16, pk_lds_bytes(ggml_type), aq, ad, n_in, 1, nullptr, vmx_stream(stream));
}

extern "C" int vmx_iq_mmvq_packed_f32(int ggml_type, const void *packed, const int8_t *aq, const float *ad, float *y, int n_out, int n_in,
                                      void *stream) {
    VMX_REQUIRE(packed && aq && ad && y, "iq_mmvq(packed): null pointer");
    VMX_REQUIRE(pk_supported(ggml_type), "iq_mmvq(packed): ggml type %d has no packed form (IQ3_XXS 18, IQ3_S 21, IQ2_S 22)", ggml_type);
    VMX_REQUIRE(n_in > 0 && n_in % 256 == 0, "iq_matmul: n_in %d not a multiple of 256", n_in);
    VMX_REQUIRE(n_out > 0, "iq_matmul: n_out %d", n_out);
    VMX_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)aq & 15) == 0 && ((uintptr_t)ad & 7) == 0, "iq_mmvq(packed): misaligned pointer");
    const VmxMmvqSeg seg = {packed, ggml_type, n_out, 0};
    return pk_launch(&seg, 1, aq, ad, y, n_in, nullptr, vmx_stream(stream));
}

extern "C" int vmx_iq_subdots_packed_i32(int ggml_type, const void *packed, const int8_t *aq, int32_t *out, int n_out, int n_in, void *stream) {
    VMX_REQUIRE(packed && aq && out, "iq_subdots(packed): null pointer");
    VMX_REQUIRE(pk_supported(ggml_type), "iq_subdots(packed): ggml type %d has no packed form", ggml_type);
    VMX_REQUIRE(n_in > 0 && n_in % 256 == 0 && n_out > 0, "iq_subdots(packed): bad shape");
    const VmxMmvqSeg seg = {packed, ggml_type, n_out, 0};
    // the activation scales are not used for the integer sub-dots: any readable f32 array of n_in / 32 entries will do
    return pk_launch(&seg, 1, aq, reinterpret_cast<const float *>(aq), nullptr, n_in, out, vmx_stream(stream));
}

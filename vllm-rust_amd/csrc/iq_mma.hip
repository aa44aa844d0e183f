// q8_1 MMVQ over GGUF I-quant weights for SEVERAL activation rows (2 .. 16: batched decode) on the int8 matrix cores.
//
// Semantics are those of mmvq_iq*_f32 / vec_dot_iq*_q8_1 (crates/core/kernels/iq_dequant.cu:274-423, M <= 16 accumulators per
// output row; CPU form crates/core/src/quantization/gguf/iq/mod.rs:556-725): y[m, o] = sum_g w_d(o, g/8) * a_d(m, g) *
// int_dot(block, g % 8, u_m), the integer sub-dots bit for bit (vmx_iq_subdots_mma_i32 exposes them); only the f32 summation
// order differs.  The one-row kernels decode a sub-block once per activation row on the vector ALU (16 rows cost 5-10 x one
// row); here a sub-block is decoded ONCE into signed int8 bytes and v_mfma_i32_16x16x32_i8 forms its integer dot with all 16
// rows: operand A = 16 activation rows x 32 k (q8_1 bytes straight from memory), operand B = 32 k x 16 weight rows (the decoded
// sub-block of each row), C[m][o] = the int32 sub-dot, which then takes the type's integer scale step and the two f32 scales.
//
// Load-time layout ("16-row tiles", vmx_iq_mma_pack): the same bytes as the GGUF blocks, permuted so that the wave that owns a
// tile of 16 rows reads, per super-block, one 16-byte entry per lane: lane (o = lane & 15, kg = lane >> 4) supplies the 8
// weights 8 kg .. 8 kg + 7 of every sub-block of row o (the MFMA's B operand takes 8 k-values per lane), i.e. it needs index
// bytes 2 kg, 2 kg + 1 of each of the 8 sub-blocks: 16 bytes per super-block.
//     IQ3_XXS  QS [lane] 16 B | AUX [row] 8 x u32 scales_and_signs | D [row] f16                       1568 B = 16 x 98
//     IQ3_S    QS [lane] 16 B | SG [lane] 8 x u8 signs | QH [row] 8 x u8 | SC [row] 4 x u8 | D [row]    1760 B = 16 x 110
//     IQ2_S    QSG [lane] 8 x u8 qs + 8 x u8 signs | QH [row] 8 x u8 | SC [row] 8 x u8 | D [row]        1312 B = 16 x 82
// A tile's super-blocks follow each other, tiles follow each other: [tile][super-block][bytes].
#include "iq_common.h"

namespace {

constexpr int MP_ROWS = 16;
typedef int mp_i32x4 __attribute__((ext_vector_type(4)));

template <int T> struct Mp;
template <> struct Mp<18> { static constexpr int TS = 98, AUX = 1024, D = 1536, BYTES = 1568, GRID_N = 256, GRID_ENTRY = 4, COPIES = 8; };
template <> struct Mp<21> { static constexpr int TS = 110, SG = 1024, QH = 1536, SC = 1664, D = 1728, BYTES = 1760, GRID_N = 512, GRID_ENTRY = 4, COPIES = 8; };
template <> struct Mp<22> { static constexpr int TS = 82, QH = 1024, SC = 1152, D = 1280, BYTES = 1312, GRID_N = 1024, GRID_ENTRY = 8, COPIES = 2; };

__host__ __device__ inline bool mp_supported(int t) { return t == 18 || t == 21 || t == 22; }
__host__ __device__ inline int mp_tile_bytes(int t) { return t == 18 ? 1568 : t == 21 ? 1760 : 1312; }

// ------------------------------------------------------------------ repack (load time): one thread per (padded row, super-block)
template <int T>
__global__ __launch_bounds__(256) void iq_mma_pack_kernel(const uint8_t *__restrict__ raw, uint8_t *__restrict__ packed, int n_out, int nsb) {
    constexpr int TS = Mp<T>::TS;
    const int rows_pad = (n_out + MP_ROWS - 1) / MP_ROWS * MP_ROWS;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows_pad * nsb) return;
    const int row = (int)(idx / nsb), b = (int)(idx % nsb), o = row & 15;
    uint8_t *U = packed + ((int64_t)(row >> 4) * nsb + b) * Mp<T>::BYTES;
    uint8_t blk[TS];
    for (int i = 0; i < TS; i++) blk[i] = row < n_out ? raw[((int64_t)row * nsb + b) * TS + i] : (uint8_t)0;   // padding rows: d = 0
    U[Mp<T>::D + 2 * o] = blk[0];
    U[Mp<T>::D + 2 * o + 1] = blk[1];
    for (int kg = 0; kg < 4; kg++) {
        uint8_t *e = U + 16 * (o + 16 * kg);
        for (int s = 0; s < 8; s++) {
            if constexpr (T == 18 || T == 21) {
                e[2 * s] = blk[2 + 8 * s + 2 * kg];
                e[2 * s + 1] = blk[2 + 8 * s + 2 * kg + 1];
            } else {
                e[s] = blk[2 + 4 * s + kg];        // qs
                e[8 + s] = blk[34 + 4 * s + kg];   // signs
            }
            if constexpr (T == 21) U[Mp<T>::SG + 8 * (o + 16 * kg) + s] = blk[74 + 4 * s + kg];
        }
    }
    if constexpr (T == 18) {
        for (int i = 0; i < 32; i++) U[Mp<T>::AUX + 32 * o + i] = blk[66 + i];
    } else if constexpr (T == 21) {
        for (int i = 0; i < 8; i++) U[Mp<T>::QH + 8 * o + i] = blk[66 + i];
        for (int i = 0; i < 4; i++) U[Mp<T>::SC + 4 * o + i] = blk[106 + i];
    } else {
        for (int i = 0; i < 8; i++) U[Mp<T>::QH + 8 * o + i] = blk[66 + i];
        for (int i = 0; i < 8; i++) U[Mp<T>::SC + 8 * o + i] = blk[74 + i];
    }
}

// ------------------------------------------------------------------ per-super-block registers of one 16-row tile
template <int T> struct MpRaw;
template <> struct MpRaw<18> { uint4 qs, aux0, aux1; uint32_t d; };
template <> struct MpRaw<21> { uint4 qs; uint2 sg, qh; uint32_t sc, d; };
template <> struct MpRaw<22> { uint4 qsg; uint2 qh, sc; uint32_t d; };

template <int T> __device__ __forceinline__ void mp_load(MpRaw<T> &r, const uint8_t *__restrict__ U, int lane) {
    const int o = lane & 15;
    if constexpr (T == 18) {
        r.qs = *reinterpret_cast<const uint4 *>(U + 16 * lane);
        r.aux0 = *reinterpret_cast<const uint4 *>(U + Mp<T>::AUX + 32 * o);
        r.aux1 = *reinterpret_cast<const uint4 *>(U + Mp<T>::AUX + 32 * o + 16);
        r.d = *reinterpret_cast<const uint16_t *>(U + Mp<T>::D + 2 * o);
    } else if constexpr (T == 21) {
        r.qs = *reinterpret_cast<const uint4 *>(U + 16 * lane);
        r.sg = *reinterpret_cast<const uint2 *>(U + Mp<T>::SG + 8 * lane);
        r.qh = *reinterpret_cast<const uint2 *>(U + Mp<T>::QH + 8 * o);
        r.sc = *reinterpret_cast<const uint32_t *>(U + Mp<T>::SC + 4 * o);
        r.d = *reinterpret_cast<const uint16_t *>(U + Mp<T>::D + 2 * o);
    } else {
        r.qsg = *reinterpret_cast<const uint4 *>(U + 16 * lane);
        r.qh = *reinterpret_cast<const uint2 *>(U + Mp<T>::QH + 8 * o);
        r.sc = *reinterpret_cast<const uint2 *>(U + Mp<T>::SC + 8 * o);
        r.d = *reinterpret_cast<const uint16_t *>(U + Mp<T>::D + 2 * o);
    }
}

__device__ __forceinline__ uint32_t mp_w4(const uint4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ uint32_t mp_byte(const uint2 &v, int i) { return ((i < 4 ? v.x : v.y) >> (8 * (i & 3))) & 0xFFu; }

// (g ^ m) + t: the bytes of g selected by m (0xFF) negated, t = m & 0x01010101 (two's complement; grid bytes are 1..127, so a
// byte never carries into its neighbour)
__device__ __forceinline__ uint32_t mp_xad(uint32_t g, uint32_t m, uint32_t t) { return (g ^ m) + t; }

// LDS tables: grid entries replicated COPIES times (copy = lane & (COPIES - 1): fewer bank conflicts between the lanes of a
// wave), then the sign entries: 8 bytes per 8-bit sign pattern = the byte masks (0xFF = negative) of weights 0..3 and 4..7.
template <int T> struct MpLds {
    static constexpr int GRID_BYTES = Mp<T>::GRID_N * Mp<T>::GRID_ENTRY * Mp<T>::COPIES;
    static constexpr int SIGN_OFF = GRID_BYTES, SIGN_BYTES = 256 * 8;
    static constexpr int AD_OFF = SIGN_OFF + SIGN_BYTES;              // per wave: [8 sub-blocks][16 rows] f32 = 512 B
    static constexpr int TOTAL_MIN = AD_OFF + 16 * 512;
    static constexpr int RED_BYTES = 16 * 2 * 64 * 16;                // cross-slice sums: [wave][tile][lane] x 4 f32 (reuses the table space)
    static constexpr int TOTAL = TOTAL_MIN > RED_BYTES ? TOTAL_MIN : RED_BYTES;
};

template <int T> __device__ __forceinline__ void mp_build_tables(unsigned char *smem, int tid) {
    constexpr int C = Mp<T>::COPIES;
    // every entry of a thread is requested before the first one is stored: one memory round trip, not one per pass
    constexpr int PASSES = Mp<T>::GRID_N * C / 1024;
    static_assert(Mp<T>::GRID_N * C % 1024 == 0, "whole passes of the 1024 threads");
    if constexpr (Mp<T>::GRID_ENTRY == 4) {
        const uint32_t *src = T == 18 ? iq3xxs_grid : iq3s_grid;
        uint32_t *g = reinterpret_cast<uint32_t *>(smem);
        uint32_t v[PASSES];
#pragma unroll
        for (int k = 0; k < PASSES; k++) v[k] = src[(tid + 1024 * k) / C];
#pragma unroll
        for (int k = 0; k < PASSES; k++) g[tid + 1024 * k] = v[k];
    } else {
        const uint2 *src = reinterpret_cast<const uint2 *>(iq2s_grid);
        uint2 *g = reinterpret_cast<uint2 *>(smem);
        uint2 v[PASSES];
#pragma unroll
        for (int k = 0; k < PASSES; k++) v[k] = src[(tid + 1024 * k) / C];
#pragma unroll
        for (int k = 0; k < PASSES; k++) g[tid + 1024 * k] = v[k];
    }
    if (tid < 256) {
        const uint32_t s8 = (uint32_t)tid;
        const uint32_t l0 = __umul24(s8 & 0xFu, 0x00204081u) & 0x01010101u, l1 = __umul24(s8 >> 4, 0x00204081u) & 0x01010101u;
        reinterpret_cast<uint2 *>(smem + MpLds<T>::SIGN_OFF)[tid] = make_uint2((l0 << 8) - l0, (l1 << 8) - l1);
    }
}

// The decoded B operand (8 signed int8 weights of row o, k-group kg, sub-block s) and the integer scale fields of the sub-block.
template <int T>
__device__ __forceinline__ void mp_decode(const MpRaw<T> &r, int s, int kg, int copy, const unsigned char *smem, uint32_t &w0, uint32_t &w1,
                                          int &ls0, int &ls1) {
    const uint2 *sign_tab = reinterpret_cast<const uint2 *>(smem + MpLds<T>::SIGN_OFF);
    ls1 = 0;
    if constexpr (T == 18) {
        const uint32_t pair = (mp_w4(r.qs, s >> 1) >> (16 * (s & 1))) & 0xFFFFu;
        const uint32_t aux = s < 4 ? mp_w4(r.aux0, s) : mp_w4(r.aux1, s - 4);
        uint32_t s7 = (aux >> (7 * kg)) & 127u;
        s7 |= (__popc(s7) & 1u) << 7;
        const uint32_t *grid = reinterpret_cast<const uint32_t *>(smem);
        const uint32_t g0 = grid[(pair & 0xFFu) * 8 + copy], g1 = grid[(pair >> 8) * 8 + copy];
        const uint2 m = sign_tab[s7];
        w0 = mp_xad(g0, m.x, m.x & 0x01010101u);
        w1 = mp_xad(g1, m.y, m.y & 0x01010101u);
        ls0 = (int)(aux >> 28);
    } else if constexpr (T == 21) {
        const uint32_t pair = (mp_w4(r.qs, s >> 1) >> (16 * (s & 1))) & 0xFFFFu;
        const uint32_t qh = mp_byte(r.qh, s);
        const uint32_t i0 = (pair & 0xFFu) | ((qh << (8 - 2 * kg)) & 256u), i1 = (pair >> 8) | ((qh << (7 - 2 * kg)) & 256u);
        const uint32_t *grid = reinterpret_cast<const uint32_t *>(smem);
        const uint32_t g0 = grid[i0 * 8 + copy], g1 = grid[i1 * 8 + copy];
        const uint2 m = sign_tab[mp_byte(r.sg, s)];
        w0 = mp_xad(g0, m.x, m.x & 0x01010101u);
        w1 = mp_xad(g1, m.y, m.y & 0x01010101u);
        ls0 = (int)((r.sc >> (4 * s)) & 0xFu);
    } else {
        const uint2 qv = make_uint2(r.qsg.x, r.qsg.y), sv = make_uint2(r.qsg.z, r.qsg.w);
        const uint32_t qh = mp_byte(r.qh, s);
        const uint32_t idx = mp_byte(qv, s) | ((qh << (8 - 2 * kg)) & 0x300u);
        const uint2 g = reinterpret_cast<const uint2 *>(smem)[idx * 2 + (copy & 1)];
        const uint2 m = sign_tab[mp_byte(sv, s)];
        w0 = mp_xad(g.x, m.x, m.x & 0x01010101u);
        w1 = mp_xad(g.y, m.y, m.y & 0x01010101u);
        const uint32_t sc = mp_byte(r.sc, s);
        ls0 = (int)(sc & 0xFu);
        ls1 = (int)(sc >> 4);
    }
}

struct MpSeg {
    const uint8_t *w;   // 16-row-tile form
    float *y;           // output of this segment (row stride ldy)
    int type, n_out, wg0;
};
struct MpArgs {
    MpSeg seg[3];
    int nseg;
};

// Workgroup = 16 waves = RT row-tile sets (TPW 16-row tiles each) x KS slices of K; wave (rt, ks).  The KS partial sums of a
// set are added through LDS in slice order.  TPW = 2 shares a wave's activation fragments between two tiles; TPW = 1 is for
// tensors with so few rows that pairs would leave compute units without a workgroup.
// GLU (TPW = 2): tile 0 of a wave is a tile of the gate tensor, tile 1 the same tile of the up tensor (glu.w_up, same type and
// shape); the epilogue forms act = bf16(bf16(gelu_erf(bf16(gate))) * bf16(up)) and its q8_1 blocks (32 rows = the tiles of two
// neighbouring tile sets of the workgroup, so KS <= 8) instead of writing the two f32 projections.
struct MpGlu {
    const uint8_t *w_up;
    uint16_t *act;     // [M][n_out] bf16
    int8_t *aq_out;    // [M][n_out]
    float *ad_out;     // [M][n_out / 32]
};

template <int T, int KS, bool DBG, int TPW, bool GLU = false>
__device__ __forceinline__ void mp_body(unsigned char *smem, const MpSeg &sg, int wg, const int8_t *__restrict__ aq, const float *__restrict__ ad,
                                        int n_in, int M, int ldy, int32_t *__restrict__ dbg, const MpGlu *glu = nullptr) {
    static_assert(!GLU || (TPW == 2 && KS <= 8 && !DBG), "the GeGLU epilogue pairs a gate and an up tile per wave and two tile sets per q8_1 block");
    constexpr int RT = 16 / KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int o = lane & 15, kg = lane >> 4, copy = lane & (Mp<T>::COPIES - 1);
    const int ks = wave % KS, rt = wave / KS;
    const int nsb = n_in >> 8, G = n_in >> 5;
    const int tiles = (sg.n_out + MP_ROWS - 1) / MP_ROWS;
    const int tile0 = GLU ? wg * RT + rt : (wg * RT + rt) * TPW;           // this wave's 16-row tiles
    // K slice of this wave: super-blocks [sb0, sb0 + nb)
    const int base = nsb / KS, rem = nsb % KS;
    const int nb = base + (ks < rem ? 1 : 0), sb0 = ks * base + (ks < rem ? ks : rem);
    mp_build_tables<T>(smem, tid);
    float *ad_s = reinterpret_cast<float *>(smem + MpLds<T>::AD_OFF + wave * 512);
    __syncthreads();

    const bool tile_ok[2] = {tile0 < tiles, GLU ? tile0 < tiles : (TPW > 1 && tile0 + 1 < tiles)};
    const int8_t *arow = aq + (int64_t)min(o, M - 1) * n_in + 8 * kg;      // A operand: lane (m = o, kg)
    const float *adrow = ad + (int64_t)min(o, M - 1) * G + kg;             // staging of a_d: lane (m = o, g = kg, kg + 4)
    mp_i32x4 zero = {0, 0, 0, 0};
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int b = 0; b < nb; b++) {
        const int sb = sb0 + b;
        MpRaw<T> raw[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++)
            mp_load<T>(raw[t], (GLU && t == 1 ? glu->w_up : sg.w) + ((int64_t)min(GLU ? tile0 : tile0 + t, tiles - 1) * nsb + sb) * Mp<T>::BYTES, lane);
        long au[8];
#pragma unroll
        for (int s = 0; s < 8; s++) au[s] = *reinterpret_cast<const long *>(arow + 256 * sb + 32 * s);
        const float ad0 = adrow[8 * sb], ad1 = adrow[8 * sb + 4];
        __builtin_amdgcn_wave_barrier();
        ad_s[kg * 16 + o] = ad0;            // [g][m]
        ad_s[(kg + 4) * 16 + o] = ad1;
        __builtin_amdgcn_wave_barrier();
        float dw[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++) dw[t] = vmx_f16_to_f32((uint16_t)raw[t].d);
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            // software pipeline of depth one: the matrix-core product of sub-block s is issued, then sub-block s - 1 (whose product
            // has landed meanwhile) takes its integer scale step and the two f32 scales; sums stay in sub-block order
            mp_i32x4 cc[2][2];
            int lsv[2][2];
            float4 adq[2];
            float blk[4] = {0.f, 0.f, 0.f, 0.f};   // sum over the super-block's sub-blocks of a_d * sub-dot; the weight's d multiplies it once
            auto finish = [&](int s) {
                const int q = s & 1;
                const float adv[4] = {adq[q].x, adq[q].y, adq[q].z, adq[q].w};   // a_d of this lane's four rows m = 4 kg + j
                const int ls0 = lsv[q][0], ls1 = lsv[q][1];
                float sub[4];
                if constexpr (T == 22) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {   // (s0 ls0 + s1 ls1 + (s0 + s1) / 2) / 4, C truncating divisions as shifts
                        const int s0 = cc[q][0][j], s1 = cc[q][1][j], h = s0 + s1;
                        const int half = (h + (int)((uint32_t)h >> 31)) >> 1;
                        const int v = __mul24(s0, ls0) + __mul24(s1, ls1) + half;
                        sub[j] = (float)((v + (int)(((uint32_t)(v >> 31)) >> 30)) >> 2);
                    }
                } else if constexpr (T == 18) {
                    // (ls * sumi + sumi / 2) / 2 with C truncating divisions == trunc((2 ls + 1) * sumi / 4): 2 ls + 1 is odd, so
                    // the product has the parity of sumi and dropping that bit towards zero first cannot change the quotient;
                    // |product| < 2^23: exact in f32
                    const float f = 0.25f * (float)(2 * ls0 + 1);
#pragma unroll
                    for (int j = 0; j < 4; j++) sub[j] = truncf((float)cc[q][0][j] * f);
                } else {
                    const float f = (float)(2 * ls0 + 1);
#pragma unroll
                    for (int j = 0; j < 4; j++) sub[j] = (float)cc[q][0][j] * f;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) blk[j] = fmaf(adv[j], sub[j], blk[j]);
                if constexpr (DBG) {
                    const int row = (tile0 + t) * MP_ROWS + o;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int mi = 4 * kg + j;
                        if (tile_ok[t] && row < sg.n_out && mi < M) dbg[((int64_t)row * M + mi) * G + sb * 8 + s] = (int)sub[j];
                    }
                }
            };
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const int q = s & 1;
                adq[q] = *reinterpret_cast<const float4 *>(ad_s + s * 16 + 4 * kg);
                uint32_t w0, w1;
                mp_decode<T>(raw[t], s, kg, copy, smem, w0, w1, lsv[q][0], lsv[q][1]);
                const long bw = (long)(((unsigned long long)w1 << 32) | w0);
                if constexpr (T == 22) {
                    // the halves of a sub-block (weights 0..15 / 16..31 = k-groups 0, 1 / 2, 3) carry their own scale nibbles
                    const long b_lo = kg < 2 ? bw : 0l, b_hi = kg < 2 ? 0l : bw;
                    cc[q][0] = __builtin_amdgcn_mfma_i32_16x16x32_i8(au[s], b_lo, zero, 0, 0, 0);
                    cc[q][1] = __builtin_amdgcn_mfma_i32_16x16x32_i8(au[s], b_hi, zero, 0, 0, 0);
                } else {
                    cc[q][0] = __builtin_amdgcn_mfma_i32_16x16x32_i8(au[s], bw, zero, 0, 0, 0);
                }
                if (s > 0) finish(s - 1);
                // one sub-block at a time: without the fence the scheduler interleaves all the decodes of a super-block and the
                // kernel needs twice the registers a 16-wave workgroup may have
                __builtin_amdgcn_sched_barrier(0);
            }
            finish(7);
#pragma unroll
            for (int j = 0; j < 4; j++) acc[t][j] = fmaf(dw[t], blk[j], acc[t][j]);
        }
    }
    if constexpr (DBG) return;
    // ---- the KS slices of a row-tile pair, added in slice order
    __syncthreads();   // every wave is done with the tables: their space now holds the partial sums
    float4 *red = reinterpret_cast<float4 *>(smem);
#pragma unroll
    for (int t = 0; t < TPW; t++) red[(wave * 2 + t) * 64 + lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    __syncthreads();
    if constexpr (GLU) {
        float *pm = reinterpret_cast<float *>(smem + MpLds<T>::TOTAL);   // [RT][16]: |act| maxima of (tile set, activation row)
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
        if (ks == 0) {
            float4 g = red[((rt * KS) * 2 + 0) * 64 + lane], u = red[((rt * KS) * 2 + 1) * 64 + lane];
            for (int k = 1; k < KS; k++) {
                const float4 g2 = red[((rt * KS + k) * 2 + 0) * 64 + lane], u2 = red[((rt * KS + k) * 2 + 1) * 64 + lane];
                g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
                u.x += u2.x; u.y += u2.y; u.z += u2.z; u.w += u2.w;
            }
            const float gv[4] = {g.x, g.y, g.z, g.w}, uv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float gg = vmx_bf16_to_f32(vmx_f32_to_bf16(gv[j]));
                const float ge = vmx_bf16_to_f32(vmx_f32_to_bf16(0.5f * gg * (1.0f + erff(gg * 0.70710678118654752f))));
                a4[j] = vmx_bf16_to_f32(vmx_f32_to_bf16(ge * vmx_bf16_to_f32(vmx_f32_to_bf16(uv[j]))));
                const float am = vmx_row16_max(fabsf(a4[j]));
                if (o == 0) pm[rt * 16 + kg * 4 + j] = am;
            }
        }
        __syncthreads();
        if (ks == 0 && tile_ok[0]) {
            const int row = tile0 * MP_ROWS + o;   // n_out % 32 == 0: every row of a valid tile exists, and so does the block's other tile
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int mi = 4 * kg + j;
                if (mi < M) {
                    // q8_1 as in vmx_q8_1_emit8: d = amax / 127, q = roundf(v / d), an all-zero block has d = 0 and q = 0
                    const float amax = fmaxf(pm[rt * 16 + kg * 4 + j], pm[(rt ^ 1) * 16 + kg * 4 + j]);
                    const float d = amax / 127.0f;
                    glu->aq_out[(int64_t)mi * sg.n_out + row] = amax != 0.0f ? (int8_t)(int)roundf(a4[j] / d) : (int8_t)0;
                    glu->act[(int64_t)mi * sg.n_out + row] = vmx_f32_to_bf16(a4[j]);
                    if ((rt & 1) == 0 && o == 0) glu->ad_out[(int64_t)mi * (sg.n_out >> 5) + (row >> 5)] = d;
                }
            }
        }
        return;
    }
    if (ks != 0) return;
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        float4 v = red[((rt * KS) * 2 + t) * 64 + lane];
        for (int k = 1; k < KS; k++) {
            const float4 u = red[((rt * KS + k) * 2 + t) * 64 + lane];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const int row = (tile0 + t) * MP_ROWS + o;
        if (!tile_ok[t] || row >= sg.n_out) continue;
        const float out4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int mi = 4 * kg + j;
            if (mi < M) sg.y[(int64_t)mi * ldy + row] = out4[j];
        }
    }
}

template <int KS, bool DBG, int TPW = 2>
__global__ __launch_bounds__(1024) void iq_mma_kernel(MpArgs a, const int8_t *__restrict__ aq, const float *__restrict__ ad, int n_in, int M,
                                                       int ldy, int32_t *__restrict__ dbg) {
    extern __shared__ __align__(16) unsigned char smem[];
    int si = 0;
    for (int i = 1; i < a.nseg; i++)
        if ((int)blockIdx.x >= a.seg[i].wg0) si = i;
    const MpSeg sg = a.seg[si];
    const int wg = blockIdx.x - sg.wg0;
    if (sg.type == 18) mp_body<18, KS, DBG, TPW>(smem, sg, wg, aq, ad, n_in, M, ldy, dbg);
    else if (sg.type == 21) mp_body<21, KS, DBG, TPW>(smem, sg, wg, aq, ad, n_in, M, ldy, dbg);
    else mp_body<22, KS, DBG, TPW>(smem, sg, wg, aq, ad, n_in, M, ldy, dbg);
}

template <int KS>
__global__ __launch_bounds__(1024) void iq_mma_geglu_kernel(MpSeg gate, MpGlu glu, const int8_t *__restrict__ aq, const float *__restrict__ ad, int n_in,
                                                             int M) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (gate.type == 18) mp_body<18, KS, false, 2, true>(smem, gate, blockIdx.x, aq, ad, n_in, M, 0, nullptr, &glu);
    else if (gate.type == 21) mp_body<21, KS, false, 2, true>(smem, gate, blockIdx.x, aq, ad, n_in, M, 0, nullptr, &glu);
    else mp_body<22, KS, false, 2, true>(smem, gate, blockIdx.x, aq, ad, n_in, M, 0, nullptr, &glu);
}

int mp_lds_bytes(int t) { return t == 18 ? MpLds<18>::TOTAL : t == 21 ? MpLds<21>::TOTAL : MpLds<22>::TOTAL; }

// K slices per workgroup (16 waves): as few as still give every compute unit about two workgroups, at least one super-block per slice
int mp_pick_ks(int total_rows, int nsb) {
    const int cus = vmx_num_cus();
    for (int ks : {4, 8, 16}) {
        const int wgs = (total_rows + 32 * (16 / ks) - 1) / (32 * (16 / ks));
        if (nsb < ks) return ks == 4 ? 4 : ks / 2;
        if (wgs >= 2 * cus || ks == 16) return ks;
    }
    return 16;
}

// K slices per workgroup of the fused gate|up + GeGLU launch: four (four tile sets per workgroup) where that still gives three
// quarters of the compute units a workgroup, else eight.  A compute unit holds one of these workgroups at a time, so a launch that
// fits in one round pays the table build, the first loads and the slice sums once: Gemma-4-12B gate|up at 16 rows, 240 workgroups
// of four slices against 480 of eight, 5.35 -> 5.23 ms per step.
int mp_glu_ks(int n_out) {
    static const int ks_force = vmx_tune("VMX_IQ_MMA_GLU_KS", 0);
    const int tiles4 = (n_out / MP_ROWS + 3) / 4;
    return ks_force == 4 || ks_force == 8 ? ks_force : (4 * tiles4 >= 3 * vmx_num_cus() ? 4 : 8);
}

template <bool DBG>
int mp_launch(const VmxMmvqSeg *segs, int nseg, const int8_t *aq, const float *ad, float *y, int n_in, int m, int ldy, int32_t *dbg,
              hipStream_t st) {
    MpArgs a = {};
    a.nseg = nseg;
    int total_rows = 0, lds = 0;
    for (int i = 0; i < nseg; i++) total_rows += segs[i].n_out;
    // a gate|up pair (two tensors of one type and shape) takes the K slices of the fused GeGLU launch, so that the two forms add
    // the same sub-blocks in the same order and agree bit for bit
    const bool glu_pair = !DBG && nseg == 2 && segs[0].ggml_type == segs[1].ggml_type && segs[0].n_out == segs[1].n_out &&
                          segs[0].n_out % 32 == 0 && (n_in >> 8) >= 8;
    const int ks = glu_pair ? mp_glu_ks(segs[0].n_out) : mp_pick_ks(total_rows, n_in >> 8);
    // one tile per wave where tile pairs would leave compute units without a workgroup (Gemma-4-12B at 16 rows: q|k|v 128, o_proj
    // and down 120 workgroups of pairs on 256 compute units)
    static const int tpw_force = vmx_tune("VMX_IQ_MMA_TPW", 0);
    const int pair_wgs = (total_rows + 32 * (16 / ks) - 1) / (32 * (16 / ks));
    const int tpw = DBG ? 2 : (tpw_force == 1 || tpw_force == 2 ? tpw_force : (pair_wgs < vmx_num_cus() ? 1 : 2));
    const int rows_per_wg = 16 * tpw * (16 / ks);
    int wg = 0;
    for (int i = 0; i < nseg; i++) {
        a.seg[i] = MpSeg{(const uint8_t *)segs[i].w, y ? y + segs[i].y_off : nullptr, segs[i].ggml_type, segs[i].n_out, wg};
        wg += (segs[i].n_out + rows_per_wg - 1) / rows_per_wg;
        lds = lds > mp_lds_bytes(segs[i].ggml_type) ? lds : mp_lds_bytes(segs[i].ggml_type);
    }
#define MP_GO_T(KSV, TPWV)                                                                                                  \
    do {                                                                                                                    \
        if (lds > 64 * 1024)                                                                                                \
            VMX_CHECK_HIP(hipFuncSetAttribute((const void *)iq_mma_kernel<KSV, DBG, TPWV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        hipLaunchKernelGGL((iq_mma_kernel<KSV, DBG, TPWV>), dim3(wg), dim3(1024), lds, st, a, aq, ad, n_in, m, ldy, dbg);    \
    } while (0)
#define MP_GO(KSV) do { if (tpw == 1) MP_GO_T(KSV, (DBG ? 2 : 1)); else MP_GO_T(KSV, 2); } while (0)
    if (ks == 4) MP_GO(4);
    else if (ks == 8) MP_GO(8);
    else MP_GO(16);
#undef MP_GO
#undef MP_GO_T
    VMX_CHECK_LAUNCH("iq_mmvq(mma)");
    return VMX_OK;
}

}  // namespace

bool vmx_iq_mma_supported(int ggml_type) { return mp_supported(ggml_type); }

extern "C" size_t vmx_iq_mma_packed_bytes(int ggml_type, int n_out, int n_in) {
    if (!mp_supported(ggml_type) || n_out <= 0 || n_in <= 0 || n_in % 256) return 0;
    return (size_t)((n_out + MP_ROWS - 1) / MP_ROWS) * (n_in / 256) * mp_tile_bytes(ggml_type);
}

extern "C" int vmx_iq_mma_pack(int ggml_type, const void *raw, void *packed, int n_out, int n_in, void *stream) {
    VMX_REQUIRE(raw && packed, "iq_mma_pack: null pointer");
    VMX_REQUIRE(mp_supported(ggml_type), "iq_mma_pack: ggml type %d has no 16-row-tile form (IQ3_XXS, IQ3_S, IQ2_S)", ggml_type);
    VMX_REQUIRE(n_out > 0 && n_in > 0 && n_in % 256 == 0, "iq_mma_pack: n_in (%d) must be a positive multiple of 256", n_in);
    VMX_REQUIRE(((uintptr_t)packed & 15) == 0, "iq_mma_pack: packed must be 16-byte aligned");
    const int nsb = n_in / 256;
    const int64_t total = (int64_t)((n_out + MP_ROWS - 1) / MP_ROWS * MP_ROWS) * nsb;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (ggml_type == 18) hipLaunchKernelGGL(iq_mma_pack_kernel<18>, grid, dim3(256), 0, vmx_stream(stream), (const uint8_t *)raw, (uint8_t *)packed, n_out, nsb);
    else if (ggml_type == 21) hipLaunchKernelGGL(iq_mma_pack_kernel<21>, grid, dim3(256), 0, vmx_stream(stream), (const uint8_t *)raw, (uint8_t *)packed, n_out, nsb);
    else hipLaunchKernelGGL(iq_mma_pack_kernel<22>, grid, dim3(256), 0, vmx_stream(stream), (const uint8_t *)raw, (uint8_t *)packed, n_out, nsb);
    VMX_CHECK_LAUNCH("iq_mma_pack");
    return VMX_OK;
}

// up to three tensors that share the activation rows (q|k|v, gate|up): segs[i].w = 16-row-tile form, outputs at y + segs[i].y_off
int vmx_iq_mmvq_mma_multi(const VmxMmvqSeg *segs, int nseg, const int8_t *aq, const float *ad, float *y, int n_in, int m, int ldy, hipStream_t st) {
    return mp_launch<false>(segs, nseg, aq, ad, y, n_in, m, ldy, nullptr, st);
}

// gate|up + GeGLU + q8_1 of the result in one launch (2 .. 16 rows): both tensors of one type and shape, whole 32-row q8_1 blocks,
// at least one super-block per K slice.  The values are those of vmx_iq_mmvq_mma_multi followed by the stand-alone GeGLU kernel,
// bit for bit (same slice order, same roundings).
bool vmx_iq_mma_geglu_ok(const VmxMmvqSeg *gate, const VmxMmvqSeg *up, int n_in, int m) {
    return gate && up && gate->w && up->w && mp_supported(gate->ggml_type) && gate->ggml_type == up->ggml_type && gate->n_out == up->n_out &&
           gate->n_out > 0 && gate->n_out % 32 == 0 && n_in > 0 && n_in % 256 == 0 && (n_in >> 8) >= 8 && m >= 1 && m <= 16;
}
int vmx_iq_mmvq_mma_geglu(const VmxMmvqSeg *gate, const VmxMmvqSeg *up, const int8_t *aq, const float *ad, uint16_t *act, int8_t *aq_out,
                          float *ad_out, int n_in, int m, hipStream_t st) {
    VMX_REQUIRE(vmx_iq_mma_geglu_ok(gate, up, n_in, m) && aq && ad && act && aq_out && ad_out, "iq_mmvq_mma_geglu: unsupported shapes or null pointer");
    VMX_REQUIRE((const void *)aq != (const void *)aq_out, "iq_mmvq_mma_geglu: the q8_1 result must not overwrite the q8_1 input");
    const int ks = mp_glu_ks(gate->n_out);
    const int rt = 16 / ks, tiles = gate->n_out / MP_ROWS, wgs = (tiles + rt - 1) / rt;
    const int lds = mp_lds_bytes(gate->ggml_type) + 256;
    const MpSeg g = {(const uint8_t *)gate->w, nullptr, gate->ggml_type, gate->n_out, 0};
    const MpGlu glu = {(const uint8_t *)up->w, act, aq_out, ad_out};
    if (ks == 4) {
        if (lds > 64 * 1024) VMX_CHECK_HIP(hipFuncSetAttribute((const void *)iq_mma_geglu_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(iq_mma_geglu_kernel<4>, dim3(wgs), dim3(1024), lds, st, g, glu, aq, ad, n_in, m);
    } else {
        if (lds > 64 * 1024) VMX_CHECK_HIP(hipFuncSetAttribute((const void *)iq_mma_geglu_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(iq_mma_geglu_kernel<8>, dim3(wgs), dim3(1024), lds, st, g, glu, aq, ad, n_in, m);
    }
    VMX_CHECK_LAUNCH("iq_mmvq(mma, geglu)");
    return VMX_OK;
}

extern "C" int vmx_iq_mmvq_mma_f32(int ggml_type, const void *packed, const int8_t *aq, const float *ad, float *y, int n_out, int n_in, int m,
                                   void *stream) {
    VMX_REQUIRE(packed && aq && ad && y, "iq_mmvq_mma: null pointer");
    VMX_REQUIRE(mp_supported(ggml_type), "iq_mmvq_mma: ggml type %d has no 16-row-tile form (IQ3_XXS, IQ3_S, IQ2_S)", ggml_type);
    VMX_REQUIRE(m >= 1 && m <= 16, "iq_matmul: m %d exceeds IQ_GEMV_MAX_M (16)", m);
    VMX_REQUIRE(n_in > 0 && n_in % 256 == 0, "iq_matmul: n_in %d not a multiple of 256", n_in);
    VMX_REQUIRE(n_out > 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)aq & 7) == 0, "iq_mmvq_mma: n_out %d / alignment (packed 16, aq 8 bytes)", n_out);
    const VmxMmvqSeg s = {packed, ggml_type, n_out, 0};
    return mp_launch<false>(&s, 1, aq, ad, y, n_in, m, n_out, nullptr, vmx_stream(stream));
}

extern "C" int vmx_iq_subdots_mma_i32(int ggml_type, const void *packed, const int8_t *aq, const float *ad, int32_t *out, int n_out, int n_in, int m,
                                      void *stream) {
    VMX_REQUIRE(packed && aq && ad && out, "iq_subdots_mma: null pointer");
    VMX_REQUIRE(mp_supported(ggml_type) && m >= 1 && m <= 16 && n_out > 0 && n_in > 0 && n_in % 256 == 0, "iq_subdots_mma: bad arguments");
    const VmxMmvqSeg s = {packed, ggml_type, n_out, 0};
    return mp_launch<true>(&s, 1, aq, ad, nullptr, n_in, m, n_out, out, vmx_stream(stream));
}

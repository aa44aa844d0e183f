// EXL3 (ExLlamaV3 trellis-coded weights) for gfx950: tile reconstruction, 128-point Hadamard transforms and the
// small-M GEMM  y = had(had(x * suh) @ decode(trellis)) * svh  (crates/core/src/quantization/exl3.rs:216-286,
// exl3_cuda.rs:271-416; kernels/exl3/{exl3_dq.cuh,codebook.cuh,reconstruct.cu,hadamard_inner.cuh,exl3_gemm_*}).
//
// Format (kept as the checkpoint stores it): trellis i16 [K/16][N/16][16*bits]; a 16x16 tile is a cyclic bit stream
// of 256*bits bits (u32 words, earlier word / higher bit first); weight t reads the 16-bit window that ends at stream
// bit (t+257)*bits; the window goes through one of three multiplicative-hash codebooks evaluated in fp16; weight
// t = 8*lane + i, lane = 8c + 4h + q sits at row 2q + (i&1) + 8*((i>>1)&1), column 2c + h + 8*(i>>2) of the tile
// (NVIDIA's mma.m16n8k16 B-fragment order, reconstruct.cu:36-72).
//
// The GEMM here is gfx950's own: one wave owns a 16-column tile and walks pairs of k-tiles; the two tiles' words are
// fetched with one coalesced load each (lane = word), every lane picks the words its 8 weights need with wave
// permutes, decodes them (5 integer/fp16 ops per weight) straight into a v_mfma_f32_16x16x32_f16 B fragment
// (lane (n, g) holds rows 8g..8g+7 of column n, i.e. four adjacent-t pairs), A = Hadamard-transformed activations
// from LDS.  Accumulation is f32 and deterministic (the reference's stream-K chain rounds partial sums to fp16 in an
// SM-count-dependent order); the output Hadamard runs on the f32 sums as in had_fh_r_128.
#include "vmx_kernels.h"

namespace {

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

constexpr float R_SCALE = 0.088388347648f;  // 1/sqrt(128)

__device__ __forceinline__ half_t bits_to_half(uint32_t b) { return __builtin_bit_cast(half_t, (uint16_t)b); }
__device__ __forceinline__ uint16_t half_to_bits(half_t h) { return __builtin_bit_cast(uint16_t, h); }

template <int CB> __device__ __forceinline__ half_t exl3_decode(uint32_t w) {
    uint32_t x = w;
    if constexpr (CB == 0 || CB == 1) {
        if constexpr (CB == 0) { x *= 89226354u; x += 64248484u; }
        else x *= 0xCBAC1FEDu;
        x = (x & 0x8fff8fffu) ^ 0x3b603b60u;
        return bits_to_half(x & 0xffffu) + bits_to_half(x >> 16);
    } else {
        x *= 0x83DCD12Du;
        const uint32_t sum = (x & 0xff) + ((x >> 8) & 0xff) + ((x >> 16) & 0xff) + (x >> 24) + 0x6400u;
        return __builtin_fmaf16(bits_to_half(sum), bits_to_half(0x1eee), bits_to_half(0xc931));
    }
}

__device__ __forceinline__ half_t exl3_decode_rt(uint32_t w, int cb) {
    return cb == 0 ? exl3_decode<0>(w) : (cb == 1 ? exl3_decode<1>(w) : exl3_decode<2>(w));
}

// 16-bit window of weight t given accessor word(i) over the tile's W = 8*bits words.
template <typename F> __device__ __forceinline__ uint32_t exl3_window(F word, int bits, int t) {
    const int W = 8 * bits;
    const int b1 = (t + 257) * bits, b0 = b1 - 16;       // [b0, b1) in stream bits (before the cyclic wrap)
    const int i0 = b0 >> 5, i1 = (b1 - 1) >> 5;
    const uint32_t a = word(i0 % W), b = word(i1 % W);
    const uint64_t merged = ((uint64_t)a << 32) | b;     // i0 == i1: both halves equal, the shift stays below 32
    return (uint32_t)(merged >> ((i1 + 1) * 32 - b1)) & 0xffffu;
}

// ---------------------------------------------------------------- reconstruct: one thread per weight
__global__ __launch_bounds__(256) void exl3_reconstruct_kernel(const uint16_t *__restrict__ trellis, uint16_t *__restrict__ dense,
                                                                int K, int N, int bits, int cb) {
    const int tn = N / 16;
    const int64_t tile = blockIdx.x;
    const int kt = (int)(tile / tn), nt = (int)(tile % tn);
    const uint32_t *words = reinterpret_cast<const uint32_t *>(trellis + tile * 16 * bits);
    const int t = threadIdx.x;
    const uint32_t w = exl3_window([&](int i) { return words[i]; }, bits, t);
    const int lane = t >> 3, i = t & 7, c = lane >> 3, h = (lane >> 2) & 1, q = lane & 3;
    const int row = 2 * q + (i & 1) + 8 * ((i >> 1) & 1), col = 2 * c + h + 8 * (i >> 2);
    dense[(int64_t)(kt * 16 + row) * N + nt * 16 + col] = half_to_bits(exl3_decode_rt(w, cb));
}

// ---------------------------------------------------------------- Hadamard, one wave per 128-element block
// lane l holds elements 2l, 2l+1.  Stages in stride order 1, 2, ..., 64 with v' = (bit ? -v : v) + partner, all f32
// (the reference's lane-local 4-point butterfly followed by xor-shuffles is the same sequence of additions).
__device__ __forceinline__ void had128_wave(float &e0, float &e1, int lane) {
    const float s = e0 + e1, d = e0 - e1;  // stride 1: element 2l+1 has the bit set: -e1 + e0
    e0 = s;
    e1 = d;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {     // stride 2m in elements = lane bit m
        const float p0 = vmx_lane_xor_m(e0, m), p1 = vmx_lane_xor_m(e1, m);
        const bool neg = (lane & m) != 0;
        e0 = (neg ? -e0 : e0) + p0;
        e1 = (neg ? -e1 : e1) + p1;
    }
}

// mode 0 none, 1 pre-scale (fp16 multiply before), 2 post-scale (fp16 multiply after rounding).  grid (cols/128, rows)
// x_bf16: the input row is bf16 and is cast to fp16 first (Exl3Linear::forward on a bf16 model, exl3.rs:222-228)
__global__ __launch_bounds__(64) void exl3_had_f16_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ scale,
                                                           uint16_t *__restrict__ y, int cols, int mode, int x_bf16) {
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.y * cols + blockIdx.x * 128 + 2 * lane;
    const uint32_t in = *reinterpret_cast<const uint32_t *>(x + base);
    half_t a = bits_to_half(in & 0xffffu), b = bits_to_half(in >> 16);
    if (x_bf16) { a = (half_t)__uint_as_float(in << 16); b = (half_t)__uint_as_float(in & 0xffff0000u); }
    uint32_t sc = 0;
    if (mode) sc = *reinterpret_cast<const uint32_t *>(scale + blockIdx.x * 128 + 2 * lane);
    if (mode == 1) { a = a * bits_to_half(sc & 0xffffu); b = b * bits_to_half(sc >> 16); }
    float e0 = (float)a, e1 = (float)b;
    had128_wave(e0, e1, lane);
    half_t o0 = (half_t)(e0 * R_SCALE), o1 = (half_t)(e1 * R_SCALE);
    if (mode == 2) { o0 = o0 * bits_to_half(sc & 0xffffu); o1 = o1 * bits_to_half(sc >> 16); }
    *reinterpret_cast<uint32_t *>(y + base) = (uint32_t)half_to_bits(o0) | ((uint32_t)half_to_bits(o1) << 16);
}

// Output side: sum of f32 split-K slices -> Hadamard in f32 -> * 1/sqrt(128) -> fp16 -> * svh (fp16).  grid (N/128, M)
// y32 != NULL: instead of fp16 into y, the fp16 value is written as f32 at y32[row * ld32 + col] for consumers that take f32
// linear outputs; every such consumer rounds to the model dtype first (the cast back of exl3.rs:272-277 on a bf16 model,
// nothing on an fp16 model).
__global__ __launch_bounds__(64) void exl3_out_had_kernel(const float *__restrict__ partial, int splits, int64_t slice,
                                                           const uint16_t *__restrict__ svh, uint16_t *__restrict__ y, int N,
                                                           float *__restrict__ y32, int ld32) {
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.y * N + blockIdx.x * 128 + 2 * lane;
    float e0 = 0.f, e1 = 0.f;
    for (int s = 0; s < splits; s++) {
        const float2 v = *reinterpret_cast<const float2 *>(partial + s * slice + base);
        e0 += v.x;
        e1 += v.y;
    }
    had128_wave(e0, e1, lane);
    half_t o0 = (half_t)(e0 * R_SCALE), o1 = (half_t)(e1 * R_SCALE);
    if (svh) {
        const uint32_t sc = *reinterpret_cast<const uint32_t *>(svh + blockIdx.x * 128 + 2 * lane);
        o0 = o0 * bits_to_half(sc & 0xffffu);
        o1 = o1 * bits_to_half(sc >> 16);
    }
    if (y32) {
        float *d = y32 + (int64_t)blockIdx.y * ld32 + blockIdx.x * 128 + 2 * lane;
        *reinterpret_cast<float2 *>(d) = make_float2((float)o0, (float)o1);
    } else {
        *reinterpret_cast<uint32_t *>(y + base) = (uint32_t)half_to_bits(o0) | ((uint32_t)half_to_bits(o1) << 16);
    }
}

// ---------------------------------------------------------------- GEMM core: partial[ks][M][N] = A_had @ decode(trellis)
// Workgroup = 4 waves = 4 adjacent column tiles (one per wave) x KT k-tiles (one split-K slice).  A k-tile row of the four
// tiles is contiguous in the checkpoint layout, so the slice is fetched with 16-byte loads, all requested before anything
// else happens, and staged in LDS in four chunks (a chunk is decoded while the later ones are still in flight).
// LDS holds each tile's bit stream word-reversed: stream word i at slot W-1-i, plus a copy of word W-1 at slot W, so that
// the <= 24-bit window pair of weights (t, t+1) is always inside the two consecutive slots {s, s+1} = stream words
// {i, i-1 (cyclic)}: one two-dword LDS read and one funnel shift per pair, with per-lane slot/shift constants computed
// once.  (The first version moved words between lanes with 16 wave permutes per k-step.)
template <int CB> __device__ __forceinline__ uint32_t exl3_hash(uint32_t w) {  // w < 2^16
    // 32-bit products are quarter rate on CDNA: w * C (mod 2^32) comes from two full-rate 24-bit multiplies,
    // w * (C & 0xFFFFFF) [+ ADD] + ((w * (C >> 24)) << 24).  Written as instructions because the optimiser folds the
    // expression back into one v_mul_lo_u32.
    constexpr uint32_t C = CB == 0 ? 89226354u : (CB == 1 ? 0xCBAC1FEDu : 0x83DCD12Du);
    constexpr uint32_t ADD = CB == 0 ? 64248484u : 0u;
    uint32_t lo, hi, x;
    if constexpr (ADD != 0) asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(lo) : "v"(w), "s"(C & 0xFFFFFFu), "v"(ADD));
    else asm("v_mul_u32_u24 %0, %1, %2" : "=v"(lo) : "s"(C & 0xFFFFFFu), "v"(w));
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(hi) : "s"(C >> 24), "v"(w));
    asm("v_lshl_add_u32 %0, %1, 24, %2" : "=v"(x) : "v"(hi), "v"(lo));
    return x;
}

// decoded weights of the windows w0, w1 as a packed fp16 pair (w0 low)
template <int CB> __device__ __forceinline__ uint32_t exl3_decode_pair(uint32_t w0, uint32_t w1) {
    if constexpr (CB == 0 || CB == 1) {
#ifdef VMX_EXL3_PERM_ADD
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
#endif
        const uint32_t x0 = (exl3_hash<CB>(w0) & 0x8fff8fffu) ^ 0x3b603b60u, x1 = (exl3_hash<CB>(w1) & 0x8fff8fffu) ^ 0x3b603b60u;
#ifdef VMX_EXL3_PERM_ADD
        const uint32_t lo = __builtin_amdgcn_perm(x1, x0, 0x05040100u), hi = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
        return __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, lo) + __builtin_bit_cast(h2, hi));
#else
        // half(lo16) + half(hi16) of each word with the operand halves picked by sub-dword addressing (SDWA): two instructions for the pair instead of
        // two byte permutes and a packed add.  The second add writes the high half of d and keeps the low one.
        uint32_t d;
        asm("v_add_f16_sdwa %0, %1, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(d) : "v"(x0));
        asm("v_add_f16_sdwa %0, %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(d) : "v"(x1));
        return d;
#endif
    } else {
        return (uint32_t)half_to_bits(exl3_decode<2>(w0)) | ((uint32_t)half_to_bits(exl3_decode<2>(w1)) << 16);
    }
}

// HAD: a_had is the raw activation row block x [M][K] (fp16, or bf16 when x_bf16) and suh its sign/scale vector; the
// workgroup transforms the KT/8 128-blocks of its slice itself (fp16(x) * suh -> 128-point Hadamard in f32 -> * 1/sqrt(128)
// -> fp16, the arithmetic of exl3_had_f16_kernel), after the weight requests are out.  Meant for M <= 4 (decode): the
// transform is repeated by every column group.
// Up to three tensors that share the activation rows (q|k|v, gate|up) go out as ONE launch: the column groups (64 columns)
// [cg_end[s-1], cg_end[s]) of the merged output belong to tensor s, which has its own trellis and input sign vector.
// norm_w != NULL (HAD form, bf16 rows): the rows are a residual stream h and the activation is its RMSNorm,
// x = bf16(h * rsqrt(sum(blk_ssq[row][0..K/128)) / K + eps) * norm_w), with the per-128-column sums of squares of h left by
// the kernel that wrote h; the normed row is never stored.
struct Exl3Segs {
    const uint16_t *tr[3];
    const uint16_t *suh[3];
    int cg_end[3];
    const uint16_t *norm_w = nullptr;
    const float *blk_ssq = nullptr;
    float eps = 0.f;
    // attn_tmp != NULL (HAD form, K = heads * 128): the rows are the decode-attention output still split into attn_p <= 4
    // partitions per (row, head) (split_decode_attn_kernel): tmp [M][K/128][P][128] f32, max logit / sum [M][K/128][P]; the
    // prologue merges them, x = bf16(sum_p tmp_p w_p / (sum_p w_p + 1e-6)), w_p = sum_p * exp(max_p - max over p)
    // (paged_attention.cu:840-953) -- what the separate merge kernel would have written.
    const float *attn_tmp = nullptr, *attn_mx = nullptr, *attn_sm = nullptr;
    int attn_p = 0;
};

template <int BITS, int CB, int KT, bool HAD, bool ATTN = false>
__global__ __launch_bounds__(256) void exl3_gemm_kernel(Exl3Segs segs, const uint16_t *__restrict__ a_had, int x_bf16,
                                                         float *__restrict__ partial, int M, int N, int K) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int W = 8 * BITS;            // u32 words per tile
    constexpr int TS = W + 4;              // LDS words per staged tile (W + 1 used)
    constexpr int CT = 4, NCH = 4, KTC = KT / NCH;   // column tiles per workgroup, chunks, k-tiles per chunk
    constexpr int RP = 8 * BITS;           // 16-byte pieces per k-tile row of the workgroup (CT * W / 4)
    constexpr int PPC = KTC * RP;          // pieces per chunk
    constexpr int LPT = (PPC + 255) / 256; // loads per thread per chunk
    constexpr int KS = KT * 16;            // k-values per workgroup
    constexpr int XROW = KS + 8;
    constexpr int XCH = 2;
    uint32_t *tb = reinterpret_cast<uint32_t *>(smem);                              // [KT][CT][TS]
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem + (size_t)KT * CT * TS * 4);   // [M][XROW] fp16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ktiles = K / 16, ncg = N / (16 * CT);   // N: columns of the merged output
    const int cg = blockIdx.x % ncg, ks = blockIdx.x / ncg;
    int sgi = 0, cg0 = 0;
    if (cg >= segs.cg_end[0]) { sgi = 1; cg0 = segs.cg_end[0]; }
    if (cg >= segs.cg_end[1]) { sgi = 2; cg0 = segs.cg_end[1]; }
    const uint16_t *__restrict__ trellis = segs.tr[sgi];
    const uint16_t *__restrict__ suh = segs.suh[sgi];
    const int tn = (segs.cg_end[sgi] - cg0) * CT;   // column tiles of this tensor
    const int kt0 = ks * KT, nt0 = (cg - cg0) * CT, k0 = kt0 * 16;
    const int n = lane & 15, g = lane >> 4;

    // activations first (L2 hits), then every weight piece of the slice
    constexpr int CPR = KS / 8;
    const int total = M * CPR;
    uint4 xv[XCH];
#pragma unroll
    for (int j = 0; j < XCH; j++) {
        const int c = tid + 256 * j;
        xv[j] = make_uint4(0, 0, 0, 0);
        if (!HAD && c < total) {
            const int m = c / CPR, k = k0 + (c % CPR) * 8;
            if (k < K) xv[j] = *reinterpret_cast<const uint4 *>(a_had + (int64_t)m * K + k);
        }
    }
    // fused input transform: this wave's (row, 128-block) items, requested ahead of the weights (loads retire in issue order)
    constexpr int NB = KT / 8;   // 128-blocks of the slice
    uint32_t hin[NB], hsc[NB];   // M <= 4: at most NB items per wave
    uint32_t hnw[NB];
    float hss[NB];
    constexpr int PMX = 4, NBA = ATTN ? NB : 1;
    float2 pt[NBA][PMX];
    float pmx[NBA][PMX], psm[NBA][PMX];
    if constexpr (HAD) {
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int it = min(wave + 4 * i, M * NB - 1), m = it / NB, kb = min(k0 + (it % NB) * 128, K - 128);
            if constexpr (ATTN) {
                const int64_t rh = ((int64_t)m * (K >> 7) + (kb >> 7)) * segs.attn_p;
#pragma unroll
                for (int p = 0; p < PMX; p++) {   // partitions past attn_p re-read the last one; their weight is zeroed below
                    const int64_t rp = rh + min(p, segs.attn_p - 1);
                    pt[i][p] = *reinterpret_cast<const float2 *>(segs.attn_tmp + rp * 128 + 2 * lane);
                    pmx[i][p] = segs.attn_mx[rp];
                    psm[i][p] = segs.attn_sm[rp];
                }
                hin[i] = 0;
            } else
            hin[i] = *reinterpret_cast<const uint32_t *>(a_had + (int64_t)m * K + kb + 2 * lane);
            hsc[i] = suh ? *reinterpret_cast<const uint32_t *>(suh + kb + 2 * lane) : 0x3c003c00u;
            hnw[i] = 0;
            hss[i] = 0.f;
            if (segs.norm_w) {   // uniform; K <= 64 * 128 (checked on the host): one block sum per lane
                hnw[i] = *reinterpret_cast<const uint32_t *>(segs.norm_w + kb + 2 * lane);
                hss[i] = segs.blk_ssq[m * (K >> 7) + min(lane, (K >> 7) - 1)];
            }
        }
    }
    uint4 wr[NCH][LPT];
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int j = 0; j < LPT; j++) {
            // branch-free (a conditional load makes the compiler drain the queue at every join): out-of-range pieces re-read
            // the last valid one and are never staged or decoded
            const int p = min(tid + 256 * j, PPC - 1), kt = min(kt0 + c * KTC + p / RP, ktiles - 1);
            wr[c][j] = *reinterpret_cast<const uint4 *>(trellis + ((int64_t)kt * tn + nt0) * 16 * BITS + (p % RP) * 8);
        }
    if constexpr (HAD) {
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int it = wave + 4 * i;
            if (it < M * NB) {   // wave-uniform
                const int m = it / NB, kb = k0 + (it % NB) * 128;
                uint32_t in = hin[i];
                const uint32_t sc = hsc[i];
                if constexpr (ATTN) {
                    float gmax = -3.402823466e+38f;
#pragma unroll
                    for (int p = 0; p < PMX; p++) gmax = fmaxf(gmax, p < segs.attn_p ? pmx[i][p] : -3.402823466e+38f);
                    float a0 = 0.f, a1 = 0.f, gsum = 0.f;
#pragma unroll
                    for (int p = 0; p < PMX; p++) {
                        const float wt = p < segs.attn_p ? psm[i][p] * __expf(pmx[i][p] - gmax) : 0.f;
                        gsum += wt;
                        a0 += pt[i][p].x * wt;
                        a1 += pt[i][p].y * wt;
                    }
                    const float ginv = 1.0f / (gsum + 1e-6f);
                    // the merged attention row in the activation dtype (what the separate merge kernel would have stored)
                    in = x_bf16 ? ((uint32_t)vmx_f32_to_bf16(a0 * ginv) | ((uint32_t)vmx_f32_to_bf16(a1 * ginv) << 16))
                                : ((uint32_t)vmx_st16<true>(a0 * ginv) | ((uint32_t)vmx_st16<true>(a1 * ginv) << 16));
                }
                if (segs.norm_w) {   // rows and norm weights in the activation dtype (bf16, or fp16 when !x_bf16)
                    const float inv = rsqrtf(vmx_wave_sum(lane < (K >> 7) ? hss[i] : 0.f) / (float)K + segs.eps);
                    if (x_bf16) {
                        const float x0 = __uint_as_float(in << 16) * inv * __uint_as_float(hnw[i] << 16);
                        const float x1 = __uint_as_float(in & 0xffff0000u) * inv * __uint_as_float(hnw[i] & 0xffff0000u);
                        in = (uint32_t)vmx_f32_to_bf16(x0) | ((uint32_t)vmx_f32_to_bf16(x1) << 16);
                    } else {
                        const float x0 = vmx_ld16<true>((uint16_t)(in & 0xffffu)) * inv * vmx_ld16<true>((uint16_t)(hnw[i] & 0xffffu));
                        const float x1 = vmx_ld16<true>((uint16_t)(in >> 16)) * inv * vmx_ld16<true>((uint16_t)(hnw[i] >> 16));
                        in = (uint32_t)vmx_st16<true>(x0) | ((uint32_t)vmx_st16<true>(x1) << 16);
                    }
                }
                half_t a, b;
                if (x_bf16) { a = (half_t)__uint_as_float(in << 16); b = (half_t)__uint_as_float(in & 0xffff0000u); }
                else { a = bits_to_half(in & 0xffffu); b = bits_to_half(in >> 16); }
                if (suh) { a = a * bits_to_half(sc & 0xffffu); b = b * bits_to_half(sc >> 16); }
                float e0 = (float)a, e1 = (float)b;
                had128_wave(e0, e1, lane);
                uint32_t o = (uint32_t)half_to_bits((half_t)(e0 * R_SCALE)) | ((uint32_t)half_to_bits((half_t)(e1 * R_SCALE)) << 16);
                if (kb >= K) o = 0;
                *reinterpret_cast<uint32_t *>(xs + m * XROW + (it % NB) * 128 + 2 * lane) = o;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < XCH; j++) {
            const int c = tid + 256 * j;
            if (c < total) *reinterpret_cast<uint4 *>(xs + (c / CPR) * XROW + (c % CPR) * 8) = xv[j];
        }
        for (int c = XCH * 256 + tid; c < total; c += 256) {
            const int m = c / CPR, k = k0 + (c % CPR) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (k < K) v = *reinterpret_cast<const uint4 *>(a_had + (int64_t)m * K + k);
            *reinterpret_cast<uint4 *>(xs + m * XROW + (c % CPR) * 8) = v;
        }
    }

    // this lane's weights: column n, rows 8*(g&1) .. +7 of k-tile (g>>1) of a step: t = t_base + 8q + {0,1}, q = 0..3.
    // The pair's bits end at stream bit (t+258)*BITS (cyclic); slot = reversed index of the word holding the last bit.
    const int c2 = (n & 7) >> 1, h2 = n & 1, hb = g & 1;
    const int t_base = 64 * c2 + 32 * h2 + 4 * (n >> 3) + 2 * hb;
    int slot[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int b2 = (t_base + 8 * q + 258) * BITS;
        const int em = (b2 - 1) % (256 * BITS) + 1;   // end of the pair's bits, in (0, 256*BITS]
        const int i1 = (em - 1) >> 5;
        slot[q] = W - 1 - i1;
        sh[q] = (i1 + 1) * 32 - em;
    }
    const uint16_t *xrow = xs + min(n, M - 1) * XROW + 8 * g;
    vmx_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; c++) {
#pragma unroll
        for (int j = 0; j < LPT; j++) {
            const int p = tid + 256 * j;
            if (p < PPC) {
                const int r = p % RP, tile = r / (2 * BITS), i = (r % (2 * BITS)) * 4;   // stream words i..i+3 of the tile
                uint32_t *dst = tb + ((c * KTC + p / RP) * CT + tile) * TS;
                *reinterpret_cast<uint4 *>(dst + (W - 4 - i)) = make_uint4(wr[c][j].w, wr[c][j].z, wr[c][j].y, wr[c][j].x);
                if (i == W - 4) dst[W] = wr[c][j].w;
            }
        }
        __syncthreads();
        for (int s = 0; s < KTC / 2; s++) {
            const int kt = c * KTC + 2 * s;   // first k-tile of the step inside the slice
            if (kt0 + kt < ktiles) {          // wave-uniform; K % 32 == 0, so a step is all in or all out
                const uint32_t *tile = tb + ((kt + (g >> 1)) * CT + wave) * TS;
                uint4 bw;
                uint32_t *bwp = reinterpret_cast<uint32_t *>(&bw);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t lo = tile[slot[q]], hi = tile[slot[q] + 1];
                    const uint32_t v = __builtin_amdgcn_alignbit(hi, lo, sh[q]);
                    bwp[q] = exl3_decode_pair<CB>((v >> BITS) & 0xffffu, v & 0xffffu);
                }
                const uint4 au = *reinterpret_cast<const uint4 *>(xrow + kt * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, au), __builtin_bit_cast(f16x8_t, bw), acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int mrow = 4 * g + r;
        if (mrow < M) partial[((int64_t)ks * M + mrow) * N + (cg * CT + wave) * 16 + n] = acc[r];
    }
}

// k-tiles per workgroup: 32 when that still gives every CU a few workgroups, 16 for narrow outputs.
struct Exl3Plan { int kt, splits; };
inline Exl3Plan exl3_plan(int N, int K, int bits) {
    (void)bits;
    const int ktiles = K / 16, kt_long = 32;   // measured on 4096x12288 at 4 bpw: 32 k-tiles (16 KiB work items) beat 64 and 16
    Exl3Plan p;
    p.kt = ((int64_t)(N / 64) * ((ktiles + kt_long - 1) / kt_long) >= 512) ? kt_long : 16;
    p.splits = (ktiles + p.kt - 1) / p.kt;
    return p;
}

template <int BITS, int CB, int KT>
int launch_exl3_gemm_kt(const Exl3Segs &segs, const uint16_t *a, int fuse_had, int x_bf16, float *partial, int M, int N, int K, int splits,
                        hipStream_t st) {
    const size_t lds = (size_t)KT * 4 * (8 * BITS + 4) * 4 + (size_t)M * (KT * 16 + 8) * 2;
    const dim3 grid((unsigned)((N / 64) * splits));
#define EXL3_GO(H)                                                                                                                       \
    do {                                                                                                                                 \
        if (lds > 64 * 1024)                                                                                                             \
            VMX_CHECK_HIP(hipFuncSetAttribute((const void *)exl3_gemm_kernel<BITS, CB, KT, H>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              (int)lds));                                                                                \
        hipLaunchKernelGGL((exl3_gemm_kernel<BITS, CB, KT, H>), grid, dim3(256), lds, st, segs, a, x_bf16, partial, M, N, K);             \
    } while (0)
    if (fuse_had && segs.attn_tmp) {
        hipLaunchKernelGGL((exl3_gemm_kernel<BITS, CB, KT, true, true>), grid, dim3(256), lds, st, segs, a, x_bf16, partial, M, N, K);
    } else if (fuse_had) EXL3_GO(true);
    else EXL3_GO(false);
#undef EXL3_GO
    VMX_CHECK_LAUNCH("exl3_gemm");
    return VMX_OK;
}

// a: Hadamard-transformed fp16 activations (fuse_had == 0) or the raw activations (fuse_had != 0, M <= 4)
template <int BITS>
int launch_exl3_gemm(int cb, const Exl3Segs &segs, const uint16_t *a, int fuse_had, int x_bf16, float *partial, int M, int N, int K,
                     hipStream_t st) {
    const Exl3Plan p = exl3_plan(N, K, BITS);
#define EXL3_L(CBV)                                                                                                                \
    return p.kt == 16 ? launch_exl3_gemm_kt<BITS, CBV, 16>(segs, a, fuse_had, x_bf16, partial, M, N, K, p.splits, st)      \
                      : launch_exl3_gemm_kt<BITS, CBV, 32>(segs, a, fuse_had, x_bf16, partial, M, N, K, p.splits, st)
    switch (cb) {
        case 0: EXL3_L(0);
        case 1: EXL3_L(1);
        default: EXL3_L(2);
    }
#undef EXL3_L
}

int exl3_gemm_dispatch_segs(int bits, int cb, const Exl3Segs &sg, const uint16_t *a, int fuse_had, int x_bf16, float *partial, int M, int N,
                            int K, hipStream_t st) {
    switch (bits) {
        case 1: return launch_exl3_gemm<1>(cb, sg, a, fuse_had, x_bf16, partial, M, N, K, st);
        case 2: return launch_exl3_gemm<2>(cb, sg, a, fuse_had, x_bf16, partial, M, N, K, st);
        case 3: return launch_exl3_gemm<3>(cb, sg, a, fuse_had, x_bf16, partial, M, N, K, st);
        case 4: return launch_exl3_gemm<4>(cb, sg, a, fuse_had, x_bf16, partial, M, N, K, st);
        case 5: return launch_exl3_gemm<5>(cb, sg, a, fuse_had, x_bf16, partial, M, N, K, st);
        case 6: return launch_exl3_gemm<6>(cb, sg, a, fuse_had, x_bf16, partial, M, N, K, st);
        case 7: return launch_exl3_gemm<7>(cb, sg, a, fuse_had, x_bf16, partial, M, N, K, st);
        default: return launch_exl3_gemm<8>(cb, sg, a, fuse_had, x_bf16, partial, M, N, K, st);
    }
}

int exl3_gemm_dispatch(int bits, int cb, const uint16_t *tr, const uint16_t *a, const uint16_t *suh, int fuse_had, int x_bf16,
                       float *partial, int M, int N, int K, hipStream_t st) {
    const int ncg = N / 64;
    const Exl3Segs sg = {{tr, tr, tr}, {suh, suh, suh}, {ncg, ncg, ncg}};
    return exl3_gemm_dispatch_segs(bits, cb, sg, a, fuse_had, x_bf16, partial, M, N, K, st);
}

}  // namespace

extern "C" int vmx_exl3_reconstruct_f16(const void *trellis, uint16_t *dense, int K, int N, int bits, int codebook, void *stream) {
    VMX_REQUIRE(trellis && dense, "exl3_reconstruct: null pointer");
    VMX_REQUIRE(K > 0 && N > 0 && K % 16 == 0 && N % 16 == 0, "exl3_reconstruct: K (%d) and N (%d) must be multiples of 16", K, N);
    VMX_REQUIRE(bits >= 1 && bits <= 8, "exl3_reconstruct: bits per weight (%d) must be in 1..8", bits);
    VMX_REQUIRE(codebook >= 0 && codebook <= 2, "exl3_reconstruct: codebook %d unknown (0 default, 1 mcg, 2 mul1)", codebook);
    hipLaunchKernelGGL(exl3_reconstruct_kernel, dim3((unsigned)((K / 16) * (N / 16))), dim3(256), 0, vmx_stream(stream),
                       (const uint16_t *)trellis, dense, K, N, bits, codebook);
    VMX_CHECK_LAUNCH("exl3_reconstruct");
    return VMX_OK;
}

extern "C" int vmx_exl3_had_r_128_f16(uint16_t *y, const uint16_t *x, const uint16_t *scale, int rows, int cols, int mode, void *stream) {
    VMX_REQUIRE(y && x, "had_r_128: null pointer");
    VMX_REQUIRE(cols > 0 && cols % 128 == 0, "had_r_128: cols (%d) must be a multiple of 128", cols);
    VMX_REQUIRE(mode >= 0 && mode <= 2 && (mode == 0 || scale), "had_r_128: scaled modes need a scale vector");
    if (rows <= 0) return VMX_OK;
    hipLaunchKernelGGL(exl3_had_f16_kernel, dim3(cols / 128, rows), dim3(64), 0, vmx_stream(stream), x, scale, y, cols, mode, 0);
    VMX_CHECK_LAUNCH("had_r_128");
    return VMX_OK;
}

extern "C" size_t vmx_exl3_gemm_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    const size_t mc = M < 16 ? M : 16;
    const size_t splits = (size_t)((K / 16 + 15) / 16);   // the plan depends on the bit width: sized for the shortest slice (16 k-tiles)
    return ((mc * K * 2 + 255) & ~(size_t)255) + splits * mc * N * 4;
}

extern "C" int vmx_exl3_gemm_f16(uint16_t *y, const uint16_t *x, const void *trellis, const uint16_t *suh, const uint16_t *svh, int M,
                                 int K, int N, int bits, int codebook, void *workspace, size_t workspace_bytes, void *stream) {
    VMX_REQUIRE(y && x && trellis && workspace, "exl3_gemm: null pointer");
    VMX_REQUIRE(M >= 1, "exl3_gemm: M (%d) must be positive", M);
    VMX_REQUIRE(K % 128 == 0 && N % 128 == 0 && K > 0 && N > 0, "exl3_gemm: K (%d) and N (%d) must be multiples of 128", K, N);
    VMX_REQUIRE(bits >= 1 && bits <= 8, "exl3_gemm: bits per weight (%d) must be in 1..8", bits);
    VMX_REQUIRE(codebook >= 0 && codebook <= 2, "exl3_gemm: codebook %d unknown", codebook);
    VMX_REQUIRE(workspace_bytes >= vmx_exl3_gemm_workspace_bytes(M, K, N), "exl3_gemm: workspace too small");
    hipStream_t st = vmx_stream(stream);
    const int mc = M < 16 ? M : 16;
    uint16_t *a_had = (uint16_t *)workspace;
    float *partial = (float *)((char *)workspace + (((size_t)mc * K * 2 + 255) & ~(size_t)255));
    const int splits = exl3_plan(N, K, bits).splits;
    const uint16_t *tr = (const uint16_t *)trellis;
    for (int m0 = 0; m0 < M; m0 += 16) {
        const int mm = M - m0 < 16 ? M - m0 : 16;
        const uint16_t *xc = x + (int64_t)m0 * K;
        if (mm <= 4) {   // decode: the input transform runs inside the GEMM workgroups
            if (int rc = exl3_gemm_dispatch(bits, codebook, tr, xc, suh, 1, 0, partial, mm, N, K, st)) return rc;
        } else {
            hipLaunchKernelGGL(exl3_had_f16_kernel, dim3(K / 128, mm), dim3(64), 0, st, xc, suh, a_had, K, suh ? 1 : 0, 0);
            VMX_CHECK_LAUNCH("exl3_gemm(input hadamard)");
            if (int rc = exl3_gemm_dispatch(bits, codebook, tr, a_had, nullptr, 0, 0, partial, mm, N, K, st)) return rc;
        }
        hipLaunchKernelGGL(exl3_out_had_kernel, dim3(N / 128, mm), dim3(64), 0, st, partial, splits, (int64_t)mm * N, svh,
                           y + (int64_t)m0 * N, N, (float *)nullptr, 0);
        VMX_CHECK_LAUNCH("exl3_gemm(output hadamard)");
    }
    return VMX_OK;
}

// ---- internal (C++) entry point used by the decoder: Exl3Linear::forward (exl3.rs:216-286) on `M` rows of a bf16 or fp16
// activation tensor; the fp16 result cast back to the model dtype (bf16) is written as f32 at y32[row * ld + col].
int vmx_exl3_forward_f32(const uint16_t *x, int x_bf16, const void *trellis, const uint16_t *suh, const uint16_t *svh, float *y32,
                         int ld, int M, int K, int N, int bits, int codebook, void *workspace, size_t workspace_bytes, hipStream_t st) {
    VMX_REQUIRE(workspace_bytes >= vmx_exl3_gemm_workspace_bytes(M, K, N), "exl3_forward: workspace too small");
    const int mc = M < 16 ? M : 16;
    uint16_t *a_had = (uint16_t *)workspace;
    float *partial = (float *)((char *)workspace + (((size_t)mc * K * 2 + 255) & ~(size_t)255));
    const int splits = exl3_plan(N, K, bits).splits;
    const uint16_t *tr = (const uint16_t *)trellis;
    for (int m0 = 0; m0 < M; m0 += 16) {
        const int mm = M - m0 < 16 ? M - m0 : 16;
        const uint16_t *xc = x + (int64_t)m0 * K;
        if (mm <= 4) {
            if (int rc = exl3_gemm_dispatch(bits, codebook, tr, xc, suh, 1, x_bf16, partial, mm, N, K, st)) return rc;
        } else {
            hipLaunchKernelGGL(exl3_had_f16_kernel, dim3(K / 128, mm), dim3(64), 0, st, xc, suh, a_had, K, suh ? 1 : 0, x_bf16);
            VMX_CHECK_LAUNCH("exl3_forward(input hadamard)");
            if (int rc = exl3_gemm_dispatch(bits, codebook, tr, a_had, nullptr, 0, 0, partial, mm, N, K, st)) return rc;
        }
        hipLaunchKernelGGL(exl3_out_had_kernel, dim3(N / 128, mm), dim3(64), 0, st, partial, splits, (int64_t)mm * N, svh,
                           (uint16_t *)nullptr, N, y32 + (int64_t)m0 * ld, ld);
        VMX_CHECK_LAUNCH("exl3_forward(output hadamard)");
    }
    return VMX_OK;
}

// Up to three Exl3Linear tensors of equal bit width and codebook that share the input (q|k|v, gate|up), M <= 4 rows, in one
// GEMM launch and one output-transform launch: column c of tensor s lands at y32[row * ld + (columns of tensors < s) + c].
// svh_merged: the tensors' svh vectors back to back.  Workspace as for one linear of the summed width.
// y32 == NULL: only the GEMM runs; *partial_out = its split-K sums f32 [*splits_out][M][N] (inside the workspace) for a
// consumer that applies the output transform itself.
int vmx_exl3_forward_multi_f32(const uint16_t *x, int x_bf16, int nseg, const void *const *trellis, const uint16_t *const *suh,
                               const int *n_out, const uint16_t *svh_merged, float *y32, int ld, int M, int K, int bits, int codebook,
                               void *workspace, size_t workspace_bytes, hipStream_t st, const float **partial_out, int *splits_out,
                               const uint16_t *norm_w, const float *blk_ssq, float eps, const VmxAttnParts *attn_parts) {
    VMX_REQUIRE(nseg >= 1 && nseg <= 3 && M >= 1 && M <= 4, "exl3_forward_multi: %d tensors / %d rows unsupported", nseg, M);
    VMX_REQUIRE(!norm_w || (blk_ssq && K <= 8192), "exl3_forward_multi: the norm form takes rows of at most 8192 columns");
    int N = 0;
    Exl3Segs sg;
    for (int i = 0; i < 3; i++) {
        const int j = i < nseg ? i : nseg - 1;
        if (i < nseg) {
            VMX_REQUIRE(n_out[i] > 0 && n_out[i] % 128 == 0, "exl3_forward_multi: out_features (%d) must be a multiple of 128", n_out[i]);
            N += n_out[i];
        }
        sg.tr[i] = (const uint16_t *)trellis[j];
        sg.suh[i] = suh[j];
        sg.cg_end[i] = N / 64;
    }
    sg.norm_w = norm_w;
    sg.blk_ssq = blk_ssq;
    sg.eps = eps;
    if (attn_parts) {
        VMX_REQUIRE(!norm_w && K % 128 == 0 && attn_parts->max_parts >= 1 && attn_parts->max_parts <= 4 && attn_parts->part_tokens == 0,
                    "exl3_forward_multi: attention partitions (%d) not mergeable here", attn_parts->max_parts);
        sg.attn_tmp = attn_parts->tmp_out;
        sg.attn_mx = attn_parts->max_logits;
        sg.attn_sm = attn_parts->exp_sums;
        sg.attn_p = attn_parts->max_parts;
    }
    VMX_REQUIRE(workspace_bytes >= vmx_exl3_gemm_workspace_bytes(M, K, N), "exl3_forward_multi: workspace too small");
    float *partial = (float *)((char *)workspace + (((size_t)M * K * 2 + 255) & ~(size_t)255));
    const int splits = exl3_plan(N, K, bits).splits;
    if (int rc = exl3_gemm_dispatch_segs(bits, codebook, sg, x, 1, x_bf16, partial, M, N, K, st)) return rc;
    if (partial_out) *partial_out = partial;
    if (splits_out) *splits_out = splits;
    if (!y32) return VMX_OK;
    hipLaunchKernelGGL(exl3_out_had_kernel, dim3(N / 128, M), dim3(64), 0, st, partial, splits, (int64_t)M * N, svh_merged, (uint16_t *)nullptr, N,
                       y32, ld);
    VMX_CHECK_LAUNCH("exl3_forward_multi(output hadamard)");
    return VMX_OK;
}

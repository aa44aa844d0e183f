// GGUF I-quant kernels for gfx950: block dequantisation, q8_1 activation
// quantisation and the q8_1 MMVQ (matrix x quantised-vector) decode kernel.
//
// Semantics follow crates/core/kernels/iq_dequant.cu (dequantize_iq*_f32 :42-194,
// quantize_q8_1_rows :253-267, vec_dot_iq*_q8_1 :274-386, mmvq_iq*_f32 :388-423)
// and their CPU restatement crates/core/src/quantization/gguf/iq/mod.rs:136-725.
// The structure is gfx950's own: wave64, codebooks staged in LDS, v_dot4_i32_i8
// for the integer dots, rows streamed straight into VGPRs with several rows in
// flight per wave, signs applied with a carry-free xor/add (codebook bytes are
// never 0, so two's-complement negation of a byte cannot carry into its neighbour).
#include "iq_common.h"

namespace {

// ------------------------------------------------------------------ dequant
// One thread per 32-weight sub-block; 8 float4 stores per thread.
template <int T>
__global__ __launch_bounds__(256) void iq_dequant_kernel(const uint8_t *__restrict__ blocks, float *__restrict__ out,
                                                          int64_t n_blocks) {
    __shared__ uint32_t cb[IqTraits<T>::LDS_WORDS];
    __shared__ uint2 sg[SignLut<T>::N];
    load_tables<T>(cb, sg);
    const int64_t total = n_blocks * 8;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t blk = s >> 3;
        const int ib = (int)(s & 7);
        const uint8_t *b = blocks + blk * IqTraits<T>::TS;
        const float d = vmx_f16_to_f32((uint16_t)ld16(b));
        int q[8];
        SubScale sc;
        decode_sub<T>(b, ib, cb, sg, q, sc);
        float db0, db1;
        if constexpr (T == 17 || T == 22) {
            db0 = d * (0.5f + (float)sc.ls0) * 0.25f;
            db1 = d * (0.5f + (float)sc.ls1) * 0.25f;
        } else if constexpr (T == 18) {
            db0 = db1 = d * (0.5f + (float)sc.ls0) * 0.5f;
        } else if constexpr (T == 21) {
            db0 = db1 = d * (float)(1 + 2 * sc.ls0);
        } else {
            db0 = db1 = d * (float)(sc.ls0 - 32);
        }
        float4 *y = reinterpret_cast<float4 *>(out + blk * QK_K + ib * 32);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float db = k < 4 ? db0 : db1;
            const int w = q[k];
            float4 v;
            v.x = db * (float)(int8_t)(w & 0xFF);
            v.y = db * (float)(int8_t)((w >> 8) & 0xFF);
            v.z = db * (float)(int8_t)((w >> 16) & 0xFF);
            v.w = db * (float)(int8_t)((w >> 24) & 0xFF);
            y[k] = v;
        }
    }
}

// Dequantise straight into the two bf16 halves of the prompt-sized GEMM (hi = bf16(w), lo = bf16(w - hi)): one thread per
// 32-weight sub-block as above, but the 64 + 64 bytes a thread produces go through LDS so that the workgroup's 16 KiB of each
// half leave as whole 16-byte-per-lane contiguous stores (a thread-per-sub-block store pattern touches 64 cache lines per
// instruction).
template <int T>
__global__ __launch_bounds__(256) void iq_dequant_split_kernel(const uint8_t *__restrict__ blocks, uint16_t *__restrict__ hi,
                                                                uint16_t *__restrict__ lo, int64_t n_blocks) {
    __shared__ uint32_t cb[IqTraits<T>::LDS_WORDS];
    __shared__ uint2 sg[SignLut<T>::N];
    __shared__ __align__(16) uint16_t stage[2][256 * 32];
    load_tables<T>(cb, sg);
    const int64_t total = n_blocks * 8;
    for (int64_t base = (int64_t)blockIdx.x * 256; base < total; base += (int64_t)gridDim.x * 256) {   // uniform per workgroup
        const int64_t s = base + threadIdx.x;
        if (s < total) {
            const int64_t blk = s >> 3;
            const int ib = (int)(s & 7);
            const uint8_t *b = blocks + blk * IqTraits<T>::TS;
            const float d = vmx_f16_to_f32((uint16_t)ld16(b));
            int q[8];
            SubScale sc;
            decode_sub<T>(b, ib, cb, sg, q, sc);
            float db0, db1;
            if constexpr (T == 17 || T == 22) {
                db0 = d * (0.5f + (float)sc.ls0) * 0.25f;
                db1 = d * (0.5f + (float)sc.ls1) * 0.25f;
            } else if constexpr (T == 18) {
                db0 = db1 = d * (0.5f + (float)sc.ls0) * 0.5f;
            } else if constexpr (T == 21) {
                db0 = db1 = d * (float)(1 + 2 * sc.ls0);
            } else {
                db0 = db1 = d * (float)(sc.ls0 - 32);
            }
            uint32_t h[16], l[16];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float db = k < 4 ? db0 : db1;
                const int w = q[k];
                uint16_t hh[4], ll[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float v = db * (float)(int8_t)((w >> (8 * j)) & 0xFF);
                    hh[j] = vmx_f32_to_bf16(v);
                    ll[j] = vmx_f32_to_bf16(v - vmx_bf16_to_f32(hh[j]));
                }
                h[2 * k] = (uint32_t)hh[0] | ((uint32_t)hh[1] << 16);
                h[2 * k + 1] = (uint32_t)hh[2] | ((uint32_t)hh[3] << 16);
                l[2 * k] = (uint32_t)ll[0] | ((uint32_t)ll[1] << 16);
                l[2 * k + 1] = (uint32_t)ll[2] | ((uint32_t)ll[3] << 16);
            }
            uint4 *dh = reinterpret_cast<uint4 *>(stage[0] + threadIdx.x * 32), *dl = reinterpret_cast<uint4 *>(stage[1] + threadIdx.x * 32);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                dh[r] = make_uint4(h[4 * r], h[4 * r + 1], h[4 * r + 2], h[4 * r + 3]);
                dl[r] = make_uint4(l[4 * r], l[4 * r + 1], l[4 * r + 2], l[4 * r + 3]);
            }
        }
        __syncthreads();
        const int64_t live = (total - base < 256 ? total - base : 256) * 4;   // 16-byte pieces of this chunk per half
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = r * 256 + threadIdx.x;
            if (i < live) {
                reinterpret_cast<uint4 *>(hi + base * 32)[i] = reinterpret_cast<const uint4 *>(stage[0])[i];
                reinterpret_cast<uint4 *>(lo + base * 32)[i] = reinterpret_cast<const uint4 *>(stage[1])[i];
            }
        }
        __syncthreads();
    }
}

// Dequantise into the EXACT integer form of the prompt-sized GEMM.  A dequantised I-quant weight is d * c * I with d the f16
// super-block scale, c a power of two and I = (2 ls + 1) * (signed codebook byte) an integer of at most 11 bits (IQ3_XXS: 31 * 62),
// so I is exact in fp16 and the f32 product the reference forms (iq_dequant.cu: db * grid byte, no rounding: 11 + 5 + 6 bits)
// equals (d * c) * I.  Q [rows][K] fp16 = I, Dt [K / 256][rows] f32 = d * c (transposed: a wave reads the scales of 16
// neighbouring columns in one line).  Same LDS staging of the stores as the split kernel.  Not for IQ4_XS (its integers need 13 bits).
template <int T>
__global__ __launch_bounds__(256) void iq_dequant_int_kernel(const uint8_t *__restrict__ blocks, uint16_t *__restrict__ Q,
                                                              float *__restrict__ Dt, int64_t n_blocks, int rows, int nsb) {
    __shared__ uint32_t cb[IqTraits<T>::LDS_WORDS];
    __shared__ uint2 sg[SignLut<T>::N];
    __shared__ __align__(16) uint16_t stage[256 * 32];
    load_tables<T>(cb, sg);
    const int64_t total = n_blocks * 8;
    for (int64_t base = (int64_t)blockIdx.x * 256; base < total; base += (int64_t)gridDim.x * 256) {   // uniform per workgroup
        const int64_t s = base + threadIdx.x;
        if (s < total) {
            const int64_t blk = s >> 3;
            const int ib = (int)(s & 7);
            const uint8_t *b = blocks + blk * IqTraits<T>::TS;
            int q[8];
            SubScale sc;
            decode_sub<T>(b, ib, cb, sg, q, sc);
            const int m0 = 2 * sc.ls0 + 1, m1 = (T == 17 || T == 22) ? 2 * sc.ls1 + 1 : m0;
            uint32_t h[16];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int mul = k < 4 ? m0 : m1, w = q[k];
                uint16_t hh[4];
#pragma unroll
                for (int j = 0; j < 4; j++) hh[j] = __builtin_bit_cast(uint16_t, (_Float16)(float)(mul * (int)(int8_t)((w >> (8 * j)) & 0xFF)));
                h[2 * k] = (uint32_t)hh[0] | ((uint32_t)hh[1] << 16);
                h[2 * k + 1] = (uint32_t)hh[2] | ((uint32_t)hh[3] << 16);
            }
            uint4 *dh = reinterpret_cast<uint4 *>(stage + threadIdx.x * 32);
#pragma unroll
            for (int r = 0; r < 4; r++) dh[r] = make_uint4(h[4 * r], h[4 * r + 1], h[4 * r + 2], h[4 * r + 3]);
            if (ib == 0) {
                const float d = vmx_f16_to_f32((uint16_t)ld16(b));
                const float c = (T == 17 || T == 22) ? 0.125f : (T == 18 ? 0.25f : 1.0f);
                Dt[(blk % nsb) * rows + blk / nsb] = d * c;
            }
        }
        __syncthreads();
        const int64_t live = (total - base < 256 ? total - base : 256) * 4;   // 16-byte pieces of this chunk
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = r * 256 + threadIdx.x;
            if (i < live) reinterpret_cast<uint4 *>(Q + base * 32)[i] = reinterpret_cast<const uint4 *>(stage)[i];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ q8_1
// 8 lanes per 32-value block: one float4 load, one packed int8x4 store per lane.
__global__ __launch_bounds__(256) void quantize_q8_1_kernel(const float *__restrict__ x, int8_t *__restrict__ aq,
                                                             float *__restrict__ ad, int64_t total_blocks) {
    const int64_t lane4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // index of a 4-value group
    const int64_t blk = lane4 >> 3;
    const bool live = blk < total_blocks;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) v = reinterpret_cast<const float4 *>(x)[lane4];
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = fmaxf(amax, vmx_lane_xor<1>(amax));
    amax = fmaxf(amax, vmx_lane_xor<2>(amax));
    amax = fmaxf(amax, vmx_lane_xor<4>(amax));
    const float d = amax / 127.0f;
    uint32_t packed = 0;
    if (amax != 0.0f) {
        int q0 = (int)roundf(v.x / d), q1 = (int)roundf(v.y / d), q2 = (int)roundf(v.z / d), q3 = (int)roundf(v.w / d);
        packed = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) |
                 ((uint32_t)(q3 & 0xFF) << 24);
    }
    if (live) {
        reinterpret_cast<uint32_t *>(aq)[lane4] = packed;
        if ((lane4 & 7) == 0) ad[blk] = d;
    }
}

// Same quantiser fed with bf16 activations (the cast bf16 -> f32 in IqLinear::forward_inner is exact).
__global__ __launch_bounds__(256) void quantize_q8_1_bf16_kernel(const uint16_t *__restrict__ x, int8_t *__restrict__ aq,
                                                                  float *__restrict__ ad, int64_t total_blocks) {
    const int64_t lane4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t blk = lane4 >> 3;
    const bool live = blk < total_blocks;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const uint2 p = reinterpret_cast<const uint2 *>(x)[lane4];
        v = make_float4(__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xFFFF0000u), __uint_as_float(p.y << 16),
                        __uint_as_float(p.y & 0xFFFF0000u));
    }
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = fmaxf(amax, vmx_lane_xor<1>(amax));
    amax = fmaxf(amax, vmx_lane_xor<2>(amax));
    amax = fmaxf(amax, vmx_lane_xor<4>(amax));
    const float d = amax / 127.0f;
    uint32_t packed = 0;
    if (amax != 0.0f) {
        int q0 = (int)roundf(v.x / d), q1 = (int)roundf(v.y / d), q2 = (int)roundf(v.z / d), q3 = (int)roundf(v.w / d);
        packed = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
    }
    if (live) {
        reinterpret_cast<uint32_t *>(aq)[lane4] = packed;
        if ((lane4 & 7) == 0) ad[blk] = d;
    }
}

// ------------------------------------------------------------------ MMVQ
// One wave owns R consecutive output rows per step; lane l walks the row's 32-weight sub-blocks l, l+64, ... G at a
// time.  A lane's position inside its super-block (ib = l & 7) never changes, so the field offsets are per-lane constants,
// row bases are wave-uniform (scalar registers) and a load needs no per-load address arithmetic.  Per step every global
// load is issued before anything is decoded, in the order the data is needed back (vmcnt retires in issue order): first
// the q8_1 activations of the lane's sub-blocks (L2-resident), then the raw weight fields of all R x G sub-blocks; only
// then are the sub-blocks decoded (codebook and sign-mask gathers from LDS) and dotted.  The decode is what bounds this
// kernel (VALU, then LDS), so everything per sub-block is kept to ~60 vector instructions.
// Accumulation: acc += (w_d * a_d) * (float)sub, lanes summed by a 64-wide butterfly.
template <int T, int MT, int R, int G>
__device__ __forceinline__ void mmvq_body(uint32_t *cb, uint2 *sg, const uint8_t *__restrict__ w, const int8_t *__restrict__ aq,
                                          const float *__restrict__ ad, float *__restrict__ y, int n_out, int n_in, int m, int ldy,
                                          int block_index, int num_blocks) {
    const TableRegs<T> treg = tables_fetch<T>();   // committed to LDS after the first step's weight loads are issued
    bool tables_pending = true;
    constexpr int TS = IqTraits<T>::TS;
    constexpr bool ACT_FIRST = MT <= 2;       // activations of the step kept in registers
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ib = lane & 7;
    const int nb256 = n_in >> 8, nblk32 = n_in >> 5;
    const int64_t row_bytes = (int64_t)nb256 * TS;
    const int waves_total = num_blocks * 4;
    for (int row0 = (block_index * 4 + wave) * R; row0 < n_out; row0 += waves_total * R) {
        const uint8_t *rb[R];
#pragma unroll
        for (int r = 0; r < R; r++) rb[r] = w + (int64_t)min(row0 + r, n_out - 1) * row_bytes;
        float acc[R][MT];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int mi = 0; mi < MT; mi++) acc[r][mi] = 0.0f;

        for (int gb = 0; gb < nblk32; gb += 64 * G) {   // wave-uniform
            int4 ua[G][ACT_FIRST ? MT : 1], ub[G][ACT_FIRST ? MT : 1];
            float da[G][ACT_FIRST ? MT : 1];
            int gcl[G];      // the lane's sub-block index, clamped to the row
            int boff[G];     // byte offset of its super-block inside the row
#pragma unroll
            for (int gi = 0; gi < G; gi++) {
                gcl[gi] = min(gb + lane + 64 * gi, nblk32 - 1);   // a clamped duplicate is discarded below
                boff[gi] = (gcl[gi] >> 3) * TS;
            }
            if constexpr (ACT_FIRST) {
#pragma unroll
                for (int gi = 0; gi < G; gi++) {
#pragma unroll
                    for (int mi = 0; mi < MT; mi++) {
                        const int mm = min(mi, m - 1);
                        const int4 *up = reinterpret_cast<const int4 *>(aq + (int64_t)mm * n_in + gcl[gi] * 32);
                        ua[gi][mi] = up[0];
                        ub[gi][mi] = up[1];
                        da[gi][mi] = ad[(int64_t)mm * nblk32 + gcl[gi]];
                    }
                }
            }
            RawSub raw[R][G];
#pragma unroll
            for (int gi = 0; gi < G; gi++)
#pragma unroll
                for (int r = 0; r < R; r++) raw[r][gi] = load_raw<T>(rb[r] + boff[gi], ib);
            if (tables_pending) {
                tables_commit<T>(treg, cb, sg);
                tables_pending = false;
            }
#pragma unroll
            for (int gi = 0; gi < G; gi++) {
                const bool live = gb + lane + 64 * gi < nblk32;
                const int gc = gcl[gi];
                int q[R][8];
                SubScale sc[R];
#pragma unroll
                for (int r = 0; r < R; r++) decode_raw<T>(raw[r][gi], ib, cb, sg, q[r], sc[r]);
#pragma unroll
                for (int mi = 0; mi < MT; mi++) {
                    if (mi < m) {
                        int4 va, vb;
                        float dav;
                        if constexpr (ACT_FIRST) {
                            va = ua[gi][mi]; vb = ub[gi][mi]; dav = da[gi][mi];
                        } else {
                            const int4 *up = reinterpret_cast<const int4 *>(aq + (int64_t)mi * n_in + gc * 32);
                            va = up[0]; vb = up[1]; dav = ad[(int64_t)mi * nblk32 + gc];
                        }
                        if (!live) dav = 0.0f;
                        const int u[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            int s1;
                            const int s0 = subdot_halves(q[r], u, s1);
                            const int sub = scale_int<T>(s0, s1, sc[r]);
                            const float wd = vmx_f16_to_f32((uint16_t)raw[r][gi].d);
                            acc[r][mi] += (wd * dav) * (float)sub;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int mi = 0; mi < MT; mi++) {
                const float v = vmx_wave_sum(acc[r][mi]);
                if (lane == 0 && mi < m && row0 + r < n_out) y[(int64_t)mi * ldy + row0 + r] = v;
            }
    }
    if (tables_pending) tables_commit<T>(treg, cb, sg);   // every wave of the workgroup reaches exactly one barrier
}

template <int T, int MT, int R, int G>
__global__ __launch_bounds__(256) void iq_mmvq_kernel(const uint8_t *__restrict__ w, const int8_t *__restrict__ aq,
                                                       const float *__restrict__ ad, float *__restrict__ y, int n_out,
                                                       int n_in, int m, int ldy) {
    __shared__ uint32_t cb[IqTraits<T>::LDS_WORDS];
    __shared__ uint2 sg[SignLut<T>::N];
    mmvq_body<T, MT, R, G>(cb, sg, w, aq, ad, y, n_out, n_in, m, ldy, blockIdx.x, gridDim.x);
}

// Up to three weights of possibly different I-quant types that share one activation row (q | k | v, gate | up), one
// launch: workgroups [blk0[i], blk0[i+1]) work on segment i and write y + y_off[i].  One launch boundary instead of
// three for projections that are each only a few microseconds long.
struct MmvqSegs {
    const uint8_t *w[3];
    int type[3], n_out[3], y_off[3], blk0[4];
};

// TALL: 0 = the segments' types differ (sized for the largest tables and register need), else the type they all share.
template <int TALL>
__global__ __launch_bounds__(256) void iq_mmvq_multi_kernel(MmvqSegs s, const int8_t *__restrict__ aq, const float *__restrict__ ad,
                                                             float *__restrict__ y, int n_in) {
    constexpr int TT = TALL ? TALL : 18;
    __shared__ uint32_t cb[TALL ? IqTraits<TT>::LDS_WORDS : 2048];
    __shared__ uint2 sg[TALL ? SignLut<TT>::N : 256];
    const int b = blockIdx.x;
    const int seg = b >= s.blk0[2] ? 2 : (b >= s.blk0[1] ? 1 : 0);
    const int bi = b - s.blk0[seg], nb = s.blk0[seg + 1] - s.blk0[seg];
    const uint8_t *w = s.w[seg];
    float *yo = y + s.y_off[seg];
    const int n_out = s.n_out[seg];
    if constexpr (TALL != 0) {
        mmvq_body<TT, 1, 2, 2>(cb, sg, w, aq, ad, yo, n_out, n_in, 1, 0, bi, nb);
    } else {
        switch (s.type[seg]) {
            case 17: mmvq_body<17, 1, 2, 2>(cb, sg, w, aq, ad, yo, n_out, n_in, 1, 0, bi, nb); break;
            case 22: mmvq_body<22, 1, 2, 2>(cb, sg, w, aq, ad, yo, n_out, n_in, 1, 0, bi, nb); break;
            case 18: mmvq_body<18, 1, 2, 2>(cb, sg, w, aq, ad, yo, n_out, n_in, 1, 0, bi, nb); break;
            case 21: mmvq_body<21, 1, 2, 2>(cb, sg, w, aq, ad, yo, n_out, n_in, 1, 0, bi, nb); break;
            default: mmvq_body<23, 1, 2, 2>(cb, sg, w, aq, ad, yo, n_out, n_in, 1, 0, bi, nb); break;
        }
    }
}

// ------------------------------------------------------------------ MMVQ, m == 1: pipelined
// Sum of v over the wave, left in lane 63 (other lanes hold partial sums): four cross-lane adds inside each row of 16
// lanes, then the two row broadcasts.  All DPP, no LDS traffic.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);   // row_mirror: every lane holds its row's sum
    v = dpp_add<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
    return v;
}

// A wave owns one 4096-weight slice of the row (128 sub-blocks, two per lane: l and l + 64), so its q8_1 activations are
// loaded ONCE and stay in registers for every row it processes (re-reading them per row costs three times the weight
// bytes in L1 traffic).  Rows longer than 4096 weights are split over KW = 2 or 4 waves of the workgroup and combined
// through LDS at the end.  Few waves per CU, each streaming many rows through a ring of D row slots: the loads of row
// k + D are issued right after row k is decoded, so a wave's rows return in the order it decodes them and the decode
// (VALU and LDS bound, as long as the weight stream itself) runs beside the stream instead of after it.  With every wave
// requesting all its rows at once the memory pipeline serves all waves' first loads, then all second loads ..., each
// wave's last bytes arrive at the end, and the decode time adds to the stream time (measured: 9.1 us = 6.4 + 3.9 - 1.2).
constexpr int IQ_PIPE_MAX_STEPS = 64;

template <int T, int KW, int D>
__global__ __launch_bounds__(256) void iq_mmvq_pipe_kernel(const uint8_t *__restrict__ w, const int8_t *__restrict__ aq,
                                                            const float *__restrict__ ad, float *__restrict__ y, int n_out,
                                                            int n_in, int steps) {
    __shared__ uint32_t cb[IqTraits<T>::LDS_WORDS];
    __shared__ uint2 sg[SignLut<T>::N];
    __shared__ float part[KW > 1 ? IQ_PIPE_MAX_STEPS * 4 : 1];
    const TableRegs<T> treg = tables_fetch<T>();
    constexpr int TS = IqTraits<T>::TS;
    constexpr int RPS = 4 / KW;            // rows per workgroup step
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = wave % KW, rsub = wave / KW;
    const int ib = lane & 7;
    const int nb256 = n_in >> 8, nblk32 = n_in >> 5;
    const int64_t row_bytes = (int64_t)nb256 * TS;

    // the lane's two sub-blocks and their activations
    int boff[2];
    int4 ua[2], ub[2];
    float da[2];
#pragma unroll
    for (int gi = 0; gi < 2; gi++) {
        const int g = slice * 128 + lane + 64 * gi;
        const int gc = min(g, nblk32 - 1);   // a clamped duplicate is weighted by zero
        boff[gi] = (gc >> 3) * TS;
        const int4 *up = reinterpret_cast<const int4 *>(aq + gc * 32);
        ua[gi] = up[0];
        ub[gi] = up[1];
        da[gi] = g < nblk32 ? ad[gc] : 0.0f;
    }

    auto step_row = [&](int k) { return (k * (int)gridDim.x + (int)blockIdx.x) * RPS + rsub; };
    auto load_step = [&](int k, RawSub (&raw)[2]) {
        // Steps past the end still issue their loads (keeps the loop free of branches around loads, so every wait counts
        // exactly the loads of the D - 1 younger slots) but all lanes then read the first super-block of the matrix:
        // one cache line, nothing streamed.
        const bool real = k < steps;
        const uint8_t *rp = w + (real ? (int64_t)min(step_row(k), n_out - 1) * row_bytes : (int64_t)0);
#pragma unroll
        for (int gi = 0; gi < 2; gi++) raw[gi] = load_raw<T>(rp + (real ? boff[gi] : 0), ib);
    };
    auto compute_step = [&](int k, const RawSub (&raw)[2]) {
        float acc = 0.0f;
#pragma unroll
        for (int gi = 0; gi < 2; gi++) {
            const int u[8] = {ua[gi].x, ua[gi].y, ua[gi].z, ua[gi].w, ub[gi].x, ub[gi].y, ub[gi].z, ub[gi].w};
            int q[8];
            SubScale sc;
            decode_raw<T>(raw[gi], ib, cb, sg, q, sc);
            int s1;
            const int s0 = subdot_halves(q, u, s1);
            const int sub = scale_int<T>(s0, s1, sc);
            acc += (vmx_f16_to_f32((uint16_t)raw[gi].d) * da[gi]) * (float)sub;
        }
        const float v = wave_sum_lane63(acc);
        if (lane == 63) {
            if constexpr (KW == 1) {
                const int row = step_row(k);
                if (row < n_out) y[row] = v;
            } else {
                part[k * 4 + wave] = v;
            }
        }
    };

    RawSub ring[D][2];
#pragma unroll
    for (int j = 0; j < D; j++) load_step(j, ring[j]);
    tables_commit<T>(treg, cb, sg);
    for (int k0 = 0; k0 < steps; k0 += D) {   // wave-uniform; `steps` is the same for every wave of the grid
#pragma unroll
        for (int j = 0; j < D; j++) {
            if (k0 + j < steps) compute_step(k0 + j, ring[j]);
            load_step(k0 + j + D, ring[j]);
        }
    }
    if constexpr (KW > 1) {
        __syncthreads();
        const int t = threadIdx.x;
        if (t < steps * RPS) {
            const int k = t / RPS, rs = t % RPS;
            const int row = (k * (int)gridDim.x + (int)blockIdx.x) * RPS + rs;
            float v = 0.0f;
#pragma unroll
            for (int sl = 0; sl < KW; sl++) v += part[k * 4 + rs * KW + sl];
            if (row < n_out) y[row] = v;
        }
    }
}

// Integer sub-dots only (parity aid): out[(o*m + mi)*nblk32 + g].
template <int T>
__global__ __launch_bounds__(256) void iq_subdots_kernel(const uint8_t *__restrict__ w, const int8_t *__restrict__ aq,
                                                          int32_t *__restrict__ out, int n_out, int n_in, int m) {
    __shared__ uint32_t cb[IqTraits<T>::LDS_WORDS];
    __shared__ uint2 sg[SignLut<T>::N];
    load_tables<T>(cb, sg);
    const int nb256 = n_in >> 8, nblk32 = n_in >> 5;
    const int64_t total = (int64_t)n_out * nblk32;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i / nblk32), g = (int)(i % nblk32);
        const uint8_t *b = w + ((int64_t)o * nb256 + (g >> 3)) * IqTraits<T>::TS;
        int q[8];
        SubScale sc;
        decode_sub<T>(b, g & 7, cb, sg, q, sc);
        for (int mi = 0; mi < m; mi++) {
            const int *up = reinterpret_cast<const int *>(aq + (int64_t)mi * n_in + g * 32);
            int u[8];
#pragma unroll
            for (int k = 0; k < 8; k++) u[k] = up[k];
            int s1;
            const int s0 = subdot_halves(q, u, s1);
            out[((int64_t)o * m + mi) * nblk32 + g] = scale_int<T>(s0, s1, sc);
        }
    }
}

// ------------------------------------------------------------------ Q6_K gather
// One 64-thread workgroup per (id, super-block): thread t -> half t/32, l = t%32, 4 outputs.
__global__ __launch_bounds__(64) void gather_q6k_kernel(const uint8_t *__restrict__ blocks, const int32_t *__restrict__ ids,
                                                         float *__restrict__ out, int num_ids, int blocks_per_row,
                                                         int num_embeddings) {
    const int gb = blockIdx.x;
    if (gb >= num_ids * blocks_per_row) return;
    const int out_row = gb / blocks_per_row, bir = gb % blocks_per_row;
    const int id = ids[out_row];
    const int t = threadIdx.x, half = t >> 5, l = t & 31;
    float *yo = out + ((int64_t)out_row * blocks_per_row + bir) * QK_K + 128 * half;
    if (id < 0 || id >= num_embeddings) {
        yo[l] = 0.f; yo[l + 32] = 0.f; yo[l + 64] = 0.f; yo[l + 96] = 0.f;
        return;
    }
    const uint8_t *b = blocks + ((int64_t)id * blocks_per_row + bir) * 210;
    const uint8_t *ql = b + 64 * half, *qh = b + 128 + 32 * half;
    const int8_t *sc = reinterpret_cast<const int8_t *>(b + 192) + 8 * half;
    const float d = vmx_f16_to_f32((uint16_t)ld16(b + 208));
    const int is = l >> 4;
    const int q1 = (int)((ql[l] & 0xF) | ((qh[l] & 3) << 4)) - 32;
    const int q2 = (int)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
    const int q3 = (int)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
    const int q4 = (int)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
    yo[l] = d * (float)sc[is] * (float)q1;
    yo[l + 32] = d * (float)sc[is + 2] * (float)q2;
    yo[l + 64] = d * (float)sc[is + 4] * (float)q3;
    yo[l + 96] = d * (float)sc[is + 6] * (float)q4;
}

inline int grid_for(int64_t work_items, int per_block, int cap) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

template <int T, int MT, int R, int G>
void launch_mmvq_cfg(const uint8_t *w, const int8_t *aq, const float *ad, float *y, int n_out, int n_in, int m, int ldy, hipStream_t st) {
    // rows per workgroup = 4 waves * R; cap the grid and stride the rest.
    hipLaunchKernelGGL((iq_mmvq_kernel<T, MT, R, G>), dim3(grid_for(n_out, 4 * R, 16384)), dim3(256), 0, st, w, aq, ad, y, n_out, n_in, m, ldy);
}

template <int T, int KW>
void launch_ring(const uint8_t *w, const int8_t *aq, const float *ad, float *y, int n_out, int n_in, hipStream_t st) {
    constexpr int RPS = 4 / KW;
    const int step_rows = (n_out + RPS - 1) / RPS;                 // workgroup-steps of work
    int grid = min(step_rows, vmx_num_cus() * 4);
    int steps = (step_rows + grid - 1) / grid;
    if (steps > IQ_PIPE_MAX_STEPS) steps = IQ_PIPE_MAX_STEPS;
    grid = (step_rows + steps - 1) / steps;                        // even share: every workgroup runs `steps` steps
    hipLaunchKernelGGL((iq_mmvq_pipe_kernel<T, KW, 2>), dim3(grid), dim3(256), 0, st, w, aq, ad, y, n_out, n_in, steps);
}

template <int T>
int launch_mmvq(const uint8_t *w, const int8_t *aq, const float *ad, float *y, int n_out, int n_in, int m, int ldy, hipStream_t st) {
    if (m == 1 && n_in > 4096 && n_in <= 16384) {   // (with one activation row ldy is unused)
        // long rows: split over 2 or 4 waves, activations resident in registers (measured 9.3 vs 11.3 us on 3840 x 15360)
        if (n_in <= 8192) launch_ring<T, 2>(w, aq, ad, y, n_out, n_in, st);
        else launch_ring<T, 4>(w, aq, ad, y, n_out, n_in, st);
    } else if (m == 1) {
        launch_mmvq_cfg<T, 1, 2, 2>(w, aq, ad, y, n_out, n_in, m, ldy, st);
    } else if (m == 2) {
        launch_mmvq_cfg<T, 2, 2, 2>(w, aq, ad, y, n_out, n_in, m, ldy, st);
    } else if (m <= 4) {
        launch_mmvq_cfg<T, 4, 2, 2>(w, aq, ad, y, n_out, n_in, m, ldy, st);
    } else if (m <= 8) {
        launch_mmvq_cfg<T, 8, 1, 2>(w, aq, ad, y, n_out, n_in, m, ldy, st);
    } else {
        launch_mmvq_cfg<T, 16, 1, 2>(w, aq, ad, y, n_out, n_in, m, ldy, st);
    }
    VMX_CHECK_LAUNCH("iq_mmvq");
    return VMX_OK;
}

}  // namespace

#define IQ_DISPATCH(type, CALL)                                   \
    switch (type) {                                               \
        case 17: CALL(17); break;                                 \
        case 22: CALL(22); break;                                 \
        case 18: CALL(18); break;                                 \
        case 21: CALL(21); break;                                 \
        case 23: CALL(23); break;                                 \
        default:                                                  \
            vmx_set_error("unsupported ggml type id %d", type);  \
            return VMX_EUNSUPPORTED;                              \
    }

extern "C" int vmx_iq_type_size(int t) {
    switch (t) {
        case 17: return 74;
        case 22: return 82;
        case 18: return 98;
        case 21: return 110;
        case 23: return 136;
        default: return -1;
    }
}

extern "C" int vmx_iq_dequantize_f32(int ggml_type, const void *blocks, float *out, int64_t n_blocks, void *stream) {
    VMX_REQUIRE(blocks && out, "iq_dequantize: null pointer");
    VMX_REQUIRE(n_blocks >= 0, "iq_dequantize: n_blocks %lld < 0", (long long)n_blocks);
    if (n_blocks == 0) return VMX_OK;
    const int grid = grid_for(n_blocks * 8, 256, 16384);
#define CALL(T) hipLaunchKernelGGL((iq_dequant_kernel<T>), dim3(grid), dim3(256), 0, vmx_stream(stream), (const uint8_t *)blocks, out, n_blocks)
    IQ_DISPATCH(ggml_type, CALL)
#undef CALL
    VMX_CHECK_LAUNCH("iq_dequantize");
    return VMX_OK;
}

int vmx_iq_dequantize_split_bf16(int ggml_type, const void *blocks, uint16_t *hi, uint16_t *lo, int64_t n_blocks, hipStream_t st) {
    if (n_blocks <= 0) return VMX_OK;
    const int grid = grid_for(n_blocks * 8, 256, 16384);
#define CALL(T) hipLaunchKernelGGL((iq_dequant_split_kernel<T>), dim3(grid), dim3(256), 0, st, (const uint8_t *)blocks, hi, lo, n_blocks)
    IQ_DISPATCH(ggml_type, CALL)
#undef CALL
    VMX_CHECK_LAUNCH("iq_dequantize_split");
    return VMX_OK;
}

bool vmx_iq_int_form_ok(int ggml_type) { return ggml_type == 17 || ggml_type == 18 || ggml_type == 21 || ggml_type == 22; }

int vmx_iq_dequantize_int_f16(int ggml_type, const void *blocks, uint16_t *Q, float *Dt, int rows, int nsb, hipStream_t st) {
    const int64_t n_blocks = (int64_t)rows * nsb;
    if (n_blocks <= 0) return VMX_OK;
    const int grid = grid_for(n_blocks * 8, 256, 16384);
#define CALL(T) hipLaunchKernelGGL((iq_dequant_int_kernel<T>), dim3(grid), dim3(256), 0, st, (const uint8_t *)blocks, Q, Dt, n_blocks, rows, nsb)
    switch (ggml_type) {
        case 17: CALL(17); break;
        case 18: CALL(18); break;
        case 21: CALL(21); break;
        case 22: CALL(22); break;
        default: vmx_set_error("iq_dequantize_int: GGML type %d has no exact fp16 integer form", ggml_type); return VMX_EUNSUPPORTED;
    }
#undef CALL
    VMX_CHECK_LAUNCH("iq_dequantize_int");
    return VMX_OK;
}

extern "C" int vmx_quantize_q8_1_rows(const float *x, int8_t *aq, float *ad, int64_t total_blocks, void *stream) {
    VMX_REQUIRE(x && aq && ad, "quantize_q8_1: null pointer");
    VMX_REQUIRE(total_blocks >= 0, "quantize_q8_1: total_blocks < 0");
    if (total_blocks == 0) return VMX_OK;
    const int64_t groups = total_blocks * 8;
    hipLaunchKernelGGL(quantize_q8_1_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, vmx_stream(stream), x, aq, ad,
                       total_blocks);
    VMX_CHECK_LAUNCH("quantize_q8_1");
    return VMX_OK;
}

extern "C" int vmx_iq_mmvq_f32(int ggml_type, const void *w, const int8_t *aq, const float *ad, float *y, int n_out, int n_in,
                               int m, void *stream) {
    VMX_REQUIRE(w && aq && ad && y, "iq_mmvq: null pointer");
    VMX_REQUIRE(m >= 1 && m <= 16, "iq_matmul: m %d exceeds IQ_GEMV_MAX_M (16)", m);
    VMX_REQUIRE(n_in > 0 && n_in % QK_K == 0, "iq_matmul: n_in %d not a multiple of 256", n_in);
    VMX_REQUIRE(n_out > 0, "iq_matmul: n_out %d", n_out);
#define CALL(T) { int rc = launch_mmvq<T>((const uint8_t *)w, aq, ad, y, n_out, n_in, m, n_out, vmx_stream(stream)); if (rc) return rc; }
    IQ_DISPATCH(ggml_type, CALL)
#undef CALL
    return VMX_OK;
}

extern "C" size_t vmx_iq_matmul_workspace_bytes(int n_in, int m) {
    if (n_in <= 0 || m <= 0) return 0;
    size_t q = ((size_t)m * n_in + 255) & ~(size_t)255;
    return q + (size_t)m * (n_in / 32) * sizeof(float);
}

extern "C" int vmx_iq_matmul_f32(int ggml_type, const void *w, const float *x, float *y, int n_out, int n_in, int m,
                                 void *workspace, size_t workspace_bytes, void *stream) {
    VMX_REQUIRE(w && x && y && workspace, "iq_matmul: null pointer");
    VMX_REQUIRE(m >= 1 && m <= 16, "iq_matmul: m %d exceeds IQ_GEMV_MAX_M (16)", m);
    VMX_REQUIRE(n_in > 0 && n_in % QK_K == 0, "iq_matmul: n_in %d not a multiple of 256", n_in);
    VMX_REQUIRE(workspace_bytes >= vmx_iq_matmul_workspace_bytes(n_in, m), "iq_matmul: workspace too small");
    int8_t *aq = (int8_t *)workspace;
    float *ad = (float *)((char *)workspace + (((size_t)m * n_in + 255) & ~(size_t)255));
    int rc = vmx_quantize_q8_1_rows(x, aq, ad, (int64_t)m * (n_in / 32), stream);
    if (rc) return rc;
    return vmx_iq_mmvq_f32(ggml_type, w, aq, ad, y, n_out, n_in, m, stream);
}

extern "C" int vmx_iq_subdots_i32(int ggml_type, const void *w, const int8_t *aq, int32_t *out, int n_out, int n_in, int m,
                                  void *stream) {
    VMX_REQUIRE(w && aq && out, "iq_subdots: null pointer");
    VMX_REQUIRE(n_in > 0 && n_in % QK_K == 0 && n_out > 0 && m >= 1, "iq_subdots: bad shape");
    const int grid = grid_for((int64_t)n_out * (n_in / 32), 256, 4096);
#define CALL(T) hipLaunchKernelGGL((iq_subdots_kernel<T>), dim3(grid), dim3(256), 0, vmx_stream(stream), (const uint8_t *)w, aq, out, n_out, n_in, m)
    IQ_DISPATCH(ggml_type, CALL)
#undef CALL
    VMX_CHECK_LAUNCH("iq_subdots");
    return VMX_OK;
}

extern "C" int vmx_gather_dequant_q6k_f32(const void *blocks, const int32_t *ids, float *out, int num_ids, int dim,
                                          int num_embeddings, void *stream) {
    VMX_REQUIRE(blocks && ids && out, "gather_dequant_q6k: null pointer");
    VMX_REQUIRE(dim > 0 && dim % QK_K == 0, "gather_dequant_q6k: dim %d not a multiple of 256", dim);
    VMX_REQUIRE(num_ids >= 0 && num_embeddings > 0, "gather_dequant_q6k: bad counts");
    if (num_ids == 0) return VMX_OK;
    const int bpr = dim / QK_K;
    hipLaunchKernelGGL(gather_q6k_kernel, dim3(num_ids * bpr), dim3(64), 0, vmx_stream(stream), (const uint8_t *)blocks, ids, out,
                       num_ids, bpr, num_embeddings);
    VMX_CHECK_LAUNCH("gather_dequant_q6k");
    return VMX_OK;
}

// ---- internal (C++) entry points used by the decoder
int vmx_quantize_q8_1_bf16(const uint16_t *x, int8_t *aq, float *ad, int64_t total_blocks, hipStream_t st) {
    const int64_t groups = total_blocks * 8;
    hipLaunchKernelGGL(quantize_q8_1_bf16_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, x, aq, ad, total_blocks);
    VMX_CHECK_LAUNCH("quantize_q8_1(bf16)");
    return VMX_OK;
}

// y[mi*ldy + row]: lets several projections write into one merged activation row
int vmx_iq_mmvq_ld(int ggml_type, const void *w, const int8_t *aq, const float *ad, float *y, int n_out, int n_in, int m, int ldy,
                   hipStream_t st) {
#define CALL(T) { int rc = launch_mmvq<T>((const uint8_t *)w, aq, ad, y, n_out, n_in, m, ldy, st); if (rc) return rc; }
    IQ_DISPATCH(ggml_type, CALL)
#undef CALL
    return VMX_OK;
}

// Several projections of one activation (m rows, q8_1 in aq/ad), outputs at y + y_off[i] with row stride ldy.  With one
// activation row and short weight rows everything goes out as one launch; otherwise one launch per projection.
int vmx_iq_mmvq_multi(const VmxMmvqSeg *segs, int nseg, const int8_t *aq, const float *ad, float *y, int n_in, int m, int ldy,
                      hipStream_t st) {
    if (nseg < 1 || nseg > 3) {
        vmx_set_error("iq_mmvq_multi: %d segments (1..3)", nseg);
        return VMX_EINVAL;
    }
    for (int i = 0; i < nseg; i++)
        if (vmx_iq_type_size(segs[i].ggml_type) < 0) {
            vmx_set_error("unsupported ggml type id %d", segs[i].ggml_type);
            return VMX_EUNSUPPORTED;
        }
    if (m != 1 || n_in > 4096 || nseg == 1) {
        for (int i = 0; i < nseg; i++)
            if (int rc = vmx_iq_mmvq_ld(segs[i].ggml_type, segs[i].w, aq, ad, y + segs[i].y_off, segs[i].n_out, n_in, m, ldy, st)) return rc;
        return VMX_OK;
    }
    MmvqSegs s = {};
    int blk = 0;
    for (int i = 0; i < 3; i++) {
        s.blk0[i] = blk;
        if (i < nseg) {
            s.w[i] = (const uint8_t *)segs[i].w;
            s.type[i] = segs[i].ggml_type;
            s.n_out[i] = segs[i].n_out;
            s.y_off[i] = segs[i].y_off;
            blk += (segs[i].n_out + 7) / 8;   // 4 waves x 2 rows per workgroup
        } else {
            s.type[i] = 18;
        }
    }
    s.blk0[3] = blk;
    for (int i = nseg; i < 3; i++) s.blk0[i] = blk;
    bool same = true;
    for (int i = 1; i < nseg; i++) same = same && segs[i].ggml_type == segs[0].ggml_type;
#define CALL(T) hipLaunchKernelGGL((iq_mmvq_multi_kernel<T>), dim3(blk), dim3(256), 0, st, s, aq, ad, y, n_in)
    if (!same) {
        CALL(0);
    } else {
        IQ_DISPATCH(segs[0].ggml_type, CALL)
    }
#undef CALL
    VMX_CHECK_LAUNCH("iq_mmvq(multi)");
    return VMX_OK;
}

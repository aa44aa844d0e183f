// AWQ / GPTQ int4 kernels for gfx950: load-time repacks and the small-M
// dequant-fused GEMM that replaces the reference's awq_gemv_* / marlin_gemm_int4*
// kernels (crates/core/kernels/awq_gemv.cu:46-665, marlin_gemm.cu:67-187,301-430).
//
// Arithmetic contract (awq_gemv.cu:70-117): y[m,n] = bf16(sum_k x[m,k] * ((nib - zp) * s) + bias),
// bf16 x and s, f32 accumulation.  This kernel evaluates the same sum re-associated per
// quantisation group on the matrix cores:
//     sum_k x*(nib - zp)*s = s * ( sum_k x*(128 + nib)  -  (128 + zp) * sum_k x )
// 0x4300|nib is the bf16 value 128+nib exactly, so a packed word unpacks to an MFMA
// B-fragment with one and-or per pair of weights; products are exact in f32 and only the
// summation order differs from the reference.
//
// HBM layout ("tile stream", built once at load by vmx_int4_pack):
//   W  [N/16][K/128][64][4] u32 : for column tile nt and k-block kb one contiguous 1 KiB unit;
//       lane l of a wave reads its dwordx4 at l*16.  Word i of lane l holds the 8 weights
//       k = kb*128 + 32*i + 8*(l>>4) + j (j = 0..7) of column nt*16 + (l&15), weight j at
//       nibble position {0,4,1,5,2,6,3,7}[j], so (w >> 4t) & 0x000F000F yields weights 2t, 2t+1.
//   SZ [N/16][G][16] u32 : low half = bf16 scale bits, high half = bf16 bits of 128 + zp.
// One wave-load is therefore 1 KiB contiguous = 16 columns x one 128-deep group slice,
// and feeds four v_mfma_f32_16x16x32_bf16 (A = activations, rows = m).
#include "vmx_kernels.h"
#include <stdlib.h>

namespace {

constexpr int KB = 128;     // k-values per unit
constexpr int NT = 16;      // columns per unit

// Registers (vector + accumulation) are capped so that at least this many waves fit a SIMD: a launch of ~6 workgroups per
// CU must be resident all at once, or the second half of the grid only starts when the first retires (measured: at 120
// registers exactly half of the 1536 gate|up workgroups of a Qwen3-8B layer entered 4.5 us after the others).
#ifndef WIDE_WAVES_PER_EU
#define WIDE_WAVES_PER_EU 6
#endif

__host__ __device__ inline size_t sz_offset_bytes(int K, int N) { return ((size_t)K * N / 2 + 255) & ~(size_t)255; }

// ---------------------------------------------------------------- repacks
// AWQ [K, N/8] (interleaved along N) -> GPTQ-sequential [K/8, N] (awq_marlin.rs:170-240).
__global__ __launch_bounds__(256) void awq_to_gptq_kernel(const uint32_t *__restrict__ awq, uint32_t *__restrict__ out, int K,
                                                           int N, int ldo) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [K/8, N]
    if (idx >= (int64_t)(K / 8) * N) return;
    const int n = (int)(idx % N), kb = (int)(idx / N);
    const int shift = ((n & 1) * 16) + ((n & 7) >> 1) * 4;  // {0,16,4,20,8,24,12,28}[n%8]
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t src = awq[(int64_t)(kb * 8 + i) * (N / 8) + (n >> 3)];
        word |= ((src >> shift) & 0xFu) << (4 * i);
    }
    out[(int64_t)kb * ldo + n] = word;
}

// rows x cols words; out row stride ldo (for writing into a wider, merged tensor)
__global__ __launch_bounds__(256) void awq_nibble_reorder_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                                  int64_t n, int cols, int ldo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = in[i];
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int shift = ((j & 1) * 16) + (j >> 1) * 4;
        r |= ((w >> shift) & 0xFu) << (4 * j);
    }
    out[(i / cols) * ldo + (i % cols)] = r;
}

// Source column of packed column (nt, nl).  glu_half == 0: identity.  glu_half = N/2 (merged gate|up linear whose GEMM
// epilogue applies the gated activation): tile nt holds gate columns 8nt..8nt+7 in its lanes 0..7 and the matching up
// columns in lanes 8..15, so both operands of act[j] = silu(gate[j]) * up[j] end in one wave.
__device__ __forceinline__ int pack_src_col(int nt, int nl, int glu_half) {
    if (glu_half == 0) return nt * NT + nl;
    return (nl < 8 ? 0 : glu_half) + nt * 8 + (nl & 7);
}

__global__ __launch_bounds__(256) void int4_pack_w_kernel(const uint32_t *__restrict__ qweight, uint32_t *__restrict__ W, int K,
                                                           int N, int glu_half) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over K*N/8 output words
    if (idx >= (int64_t)K * N / 8) return;
    const int i = (int)(idx & 3), l = (int)((idx >> 2) & 63);
    const int64_t unit = idx >> 8;
    const int kblocks = K / KB;
    const int kb = (int)(unit % kblocks), nt = (int)(unit / kblocks);
    const int n = pack_src_col(nt, l & 15, glu_half);
    const int c = 4 * i + (l >> 4);  // 8-wide k chunk inside the unit
    const uint32_t src = qweight[(int64_t)(kb * 16 + c) * N + n];
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int pos = (j >> 1) + 4 * (j & 1);  // {0,4,1,5,2,6,3,7}[j]
        r |= ((src >> (4 * j)) & 0xFu) << (4 * pos);
    }
    W[idx] = r;
}

__global__ __launch_bounds__(256) void int4_pack_sz_kernel(const uint16_t *__restrict__ scales, const uint32_t *__restrict__ qzeros,
                                                            uint32_t *__restrict__ SZ, int G, int N, int glu_half) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [N/16][G][16]
    if (idx >= (int64_t)G * N) return;
    const int nl = (int)(idx & 15);
    const int g = (int)((idx >> 4) % G), nt = (int)((idx >> 4) / G);
    const int n = pack_src_col(nt, nl, glu_half);
    uint32_t zp = 8;
    if (qzeros) zp = (qzeros[(int64_t)g * (N / 8) + (n >> 3)] >> (4 * (n & 7))) & 0xFu;
    SZ[idx] = (uint32_t)scales[(int64_t)g * N + n] | ((0x4300u | zp) << 16);
}

// tile stream -> dense bf16 [K, N]; value bf16((nib - zp) * scale) as awq_marlin_dequant.cu:35-69.
__global__ __launch_bounds__(256) void int4_dequant_kernel(const uint32_t *__restrict__ W, const uint32_t *__restrict__ SZ,
                                                            uint16_t *__restrict__ dense, int K, int N, int group) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)K * N / 8) return;
    const int i = (int)(idx & 3), l = (int)((idx >> 2) & 63);
    const int64_t unit = idx >> 8;
    const int kblocks = K / KB, G = K / group;
    const int kb = (int)(unit % kblocks), nt = (int)(unit / kblocks);
    const int n = nt * NT + (l & 15);
    const int k0 = kb * KB + 32 * i + 8 * (l >> 4);
    const uint32_t sz = SZ[((int64_t)nt * G + k0 / group) * 16 + (l & 15)];
    const float s = vmx_bf16_to_f32((uint16_t)(sz & 0xFFFFu));
    const int zp = (int)((sz >> 16) & 0xFu);
    const uint32_t w = W[idx];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int pos = (j >> 1) + 4 * (j & 1);
        const int nib = (int)((w >> (4 * pos)) & 0xFu);
        dense[(int64_t)(k0 + j) * N + n] = vmx_f32_to_bf16((float)(nib - zp) * s);
    }
}

// ---------------------------------------------------------------- small-M GEMM
// Workgroup = 4 waves = one 16-column tile x (4*UPW) consecutive k-blocks (one split-K slice);
// wave w owns k-blocks [w*UPW, (w+1)*UPW) of the slice.  One work item per workgroup and every
// weight byte of the matrix requested at kernel start: measured on MI355X, a launch streams
// fastest with the whole matrix in flight (several workgroups resident per CU).
//
// Order of the global loads matters: vmcnt retires loads in issue order, so anything requested
// AFTER the weight units cannot be consumed before all of them have landed.  The activation chunks
// and the scale|zero words (L2-resident, quick) are therefore requested first, the weight units
// second; staging then waits only for the former (counted vmcnt) and the unpack/MFMA work of unit
// u starts as soon as unit u is back while units u+1.. are still in flight.
//
// TAIL: the last split-K slice may run past K (k-blocks not a multiple of 4*UPW); only then are the
// per-unit bounds checks compiled in.
constexpr int XCH = 4;  // activation chunks (16 B) a thread keeps in registers across the weight requests

// Weight units are read once per launch by one CU: the non-temporal policy keeps them from displacing the activations and
// partial sums that ARE re-read from the caches (VMX_INT4_NT_OFF at build time restores the default policy for A/B runs).
__device__ __forceinline__ uint4 ld_weights(const uint4 *p) {
#ifdef VMX_INT4_NT_OFF
    return *p;
#else
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#endif
}

__device__ __forceinline__ void bf16x8_to_f32(const uint4 v, float f[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f[2 * j] = __uint_as_float(w[j] << 16);
        f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
    }
}

// Sum of the 8 bf16 values of a packed chunk, then across the 16 consecutive lanes that hold one
// (row, k-block); every lane of the wave must call this.
__device__ __forceinline__ float chunk_sum16(const uint4 v) {
    float f[8];
    bf16x8_to_f32(v, f);
    float s = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
    return vmx_row16_sum(s);
}

template <bool GLU>
__device__ __forceinline__ void int4_gemm_tile_store(const float *red, const uint16_t *__restrict__ bias, uint16_t *__restrict__ y,
                                                     float *__restrict__ partial, int64_t slice_stride, int M, int N, int nt, int ks,
                                                     const int32_t *omap);

// Sum of the 4 waves' accumulators and the store.  Lane holds rows m = 4h + r, column n = nt*16 + nl.
// omap (token-grouped experts): output row of block row m (LDS), instead of m itself.
template <bool GLU>
__device__ __forceinline__ void int4_gemm_epilogue(float *red, const vmx_f32x4 fin, const uint16_t *__restrict__ bias,
                                                   uint16_t *__restrict__ y, float *__restrict__ partial, int64_t slice_stride, int M,
                                                   int N, int nt, int ks, const int32_t *omap = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    *reinterpret_cast<float4 *>(red + (wave * 64 + lane) * 4) = make_float4(fin[0], fin[1], fin[2], fin[3]);
    __syncthreads();
    if (wave != 0) return;
    int4_gemm_tile_store<GLU>(red, bias, y, partial, slice_stride, M, N, nt, ks, omap);
}

// One wave: sum of the 4 waves' accumulators of one column tile (red [4][64][4]) and the store.
template <bool GLU>
__device__ __forceinline__ void int4_gemm_tile_store(const float *red, const uint16_t *__restrict__ bias, uint16_t *__restrict__ y,
                                                     float *__restrict__ partial, int64_t slice_stride, int M, int N, int nt, int ks,
                                                     const int32_t *omap) {
    const int lane = threadIdx.x & 63, nl = lane & 15, h = lane >> 4;
    float4 a = *reinterpret_cast<const float4 *>(red + lane * 4);
#pragma unroll
    for (int wv = 1; wv < 4; wv++) {
        const float4 b = *reinterpret_cast<const float4 *>(red + (wv * 64 + lane) * 4);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const float out4[4] = {a.x, a.y, a.z, a.w};
    const int n = nt * NT + nl;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int mrow = 4 * h + r;
        const int orow = omap ? omap[mrow] : mrow;
        if (GLU) {
            const float up = vmx_lane_xor<8>(out4[r]);
            if (mrow < M && nl < 8) {
                const float gg = vmx_bf16_to_f32(vmx_f32_to_bf16(out4[r]));
                const float uu = vmx_bf16_to_f32(vmx_f32_to_bf16(up));
                y[(int64_t)orow * (N >> 1) + nt * 8 + nl] = vmx_f32_to_bf16((gg / (1.0f + expf(-gg))) * uu);
            }
        } else if (mrow < M) {
            if (y) {
                float v = out4[r];
                if (bias) v += vmx_bf16_to_f32(bias[n]);
                y[(int64_t)orow * N + n] = vmx_f32_to_bf16(v);
            } else {
                partial[(int64_t)ks * slice_stride + (int64_t)orow * N + n] = out4[r];
            }
        }
    }
}

// GLU: the linear is a merged gate|up packed with glu_half = N/2 and runs unsplit; y is act[M][N/2] =
// bf16(silu(bf16(gate)) * bf16(up)), the rounding points of silu_and_mul on the bf16 linear output (activations.cu:148-164).
//
// NORM (M <= 2, K <= 8192, K % 64 == 0): `x` is the bf16 residual stream h [M][K] and the activations are its RMSNorm,
// computed here: x = bf16(h * rsqrt(mean(h^2) + eps) * w) (rms_norm_bf16 of the reference, layer_norm.cu), so that the norm
// is not a launch of its own.  The mean comes from ssq [M][K/16]: the sums of squares of each 16-column tile of h, left
// there by the kernel that wrote h (int4_gemm_resid_kernel); a workgroup normalises only its own k-slice.
template <int UPW, bool TAIL, bool GLU = false, bool NORM = false>
__device__ __forceinline__ void int4_gemm_body(const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ,
                                               const uint16_t *__restrict__ x, const uint16_t *__restrict__ bias,
                                               uint16_t *__restrict__ y, float *__restrict__ partial, int64_t slice_stride, int M,
                                               int N, int K, int group, int block, const uint16_t *__restrict__ norm_w = nullptr,
                                               float eps = 0.f, const float *__restrict__ ssq = nullptr) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int KS = 4 * UPW * KB;       // k-values per workgroup
    constexpr int XROW = KS + 8;           // bf16 per staged row (+16 B pad against bank conflicts)
    constexpr int CPR = KS / 8;            // 16-byte chunks per staged row (a multiple of 16)
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);                     // [M][XROW]
    float *sumx = reinterpret_cast<float *>(smem + (size_t)M * XROW * 2);  // [4*UPW][16]
    float *red = sumx + 4 * UPW * 16;                                      // [4][64][4] cross-wave sums
    uint32_t *szs = reinterpret_cast<uint32_t *>(red + 4 * 64 * 4);        // [4*UPW][16] scale|zero words of the slice

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = N / NT, kblocks = K / KB, G = K / group;
    const int nt = block % n_tiles, ks = block / n_tiles;
    const int kb_wg = ks * 4 * UPW;        // first k-block of this workgroup
    const int k0 = kb_wg * KB;
    const int nl = lane & 15, h = lane >> 4;
    const int total = M * CPR;

    // 1. request the first XCH*256 activation chunks and the scale|zero words (L2 hits) ...
    uint4 xv[XCH];
#pragma unroll
    for (int j = 0; j < XCH; j++) {
        const int c = tid + 256 * j;
        xv[j] = make_uint4(0, 0, 0, 0);
        if (c < total) {
            const int m = c / CPR, k = k0 + (c % CPR) * 8;
            if (!TAIL || k < K) xv[j] = *reinterpret_cast<const uint4 *>(x + (int64_t)m * K + k);
        }
    }
    uint4 nwv[XCH];
    float4 nss[2][2];   // NORM: per row, this lane's share of the K/16 per-tile sums of squares (K <= 8192)
    if constexpr (NORM) {
        const int nsq = K >> 4;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                // unconditional loads from clamped addresses (a branch here would put a wait in front of the weight requests);
                // the surplus ones are zeroed where they are used
                const int i = 4 * (lane + 64 * j);
                nss[m][j] = *reinterpret_cast<const float4 *>(ssq + min(m, M - 1) * nsq + min(i, nsq - 4));
            }
#pragma unroll
        for (int j = 0; j < XCH; j++) {
            const int c = tid + 256 * j, k = k0 + (c % CPR) * 8;
            nwv[j] = make_uint4(0, 0, 0, 0);
            if (c < total && (!TAIL || k < K)) nwv[j] = *reinterpret_cast<const uint4 *>(norm_w + k);
        }
    }
    uint32_t szv[(4 * UPW * 16 + 255) / 256];
#pragma unroll
    for (int j = 0; j < (4 * UPW * 16 + 255) / 256; j++) {
        const int i = tid + 256 * j, kb = kb_wg + (i >> 4);
        szv[j] = (i < 4 * UPW * 16 && (!TAIL || kb < kblocks)) ? SZ[((int64_t)nt * G + (kb * KB) / group) * 16 + (i & 15)] : 0u;
    }
    // 2. ... then this wave's weight units
    uint4 wreg[UPW];
#pragma unroll
    for (int u = 0; u < UPW; u++) {
        const int kb = kb_wg + wave * UPW + u;
        if (!TAIL || kb < kblocks) wreg[u] = W[((int64_t)nt * kblocks + kb) * 64 + lane];
        else wreg[u] = make_uint4(0, 0, 0, 0);
    }
    if constexpr (NORM) {   // mean of squares from the per-tile sums (every wave adds them up itself), then normalise the slice
        float inv[2];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const float4 a4 = nss[m][0], b4 = nss[m][1];
            const int nsq = K >> 4;
            const float ta = 4 * lane < nsq ? (a4.x + a4.y) + (a4.z + a4.w) : 0.f;
            const float tb = 4 * (lane + 64) < nsq ? (b4.x + b4.y) + (b4.z + b4.w) : 0.f;
            const float t = vmx_wave_sum(ta + tb);
            inv[m] = rsqrtf(t / (float)K + eps);
        }
#pragma unroll
        for (int j = 0; j < XCH; j++) {
            const int c = tid + 256 * j;
            float hv[8], wv[8];
            bf16x8_to_f32(xv[j], hv);
            bf16x8_to_f32(nwv[j], wv);
            const float iv = (c < total && c / CPR == 1) ? inv[1] : inv[0];
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
                pk[e] = (uint32_t)vmx_f32_to_bf16(hv[2 * e] * iv * wv[2 * e]) | ((uint32_t)vmx_f32_to_bf16(hv[2 * e + 1] * iv * wv[2 * e + 1]) << 16);
            xv[j] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
    // 3. stage activations (+ per-(k-block,row) sums) and scale|zero words in LDS
#pragma unroll
    for (int j = 0; j < XCH; j++) {
        const int c = tid + 256 * j;
        if (256 * j < total) {  // workgroup-uniform
            const bool live = c < total;
            const int m = live ? c / CPR : 0, cc = live ? c % CPR : 0;
            if (live) *reinterpret_cast<uint4 *>(xs + m * XROW + cc * 8) = xv[j];
            const float sm = chunk_sum16(xv[j]);  // 16 consecutive chunks = one (row, k-block)
            if (live && (tid & 15) == 0) sumx[(cc >> 4) * 16 + m] = sm;
        }
    }
    for (int c0 = XCH * 256; c0 < total; c0 += 256) {  // large M*K slices: the rest, behind the weight requests
        const int c = c0 + tid;
        const bool live = c < total;
        const int m = live ? c / CPR : 0, cc = live ? c % CPR : 0, k = k0 + cc * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (live && (!TAIL || k < K)) v = *reinterpret_cast<const uint4 *>(x + (int64_t)m * K + k);
        if (live) *reinterpret_cast<uint4 *>(xs + m * XROW + cc * 8) = v;
        const float sm = chunk_sum16(v);
        if (live && (tid & 15) == 0) sumx[(cc >> 4) * 16 + m] = sm;
    }
#pragma unroll
    for (int j = 0; j < (4 * UPW * 16 + 255) / 256; j++)
        if (tid + 256 * j < 4 * UPW * 16) szs[tid + 256 * j] = szv[j];
    __syncthreads();

    // 4. unpack + MFMA, unit by unit as the weights arrive.
    // A rows >= M replicate row M-1: their products land in output rows that are never stored.
    const uint16_t *xrow = xs + min(nl, M - 1) * XROW + 8 * h;
    vmx_f32x4 fin = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < UPW; u++) {
        const int kbl = wave * UPW + u;  // k-block inside the workgroup slice
        if (!TAIL || kb_wg + kbl < kblocks) {  // wave-uniform
            vmx_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const uint32_t w4[4] = {wreg[u].x, wreg[u].y, wreg[u].z, wreg[u].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t w = w4[i];
                uint4 bu;
                bu.x = (w & 0x000F000Fu) | 0x43004300u;
                bu.y = ((w >> 4) & 0x000F000Fu) | 0x43004300u;
                bu.z = ((w >> 8) & 0x000F000Fu) | 0x43004300u;
                bu.w = ((w >> 12) & 0x000F000Fu) | 0x43004300u;
                const uint4 au = *reinterpret_cast<const uint4 *>(xrow + kbl * KB + 32 * i);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, au), __builtin_bit_cast(vmx_bf16x8, bu),
                                                              acc, 0, 0, 0);
            }
            const uint32_t szw = szs[kbl * 16 + nl];
            const float s = __uint_as_float(szw << 16);
            const float zo = __uint_as_float(szw & 0xFFFF0000u);
            const float4 sx = *reinterpret_cast<const float4 *>(sumx + kbl * 16 + 4 * h);
            fin[0] += s * (acc[0] - zo * sx.x);
            fin[1] += s * (acc[1] - zo * sx.y);
            fin[2] += s * (acc[2] - zo * sx.z);
            fin[3] += s * (acc[3] - zo * sx.w);
        }
    }

    int4_gemm_epilogue<GLU>(red, fin, bias, y, partial, slice_stride, M, N, nt, ks);
}

// Several activation rows (M > 4) with a long k-slice: the body above stages M x (4*UPW*128) activations at once -- 131 KB
// at M = 16, UPW = 8, one workgroup per CU, and the launch is 3x (M = 8) to 8x (M = 16) its one-row time.  Here the slice is
// staged XS = 8 k-blocks at a time (M = 16: 33 KB, four workgroups per CU).  Every weight unit is still requested at kernel
// start; units are dealt to the waves round-robin (k-block 4u + wave), so each pass feeds all four waves.  Same arithmetic
// per k-block; only the order in which a wave meets its k-blocks differs from the body above.
template <int UPW, bool TAIL, bool GLU, int CT = 1>
__device__ __forceinline__ void int4_gemm_body_looped(const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ,
                                                      const uint16_t *__restrict__ x, const uint16_t *__restrict__ bias,
                                                      uint16_t *__restrict__ y, float *__restrict__ partial, int64_t slice_stride,
                                                      int M, int N, int K, int group, int block,
                                                      const int32_t *xmap = nullptr, const int32_t *omap = nullptr) {
    // xmap / omap (token-grouped experts, LDS): activation row and output row of block row m
    // CT: column tiles per workgroup.  A staged activation pass (M x 1024 k, 32 KB at 16 rows) serves one 1 KB weight unit per
    // k-block and tile: with one tile per workgroup a 16-row launch reads four times as many activation bytes (from L2) as
    // weight bytes -- 200 MB against 52 MB on the gate|up shape; CT tiles share the staged rows and each A fragment read from LDS.
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int XS = 4 * UPW < 8 ? 4 * UPW : 8;   // k-blocks staged per pass
    static_assert((4 * UPW) % XS == 0 && XS % 4 == 0, "slice must be whole passes, a pass whole rounds of the 4 waves");
    constexpr int PASSES = 4 * UPW / XS, UPP = XS / 4;   // units per wave per pass
    constexpr int XROW = XS * KB + 8, CPR = XS * KB / 8;
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);                     // [M][XROW]
    float *sumx = reinterpret_cast<float *>(smem + (size_t)M * XROW * 2);  // [4*UPW][16]
    float *red = sumx + 4 * UPW * 16;                                      // [CT][4][64][4]
    uint32_t *szs = reinterpret_cast<uint32_t *>(red + CT * 4 * 64 * 4);   // [CT][4*UPW][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = N / NT, kblocks = K / KB, G = K / group;
    const int n_groups = n_tiles / CT;
    const int nt0 = (block % n_groups) * CT, ks = block / n_groups;
    const int kb_wg = ks * 4 * UPW;
    const int nl = lane & 15, h = lane >> 4;
    const int total = M * CPR;

    constexpr int NSZ = (CT * 4 * UPW * 16 + 255) / 256;
    uint32_t szv[NSZ];
#pragma unroll
    for (int j = 0; j < NSZ; j++) {
        const int i = tid + 256 * j, t = i / (4 * UPW * 16), r = i % (4 * UPW * 16), kb = kb_wg + (r >> 4);
        szv[j] = (i < CT * 4 * UPW * 16 && (!TAIL || kb < kblocks)) ? SZ[((int64_t)(nt0 + t) * G + (kb * KB) / group) * 16 + (r & 15)] : 0u;
    }
    uint4 wreg[CT][UPW];   // unit u of this wave = k-block 4u + wave of the slice
#pragma unroll
    for (int u = 0; u < UPW; u++) {
        const int kb = kb_wg + 4 * u + wave;
#pragma unroll
        for (int t = 0; t < CT; t++) {
            if (!TAIL || kb < kblocks) wreg[t][u] = W[((int64_t)(nt0 + t) * kblocks + kb) * 64 + lane];
            else wreg[t][u] = make_uint4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NSZ; j++)
        if (tid + 256 * j < CT * 4 * UPW * 16) szs[tid + 256 * j] = szv[j];

    const uint16_t *xrow = xs + min(nl, M - 1) * XROW + 8 * h;
    vmx_f32x4 fin[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) fin[t] = vmx_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PASSES; p++) {
        if (p) __syncthreads();   // the previous pass has been read by every wave
        const int kp0 = (kb_wg + p * XS) * KB;
        for (int c0 = 0; c0 < total; c0 += 1024) {   // four chunks per thread requested before the first is used (a round trip each otherwise)
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = c0 + 256 * j + tid;
                const bool live = c < total;
                const int m = live ? c / CPR : 0, cc = live ? c % CPR : 0, k = kp0 + cc * 8;
                v[j] = make_uint4(0, 0, 0, 0);
                if (live && (!TAIL || k < K)) v[j] = *reinterpret_cast<const uint4 *>(x + (int64_t)(xmap ? xmap[m] : m) * K + k);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (c0 + 256 * j < total) {   // workgroup-uniform
                    const int c = c0 + 256 * j + tid;
                    const bool live = c < total;
                    const int m = live ? c / CPR : 0, cc = live ? c % CPR : 0;
                    if (live) *reinterpret_cast<uint4 *>(xs + m * XROW + cc * 8) = v[j];
                    const float sm = chunk_sum16(v[j]);
                    if (live && (tid & 15) == 0) sumx[(p * XS + (cc >> 4)) * 16 + m] = sm;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < UPP; q++) {
            const int u = p * UPP + q, kbl = 4 * u + wave, kin = kbl - p * XS;   // k-block inside the slice / inside the pass
            if (!TAIL || kb_wg + kbl < kblocks) {  // wave-uniform
                vmx_f32x4 acc[CT];
#pragma unroll
                for (int t = 0; t < CT; t++) acc[t] = vmx_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint4 au = *reinterpret_cast<const uint4 *>(xrow + kin * KB + 32 * i);
#pragma unroll
                    for (int t = 0; t < CT; t++) {
                        const uint32_t w = i == 0 ? wreg[t][u].x : i == 1 ? wreg[t][u].y : i == 2 ? wreg[t][u].z : wreg[t][u].w;
                        uint4 bu;
                        bu.x = (w & 0x000F000Fu) | 0x43004300u;
                        bu.y = ((w >> 4) & 0x000F000Fu) | 0x43004300u;
                        bu.z = ((w >> 8) & 0x000F000Fu) | 0x43004300u;
                        bu.w = ((w >> 12) & 0x000F000Fu) | 0x43004300u;
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, au), __builtin_bit_cast(vmx_bf16x8, bu),
                                                                         acc[t], 0, 0, 0);
                    }
                }
                const float4 sx = *reinterpret_cast<const float4 *>(sumx + kbl * 16 + 4 * h);
#pragma unroll
                for (int t = 0; t < CT; t++) {
                    const uint32_t szw = szs[(t * 4 * UPW + kbl) * 16 + nl];
                    const float s = __uint_as_float(szw << 16);
                    const float zo = __uint_as_float(szw & 0xFFFF0000u);
                    fin[t][0] += s * (acc[t][0] - zo * sx.x);
                    fin[t][1] += s * (acc[t][1] - zo * sx.y);
                    fin[t][2] += s * (acc[t][2] - zo * sx.z);
                    fin[t][3] += s * (acc[t][3] - zo * sx.w);
                }
            }
        }
    }
    if constexpr (CT == 1) {
        int4_gemm_epilogue<GLU>(red, fin[0], bias, y, partial, slice_stride, M, N, nt0, ks, omap);
    } else {   // every wave leaves its sums of all CT tiles; wave t < CT then adds up and stores tile t
#pragma unroll
        for (int t = 0; t < CT; t++)
            *reinterpret_cast<float4 *>(red + ((t * 4 + wave) * 64 + lane) * 4) = make_float4(fin[t][0], fin[t][1], fin[t][2], fin[t][3]);
        __syncthreads();
        for (int t = wave; t < CT; t += 4)
            int4_gemm_tile_store<GLU>(red + t * 4 * 64 * 4, bias, y, partial, slice_stride, M, N, nt0 + t, ks, omap);
    }
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression in its body
template <int... I, class F> __device__ __forceinline__ void ring_static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void ring_static_for(F &&f) { ring_static_for_impl(std::make_integer_sequence<int, N>{}, f); }
#ifndef VMX_RING_D
#define VMX_RING_D 3
#endif
constexpr int RING_D = VMX_RING_D;   // passes a request runs ahead of its use
// 5..16 activation rows, ring form.  The looped body above waits for an L2 round trip per staging batch (8 of them on a
// 4096-deep slice at 16 rows) with its matrix work in between: a wave spends most of its life waiting.  Here
//   * a pass is 4 k-blocks (one per wave); its 16 x 512 activation tile is requested D passes ahead into registers (16 per
//     lane and pass) and written to one of two 16 KB LDS stages one pass ahead, so the round trips overlap each other and
//     the matrix work; one barrier per pass;
//   * weights and scale|zero words are requested D passes ahead as well; the loop is fully unrolled, every request is a plain
//     load, so the compiler's per-use wait counts are exact (loads retire in issue order);
//   * a workgroup owns CT column tiles that share every staged tile and every A fragment;
//   * a row is 1 KiB unpadded; position p of row r holds the row's 16-byte chunk p ^ (r & 15), so the 16 rows a fragment read
//     touches lie in 16 different bank groups;
//   * the row sums of the zero-point term come from the matrix cores (the tile times a matrix of ones).
// K % (512 * UPW) == 0, (N / 16) % CT == 0; rows past M replicate row M - 1 (never stored).
// xmap / omap (token-grouped experts, LDS): activation row and output row of block row m.
template <int UPW, bool GLU, int CT>
__device__ __forceinline__ void int4_gemm_ring_body(const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ,
                                                    const uint16_t *__restrict__ x, const uint16_t *__restrict__ bias,
                                                    uint16_t *__restrict__ y, float *__restrict__ partial, int64_t slice_stride, int M,
                                                    int N, int K, int group, int block, const int32_t *xmap = nullptr,
                                                    const int32_t *omap = nullptr) {
    constexpr int PASSES = UPW, D = RING_D, ROWB = 4 * KB * 2;   // bytes per staged row
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *xs = smem;                                        // [2][16][ROWB]
    float *red = reinterpret_cast<float *>(smem);                    // [CT][4][64][4], after the last pass
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, h = lane >> 4;
    const int n_tiles = N / NT, kblocks = K / KB, G = K / group;
    const int n_groups = n_tiles / CT;
    const int nt0 = (block % n_groups) * CT, ks = block / n_groups;
    const int kb_wg = ks * 4 * UPW;

    // staging: request j of a wave is row 4 j + wave of the tile, lane l its chunk l.  Every address is a wave-uniform base
    // plus a 32-bit lane offset.
    const char *xrow_u[4];
    uint32_t xdst[4];   // byte offset of the lane's chunk inside a stage
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = 4 * j + wave, rsrc = min(row, M - 1);
        xrow_u[j] = reinterpret_cast<const char *>(x + (int64_t)(xmap ? __builtin_amdgcn_readfirstlane(xmap[rsrc]) : rsrc) * K + (int64_t)kb_wg * KB);
        xdst[j] = (uint32_t)(row * ROWB + ((lane ^ (row & 15)) << 4));
    }
    const char *w_u[CT];
    const char *z_u[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) {
        w_u[t] = reinterpret_cast<const char *>(W + ((int64_t)(nt0 + t) * kblocks + kb_wg + wave) * 64);
        z_u[t] = reinterpret_cast<const char *>(SZ + (int64_t)(nt0 + t) * G * 16);
    }
    const uint32_t woff = (uint32_t)lane << 4, zoff = (uint32_t)nl << 2;
    typedef unsigned int ring_u32x4 __attribute__((ext_vector_type(4)));
    ring_u32x4 tq[D][4];
    uint4 wq[D][CT];
    uint32_t zq[D][CT];
    auto issue = [&](auto pc) {
        constexpr int p = decltype(pc)::value, r = p % D;
#pragma unroll
        for (int j = 0; j < 4; j++) tq[r][j] = *reinterpret_cast<const ring_u32x4 *>(xrow_u[j] + (size_t)p * ROWB + woff);
#pragma unroll
        for (int t = 0; t < CT; t++) {
            wq[r][t] = ld_weights(reinterpret_cast<const uint4 *>(w_u[t] + (size_t)p * 4 * 64 * 16 + woff));
            zq[r][t] = *reinterpret_cast<const uint32_t *>(z_u[t] + (size_t)((((kb_wg + 4 * p + wave) * KB) / group) * 16) * 4 + zoff);
        }
    };
    auto write_tile = [&](auto pc) {
        constexpr int p = decltype(pc)::value, r = p % D;
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<ring_u32x4 *>(xs + (p & 1) * 16 * ROWB + xdst[j]) = tq[r][j];
    };
    ring_static_for<(D < PASSES ? D : PASSES)>([&](auto pc) { issue(pc); });
    write_tile(std::integral_constant<int, 0>{});
    const int rowc = min(nl, M - 1);
    uint32_t fa[4];   // byte offsets of this lane's four A fragments inside a stage
#pragma unroll
    for (int i = 0; i < 4; i++) fa[i] = (uint32_t)rowc * ROWB + (uint32_t)(((wave * 16 + 4 * i + h) ^ rowc) << 4);
    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    vmx_f32x4 fin[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) fin[t] = vmx_f32x4{0.f, 0.f, 0.f, 0.f};
    ring_static_for<PASSES>([&](auto pc) {
        constexpr int p = decltype(pc)::value, r = p % D;
        __syncthreads();   // tile p is in its stage; every wave has left pass p - 1 (the other stage)
        vmx_f32x4 acc[CT], acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < CT; t++) acc[t] = vmx_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 au = *reinterpret_cast<const uint4 *>(xs + (p & 1) * 16 * ROWB + fa[i]);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, au), __builtin_bit_cast(vmx_bf16x8, ones), acc1, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < CT; t++) {
                const uint4 wv = wq[r][t];
                const uint32_t w = i == 0 ? wv.x : i == 1 ? wv.y : i == 2 ? wv.z : wv.w;
                uint4 bu;
                bu.x = (w & 0x000F000Fu) | 0x43004300u;
                bu.y = ((w >> 4) & 0x000F000Fu) | 0x43004300u;
                bu.z = ((w >> 8) & 0x000F000Fu) | 0x43004300u;
                bu.w = ((w >> 12) & 0x000F000Fu) | 0x43004300u;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, au), __builtin_bit_cast(vmx_bf16x8, bu), acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < CT; t++) {
            const uint32_t szw = zq[r][t];
            const float sc = __uint_as_float(szw << 16);
            const float zo = __uint_as_float(szw & 0xFFFF0000u);
            fin[t][0] += sc * (acc[t][0] - zo * acc1[0]);
            fin[t][1] += sc * (acc[t][1] - zo * acc1[1]);
            fin[t][2] += sc * (acc[t][2] - zo * acc1[2]);
            fin[t][3] += sc * (acc[t][3] - zo * acc1[3]);
        }
        if constexpr (p + 1 < PASSES) write_tile(std::integral_constant<int, p + 1>{});   // requested D - 1 passes ago; into the stage every wave left before the barrier above
        if constexpr (p + D < PASSES) issue(std::integral_constant<int, p + D>{});        // into the registers this pass and that write have just used
    });
    __syncthreads();   // the last tile has been read: the stages become the waves' exchange buffer
#pragma unroll
    for (int t = 0; t < CT; t++)
        *reinterpret_cast<float4 *>(red + ((t * 4 + wave) * 64 + lane) * 4) = make_float4(fin[t][0], fin[t][1], fin[t][2], fin[t][3]);
    __syncthreads();
    for (int t = wave; t < CT; t += 4)
        int4_gemm_tile_store<GLU>(red + t * 4 * 64 * 4, bias, y, partial, slice_stride, M, N, nt0 + t, ks, omap);
}

template <int UPW, bool GLU, int CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void int4_gemm_ring_kernel(
    const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ, const uint16_t *__restrict__ x, const uint16_t *__restrict__ bias,
    uint16_t *__restrict__ y, float *__restrict__ partial, int M, int N, int K, int group, int slice_rows) {
    int4_gemm_ring_body<UPW, GLU, CT>(W, SZ, x, bias, y, partial, (int64_t)slice_rows * N, M, N, K, group, (int)blockIdx.x);
}

template <int UPW, bool TAIL, int CT = 1>
__global__ __launch_bounds__(256) void int4_gemm_looped_kernel(const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ,
                                                                const uint16_t *__restrict__ x, const uint16_t *__restrict__ bias,
                                                                uint16_t *__restrict__ y, float *__restrict__ partial, int M, int N,
                                                                int K, int group, int slice_rows) {
    int4_gemm_body_looped<UPW, TAIL, false, CT>(W, SZ, x, bias, y, partial, (int64_t)slice_rows * N, M, N, K, group, blockIdx.x);
}

template <int UPW, bool TAIL, int CT = 1>
__global__ __launch_bounds__(256) void int4_gemm_glu_looped_kernel(const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ,
                                                                    const uint16_t *__restrict__ x, uint16_t *__restrict__ act, int M,
                                                                    int N, int K, int group) {
    int4_gemm_body_looped<UPW, TAIL, true, CT>(W, SZ, x, nullptr, act, nullptr, 0, M, N, K, group, blockIdx.x);
}

template <int UPW, bool TAIL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WIDE_WAVES_PER_EU))) void int4_gemm_kernel(const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ,
                                                         const uint16_t *__restrict__ x, const uint16_t *__restrict__ bias,
                                                         uint16_t *__restrict__ y, float *__restrict__ partial, int M, int N, int K,
                                                         int group, int slice_rows) {
    // slice_rows: rows of one split-K slice of `partial` (>= M when this launch covers 16 of a larger call's rows)
    int4_gemm_body<UPW, TAIL>(W, SZ, x, bias, y, partial, (int64_t)slice_rows * N, M, N, K, group, blockIdx.x);
}

template <int UPW, bool TAIL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WIDE_WAVES_PER_EU))) void int4_gemm_glu_kernel(const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ,
                                                             const uint16_t *__restrict__ x, uint16_t *__restrict__ act, int M, int N,
                                                             int K, int group) {
    int4_gemm_body<UPW, TAIL, true>(W, SZ, x, nullptr, act, nullptr, 0, M, N, K, group, blockIdx.x);
}

// ---------------------------------------------------------------- decode rows (M <= 2): wide workgroups, norm in the prologue
// Every workgroup of the kernels above fetches (and here also normalises) its whole activation slice, so with one
// 16-column tile per workgroup the activations cross the L2 -> CU path once per tile: 1536 times for a 24576-wide
// gate|up.  Here a workgroup of NWV waves owns NTW adjacent column tiles over a k-slice of NWV*UPW k-blocks; wave w holds
// the units (tile t, k-block w*UPW + u) -- NTW*UPW of them, all requested at kernel start -- and one A fragment read from
// LDS feeds NTW matrix instructions.  The launcher picks (NWV, UPW, NTW, split-K) so that the workgroup count is a whole
// multiple of the CU count where the shape allows it.  No bounds checks: K is a whole number of slices, N of NTW tiles.
// The activations are the RMSNorm of the residual stream h (see NORM above).  Output: f32 split-K slices, or with GLU
// (one slice) the gated activation.
#ifdef VMX_WIDE_STAMPS
// build-time measurement variant: per-wave clock stamps of the last wide-kernel launch (8 u64 per wave, up to 8192 waves)
__device__ unsigned long long vmx_wide_stamp_buf[8192 * 8];
#define WIDE_STAMP(i)                                                                                          \
    do {                                                                                                       \
        unsigned long long t__;                                                                                \
        if ((i) == 0 || (i) == 7) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); \
        else asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                        \
        if ((threadIdx.x & 63) == 0 && blockIdx.x * NWV + (threadIdx.x >> 6) < 8192)                           \
            vmx_wide_stamp_buf[(blockIdx.x * NWV + (threadIdx.x >> 6)) * 8 + (i)] = t__;                       \
    } while (0)
#else
#define WIDE_STAMP(i) do { } while (0)
#endif

template <int NWV, int UPW, int NTW, bool GLU, int MR, bool DOT2 = false>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(MR == 1 ? WIDE_WAVES_PER_EU : (MR == 2 ? 4 : 3)))) void int4_gemm_wide_kernel(const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ,
                                                                   const uint16_t *__restrict__ hres, const uint16_t *__restrict__ norm_w,
                                                                   float eps, const float *__restrict__ ssq, float *__restrict__ partial,
                                                                   uint16_t *__restrict__ act, int N, int K, int group) {
    // MR = rows (1..4), compiled in: every load instruction a wave issues costs the CU's memory pipeline time, used or not
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int M = MR;
    constexpr int TH = 64 * NWV, KBS = NWV * UPW, KS = KBS * KB, XROW = KS + 8, CPR = KS / 8;
    constexpr int XCF = (MR * CPR + TH - 1) / TH, SZW = (NTW * KBS * 16 + TH - 1) / TH;
    // LDS is kept small on purpose: the number of workgroups a CU holds at once is what keeps its loads flowing
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);                                           // [MR][XROW]
    float *sumx = reinterpret_cast<float *>(smem + (((size_t)MR * XROW * 2 + 15) & ~(size_t)15));  // [KBS][4 rows]
    float *red = sumx + KBS * 4;                                                                 // [NTW][NWV][16 columns][4 rows]
    uint32_t *szs = reinterpret_cast<uint32_t *>(red + NTW * NWV * 64);                          // [NTW][KBS][16]
    float *invs = reinterpret_cast<float *>(szs + NTW * KBS * 16);                               // [4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 15, hh = lane >> 4;
    const int groups = (N / NT) / NTW, kblocks = K / KB, G = K / group;
    const int nt0 = ((int)blockIdx.x % groups) * NTW, ks = (int)blockIdx.x / groups;
    const int kb_wg = ks * KBS, k0 = kb_wg * KB, nsq = K >> 4;
    constexpr int total = MR * CPR;
    WIDE_STAMP(0);
    WIDE_STAMP(1);

    // The mean of squares is the business of wave 0 alone (one or two 1 KiB loads per row): the 256..512 tile sums are the
    // same few cache lines for every workgroup of the launch, and fetched by every wave they cost more than the barrier
    // that shares the result.
    float4 nss[MR][2];
    if (wave == 0) {
#pragma unroll
        for (int m = 0; m < MR; m++) {
            nss[m][0] = *reinterpret_cast<const float4 *>(ssq + m * nsq + min(4 * lane, nsq - 4));
            nss[m][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nsq > 256) nss[m][1] = *reinterpret_cast<const float4 *>(ssq + m * nsq + min(4 * (lane + 64), nsq - 4));
        }
    }
    uint4 xv[XCF], nwv[XCF];
#pragma unroll
    for (int j = 0; j < XCF; j++) {
        const int c = min(tid + TH * j, total - 1), k = k0 + (c % CPR) * 8;
        xv[j] = *reinterpret_cast<const uint4 *>(hres + (int64_t)(c / CPR) * K + k);
        nwv[j] = *reinterpret_cast<const uint4 *>(norm_w + k);
    }
    uint32_t szv[SZW];
#pragma unroll
    for (int j = 0; j < SZW; j++) {
        const int i = min(tid + TH * j, NTW * KBS * 16 - 1), t = i / (KBS * 16), kb = kb_wg + (i / 16) % KBS;
        szv[j] = SZ[((int64_t)(nt0 + t) * G + (kb * KB) / group) * 16 + (i & 15)];
    }
    uint4 wreg[NTW][UPW];
#pragma unroll
    for (int u = 0; u < UPW; u++)
#pragma unroll
        for (int t = 0; t < NTW; t++) wreg[t][u] = ld_weights(&W[((int64_t)(nt0 + t) * kblocks + kb_wg + wave * UPW + u) * 64 + lane]);
    WIDE_STAMP(2);

    if (wave == 0) {
#pragma unroll
        for (int m = 0; m < MR; m++) {
            const float4 a4 = nss[m][0], b4 = nss[m][1];
            const float ta = 4 * lane < nsq ? (a4.x + a4.y) + (a4.z + a4.w) : 0.f;
            const float tb = 4 * (lane + 64) < nsq ? (b4.x + b4.y) + (b4.z + b4.w) : 0.f;
            const float iv = rsqrtf(vmx_wave_sum(ta + tb) / (float)K + eps);
            if (lane == 0) invs[m] = iv;
        }
    }
    __syncthreads();
    float inv[MR];
#pragma unroll
    for (int m = 0; m < MR; m++) inv[m] = invs[m];
#pragma unroll
    for (int j = 0; j < XCF; j++) {
        const int c = tid + TH * j;
        const bool live = c < total;
        const int m = live ? c / CPR : 0, cc = live ? c % CPR : 0;
        float hv[8], wv[8];
        bf16x8_to_f32(xv[j], hv);
        bf16x8_to_f32(nwv[j], wv);
        float iv = inv[0];
#pragma unroll
        for (int r = 1; r < MR; r++) iv = m == r ? inv[r] : iv;
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
            pk[e] = (uint32_t)vmx_f32_to_bf16(hv[2 * e] * iv * wv[2 * e]) | ((uint32_t)vmx_f32_to_bf16(hv[2 * e + 1] * iv * wv[2 * e + 1]) << 16);
        const uint4 xn = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        if (live) *reinterpret_cast<uint4 *>(xs + m * XROW + cc * 8) = xn;
        if constexpr (DOT2) {   // one row: sums over the chunks h, 4 + h, 8 + h, 12 + h of a k-block (what lane group h multiplies)
            float f8[8];
            bf16x8_to_f32(xn, f8);
            float sh = ((f8[0] + f8[1]) + (f8[2] + f8[3])) + ((f8[4] + f8[5]) + (f8[6] + f8[7]));
            sh += vmx_lane_xor<4>(sh);
            sh += vmx_lane_xor<8>(sh);
            if (live && (tid & 15) < 4) sumx[(cc >> 4) * 4 + (tid & 3)] = sh;
        } else {
            const float sm = chunk_sum16(xn);   // 16 consecutive chunks = one (row, k-block)
            if (live && (tid & 15) == 0) sumx[(cc >> 4) * 4 + m] = sm;
        }
    }
#pragma unroll
    for (int j = 0; j < SZW; j++)
        if (tid + TH * j < NTW * KBS * 16) szs[tid + TH * j] = szv[j];
    __syncthreads();
    WIDE_STAMP(3);

    if constexpr (DOT2) {
        // The same sum on the vector ALU: lane (column nl, group hh) multiplies its 32 weights of a unit with v_dot2c_f32_bf16
        // (two bf16 products per instruction, f32 accumulate); the four groups of a column are added at the end.
        static_assert(MR == 1, "the v_dot2 form is for one activation row");
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const uint16_t *xr = xs + 8 * hh;
        float fsum[NTW];
#pragma unroll
        for (int t = 0; t < NTW; t++) fsum[t] = 0.f;
#pragma unroll
        for (int u = 0; u < UPW; u++) {
            const int kbl = wave * UPW + u;
            uint4 xa[4];
#pragma unroll
            for (int i = 0; i < 4; i++) xa[i] = *reinterpret_cast<const uint4 *>(xr + kbl * KB + 32 * i);
            const float sxh = sumx[kbl * 4 + hh];
#pragma unroll
            for (int t = 0; t < NTW; t++) {
                float acc = 0.f;
                const uint32_t w4[4] = {wreg[t][u].x, wreg[t][u].y, wreg[t][u].z, wreg[t][u].w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t xw[4] = {xa[i].x, xa[i].y, xa[i].z, xa[i].w};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t b = ((w4[i] >> (4 * q)) & 0x000F000Fu) | 0x43004300u;   // weights 2q, 2q + 1 as bf16 128 + nib
                        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xw[q]), __builtin_bit_cast(bf16x2_t, b), acc, false);
                    }
                }
                const uint32_t szw = szs[(t * KBS + kbl) * 16 + nl];
                fsum[t] += __uint_as_float(szw << 16) * (acc - __uint_as_float(szw & 0xFFFF0000u) * sxh);
            }
        }
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            fsum[t] += vmx_lane_xor<16>(fsum[t]);
            fsum[t] += vmx_lane_xor<32>(fsum[t]);
            if (hh == 0) red[((t * NWV + wave) * 16 + nl) * 4] = fsum[t];
        }
    }
    const uint16_t *xrow = xs + min(nl, M - 1) * XROW + 8 * hh;
    vmx_f32x4 fin[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) fin[t] = vmx_f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (!DOT2) {
#pragma unroll
    for (int u = 0; u < UPW; u++) {
        const int kbl = wave * UPW + u;
        uint4 au[4];
#pragma unroll
        for (int i = 0; i < 4; i++) au[i] = *reinterpret_cast<const uint4 *>(xrow + kbl * KB + 32 * i);
        const float4 sx = *reinterpret_cast<const float4 *>(sumx + kbl * 4);   // rows 0..3; the other row groups' outputs are never stored
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            vmx_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const uint32_t w4[4] = {wreg[t][u].x, wreg[t][u].y, wreg[t][u].z, wreg[t][u].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t w = w4[i];
                uint4 bu;
                bu.x = (w & 0x000F000Fu) | 0x43004300u;
                bu.y = ((w >> 4) & 0x000F000Fu) | 0x43004300u;
                bu.z = ((w >> 8) & 0x000F000Fu) | 0x43004300u;
                bu.w = ((w >> 12) & 0x000F000Fu) | 0x43004300u;
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, au[i]), __builtin_bit_cast(vmx_bf16x8, bu), acc, 0, 0, 0);
            }
            const uint32_t szw = szs[(t * KBS + kbl) * 16 + nl];
            const float sc = __uint_as_float(szw << 16);
            const float zo = __uint_as_float(szw & 0xFFFF0000u);
            fin[t][0] += sc * (acc[0] - zo * sx.x);
            fin[t][1] += sc * (acc[1] - zo * sx.y);
            fin[t][2] += sc * (acc[2] - zo * sx.z);
            fin[t][3] += sc * (acc[3] - zo * sx.w);
        }
    }
    if (hh == 0) {   // rows 0..3 of the MFMA tile (M <= 2)
#pragma unroll
        for (int t = 0; t < NTW; t++)
            *reinterpret_cast<float4 *>(red + ((t * NWV + wave) * 16 + nl) * 4) = make_float4(fin[t][0], fin[t][1], fin[t][2], fin[t][3]);
    }
    }
    WIDE_STAMP(4);
    __syncthreads();
    // thread = (tile t, row m, column c): sum over the waves in wave order
    constexpr int RSH = MR > 2 ? 6 : 5;   // 32 (two rows) or 64 (four rows) threads per tile
    for (int i = tid; i < (NTW << RSH); i += TH) {
        const int t = i >> RSH, m = (i >> 4) & (MR > 2 ? 3 : 1), c = i & 15;
        if (m >= M) continue;
        if (GLU) {
            if (c >= 8) continue;
            float g = 0.f, up = 0.f;
#pragma unroll
            for (int wv = 0; wv < NWV; wv++) {
                g += red[((t * NWV + wv) * 16 + c) * 4 + m];
                up += red[((t * NWV + wv) * 16 + c + 8) * 4 + m];
            }
            const float gg = vmx_bf16_to_f32(vmx_f32_to_bf16(g)), uu = vmx_bf16_to_f32(vmx_f32_to_bf16(up));
            act[(int64_t)m * (N >> 1) + (nt0 + t) * 8 + c] = vmx_f32_to_bf16((gg / (1.0f + expf(-gg))) * uu);
        } else {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < NWV; wv++) v += red[((t * NWV + wv) * 16 + c) * 4 + m];
            partial[((int64_t)ks * M + m) * N + (nt0 + t) * NT + c] = v;
        }
    }
    WIDE_STAMP(7);
}

// ---------------------------------------------------------------- whole-K workgroups with the residual add in the epilogue
// The two linears whose output goes back into the residual stream (o_proj, down_proj), for M <= 2 rows: one workgroup of 16
// waves owns a 16-column tile over ALL of K (wave w the k-blocks [w*UPW, (w+1)*UPW)), so the sum needs no split-K slices
// and the epilogue can finish the job of the reference's separate add: h = bf16(h + bf16(y)) (the linear's bf16 output,
// then add_bf16).  N/16 workgroups of 1024 threads: the launch keeps the whole matrix in flight like the kernel above.
//
// PM > 0: the activations are the decode-attention output still split into P <= PM partitions per (row, head)
// (split_decode_attn_kernel): tmp [M][K/HD][P][HD] f32 with max logit / sum [M][K/HD][P]; staging merges them,
//   x = bf16( sum_p tmp_p w_p / (sum_p w_p + 1e-6) ),  w_p = sum_p * exp(max_p - max over p)   (paged_attention.cu:840-953),
// which is what the separate merge kernel writes.
struct AttnMergeArgs {
    const float *tmp, *mx, *sm;
    int P, HD;
};

template <int UPW, int PM, int MRR = 2>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(1, 4))) void int4_gemm_resid_kernel(
    const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ, const uint16_t *__restrict__ x, uint16_t *__restrict__ h,
    float *__restrict__ ssq, int M, int N, int K, int group, AttnMergeArgs am) {
    // K == 16 * UPW * 128 exactly (checked by the launcher): no bounds checks, so every load below is unconditional and
    // issued back to back (surplus threads re-read a valid address and drop the value)
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NWV = 16, KBLK = NWV * UPW, KK = KBLK * KB, XROW = KK + 8, CPR = KK / 8;
    constexpr int XCF = (MRR * CPR + 1023) / 1024;   // chunks per thread for MRR (2 or 4) rows
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);                                   // [MRR][XROW]
    float *sumx = reinterpret_cast<float *>(smem + (((size_t)MRR * XROW * 2 + 15) & ~(size_t)15));   // [KBLK][16]
    float *red = sumx + KBLK * 16;                                                        // [16 waves][16 columns][4 rows]
    uint32_t *szs = reinterpret_cast<uint32_t *>(red + NWV * 16 * 4);                     // [KBLK][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x, nl = lane & 15, hh = lane >> 4;
    const int G = K / group, total = M * CPR;

    // the residual values of the epilogue, requested first: thread (row m, column c) for tid < 16 * M
    const int64_t hidx = (int64_t)min(tid >> 4, M - 1) * N + nt * NT + (tid & 15);
    const uint16_t hold = h[hidx];
    uint4 xv[XCF];
    constexpr int PMA = PM > 0 ? PM : 1;
    float4 pa[PM > 0 ? XCF : 1][PMA], pb[PM > 0 ? XCF : 1][PMA];
    float pmx[PM > 0 ? XCF : 1][PMA], psm[PM > 0 ? XCF : 1][PMA];
    if constexpr (PM == 0) {
#pragma unroll
        for (int j = 0; j < XCF; j++) {
            const int c = min(tid + 1024 * j, total - 1);
            xv[j] = *reinterpret_cast<const uint4 *>(x + (int64_t)(c / CPR) * K + (c % CPR) * 8);
        }
    } else {
#pragma unroll
        for (int j = 0; j < XCF; j++) {
            const int c = min(tid + 1024 * j, total - 1), k = (c % CPR) * 8;
            const int64_t row = ((int64_t)(c / CPR) * (K / am.HD) + k / am.HD) * am.P;   // (activation row, head)
#pragma unroll
            for (int p = 0; p < PM; p++) {   // partitions past P re-read the last one; their weight is zeroed below
                const int64_t rp = row + min(p, am.P - 1);
                const float *src = am.tmp + rp * am.HD + k % am.HD;
                pa[j][p] = *reinterpret_cast<const float4 *>(src);
                pb[j][p] = *reinterpret_cast<const float4 *>(src + 4);
                pmx[j][p] = am.mx[rp];
                psm[j][p] = am.sm[rp];
            }
        }
    }
    constexpr int SZW = (KBLK * 16 + 1023) / 1024;
    uint32_t szv[SZW];
#pragma unroll
    for (int j = 0; j < SZW; j++) {
        const int i = min(tid + 1024 * j, KBLK * 16 - 1), kb = i >> 4;
        szv[j] = SZ[((int64_t)nt * G + (kb * KB) / group) * 16 + (i & 15)];
    }
    uint4 wreg[UPW];
#pragma unroll
    for (int u = 0; u < UPW; u++) wreg[u] = ld_weights(&W[((int64_t)nt * KBLK + wave * UPW + u) * 64 + lane]);

    if constexpr (PM > 0) {
#pragma unroll
        for (int j = 0; j < XCF; j++) {
            float gmax = -3.402823466e+38f;
#pragma unroll
            for (int p = 0; p < PM; p++) gmax = fmaxf(gmax, p < am.P ? pmx[j][p] : -3.402823466e+38f);
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gsum = 0.f;
#pragma unroll
            for (int p = 0; p < PM; p++) {
                const float wt = p < am.P ? psm[j][p] * __expf(pmx[j][p] - gmax) : 0.f;
                gsum += wt;
                acc[0] += pa[j][p].x * wt; acc[1] += pa[j][p].y * wt; acc[2] += pa[j][p].z * wt; acc[3] += pa[j][p].w * wt;
                acc[4] += pb[j][p].x * wt; acc[5] += pb[j][p].y * wt; acc[6] += pb[j][p].z * wt; acc[7] += pb[j][p].w * wt;
            }
            const float inv = 1.0f / (gsum + 1e-6f);
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
                pk[e] = (uint32_t)vmx_f32_to_bf16(acc[2 * e] * inv) | ((uint32_t)vmx_f32_to_bf16(acc[2 * e + 1] * inv) << 16);
            xv[j] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
#pragma unroll
    for (int j = 0; j < XCF; j++) {
        const int c = tid + 1024 * j;
        const bool live = c < total;
        const int m = live ? c / CPR : 0, cc = live ? c % CPR : 0;
        if (live) *reinterpret_cast<uint4 *>(xs + m * XROW + cc * 8) = xv[j];
        const float sm = chunk_sum16(xv[j]);   // 16 consecutive chunks = one (row, k-block); CPR % 16 == 0
        if (live && (tid & 15) == 0) sumx[(cc >> 4) * 16 + m] = sm;
    }
#pragma unroll
    for (int j = 0; j < SZW; j++)
        if (tid + 1024 * j < KBLK * 16) szs[tid + 1024 * j] = szv[j];
    __syncthreads();

    const uint16_t *xrow = xs + min(nl, M - 1) * XROW + 8 * hh;
    vmx_f32x4 fin = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < UPW; u++) {
        const int kb = wave * UPW + u;
        vmx_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const uint32_t w4[4] = {wreg[u].x, wreg[u].y, wreg[u].z, wreg[u].w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t w = w4[i];
            uint4 bu;
            bu.x = (w & 0x000F000Fu) | 0x43004300u;
            bu.y = ((w >> 4) & 0x000F000Fu) | 0x43004300u;
            bu.z = ((w >> 8) & 0x000F000Fu) | 0x43004300u;
            bu.w = ((w >> 12) & 0x000F000Fu) | 0x43004300u;
            const uint4 au = *reinterpret_cast<const uint4 *>(xrow + kb * KB + 32 * i);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, au), __builtin_bit_cast(vmx_bf16x8, bu), acc, 0, 0, 0);
        }
        const uint32_t szw = szs[kb * 16 + nl];
        const float s = __uint_as_float(szw << 16);
        const float zo = __uint_as_float(szw & 0xFFFF0000u);
        const float4 sx = *reinterpret_cast<const float4 *>(sumx + kb * 16 + 4 * hh);
        fin[0] += s * (acc[0] - zo * sx.x);
        fin[1] += s * (acc[1] - zo * sx.y);
        fin[2] += s * (acc[2] - zo * sx.z);
        fin[3] += s * (acc[3] - zo * sx.w);
    }
    if (hh == 0) {   // rows 0 .. MRR - 1 of the MFMA tile, laid out [row][column][wave] so that the sum below reads 4 x 16 bytes
#pragma unroll
        for (int r = 0; r < MRR; r++) red[(r * 16 + nl) * 16 + wave] = fin[r];
    }
    __syncthreads();
    if (tid < 64) {   // wave 0; thread = (row m, column c): sum of the 16 waves in wave order
        const int m = (tid >> 4) & (MRR - 1), c = tid & 15;
        float v = 0.f;
        const float4 *rp = reinterpret_cast<const float4 *>(red + (m * 16 + c) * 16);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 r4 = rp[q];
            v += r4.x; v += r4.y; v += r4.z; v += r4.w;
        }
        const uint16_t hn = vmx_f32_to_bf16(vmx_bf16_to_f32(hold) + vmx_bf16_to_f32(vmx_f32_to_bf16(v)));
        if (tid < 16 * M) h[hidx] = hn;
        float sq = vmx_bf16_to_f32(hn);   // sum of squares of the tile's 16 new residual values, for the norm in the next linear
        sq *= sq;
        sq = vmx_row16_sum(sq);
        if (tid < 16 * M && c == 0) ssq[m * (N / NT) + nt] = sq;
    }
}

// ---------------------------------------------------------------- prompt-shaped M (more than one 16-row tile)
// The small-M kernel streams the whole weight matrix once per 16 rows.  For prompt chunks a workgroup instead owns a
// 64-row x 128-column output tile and walks k in 128-deep blocks: the activation tile goes through LDS (double buffered,
// rows padded by 16 bytes against bank conflicts) and is shared by the 4 waves; a wave owns two column tiles, so one
// A-fragment read feeds two MFMAs and an unpacked weight unit is used for 4 row tiles: the weights are streamed once per
// 64 rows and the kernel is MFMA-bound.  Same arithmetic as above: per k-block s * (sum x*(128+nib) - (128+zp) * sum x).
// grid (N/128, ceil(M/64), splits); partial [splits][slice_rows][N] f32, or with GLU (splits == 1, gate|up interleaved
// per tile) act [M][N/2] bf16.
constexpr int PF_BM = 64, PF_XROW = KB + 8;

// GROUPED (token-grouped mixture-of-experts GEMM; the grouped GEMM over sorted_token_ids / expert_ids of
// crates/core/kernels/fused_moe_gemm.cu:72-545 with int4 experts): row block blockIdx.y belongs to expert
// block_expert[blockIdx.y] and holds the (token, slot) pairs sorted[64 * blockIdx.y ...] (an entry >= numel is padding), as
// moe_align_block_size with block_size 64 leaves them.  Pair p reads activation row p / x_row_div and writes output row p, so an
// expert's tile stream is read once per 64 of its rows instead of once per pair.
struct Int4GroupedArgs {
    const void *const *table;        // [num_experts] tile streams; a null entry = expert held by another rank
    const int32_t *sorted;           // [blocks * 64] pair indices
    const int32_t *block_expert;     // [blocks]
    const int32_t *total_post_pad;   // [1] padded length: blocks beyond it have nothing to do
    int numel, x_row_div, num_experts;
};

template <bool GLU, bool GROUPED = false>
__global__ __launch_bounds__(256) void int4_gemm_tiled_kernel(const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ,
                                                               const uint16_t *__restrict__ x, float *__restrict__ partial,
                                                               uint16_t *__restrict__ act, int M, int N, int K, int group,
                                                               int kb_per_split, int slice_rows, Int4GroupedArgs ga) {
    __shared__ __align__(16) uint16_t xs[2][PF_BM * PF_XROW];
    __shared__ float sumx[2][PF_BM];
    __shared__ int32_t prow_s[PF_BM];   // GROUPED: pair index of each row of the block (-1: padding)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & 15, h = lane >> 4;
    const int kblocks = K / KB, G = K / group;
    const int row0 = blockIdx.y * PF_BM, ks = blockIdx.z;
    const int nt0 = blockIdx.x * 8 + wave * 2;   // this wave's two column tiles
    const int kb_begin = ks * kb_per_split, kb_end = min(kblocks, kb_begin + kb_per_split);
    int xrow[4] = {-1, -1, -1, -1};   // GROUPED: activation row of this thread's four staging chunks
    if constexpr (GROUPED) {
        if (row0 >= ga.total_post_pad[0]) return;
        int e = ga.block_expert[blockIdx.y];
        e = e < 0 ? 0 : (e >= ga.num_experts ? ga.num_experts - 1 : e);
        const char *packed = static_cast<const char *>(ga.table[e]);
        if (!packed) return;
        W = reinterpret_cast<const uint4 *>(packed);
        SZ = reinterpret_cast<const uint32_t *>(packed + sz_offset_bytes(K, N));
        if (tid < PF_BM) {
            const int p = ga.sorted[row0 + tid];
            prow_s[tid] = p >= 0 && p < ga.numel ? p : -1;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int p = ga.sorted[row0 + ((tid + 256 * j) >> 4)];
            xrow[j] = p >= 0 && p < ga.numel ? p / ga.x_row_div : -1;
        }
        __syncthreads();
    }

    vmx_f32x4 fin[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int c = 0; c < 2; c++) fin[mt][c] = vmx_f32x4{0.f, 0.f, 0.f, 0.f};

    // activation tile of one k-block: 64 rows x 16 chunks of 16 bytes, 4 chunks per thread (16 consecutive lanes = one row)
    uint4 xr[4];
    uint4 wr[2];
    uint32_t szr[2];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = tid + 256 * j, row = GROUPED ? xrow[j] : row0 + (c >> 4);
            xr[j] = make_uint4(0, 0, 0, 0);
            if (GROUPED ? row >= 0 : row < M) xr[j] = *reinterpret_cast<const uint4 *>(x + (int64_t)row * K + kb * KB + (c & 15) * 8);
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            wr[c] = W[((int64_t)(nt0 + c) * kblocks + kb) * 64 + lane];
            szr[c] = SZ[((int64_t)(nt0 + c) * G + (kb * KB) / group) * 16 + nl];
        }
    };
    if (kb_begin < kb_end) fetch(kb_begin);
    int buf = 0;
    for (int kb = kb_begin; kb < kb_end; kb++) {
        uint16_t *xb = xs[buf];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = tid + 256 * j;
            *reinterpret_cast<uint4 *>(xb + (c >> 4) * PF_XROW + (c & 15) * 8) = xr[j];
            const float sm = chunk_sum16(xr[j]);
            if ((tid & 15) == 0) sumx[buf][c >> 4] = sm;
        }
        const uint4 wc[2] = {wr[0], wr[1]};
        const uint32_t szc[2] = {szr[0], szr[1]};
        __syncthreads();
        if (kb + 1 < kb_end) fetch(kb + 1);   // next block's loads fly under this block's matrix work

        vmx_f32x4 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int c = 0; c < 2; c++) acc[mt][c] = vmx_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint4 bu[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint32_t w = i == 0 ? wc[c].x : (i == 1 ? wc[c].y : (i == 2 ? wc[c].z : wc[c].w));
                bu[c].x = (w & 0x000F000Fu) | 0x43004300u;
                bu[c].y = ((w >> 4) & 0x000F000Fu) | 0x43004300u;
                bu[c].z = ((w >> 8) & 0x000F000Fu) | 0x43004300u;
                bu[c].w = ((w >> 12) & 0x000F000Fu) | 0x43004300u;
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const uint4 au = *reinterpret_cast<const uint4 *>(xb + (mt * 16 + nl) * PF_XROW + 32 * i + 8 * h);
#pragma unroll
                for (int c = 0; c < 2; c++)
                    acc[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, au), __builtin_bit_cast(vmx_bf16x8, bu[c]),
                                                                         acc[mt][c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const float4 sx = *reinterpret_cast<const float4 *>(&sumx[buf][mt * 16 + 4 * h]);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float s = __uint_as_float(szc[c] << 16), zo = __uint_as_float(szc[c] & 0xFFFF0000u);
                fin[mt][c][0] += s * (acc[mt][c][0] - zo * sx.x);
                fin[mt][c][1] += s * (acc[mt][c][1] - zo * sx.y);
                fin[mt][c][2] += s * (acc[mt][c][2] - zo * sx.z);
                fin[mt][c][3] += s * (acc[mt][c][3] - zo * sx.w);
            }
        }
        buf ^= 1;
    }
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int row = row0 + mt * 16 + 4 * h + r;
                if constexpr (GROUPED) {
                    const int p = prow_s[mt * 16 + 4 * h + r];
                    row = p < 0 ? M : p;   // padding rows fail the bound check below (M = numel)
                }
                if constexpr (GLU) {
                    const float up = vmx_lane_xor<8>(fin[mt][c][r]);
                    if (row < M && nl < 8) {
                        const float gg = vmx_bf16_to_f32(vmx_f32_to_bf16(fin[mt][c][r]));
                        const float uu = vmx_bf16_to_f32(vmx_f32_to_bf16(up));
                        act[(int64_t)row * (N >> 1) + (nt0 + c) * 8 + nl] = vmx_f32_to_bf16((gg / (1.0f + expf(-gg))) * uu);
                    }
                } else if (row < M) {
                    partial[((int64_t)ks * slice_rows + row) * N + (nt0 + c) * 16 + nl] = fin[mt][c][r];
                }
            }
}

// The same product with a 64-row x 256-column workgroup tile, built around what bounds the 64 x 128 form on gfx950 (its vector
// ALU: staging, row sums and the per-k-block epilogue cost twice the cycles of its matrix instructions):
//   * a wave owns FOUR column tiles (64 x 64 outputs): one A-fragment read from LDS feeds four matrix instructions;
//   * the activation tile goes from global memory straight into LDS (global_load_lds, no registers, no vector-ALU work).
//     Rows are 256 bytes unpadded; lane l of a 1 KiB request fills slot (row l / 16, position l % 16) with the row's 16-byte
//     chunk  position ^ (row & 15), so the 16 rows a fragment read touches lie in 16 different bank groups;
//   * the row sums of the zero-point term come from the matrix cores: wave w multiplies row tile w by a matrix of ones
//     (4 extra instructions per k-block), leaves the 16 sums in LDS, and every wave applies  -(s (128 + zp)) * sum  one
//     k-block later, after the barrier it needs anyway; the epilogue per k-block is two fused multiply-adds per element.
// 1-D grid, ordered so that the row tiles of one column block are neighbours in time AND run on the same XCD (workgroup i
// runs on XCD i % 8): the block's weights come from HBM once and from that XCD's L2 for the other row tiles.  N % 256 == 0.
// (w & 0x000F000F) | base in one instruction (the compiler emits v_and + v_or: VOP3 takes no second literal)
__device__ __forceinline__ uint32_t nib_pair(uint32_t w, uint32_t base) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(0x000F000Fu), "v"(base));
    return r;
}

template <bool GLU, bool GROUPED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void int4_gemm_tiled4_kernel(
    const uint4 *__restrict__ W, const uint32_t *__restrict__ SZ, const uint16_t *__restrict__ x, float *__restrict__ partial,
    uint16_t *__restrict__ act, int M, int N, int K, int group, int kb_per_split, int slice_rows, Int4GroupedArgs ga, int row_tiles,
    uint16_t *__restrict__ ybf, const uint16_t *__restrict__ bias) {
    // ybf (one split only): the finished bf16 output y = bf16(sum + bias) instead of f32 slices
    __shared__ __align__(16) uint16_t xs[2][PF_BM * KB];
    __shared__ __align__(16) float sumx[2][PF_BM];
    __shared__ int32_t prow_s[PF_BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, h = lane >> 4;
    const int kblocks = K / KB, G = K / group;
    const int col_blocks = N >> 8, cbg = (col_blocks + 7) >> 3;
    // id -> (split, group of 8 column blocks, row tile, column block inside the group)
    const int per_split = cbg * 8 * row_tiles;
    const int ks = (int)blockIdx.x / per_split, rid = (int)blockIdx.x % per_split;
    const int g8 = rid / (8 * row_tiles), rem = rid % (8 * row_tiles);
    const int rt = rem >> 3, cb = g8 * 8 + (rem & 7);
    if (cb >= col_blocks) return;
    const int row0 = rt * PF_BM;
    const int nt0 = cb * 16 + wave * 4;   // this wave's four column tiles
    const int kb_begin = ks * kb_per_split, kb_end = min(kblocks, kb_begin + kb_per_split);
    if constexpr (GROUPED) {
        if (row0 >= ga.total_post_pad[0]) return;
        int e = ga.block_expert[rt];
        e = e < 0 ? 0 : (e >= ga.num_experts ? ga.num_experts - 1 : e);
        const char *packed = static_cast<const char *>(ga.table[e]);
        if (!packed) return;
        W = reinterpret_cast<const uint4 *>(packed);
        SZ = reinterpret_cast<const uint32_t *>(packed + sz_offset_bytes(K, N));
        if (tid < PF_BM) {
            const int p = ga.sorted[row0 + tid];
            prow_s[tid] = p >= 0 && p < ga.numel ? p : -1;
        }
        __syncthreads();
    }
    // The four 1 KiB pieces of the activation tile this wave requests per k-block: piece q = rows 16 wave + 4 q .. + 3.  A row
    // past the end (or a padding pair) reads row 0: its outputs are never stored.
    const uint16_t *xsrc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = wave * 16 + q * 4 + (lane >> 4);
        int64_t grow;
        if constexpr (GROUPED) {
            const int p = prow_s[r];
            grow = p < 0 ? 0 : p / ga.x_row_div;
        } else {
            grow = row0 + r < M ? row0 + r : 0;
        }
        xsrc[q] = x + grow * K + (((lane & 15) ^ (r & 15)) << 3);
    }

    vmx_f32x4 fin[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int c = 0; c < 4; c++) fin[mt][c] = vmx_f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 wr[4];
    uint32_t szr[4];
    auto fetch = [&](int kb, int b) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xsrc[q] + kb * KB),
                                             (__attribute__((address_space(3))) void *)(&xs[b][(wave * 16 + q * 4) * KB]), 16, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            wr[c] = W[((int64_t)(nt0 + c) * kblocks + kb) * 64 + lane];
            szr[c] = SZ[((int64_t)(nt0 + c) * G + (kb * KB) / group) * 16 + nl];
        }
    };
    if (kb_begin < kb_end) fetch(kb_begin, 0);
    int buf = 0;
    float nsz_prev[4] = {0.f, 0.f, 0.f, 0.f};   // -(s (128 + zp)) of the previous k-block
    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    uint32_t base = 0x43004300u;
    asm volatile("" : "+v"(base));   // kept in a vector register: the third operand of v_and_or_b32
    // byte offset of this lane's A fragments inside a tile: row (mt * 16 + nl), chunk (4 i + h) at position chunk ^ nl
    for (int kb = kb_begin; kb < kb_end; kb++) {
        const uint4 wc[4] = {wr[0], wr[1], wr[2], wr[3]};
        float sc[4], nsz[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            sc[c] = __uint_as_float(szr[c] << 16);
            nsz[c] = -sc[c] * __uint_as_float(szr[c] & 0xFFFF0000u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile requested one iteration ago has landed in LDS
        __syncthreads();
        if (kb + 1 < kb_end) fetch(kb + 1, buf ^ 1);   // next block's loads fly under this block's matrix work
        const uint16_t *xb = xs[buf];

        // zero-point term of the previous k-block (its row sums became visible with the barrier above)
        if (kb > kb_begin) {
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const float4 sx = *reinterpret_cast<const float4 *>(&sumx[buf ^ 1][mt * 16 + 4 * h]);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    fin[mt][c][0] = fmaf(nsz_prev[c], sx.x, fin[mt][c][0]);
                    fin[mt][c][1] = fmaf(nsz_prev[c], sx.y, fin[mt][c][1]);
                    fin[mt][c][2] = fmaf(nsz_prev[c], sx.z, fin[mt][c][2]);
                    fin[mt][c][3] = fmaf(nsz_prev[c], sx.w, fin[mt][c][3]);
                }
            }
        }
        vmx_f32x4 acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[mt][c] = vmx_f32x4{0.f, 0.f, 0.f, 0.f};
        vmx_f32x4 rs = {0.f, 0.f, 0.f, 0.f};   // row sums of row tile `wave`
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint4 bu[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t w = i == 0 ? wc[c].x : (i == 1 ? wc[c].y : (i == 2 ? wc[c].z : wc[c].w));
                bu[c].x = nib_pair(w, base);
                bu[c].y = nib_pair(w >> 4, base);
                bu[c].z = nib_pair(w >> 8, base);
                bu[c].w = nib_pair(w >> 12, base);
            }
            {
                const uint4 aw = *reinterpret_cast<const uint4 *>(xb + (wave * 16 + nl) * KB + (((4 * i + h) ^ nl) << 3));
                rs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, aw), __builtin_bit_cast(vmx_bf16x8, ones), rs, 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const uint4 au = *reinterpret_cast<const uint4 *>(xb + (mt * 16 + nl) * KB + (((4 * i + h) ^ nl) << 3));
#pragma unroll
                for (int c = 0; c < 4; c++)
                    acc[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vmx_bf16x8, au), __builtin_bit_cast(vmx_bf16x8, bu[c]),
                                                                         acc[mt][c], 0, 0, 0);
            }
        }
        if (nl == 0) *reinterpret_cast<float4 *>(&sumx[buf][wave * 16 + 4 * h]) = make_float4(rs[0], rs[1], rs[2], rs[3]);
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) fin[mt][c][r] = fmaf(sc[c], acc[mt][c][r], fin[mt][c][r]);
#pragma unroll
        for (int c = 0; c < 4; c++) nsz_prev[c] = nsz[c];
        buf ^= 1;
    }
    if (kb_begin < kb_end) {   // zero-point term of the last k-block
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const float4 sx = *reinterpret_cast<const float4 *>(&sumx[buf ^ 1][mt * 16 + 4 * h]);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                fin[mt][c][0] = fmaf(nsz_prev[c], sx.x, fin[mt][c][0]);
                fin[mt][c][1] = fmaf(nsz_prev[c], sx.y, fin[mt][c][1]);
                fin[mt][c][2] = fmaf(nsz_prev[c], sx.z, fin[mt][c][2]);
                fin[mt][c][3] = fmaf(nsz_prev[c], sx.w, fin[mt][c][3]);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int row = row0 + mt * 16 + 4 * h + r;
                if constexpr (GROUPED) {
                    const int p = prow_s[mt * 16 + 4 * h + r];
                    row = p < 0 ? M : p;   // padding rows fail the bound check below (M = numel)
                }
                if constexpr (GLU) {
                    const float up = vmx_lane_xor<8>(fin[mt][c][r]);
                    if (row < M && nl < 8) {
                        const float gg = vmx_bf16_to_f32(vmx_f32_to_bf16(fin[mt][c][r]));
                        const float uu = vmx_bf16_to_f32(vmx_f32_to_bf16(up));
                        act[(int64_t)row * (N >> 1) + (nt0 + c) * 8 + nl] = vmx_f32_to_bf16((gg / (1.0f + expf(-gg))) * uu);
                    }
                } else if (row < M) {
                    const int n = (nt0 + c) * 16 + nl;
                    if (ybf) ybf[(int64_t)row * N + n] = vmx_f32_to_bf16(bias ? fin[mt][c][r] + vmx_bf16_to_f32(bias[n]) : fin[mt][c][r]);
                    else partial[((int64_t)ks * slice_rows + row) * N + n] = fin[mt][c][r];
                }
            }
}

// Token-grouped form for a handful of rows per expert (decode batches): row block blockIdx.y = 16 sorted (token, slot) pairs
// of one expert (moe_align_block_size with block_size 16); the expert's tile stream is read once for all of them, at the rate of
// the small-M kernel.  GLU: act [pairs][N/2]; else partial [splits][pairs][N].
template <int UPW, bool TAIL, bool GLU>
__global__ __launch_bounds__(256) void int4_gemm_grouped16_kernel(Int4GroupedArgs ga, const uint16_t *__restrict__ x,
                                                                   float *__restrict__ partial, uint16_t *__restrict__ act, int N,
                                                                   int K, int group) {
    __shared__ int32_t xmap[16], omap[16];
    __shared__ int m_s;
    const int row0 = blockIdx.y * 16, tid = threadIdx.x;
    if (row0 >= ga.total_post_pad[0]) return;
    int e = ga.block_expert[blockIdx.y];
    e = e < 0 ? 0 : (e >= ga.num_experts ? ga.num_experts - 1 : e);
    const char *packed = static_cast<const char *>(ga.table[e]);
    if (!packed) return;   // expert held by another rank
    if (tid < 64) {
        const int p = tid < 16 ? ga.sorted[row0 + tid] : -1;
        const bool valid = tid < 16 && p >= 0 && p < ga.numel;
        if (tid < 16) {
            xmap[tid] = valid ? p / ga.x_row_div : 0;
            omap[tid] = valid ? p : 0;
        }
        const unsigned long long mask = __ballot(valid);   // the valid pairs of a block come first
        if (tid == 0) m_s = __popcll(mask);
    }
    __syncthreads();
    const int M = m_s;
    if (M == 0) return;
    const uint4 *W = reinterpret_cast<const uint4 *>(packed);
    const uint32_t *SZ = reinterpret_cast<const uint32_t *>(packed + sz_offset_bytes(K, N));
    int4_gemm_body_looped<UPW, TAIL, GLU>(W, SZ, x, nullptr, act, partial, (int64_t)ga.numel * N, M, N, K, group, blockIdx.x, xmap, omap);
}

// The same row blocks through the ring form (slices of whole 512-deep passes): the looped body waits for an L2 round trip per
// staging pass, and a 16-sequence decode step of an 8-expert model is thousands of such workgroups in a dozen rounds.
template <int UPW, bool GLU, int CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void int4_gemm_grouped16_ring_kernel(
    Int4GroupedArgs ga, const uint16_t *__restrict__ x, float *__restrict__ partial, uint16_t *__restrict__ act, int N, int K, int group) {
    __shared__ int32_t xmap[16], omap[16];
    __shared__ int m_s;
    const int row0 = blockIdx.y * 16, tid = threadIdx.x;
    if (row0 >= ga.total_post_pad[0]) return;
    int e = ga.block_expert[blockIdx.y];
    e = e < 0 ? 0 : (e >= ga.num_experts ? ga.num_experts - 1 : e);
    const char *packed = static_cast<const char *>(ga.table[e]);
    if (!packed) return;   // expert held by another rank
    if (tid < 64) {
        const int p = tid < 16 ? ga.sorted[row0 + tid] : -1;
        const bool valid = tid < 16 && p >= 0 && p < ga.numel;
        if (tid < 16) {
            xmap[tid] = valid ? p / ga.x_row_div : 0;
            omap[tid] = valid ? p : 0;
        }
        const unsigned long long mask = __ballot(valid);   // the valid pairs of a block come first
        if (tid == 0) m_s = __popcll(mask);
    }
    __syncthreads();
    const int M = m_s;
    if (M == 0) return;
    const uint4 *W = reinterpret_cast<const uint4 *>(packed);
    const uint32_t *SZ = reinterpret_cast<const uint32_t *>(packed + sz_offset_bytes(K, N));
    int4_gemm_ring_body<UPW, GLU, CT>(W, SZ, x, nullptr, act, partial, (int64_t)ga.numel * N, M, N, K, group, (int)blockIdx.x, xmap, omap);
}

// Grouped form for mixture-of-experts decode: blockIdx.y = p selects one (token, expert slot) pair;
// its weights are packed_table[expert_ids[p]], its activation row is x[p / x_row_div] and it writes
// row p of partial[splits][pairs][N].  M = 1 per pair.
template <int UPW, bool TAIL>
__global__ __launch_bounds__(256) void int4_gemm_grouped_kernel(const void *const *__restrict__ packed_table,
                                                                 const int32_t *__restrict__ expert_ids, int num_experts,
                                                                 const uint16_t *__restrict__ x, int x_row_div,
                                                                 float *__restrict__ partial, int pairs, int N, int K, int group) {
    const int p = blockIdx.y;
    int e = expert_ids[p];
    e = e < 0 ? 0 : (e >= num_experts ? num_experts - 1 : e);  // ids come from top-k over num_experts
    const char *packed = static_cast<const char *>(packed_table[e]);
    if (!packed) return;   // expert held by another rank (expert parallelism): the combine step skips this pair
    const uint4 *W = reinterpret_cast<const uint4 *>(packed);
    const uint32_t *SZ = reinterpret_cast<const uint32_t *>(packed + sz_offset_bytes(K, N));
    int4_gemm_body<UPW, TAIL>(W, SZ, x + (int64_t)(p / x_row_div) * K, nullptr, nullptr, partial + (int64_t)p * N, (int64_t)pairs * N, 1,
                              N, K, group, blockIdx.x);
}

// Grouped gate|up with the gated activation in the epilogue (experts packed with glu = 1, one k slice): act [pairs][N/2]
template <int UPW, bool TAIL>
__global__ __launch_bounds__(256) void int4_gemm_grouped_glu_kernel(const void *const *__restrict__ packed_table,
                                                                     const int32_t *__restrict__ expert_ids, int num_experts,
                                                                     const uint16_t *__restrict__ x, int x_row_div,
                                                                     uint16_t *__restrict__ act, int N, int K, int group) {
    const int p = blockIdx.y;
    int e = expert_ids[p];
    e = e < 0 ? 0 : (e >= num_experts ? num_experts - 1 : e);
    const char *packed = static_cast<const char *>(packed_table[e]);
    if (!packed) return;
    const uint4 *W = reinterpret_cast<const uint4 *>(packed);
    const uint32_t *SZ = reinterpret_cast<const uint32_t *>(packed + sz_offset_bytes(K, N));
    int4_gemm_body<UPW, TAIL, true>(W, SZ, x + (int64_t)(p / x_row_div) * K, nullptr, act + (int64_t)p * (N / 2), nullptr, 0, 1, N, K, group,
                                    blockIdx.x);
}

__global__ __launch_bounds__(256) void int4_reduce_kernel(const float *__restrict__ partial, const uint16_t *__restrict__ bias,
                                                           uint16_t *__restrict__ y, int M, int N, int splits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    float v = 0.f;
    for (int s = 0; s < splits; s++) v += partial[(int64_t)s * M * N + i];
    if (bias) v += vmx_bf16_to_f32(bias[i % N]);
    y[i] = vmx_f32_to_bf16(v);
}

constexpr int SPLIT_MAX = 4;  // consumers that fold the split-K sum into their first pass read at most 4 slices

struct GemmPlan { int upw; int splits; };

inline int tune(const char *name, int dflt) { return vmx_tune(name, dflt); }

// Units per wave (-> k-blocks per workgroup, split-K slices).  One work item per workgroup and every
// weight byte requested at kernel start: the chip streams fastest with the whole matrix in flight,
// so prefer the largest slice that still gives >= ~3 workgroups per CU, and at most SPLIT_MAX slices.
inline GemmPlan plan_gemm(int N, int K) {
    static const int min_items = tune("VMX_INT4_MIN_ITEMS", 768), upw_max = tune("VMX_INT4_UPW_MAX", 8);
    const int n_tiles = N / NT, kblocks = K / KB;
    auto splits_for = [&](int upw) { return (kblocks + 4 * upw - 1) / (4 * upw); };
    int upw = upw_max;
    while (upw > 1 && ((int64_t)n_tiles * splits_for(upw) < min_items || 4 * upw > kblocks)) upw >>= 1;
    while (upw < 8 && splits_for(upw) > SPLIT_MAX) upw <<= 1;
    GemmPlan p;
    p.upw = upw;
    p.splits = splits_for(upw);
    return p;
}

// rows from which the activation staging is looped (VMX_INT4_LOOPED_MIN_M overrides for measurement; 0 disables)
inline bool use_looped(int M, int upw) {
    static const int min_m = vmx_tune("VMX_INT4_LOOPED_MIN_M", 5);
    return min_m > 0 && M >= min_m && upw >= 4;
}
inline size_t gemm_lds_bytes_looped(int M, int upw, int ct = 1) {
    return (size_t)M * ((4 * upw < 8 ? 4 * upw : 8) * KB + 8) * 2 + (size_t)4 * upw * 16 * 4 + (size_t)ct * 4 * 64 * 4 * 4 +
           (size_t)ct * 4 * upw * 16 * 4;
}
// Column tiles per workgroup of the looped form: two where the tile count allows (Qwen3-8B AWQ, same box, with the staging
// loads in batches of four: 16 rows 3912 -> 4346 tok/s aggregate, 8 rows 2510 -> 2588; VMX_INT4_LOOPED_CT=1 for A/B runs)
inline int looped_ct(int M, int N) {
    static const int ct = vmx_tune("VMX_INT4_LOOPED_CT", 2);
    (void)M;
    return (ct == 2 && (N / NT) % 2 == 0) ? 2 : 1;
}

// Ring form (5..16 rows, whole passes of 512 k): column tiles per workgroup, 0 = not eligible.  The most tiles (at most 3:
// registers) that still leave a workgroup for every compute unit (Qwen3-8B AWQ at 16 rows, same box: 3 / 2 / 1 tiles
// 6367 / 6155 / 5821 tok/s aggregate).
constexpr size_t RING_LDS = 2 * 16 * 4 * KB * 2;
inline int ring_ct(int M, int N, int K, const GemmPlan &p) {
    static const int on = vmx_tune("VMX_INT4_RING", 1), force_ct = vmx_tune("VMX_INT4_RING_CT", 0);
    if (!on || M < 5 || M > 16 || (p.upw != 2 && p.upw != 4 && p.upw != 8) || (K / KB) % (4 * p.upw) != 0) return 0;
    const int n_tiles = N / NT, cus = vmx_num_cus();
    if (force_ct >= 1 && force_ct <= 3 && n_tiles % force_ct == 0) return force_ct;
    for (int ct = 3; ct > 1; ct--)
        if (n_tiles % ct == 0 && (n_tiles / ct) * p.splits >= cus) return ct;
    return 1;
}

inline size_t gemm_lds_bytes(int M, int upw) {
    return (size_t)M * (4 * upw * KB + 8) * 2 + (size_t)4 * upw * 16 * 4 + 4 * 64 * 4 * 4 + (size_t)4 * upw * 16 * 4;
}

int launch_gemm(const GemmPlan &p, const uint4 *W, const uint32_t *SZ, const uint16_t *x, const uint16_t *bias, uint16_t *y,
                float *partial, int M, int N, int K, int group, hipStream_t st, int slice_rows = 0) {
    if (slice_rows <= 0) slice_rows = M;
    if (const int rct = ring_ct(M, N, K, p)) {
        const dim3 rgrid((unsigned)((N / NT / rct) * p.splits));
#define RING_C(U, C) hipLaunchKernelGGL((int4_gemm_ring_kernel<U, false, C>), rgrid, dim3(256), RING_LDS, st, W, SZ, x, bias, y, partial, M, N, K, group, slice_rows)
#define RING(U) do { if (rct == 3) RING_C(U, 3); else if (rct == 2) RING_C(U, 2); else RING_C(U, 1); } while (0)
        if (p.upw == 8) RING(8);
        else if (p.upw == 4) RING(4);
        else RING(2);
#undef RING
#undef RING_C
        VMX_CHECK_LAUNCH("int4_gemm(ring)");
        return VMX_OK;
    }
    if (use_looped(M, p.upw)) {
        const int ct = looped_ct(M, N);
        const size_t ll = gemm_lds_bytes_looped(M, p.upw, ct);
        const dim3 lgrid((unsigned)((N / NT / ct) * p.splits));
        const bool ltail = (K / KB) % (4 * p.upw) != 0;
#define LOOPED_C(U, C)                                                                                                                     \
        do {                                                                                                                               \
            if (ltail) hipLaunchKernelGGL((int4_gemm_looped_kernel<U, true, C>), lgrid, dim3(256), ll, st, W, SZ, x, bias, y, partial, M, N, K, group, slice_rows); \
            else hipLaunchKernelGGL((int4_gemm_looped_kernel<U, false, C>), lgrid, dim3(256), ll, st, W, SZ, x, bias, y, partial, M, N, K, group, slice_rows);      \
        } while (0)
#define LOOPED(U) do { if (ct == 2) LOOPED_C(U, 2); else LOOPED_C(U, 1); } while (0)
        if (p.upw == 8) LOOPED(8);
        else LOOPED(4);
#undef LOOPED
#undef LOOPED_C
        VMX_CHECK_LAUNCH("int4_gemm(looped)");
        return VMX_OK;
    }
    const size_t lds = gemm_lds_bytes(M, p.upw);
    if (lds > 160 * 1024) {
        vmx_set_error("int4_gemm: LDS need %zu B exceeds 160 KiB", lds);
        return VMX_EINVAL;
    }
    const dim3 grid((unsigned)((N / NT) * p.splits)), block(256);
    const bool tail = (K / KB) % (4 * p.upw) != 0;
#define LAUNCH_T(U, T)                                                                                                         \
    do {                                                                                                                       \
        if (lds > 64 * 1024)                                                                                                   \
            VMX_CHECK_HIP(hipFuncSetAttribute((const void *)int4_gemm_kernel<U, T>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              (int)lds));                                                                      \
        hipLaunchKernelGGL((int4_gemm_kernel<U, T>), grid, block, lds, st, W, SZ, x, bias, y, partial, M, N, K, group, slice_rows); \
    } while (0)
#define LAUNCH(U)                    \
    do {                             \
        if (tail) LAUNCH_T(U, true); \
        else LAUNCH_T(U, false);     \
    } while (0)
    switch (p.upw) {
        case 8: LAUNCH(8); break;
        case 4: LAUNCH(4); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(1); break;
    }
#undef LAUNCH
#undef LAUNCH_T
    VMX_CHECK_LAUNCH("int4_gemm");
    return VMX_OK;
}

}  // namespace

extern "C" int vmx_awq_to_gptq_qweight(const uint32_t *awq, uint32_t *out, int K, int N, void *stream) {
    VMX_REQUIRE(awq && out, "awq_to_gptq_qweight: null pointer");
    VMX_REQUIRE(K > 0 && K % 8 == 0, "awq_to_gptq_qweight: in_features (%d) must be divisible by 8", K);
    VMX_REQUIRE(N > 0 && N % 8 == 0, "awq_to_gptq_qweight: out_features (%d) must be divisible by 8", N);
    const int64_t total = (int64_t)(K / 8) * N;
    hipLaunchKernelGGL(awq_to_gptq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vmx_stream(stream), awq, out, K, N, N);
    VMX_CHECK_LAUNCH("awq_to_gptq_qweight");
    return VMX_OK;
}

extern "C" int vmx_awq_repack_nibbles(const uint32_t *in, uint32_t *out, int64_t n_words, void *stream) {
    VMX_REQUIRE(in && out && n_words >= 0 && n_words < ((int64_t)1 << 31), "awq_repack_nibbles: bad arguments");
    if (n_words == 0) return VMX_OK;
    hipLaunchKernelGGL(awq_nibble_reorder_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, vmx_stream(stream), in, out,
                       n_words, (int)n_words, (int)n_words);
    VMX_CHECK_LAUNCH("awq_repack_nibbles");
    return VMX_OK;
}

static int check_int4_shape(const char *who, int K, int N, int group) {
    VMX_REQUIRE(K > 0 && K % KB == 0, "%s: K (%d) must be a multiple of 128", who, K);
    VMX_REQUIRE(N > 0 && N % NT == 0, "%s: N (%d) must be a multiple of 16", who, N);
    VMX_REQUIRE(group > 0 && group % KB == 0 && K % group == 0, "%s: group_size (%d) must be a multiple of 128 dividing K (%d)", who,
                group, K);
    return VMX_OK;
}

extern "C" size_t vmx_int4_packed_bytes(int K, int N, int group) {
    if (K <= 0 || N <= 0 || group <= 0) return 0;
    return sz_offset_bytes(K, N) + (size_t)(K / group) * N * 4;
}

extern "C" int vmx_int4_pack(const uint32_t *qweight, const uint16_t *scales_bf16, const uint32_t *qzeros, void *packed, int K, int N,
                             int group, void *stream) {
    return vmx_int4_pack_cols(qweight, scales_bf16, qzeros, packed, K, N, group, 0, stream);
}

int vmx_int4_pack_cols(const uint32_t *qweight, const uint16_t *scales_bf16, const uint32_t *qzeros, void *packed, int K, int N,
                       int group, int glu, void *stream) {
    VMX_REQUIRE(qweight && scales_bf16 && packed, "int4_pack: null pointer");
    if (int rc = check_int4_shape("int4_pack", K, N, group)) return rc;
    VMX_REQUIRE(!glu || N % 32 == 0, "int4_pack: a gate|up linear needs N (%d) divisible by 32", N);
    const int glu_half = glu ? N / 2 : 0;
    uint32_t *W = (uint32_t *)packed;
    uint32_t *SZ = (uint32_t *)((char *)packed + sz_offset_bytes(K, N));
    const int64_t words = (int64_t)K * N / 8;
    hipLaunchKernelGGL(int4_pack_w_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, vmx_stream(stream), qweight, W, K, N, glu_half);
    VMX_CHECK_LAUNCH("int4_pack(W)");
    const int G = K / group;
    const int64_t szn = (int64_t)G * N;
    hipLaunchKernelGGL(int4_pack_sz_kernel, dim3((unsigned)((szn + 255) / 256)), dim3(256), 0, vmx_stream(stream), scales_bf16, qzeros,
                       SZ, G, N, glu_half);
    VMX_CHECK_LAUNCH("int4_pack(SZ)");
    return VMX_OK;
}

extern "C" int vmx_int4_dequant_bf16(const void *packed, uint16_t *dense, int K, int N, int group, void *stream) {
    VMX_REQUIRE(packed && dense, "int4_dequant: null pointer");
    if (int rc = check_int4_shape("int4_dequant", K, N, group)) return rc;
    const uint32_t *W = (const uint32_t *)packed;
    const uint32_t *SZ = (const uint32_t *)((const char *)packed + sz_offset_bytes(K, N));
    const int64_t words = (int64_t)K * N / 8;
    hipLaunchKernelGGL(int4_dequant_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, vmx_stream(stream), W, SZ, dense, K, N,
                       group);
    VMX_CHECK_LAUNCH("int4_dequant");
    return VMX_OK;
}

// Rows from which the prompt-shaped (tiled) kernel takes over from 16-row launches of the small-M kernels: 17 where those are
// the looped form, 49 where they are the ring form (Qwen3-8B shapes, whole call at 24 / 32 / 48 / 64 rows: down 26 / 27 / 41 / 53 us
// in 16-row ring launches against 53 / 53 / 55 / 56 us tiled, gate|up 28 / 32 / 49 / 61 against 59 / 59 / 59 / 60).
// VMX_INT4_NO_TILED=1 keeps the 16-row launches, VMX_INT4_TILED_MIN_M overrides both thresholds (measurement).
static bool tiled_ok(int M, int N, int K) {
    static const bool off = vmx_tune("VMX_INT4_NO_TILED", 0) != 0;
    static const int force_min_m = vmx_tune("VMX_INT4_TILED_MIN_M", 0);
    const int min_m = force_min_m > 0 ? force_min_m : (ring_ct(16, N, K, plan_gemm(N, K)) ? 49 : 17);
    return !off && M >= min_m && N % 128 == 0;
}

// Launch of the prompt-shaped kernel: 64 x 256 workgroup tiles where N allows (VMX_INT4_TILED_WIDE=0 keeps 64 x 128 for A/B runs)
static bool tiled_wide_ok(int N) {
    static const bool wide = vmx_tune("VMX_INT4_TILED_WIDE", 1) != 0;
    return wide && N % 256 == 0;
}

template <bool GLU, bool GROUPED>
static void launch_tiled(const uint4 *W, const uint32_t *SZ, const uint16_t *x, float *partial, uint16_t *act, int M, int N, int K, int group,
                         int kbs, int slice_rows, const Int4GroupedArgs &ga, int row_tiles, int splits, hipStream_t st,
                         uint16_t *ybf = nullptr, const uint16_t *bias = nullptr) {
    if (tiled_wide_ok(N)) {
        const int cbg = ((N >> 8) + 7) >> 3;
        hipLaunchKernelGGL((int4_gemm_tiled4_kernel<GLU, GROUPED>), dim3(cbg * 8 * row_tiles * splits), dim3(256), 0, st, W, SZ, x, partial, act, M, N,
                           K, group, kbs, slice_rows, ga, row_tiles, ybf, bias);
    } else {
        hipLaunchKernelGGL((int4_gemm_tiled_kernel<GLU, GROUPED>), dim3(N / 128, row_tiles, splits), dim3(256), 0, st, W, SZ, x, partial, act, M, N,
                           K, group, kbs, slice_rows, ga);
    }
}

// ---- internal (C++) entry points used by the decoder ---------------------------------------
int vmx_int4_splits(int N, int K) { return plan_gemm(N, K).splits; }

// partial[splits][M][N] f32, also when splits == 1.  Arguments must already satisfy check_int4_shape.
int vmx_int4_gemm_partial(const void *packed, const uint16_t *x, float *partial, int M, int N, int K, int group, hipStream_t st) {
    const GemmPlan p = plan_gemm(N, K);
    const uint4 *W = (const uint4 *)packed;
    const uint32_t *SZ = (const uint32_t *)((const char *)packed + sz_offset_bytes(K, N));
    if (tiled_ok(M, N, K)) {   // prompt chunk: 64 x 128 output tiles, the same split-K slices as the small-M plan
        const int kblocks = K / KB, kbs = (kblocks + p.splits - 1) / p.splits;
        launch_tiled<false, false>(W, SZ, x, partial, nullptr, M, N, K, group, kbs, M, Int4GroupedArgs{}, (M + PF_BM - 1) / PF_BM, p.splits, st);
        VMX_CHECK_LAUNCH("int4_gemm(tiled)");
        return VMX_OK;
    }
    for (int m0 = 0; m0 < M; m0 += 16) {   // 16 rows (one MFMA row tile) per launch; every launch re-streams the weights
        const int mm = M - m0 < 16 ? M - m0 : 16;
        if (int rc = launch_gemm(p, W, SZ, x + (int64_t)m0 * K, nullptr, nullptr, partial + (int64_t)m0 * N, mm, N, K, group, st, M)) return rc;
    }
    return VMX_OK;
}

// ---- the norm-in-prologue / residual-in-epilogue forms (M <= 2): see int4_gemm_body<NORM> and int4_gemm_resid_kernel
bool vmx_int4_resid_fusable(int M, int N, int K) {
    return M >= 1 && M <= 4 && K % (16 * KB) == 0 && K / (16 * KB) <= 8 && N % NT == 0;   // whole k-blocks for each of the 16 waves
}

namespace {
struct WidePlan { int nwv, upw, ntw, splits; };

// (waves, units per wave and tile, tiles per workgroup, k slices) for int4_gemm_wide_kernel.  Measured on MI355X (Qwen3-8B
// shapes, one row): one tile per workgroup and the largest k slice that still gives about three workgroups per CU is
// fastest -- the same rule as plan_gemm; two or more tiles per workgroup (fewer activation fetches, but fewer and longer
// workgroups) lost 3..20 %.  VMX_INT4_WIDE / VMX_INT4_WIDE_GLU=nwv,upw,ntw force a shape for measurements.
bool plan_wide(int N, int K, bool glu, WidePlan *out) {
    static const int cfgs[4][2] = {{4, 8}, {4, 4}, {4, 2}, {8, 4}};
    static const int min_items = tune("VMX_INT4_MIN_ITEMS", 768);
    const int n_tiles = N / NT, kblocks = K / KB;
    int f_nwv = 0, f_upw = 0, f_ntw = 0;
    if (const char *e = vmx_tune_str(glu ? "VMX_INT4_WIDE_GLU" : "VMX_INT4_WIDE")) sscanf(e, "%d,%d,%d", &f_nwv, &f_upw, &f_ntw);
    bool found = false;
    for (const auto &cfg : cfgs) {
        const int nwv = cfg[0], upw = cfg[1], kbs = nwv * upw, ntw = f_nwv ? f_ntw : 1;
        if (kblocks % kbs || ntw < 1 || n_tiles % ntw || upw * ntw > 24) continue;
        if (f_nwv ? (f_nwv != nwv || f_upw != upw) : nwv != 4) continue;
        const int splits = kblocks / kbs;
        if (splits > SPLIT_MAX || (glu && splits != 1)) continue;
        *out = WidePlan{nwv, upw, ntw, splits};
        found = true;
        if (f_nwv || (int64_t)n_tiles * splits >= min_items) break;   // else try the next smaller slice
    }
    return found;
}

template <int NWV, int UPW, int NTW, bool GLU, int MR>
int launch_wide_m(const WidePlan &p, const uint4 *W, const uint32_t *SZ, const uint16_t *h, const uint16_t *norm_w, float eps, const float *ssq,
                  float *partial, uint16_t *act, int N, int K, int group, hipStream_t st) {
    constexpr int KBS = NWV * UPW;
    const size_t lds = (((size_t)MR * (KBS * KB + 8) * 2 + 15) & ~(size_t)15) + (size_t)KBS * 16 + (size_t)NTW * NWV * 256 + (size_t)NTW * KBS * 64 + 16;
    if (lds > 64 * 1024)
        VMX_CHECK_HIP(hipFuncSetAttribute((const void *)int4_gemm_wide_kernel<NWV, UPW, NTW, GLU, MR>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds));
    // VMX_INT4_DOT2=1: the one-row kernel with v_dot2c_f32_bf16 on the vector ALU instead of the matrix cores (A/B measurement)
    static const bool dot2 = vmx_tune("VMX_INT4_DOT2", 0) != 0;
    if constexpr (MR == 1 && NTW == 1 && NWV == 4) {
        if (dot2) {
            hipLaunchKernelGGL((int4_gemm_wide_kernel<NWV, UPW, NTW, GLU, MR, true>), dim3((unsigned)((N / NT) / NTW * p.splits)), dim3(64 * NWV), lds,
                               st, W, SZ, h, norm_w, eps, ssq, partial, act, N, K, group);
            VMX_CHECK_LAUNCH("int4_gemm(wide, dot2)");
            return VMX_OK;
        }
    }
    hipLaunchKernelGGL((int4_gemm_wide_kernel<NWV, UPW, NTW, GLU, MR>), dim3((unsigned)((N / NT) / NTW * p.splits)), dim3(64 * NWV), lds, st, W, SZ,
                       h, norm_w, eps, ssq, partial, act, N, K, group);
    VMX_CHECK_LAUNCH("int4_gemm(wide)");
    return VMX_OK;
}

template <int NWV, int UPW, int NTW, bool GLU>
int launch_wide_t(const WidePlan &p, const uint4 *W, const uint32_t *SZ, const uint16_t *h, const uint16_t *norm_w, float eps, const float *ssq,
                  float *partial, uint16_t *act, int M, int N, int K, int group, hipStream_t st) {
    if constexpr (UPW * NTW > 24) {
        return VMX_EINVAL;
    } else {
        if (M == 1) return launch_wide_m<NWV, UPW, NTW, GLU, 1>(p, W, SZ, h, norm_w, eps, ssq, partial, act, N, K, group, st);
        if constexpr (NTW == 1) {   // three and four rows: the planner's own shapes only (one tile per workgroup)
            if (M == 3) return launch_wide_m<NWV, UPW, NTW, GLU, 3>(p, W, SZ, h, norm_w, eps, ssq, partial, act, N, K, group, st);
            if (M == 4) return launch_wide_m<NWV, UPW, NTW, GLU, 4>(p, W, SZ, h, norm_w, eps, ssq, partial, act, N, K, group, st);
        }
        if (M != 2) {
            vmx_set_error("int4_gemm(wide): %d rows with %d tiles per workgroup", M, NTW);
            return VMX_EUNSUPPORTED;
        }
        return launch_wide_m<NWV, UPW, NTW, GLU, 2>(p, W, SZ, h, norm_w, eps, ssq, partial, act, N, K, group, st);
    }
}

template <int NWV, int UPW, bool GLU>
int launch_wide_n(const WidePlan &p, const uint4 *W, const uint32_t *SZ, const uint16_t *h, const uint16_t *norm_w, float eps, const float *ssq,
                  float *partial, uint16_t *act, int M, int N, int K, int group, hipStream_t st) {
    switch (p.ntw) {
        case 6: return launch_wide_t<NWV, UPW, 6, GLU>(p, W, SZ, h, norm_w, eps, ssq, partial, act, M, N, K, group, st);
        case 4: return launch_wide_t<NWV, UPW, 4, GLU>(p, W, SZ, h, norm_w, eps, ssq, partial, act, M, N, K, group, st);
        case 3: return launch_wide_t<NWV, UPW, 3, GLU>(p, W, SZ, h, norm_w, eps, ssq, partial, act, M, N, K, group, st);
        case 2: return launch_wide_t<NWV, UPW, 2, GLU>(p, W, SZ, h, norm_w, eps, ssq, partial, act, M, N, K, group, st);
        default: return launch_wide_t<NWV, UPW, 1, GLU>(p, W, SZ, h, norm_w, eps, ssq, partial, act, M, N, K, group, st);
    }
}

template <bool GLU>
int launch_wide(const WidePlan &p, const void *packed, const uint16_t *h, const uint16_t *norm_w, float eps, const float *ssq, float *partial,
                uint16_t *act, int M, int N, int K, int group, hipStream_t st) {
    const uint4 *W = (const uint4 *)packed;
    const uint32_t *SZ = (const uint32_t *)((const char *)packed + sz_offset_bytes(K, N));
    if (p.nwv == 8) return launch_wide_n<8, 4, GLU>(p, W, SZ, h, norm_w, eps, ssq, partial, act, M, N, K, group, st);
    if (p.upw == 8) return launch_wide_n<4, 8, GLU>(p, W, SZ, h, norm_w, eps, ssq, partial, act, M, N, K, group, st);
    if (p.upw == 4) return launch_wide_n<4, 4, GLU>(p, W, SZ, h, norm_w, eps, ssq, partial, act, M, N, K, group, st);
    return launch_wide_n<4, 2, GLU>(p, W, SZ, h, norm_w, eps, ssq, partial, act, M, N, K, group, st);
}
}  // namespace

#ifdef VMX_WIDE_STAMPS
extern "C" int vmx_debug_wide_stamps_read(void *dst, size_t bytes) {
    VMX_CHECK_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(vmx_wide_stamp_buf), bytes < sizeof(vmx_wide_stamp_buf) ? bytes : sizeof(vmx_wide_stamp_buf)));
    return VMX_OK;
}
#endif

// split-K slices the norm-in-prologue form leaves in `partial` (0: shape not supported)
int vmx_int4_norm_splits(int M, int N, int K, int glu) {
    WidePlan p;
    static const int max_rows = tune("VMX_INT4_FOLD_ROWS", 4);   // measurement: 2 restores the two-row limit
    if (M < 1 || M > 4 || M > max_rows || K > 8192 || K % KB || N % NT || (glu && N % 32)) return 0;
    if (!plan_wide(N, K, glu != 0, &p)) return 0;
    if (M > 2 && p.ntw != 1) return 0;
    return p.splits;
}

int vmx_int4_gemm_partial_norm(const void *packed, const uint16_t *h, const uint16_t *norm_w, float eps, const float *ssq, float *partial,
                               int M, int N, int K, int group, hipStream_t st) {
    WidePlan p;
    VMX_REQUIRE(vmx_int4_norm_splits(M, N, K, 0) > 0 && plan_wide(N, K, false, &p), "int4_gemm_partial_norm: shape [%d x %d -> %d] not supported", M, K, N);
    return launch_wide<false>(p, packed, h, norm_w, eps, ssq, partial, nullptr, M, N, K, group, st);
}

int vmx_int4_gemm_glu_norm(const void *packed, const uint16_t *h, const uint16_t *norm_w, float eps, const float *ssq, uint16_t *act, int M,
                           int N, int K, int group, hipStream_t st) {
    WidePlan p;
    VMX_REQUIRE(vmx_int4_norm_splits(M, N, K, 1) == 1 && plan_wide(N, K, true, &p), "int4_gemm_glu_norm: shape [%d x %d -> %d] not supported", M, K, N);
    return launch_wide<true>(p, packed, h, norm_w, eps, ssq, nullptr, act, M, N, K, group, st);
}

// h[M][N] = bf16(h + bf16(x W)) in one launch; ssq[M][N/16] = per-tile sums of squares of the new h.  parts != NULL: x is
// the merge of the decode-attention partitions (head_dim-wide heads) instead of a bf16 tensor.
static int resid_launch(const void *packed, const uint16_t *x, const VmxAttnParts *parts, int head_dim, uint16_t *h, float *ssq, int M, int N,
                        int K, int group, hipStream_t st) {
    VMX_REQUIRE(vmx_int4_resid_fusable(M, N, K), "int4_gemm_resid: shape [%d x %d -> %d] not supported", M, K, N);
    const uint4 *W = (const uint4 *)packed;
    const uint32_t *SZ = (const uint32_t *)((const char *)packed + sz_offset_bytes(K, N));
    const int upw = K / (16 * KB);
    AttnMergeArgs am{nullptr, nullptr, nullptr, 1, 128};
    int pm = 0;
    if (parts) {
        VMX_REQUIRE(parts->max_parts >= 1 && parts->max_parts <= 8 && head_dim >= 8 && head_dim % 8 == 0 && K % head_dim == 0 && upw <= 4,
                    "int4_gemm_resid_attn: %d partitions / head_dim %d / K %d not supported", parts->max_parts, head_dim, K);
        am = AttnMergeArgs{parts->tmp_out, parts->max_logits, parts->exp_sums, parts->max_parts, head_dim};
        pm = parts->max_parts <= 4 ? 4 : 8;
    }
    // VMX_INT4_RESID_LDS: pad the workgroup's LDS request (bytes) to steer how many workgroups share a CU (measurement aid)
    static const int lds_pad = tune("VMX_INT4_RESID_LDS", 0);
    const int mrr = M <= 2 ? 2 : 4;   // rows the workgroup stages
    size_t lds = (((size_t)mrr * (K + 8) * 2 + 15) & ~(size_t)15) + (size_t)(K / KB) * 16 * 4 * 2 + 16 * 16 * 4 * 4;
    if ((size_t)lds_pad > lds) lds = lds_pad;
    const dim3 grid((unsigned)(N / NT)), block(1024);
#define RSD_M(U, PMV, R)                                                                                                               \
    do {                                                                                                                               \
        if (lds > 64 * 1024)                                                                                                           \
            VMX_CHECK_HIP(hipFuncSetAttribute((const void *)int4_gemm_resid_kernel<U, PMV, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((int4_gemm_resid_kernel<U, PMV, R>), grid, block, lds, st, W, SZ, x, h, ssq, M, N, K, group, am);           \
    } while (0)
#define RSD_P(U, PMV)                           \
    do {                                        \
        if (mrr == 2) RSD_M(U, PMV, 2);         \
        else RSD_M(U, PMV, 4);                  \
    } while (0)
#define RSD(U)                                  \
    do {                                        \
        if (pm == 0) RSD_P(U, 0);               \
        else if (U <= 4 && pm == 4) RSD_P((U <= 4 ? U : 1), 4); \
        else RSD_P((U <= 4 ? U : 1), 8);        \
    } while (0)
    switch (upw) {
        case 1: RSD(1); break;
        case 2: RSD(2); break;
        case 3: RSD(3); break;
        case 4: RSD(4); break;
        case 5: RSD_P(5, 0); break;
        case 6: RSD_P(6, 0); break;
        case 7: RSD_P(7, 0); break;
        default: RSD_P(8, 0); break;
    }
#undef RSD
#undef RSD_P
#undef RSD_M
    VMX_CHECK_LAUNCH("int4_gemm(resid)");
    return VMX_OK;
}

int vmx_int4_gemm_resid(const void *packed, const uint16_t *x, uint16_t *h, float *ssq, int M, int N, int K, int group, hipStream_t st) {
    return resid_launch(packed, x, nullptr, 0, h, ssq, M, N, K, group, st);
}

int vmx_int4_gemm_resid_attn(const void *packed, const VmxAttnParts *parts, int head_dim, uint16_t *h, float *ssq, int M, int N, int K,
                             int group, hipStream_t st) {
    VMX_REQUIRE(parts, "int4_gemm_resid_attn: null partitions");
    return resid_launch(packed, nullptr, parts, head_dim, h, ssq, M, N, K, group, st);
}

// act[M][N/2] = silu(gate) * up straight from the GEMM epilogue; the linear must be GLU-packed and plan to one k slice.
bool vmx_int4_glu_fusable(int N, int K) { return N % 32 == 0 && plan_gemm(N, K).splits == 1; }

int vmx_int4_gemm_glu(const void *packed, const uint16_t *x, uint16_t *act, int M, int N, int K, int group, hipStream_t st) {
    const GemmPlan p = plan_gemm(N, K);
    VMX_REQUIRE(p.splits == 1, "int4_gemm_glu: linear [%d -> %d] is split over k", K, N);
    const uint4 *W = (const uint4 *)packed;
    const uint32_t *SZ = (const uint32_t *)((const char *)packed + sz_offset_bytes(K, N));
    if (tiled_ok(M, N, K)) {
        launch_tiled<true, false>(W, SZ, x, nullptr, act, M, N, K, group, K / KB, M, Int4GroupedArgs{}, (M + PF_BM - 1) / PF_BM, 1, st);
        VMX_CHECK_LAUNCH("int4_gemm(tiled glu)");
        return VMX_OK;
    }
    const bool tail = (K / KB) % (4 * p.upw) != 0;
    const dim3 grid((unsigned)(N / NT)), block(256);
    for (int m0 = 0; m0 < M; m0 += 16) {
        const int mm = M - m0 < 16 ? M - m0 : 16;
        const uint16_t *xc = x + (int64_t)m0 * K;
        uint16_t *ac = act + (int64_t)m0 * (N / 2);
        if (const int rct = p.upw >= 4 ? ring_ct(mm, N, K, p) : 0) {
            const dim3 rgrid((unsigned)(N / NT / rct));
#define GRING_C(U, C) hipLaunchKernelGGL((int4_gemm_ring_kernel<U, true, C>), rgrid, block, RING_LDS, st, W, SZ, xc, nullptr, ac, nullptr, mm, N, K, group, mm)
#define GRING(U) do { if (rct == 3) GRING_C(U, 3); else if (rct == 2) GRING_C(U, 2); else GRING_C(U, 1); } while (0)
            if (p.upw == 8) GRING(8);
            else GRING(4);
#undef GRING
#undef GRING_C
            continue;
        }
        if (use_looped(mm, p.upw)) {
            const int ct = looped_ct(mm, N);
            const size_t ll = gemm_lds_bytes_looped(mm, p.upw, ct);
            const dim3 lgrid((unsigned)(N / NT / ct));
#define GLOOPED_C(U, C)                                                                                                            \
            do {                                                                                                                   \
                if (tail) hipLaunchKernelGGL((int4_gemm_glu_looped_kernel<U, true, C>), lgrid, block, ll, st, W, SZ, xc, ac, mm, N, K, group); \
                else hipLaunchKernelGGL((int4_gemm_glu_looped_kernel<U, false, C>), lgrid, block, ll, st, W, SZ, xc, ac, mm, N, K, group);      \
            } while (0)
#define GLOOPED(U) do { if (ct == 2) GLOOPED_C(U, 2); else GLOOPED_C(U, 1); } while (0)
            if (p.upw == 8) GLOOPED(8);
            else GLOOPED(4);
#undef GLOOPED
#undef GLOOPED_C
            continue;
        }
        const size_t lds = gemm_lds_bytes(mm, p.upw);
        VMX_REQUIRE(lds <= 160 * 1024, "int4_gemm_glu: LDS need %zu B exceeds 160 KiB", lds);
#define GLU_T(U, T)                                                                                                                \
    do {                                                                                                                           \
        if (lds > 64 * 1024)                                                                                                       \
            VMX_CHECK_HIP(hipFuncSetAttribute((const void *)int4_gemm_glu_kernel<U, T>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              (int)lds));                                                                          \
        hipLaunchKernelGGL((int4_gemm_glu_kernel<U, T>), grid, block, lds, st, W, SZ, xc, ac, mm, N, K, group);                    \
    } while (0)
#define GLU_L(U) do { if (tail) GLU_T(U, true); else GLU_T(U, false); } while (0)
        switch (p.upw) {
            case 8: GLU_L(8); break;
            case 4: GLU_L(4); break;
            case 2: GLU_L(2); break;
            default: GLU_L(1); break;
        }
#undef GLU_L
#undef GLU_T
    }
    VMX_CHECK_LAUNCH("int4_gemm(glu)");
    return VMX_OK;
}

int vmx_int4_gemm_grouped(const void *const *packed_table, const int32_t *expert_ids, int num_experts, const uint16_t *x,
                          int x_row_div, float *partial, int pairs, int N, int K, int group, hipStream_t st) {
    const GemmPlan p = plan_gemm(N, K);
    const size_t lds = gemm_lds_bytes(1, p.upw);
    const dim3 grid((unsigned)((N / NT) * p.splits), (unsigned)pairs), block(256);
    const bool tail = (K / KB) % (4 * p.upw) != 0;
#define GL_T(U, T) hipLaunchKernelGGL((int4_gemm_grouped_kernel<U, T>), grid, block, lds, st, packed_table, expert_ids, num_experts, x, x_row_div, partial, pairs, N, K, group)
#define GL(U) do { if (tail) GL_T(U, true); else GL_T(U, false); } while (0)
    switch (p.upw) {
        case 8: GL(8); break;
        case 4: GL(4); break;
        case 2: GL(2); break;
        default: GL(1); break;
    }
#undef GL
#undef GL_T
    VMX_CHECK_LAUNCH("int4_gemm(grouped)");
    return VMX_OK;
}

int vmx_int4_gemm_grouped_glu(const void *const *packed_table, const int32_t *expert_ids, int num_experts, const uint16_t *x,
                              int x_row_div, uint16_t *act, int pairs, int N, int K, int group, hipStream_t st) {
    const GemmPlan p = plan_gemm(N, K);
    VMX_REQUIRE(p.splits == 1, "int4_gemm_grouped_glu: linear [%d -> %d] is split over k", K, N);
    const size_t lds = gemm_lds_bytes(1, p.upw);
    const dim3 grid((unsigned)(N / NT), (unsigned)pairs), block(256);
    const bool tail = (K / KB) % (4 * p.upw) != 0;
#define GG_T(U, T) hipLaunchKernelGGL((int4_gemm_grouped_glu_kernel<U, T>), grid, block, lds, st, packed_table, expert_ids, num_experts, x, x_row_div, act, N, K, group)
#define GG(U) do { if (tail) GG_T(U, true); else GG_T(U, false); } while (0)
    switch (p.upw) {
        case 8: GG(8); break;
        case 4: GG(4); break;
        case 2: GG(2); break;
        default: GG(1); break;
    }
#undef GG
#undef GG_T
    VMX_CHECK_LAUNCH("int4_gemm(grouped glu)");
    return VMX_OK;
}

// Token-grouped form: `sorted` / `block_expert` / `total_post_pad` as vmx_moe_align_block_size(block_size = 64) leaves them;
// max_blocks = the host-side bound on the block count (blocks past total_post_pad exit at once).  act != NULL: experts packed
// with glu = 1, act [pairs][N/2]; else partial [splits][pairs][N].
bool vmx_int4_token_grouped16_ok(int N, int K) { return N % NT == 0 && K % KB == 0; }

int vmx_int4_gemm_token_grouped(const void *const *packed_table, const int32_t *sorted, const int32_t *block_expert,
                                const int32_t *total_post_pad, int max_blocks, int block_rows, int num_experts, const uint16_t *x,
                                int x_row_div, float *partial, uint16_t *act, int pairs, int N, int K, int group, hipStream_t st) {
    VMX_REQUIRE(block_rows == 16 || block_rows == 64, "int4_gemm_token_grouped: block_rows %d (16 or 64)", block_rows);
    const GemmPlan p = plan_gemm(N, K);
    VMX_REQUIRE(!act || p.splits == 1, "int4_gemm_token_grouped: gated linear [%d -> %d] is split over k", K, N);
    const int kblocks = K / KB, kbs = (kblocks + p.splits - 1) / p.splits;
    Int4GroupedArgs ga{packed_table, sorted, block_expert, total_post_pad, pairs, x_row_div, num_experts};
    if (block_rows == 16) {
        // ring form where the plan's k slices are whole 512-deep passes (4, 7 or 8 of them) and the tiles pair up
        static const int ring_on = vmx_tune("VMX_INT4_RING_GROUPED", 1);
        const int per_slice = kblocks % p.splits == 0 ? kblocks / p.splits : 0;
        const int rupw = per_slice % 4 == 0 ? per_slice / 4 : 0;
        if (ring_on && vmx_tune("VMX_INT4_RING", 1) && (rupw == 4 || rupw == 7 || rupw == 8) && (N / NT) % 2 == 0) {
            const dim3 rgrid((unsigned)((N / NT / 2) * p.splits), (unsigned)max_blocks);
#define G16R(U)                                                                                                                       \
            do {                                                                                                                      \
                if (act) hipLaunchKernelGGL((int4_gemm_grouped16_ring_kernel<U, true, 2>), rgrid, dim3(256), RING_LDS, st, ga, x, partial, act, N, K, group); \
                else hipLaunchKernelGGL((int4_gemm_grouped16_ring_kernel<U, false, 2>), rgrid, dim3(256), RING_LDS, st, ga, x, partial, act, N, K, group);    \
            } while (0)
            if (rupw == 8) G16R(8);
            else if (rupw == 7) G16R(7);
            else G16R(4);
#undef G16R
            VMX_CHECK_LAUNCH("int4_gemm(token grouped, ring)");
            return VMX_OK;
        }
        const size_t lds = gemm_lds_bytes_looped(16, p.upw);
        const dim3 grid16((unsigned)((N / NT) * p.splits), (unsigned)max_blocks);
        const bool tail = kblocks % (4 * p.upw) != 0;
#define G16_T(U, T)                                                                                                              \
        do {                                                                                                                     \
            if (act) hipLaunchKernelGGL((int4_gemm_grouped16_kernel<U, T, true>), grid16, dim3(256), lds, st, ga, x, partial, act, N, K, group); \
            else hipLaunchKernelGGL((int4_gemm_grouped16_kernel<U, T, false>), grid16, dim3(256), lds, st, ga, x, partial, act, N, K, group);    \
        } while (0)
#define G16(U) do { if (tail) G16_T(U, true); else G16_T(U, false); } while (0)
        if (p.upw == 8) G16(8);
        else if (p.upw == 4) G16(4);
        else if (p.upw == 2) G16(2);
        else G16(1);
#undef G16
#undef G16_T
        VMX_CHECK_LAUNCH("int4_gemm(token grouped, 16 rows)");
        return VMX_OK;
    }
    VMX_REQUIRE(N % 128 == 0, "int4_gemm_token_grouped: N (%d) must be a multiple of 128", N);
    if (act)
        launch_tiled<true, true>(nullptr, nullptr, x, nullptr, act, pairs, N, K, group, kblocks, pairs, ga, max_blocks, 1, st);
    else
        launch_tiled<false, true>(nullptr, nullptr, x, partial, nullptr, pairs, N, K, group, kbs, pairs, ga, max_blocks, p.splits, st);
    VMX_CHECK_LAUNCH("int4_gemm(token grouped)");
    return VMX_OK;
}

// AWQ checkpoint tensors -> columns [col0, col0+N) of merged GPTQ-sequential tensors with Ntot columns.
int vmx_awq_convert_into(const uint32_t *awq_qweight, const uint32_t *awq_qzeros, uint32_t *qweight_out, uint32_t *qzeros_out,
                         int K, int N, int group, int Ntot, int col0, hipStream_t st) {
    const int64_t total = (int64_t)(K / 8) * N;
    hipLaunchKernelGGL(awq_to_gptq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, awq_qweight, qweight_out + col0, K,
                       N, Ntot);
    VMX_CHECK_LAUNCH("awq_to_gptq_qweight");
    const int64_t zw = (int64_t)(K / group) * (N / 8);
    hipLaunchKernelGGL(awq_nibble_reorder_kernel, dim3((unsigned)((zw + 255) / 256)), dim3(256), 0, st, awq_qzeros,
                       qzeros_out + col0 / 8, zw, N / 8, Ntot / 8);
    VMX_CHECK_LAUNCH("awq_repack_nibbles");
    return VMX_OK;
}

extern "C" size_t vmx_int4_gemm_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % KB || N % NT) return 0;
    const GemmPlan p = plan_gemm(N, K);
    if (tiled_ok(M, N, K)) return (size_t)p.splits * M * N * sizeof(float);   // prompt-shaped M: f32 slices of every row
    return p.splits > 1 ? (size_t)p.splits * (M < 16 ? M : 16) * N * sizeof(float) : 0;
}

extern "C" int vmx_int4_gemm_bf16(uint16_t *y, const uint16_t *x, const void *packed, const uint16_t *bias, int M, int N, int K,
                                  int group, void *workspace, size_t workspace_bytes, void *stream) {
    VMX_REQUIRE(y && x && packed, "int4_gemm: null pointer");
    VMX_REQUIRE(M >= 1, "int4_gemm: M (%d) must be positive", M);
    if (int rc = check_int4_shape("int4_gemm", K, N, group)) return rc;
    const GemmPlan p = plan_gemm(N, K);
    const int mc = M < 16 ? M : 16;  // rows per launch of the small-M kernel
    const size_t need = vmx_int4_gemm_workspace_bytes(M, N, K);
    VMX_REQUIRE(workspace_bytes >= need && (need == 0 || workspace), "int4_gemm: workspace %zu B < required %zu B", workspace_bytes,
                need);
    const uint4 *W = (const uint4 *)packed;
    const uint32_t *SZ = (const uint32_t *)((const char *)packed + sz_offset_bytes(K, N));
    hipStream_t st = vmx_stream(stream);
    float *partial = (float *)workspace;
    (void)mc;
    if (tiled_ok(M, N, K)) {   // prompt chunk: 64 x 128 output tiles, then the slices are summed (+ bias) into bf16
        const int kblocks = K / KB, row_tiles = (M + PF_BM - 1) / PF_BM;
        // split-K only while the tiles alone leave compute units idle (the plan's slice count is sized for one row tile)
        int splits = p.splits;
        if (tiled_wide_ok(N)) {
            const int tiles = (N >> 8) * row_tiles, want = (2 * vmx_num_cus() + tiles - 1) / tiles;
            splits = want < splits ? (want < 1 ? 1 : want) : splits;
        }
        const int kbs = (kblocks + splits - 1) / splits;
        if (splits == 1 && tiled_wide_ok(N)) {   // one slice: the kernel writes the bf16 output itself
            launch_tiled<false, false>(W, SZ, x, nullptr, nullptr, M, N, K, group, kbs, M, Int4GroupedArgs{}, row_tiles, 1, st, y, bias);
            VMX_CHECK_LAUNCH("int4_gemm(tiled)");
            return VMX_OK;
        }
        launch_tiled<false, false>(W, SZ, x, partial, nullptr, M, N, K, group, kbs, M, Int4GroupedArgs{}, row_tiles, splits, st);
        VMX_CHECK_LAUNCH("int4_gemm(tiled)");
        const int64_t total = (int64_t)M * N;
        hipLaunchKernelGGL(int4_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partial, bias, y, M, N, splits);
        VMX_CHECK_LAUNCH("int4_reduce");
        return VMX_OK;
    }
    for (int m0 = 0; m0 < M; m0 += 16) {
        const int mm = M - m0 < 16 ? M - m0 : 16;
        uint16_t *yc = y + (int64_t)m0 * N;
        if (int rc = launch_gemm(p, W, SZ, x + (int64_t)m0 * K, bias, p.splits == 1 ? yc : nullptr, partial, mm, N, K, group, st)) return rc;
        if (p.splits > 1) {
            const int64_t total = (int64_t)mm * N;
            hipLaunchKernelGGL(int4_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partial, bias, yc, mm, N,
                               p.splits);
            VMX_CHECK_LAUNCH("int4_reduce");
        }
    }
    return VMX_OK;
}

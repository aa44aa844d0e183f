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
#include "vmx_internal.h"

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
        const float p0 = __shfl_xor(e0, m, 64), p1 = __shfl_xor(e1, m, 64);
        const bool neg = (lane & m) != 0;
        e0 = (neg ? -e0 : e0) + p0;
        e1 = (neg ? -e1 : e1) + p1;
    }
}

// mode 0 none, 1 pre-scale (fp16 multiply before), 2 post-scale (fp16 multiply after rounding).  grid (cols/128, rows)
__global__ __launch_bounds__(64) void exl3_had_f16_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ scale,
                                                           uint16_t *__restrict__ y, int cols, int mode) {
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.y * cols + blockIdx.x * 128 + 2 * lane;
    const uint32_t in = *reinterpret_cast<const uint32_t *>(x + base);
    half_t a = bits_to_half(in & 0xffffu), b = bits_to_half(in >> 16);
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
__global__ __launch_bounds__(64) void exl3_out_had_kernel(const float *__restrict__ partial, int splits, int64_t slice,
                                                           const uint16_t *__restrict__ svh, uint16_t *__restrict__ y, int N) {
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
    *reinterpret_cast<uint32_t *>(y + base) = (uint32_t)half_to_bits(o0) | ((uint32_t)half_to_bits(o1) << 16);
}

// ---------------------------------------------------------------- GEMM core: partial[ks][M][N] = A_had @ decode(trellis)
// Workgroup = 4 waves = one 16-column tile x (4*U) k-steps of 32; wave w owns steps [w*U, (w+1)*U) of the slice.
template <int BITS, int CB, int U>
__global__ __launch_bounds__(256) void exl3_gemm_kernel(const uint16_t *__restrict__ trellis, const uint16_t *__restrict__ a_had,
                                                         float *__restrict__ partial, int M, int N, int K) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int W = 8 * BITS;            // u32 words per tile
    constexpr int KS = 4 * U * 32;         // k-values per workgroup
    constexpr int XROW = KS + 8;
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);                       // [M][XROW] fp16
    float *red = reinterpret_cast<float *>(smem + (size_t)M * XROW * 2);     // [4][64][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tn = N / 16, steps = K / 32;
    const int nt = blockIdx.x % tn, ks = blockIdx.x / tn;
    const int step0 = ks * 4 * U;
    const int k0 = step0 * 32;
    const int n = lane & 15, g = lane >> 4;

    // activations first (L2), then the tiles' words: lane = word index
    constexpr int CPR = KS / 8;
    const int total = M * CPR;
    uint4 xv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = tid + 256 * j;
        xv[j] = make_uint4(0, 0, 0, 0);
        if (c < total) {
            const int m = c / CPR, k = k0 + (c % CPR) * 8;
            if (k < K) xv[j] = *reinterpret_cast<const uint4 *>(a_had + (int64_t)m * K + k);
        }
    }
    uint32_t wa[U], wb[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int st = step0 + wave * U + u;
        wa[u] = wb[u] = 0;
        if (st < steps && lane < W) {
            wa[u] = reinterpret_cast<const uint32_t *>(trellis + ((int64_t)(2 * st) * tn + nt) * 16 * BITS)[lane];
            wb[u] = reinterpret_cast<const uint32_t *>(trellis + ((int64_t)(2 * st + 1) * tn + nt) * 16 * BITS)[lane];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = tid + 256 * j;
        if (c < total) *reinterpret_cast<uint4 *>(xs + (c / CPR) * XROW + (c % CPR) * 8) = xv[j];
    }
    for (int c = 1024 + tid; c < total; c += 256) {
        const int m = c / CPR, k = k0 + (c % CPR) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k < K) v = *reinterpret_cast<const uint4 *>(a_had + (int64_t)m * K + k);
        *reinterpret_cast<uint4 *>(xs + m * XROW + (c % CPR) * 8) = v;
    }
    __syncthreads();

    // this lane's weights: column n, rows 8*(g&1) .. +7 of tile (g>>1): t = t_base + 8q + {0,1}, q = 0..3
    const int c2 = (n & 7) >> 1, h2 = n & 1, hb = g & 1;
    const int t_base = 64 * c2 + 32 * h2 + 4 * (n >> 3) + 2 * hb;
    const uint16_t *xrow = xs + min(n, M - 1) * XROW + 8 * g;
    vmx_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int st = step0 + wave * U + u;
        if (st < steps) {  // wave-uniform
            const uint32_t src = (g >> 1) ? wb[u] : wa[u];  // every lane evaluates both permutes' sources below
            f16x8_t bf;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int t = t_base + 8 * q;
                // windows of t and t+1 share the words that hold stream bits [(t+257)*BITS - 16, (t+258)*BITS)
                const int b2 = (t + 258) * BITS, b0 = (t + 257) * BITS - 16;
                const int i0 = (b0 >> 5) % W, i2 = ((b2 - 1) >> 5) % W;
                // words come from the lanes that loaded them; tile A's words live in wa, tile B's in wb
                const uint32_t a_a = __shfl(wa[u], i0, 64), a_b = __shfl(wb[u], i0, 64);
                const uint32_t b_a = __shfl(wa[u], i2, 64), b_b = __shfl(wb[u], i2, 64);
                const uint32_t a = (g >> 1) ? a_b : a_a, b = (g >> 1) ? b_b : b_a;
                const uint64_t merged = ((uint64_t)a << 32) | b;
                const int s2 = (((b2 - 1) >> 5) + 1) * 32 - b2;
                const uint32_t w1 = (uint32_t)(merged >> s2) & 0xffffu;
                const uint32_t w0 = (uint32_t)(merged >> (s2 + BITS)) & 0xffffu;
                bf[2 * q] = exl3_decode<CB>(w0);
                bf[2 * q + 1] = exl3_decode<CB>(w1);
            }
            (void)src;
            const uint4 au = *reinterpret_cast<const uint4 *>(xrow + (wave * U + u) * 32);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, au), bf, acc, 0, 0, 0);
        }
    }
    *reinterpret_cast<float4 *>(red + (wave * 64 + lane) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (wave == 0) {
        float4 s = *reinterpret_cast<const float4 *>(red + lane * 4);
#pragma unroll
        for (int wv = 1; wv < 4; wv++) {
            const float4 b = *reinterpret_cast<const float4 *>(red + (wv * 64 + lane) * 4);
            s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        }
        const float o4[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int mrow = 4 * g + r;
            if (mrow < M) partial[((int64_t)ks * M + mrow) * N + nt * 16 + n] = o4[r];
        }
    }
}

constexpr int EXL3_U = 4;
inline int exl3_splits(int K) { return (K / 32 + 4 * EXL3_U - 1) / (4 * EXL3_U); }

template <int BITS>
int launch_exl3_gemm(int cb, const uint16_t *trellis, const uint16_t *a_had, float *partial, int M, int N, int K, hipStream_t st) {
    const size_t lds = (size_t)M * (4 * EXL3_U * 32 + 8) * 2 + 4 * 64 * 4 * 4;
    const dim3 grid((unsigned)((N / 16) * exl3_splits(K))), block(256);
    switch (cb) {
        case 0: hipLaunchKernelGGL((exl3_gemm_kernel<BITS, 0, EXL3_U>), grid, block, lds, st, trellis, a_had, partial, M, N, K); break;
        case 1: hipLaunchKernelGGL((exl3_gemm_kernel<BITS, 1, EXL3_U>), grid, block, lds, st, trellis, a_had, partial, M, N, K); break;
        default: hipLaunchKernelGGL((exl3_gemm_kernel<BITS, 2, EXL3_U>), grid, block, lds, st, trellis, a_had, partial, M, N, K); break;
    }
    VMX_CHECK_LAUNCH("exl3_gemm");
    return VMX_OK;
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
    hipLaunchKernelGGL(exl3_had_f16_kernel, dim3(cols / 128, rows), dim3(64), 0, vmx_stream(stream), x, scale, y, cols, mode);
    VMX_CHECK_LAUNCH("had_r_128");
    return VMX_OK;
}

extern "C" size_t vmx_exl3_gemm_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    const size_t mc = M < 16 ? M : 16;
    return ((mc * K * 2 + 255) & ~(size_t)255) + (size_t)exl3_splits(K) * mc * N * 4;
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
    const int splits = exl3_splits(K);
    for (int m0 = 0; m0 < M; m0 += 16) {
        const int mm = M - m0 < 16 ? M - m0 : 16;
        hipLaunchKernelGGL(exl3_had_f16_kernel, dim3(K / 128, mm), dim3(64), 0, st, x + (int64_t)m0 * K, suh, a_had, K, suh ? 1 : 0);
        VMX_CHECK_LAUNCH("exl3_gemm(input hadamard)");
        int rc;
        const uint16_t *tr = (const uint16_t *)trellis;
        switch (bits) {
            case 1: rc = launch_exl3_gemm<1>(codebook, tr, a_had, partial, mm, N, K, st); break;
            case 2: rc = launch_exl3_gemm<2>(codebook, tr, a_had, partial, mm, N, K, st); break;
            case 3: rc = launch_exl3_gemm<3>(codebook, tr, a_had, partial, mm, N, K, st); break;
            case 4: rc = launch_exl3_gemm<4>(codebook, tr, a_had, partial, mm, N, K, st); break;
            case 5: rc = launch_exl3_gemm<5>(codebook, tr, a_had, partial, mm, N, K, st); break;
            case 6: rc = launch_exl3_gemm<6>(codebook, tr, a_had, partial, mm, N, K, st); break;
            case 7: rc = launch_exl3_gemm<7>(codebook, tr, a_had, partial, mm, N, K, st); break;
            default: rc = launch_exl3_gemm<8>(codebook, tr, a_had, partial, mm, N, K, st); break;
        }
        if (rc) return rc;
        hipLaunchKernelGGL(exl3_out_had_kernel, dim3(N / 128, mm), dim3(64), 0, st, partial, splits, (int64_t)mm * N, svh,
                           y + (int64_t)m0 * N, N);
        VMX_CHECK_LAUNCH("exl3_gemm(output hadamard)");
    }
    return VMX_OK;
}

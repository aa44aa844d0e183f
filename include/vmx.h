/*
 * libvmx -- MI355X (gfx950) quantized-inference hot path, C ABI.
 *
 * Every entry point takes raw device pointers, explicit shapes and a HIP
 * stream (passed as void*, NULL = the null stream).  No entry point allocates,
 * synchronises or reads device memory on the host unless its comment says so,
 * so all "forward" calls may be captured into a hipGraph.  Return value: 0 on
 * success, a negative VMX_E* code otherwise; vmx_last_error() returns a
 * thread-local message for the last failure.  Nothing aborts.
 *
 * Each declaration cites the reference interface it replaces (paths relative
 * to the vasia123/vllm-rust tree).
 */
#ifndef VMX_H
#define VMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMX_OK 0
#define VMX_EINVAL (-1)   /* shape / dtype / alignment precondition violated */
#define VMX_ELAUNCH (-2)  /* HIP runtime reported an error */
#define VMX_ENOMEM (-3)   /* allocation failed (create/load-time calls only) */
#define VMX_EUNSUPPORTED (-4)

/* GGML tensor type ids accepted by the I-quant entry points
 * (crates/core/src/quantization/gguf/iq/mod.rs:55-75). */
#define VMX_GGML_IQ2_XS 17
#define VMX_GGML_IQ3_XXS 18
#define VMX_GGML_IQ3_S 21
#define VMX_GGML_IQ2_S 22
#define VMX_GGML_IQ4_XS 23

/* Activation / storage dtypes. */
#define VMX_F32 0
#define VMX_BF16 1
#define VMX_F16 2

/* KV-cache dtypes (crates/core/src/kv_cache/quantization.rs:16). */
#define VMX_KV_AUTO 0     /* same 16-bit type as the activations */
#define VMX_KV_FP8_E4M3 1
#define VMX_KV_INT8 2     /* symmetric int8 stored as value + 128 (quantize_int8, kv_cache/quantization.rs:330-345): the standalone cache
                           * write and decode attention take it; the whole-token executors keep bf16 / FP8 e4m3 caches */
#define VMX_KV_FP8_E5M2 4  /* FP8 e5m2 bytes (KVCacheDtype::Fp8E5m2; read side load_kv_to_f32<fp8_e5m2_byte>, kernels/paged_attention.cu:77-82):
                           * standalone cache write and decode attention.  The write encodes e5m2 (round to nearest even, clamp to
                           * +-57344); the reference's own write path still stores e4m3 bytes in this mode (cache_engine.rs:218-224) */

/* ------------------------------------------------------------------ runtime */

/* Select `device` for the calling thread and load the code object.  Idempotent. */
int vmx_init(int device);
int vmx_abi_version(void);
const char *vmx_last_error(void);

/* ------------------------------------------------------------ GGUF I-quants */

/* Bytes per 256-weight super-block, or -1 (iq/mod.rs:85-93). */
int vmx_iq_type_size(int ggml_type);

/* Replaces dequantize_iq{2_xs,2_s,3_xxs,3_s,4_xs}_f32
 * (crates/core/kernels/iq_dequant.cu:182-194; launcher iq/mod.rs:382-436):
 * n_blocks super-blocks -> n_blocks*256 f32. */
int vmx_iq_dequantize_f32(int ggml_type, const void *blocks, float *out, int64_t n_blocks, void *stream);

/* IqLinear::forward for more than IQ_GEMV_MAX_M = 16 rows (crates/core/src/quantization/gguf/mod.rs:358-365): the weight is
 * dequantised to dense f32 once and multiplied with the UNQUANTISED activations, y[m, n] = sum_k x[m,k] * w[n,k] in f32.
 * x bf16 [m, n_in], y f32 [m, n_out]; workspace of vmx_iq_gemm_workspace_bytes(n_out, n_in) bytes.  The f32 weight runs on the bf16
 * matrix cores as an exact two-term split (w = hi + lo), so the result differs from an f32 GEMM only by summation order. */
size_t vmx_iq_gemm_workspace_bytes(int n_out, int n_in);
int vmx_iq_gemm_f32(int ggml_type, const void *blocks, const uint16_t *x, float *y, int m, int n_out, int n_in, void *workspace,
                    size_t workspace_bytes, void *stream);

/* Replaces quantize_q8_1_rows (iq_dequant.cu:253-267; launcher iq/mod.rs:797-817):
 * x f32 [total_blocks*32] -> aq int8 [total_blocks*32], ad f32 [total_blocks]. */
int vmx_quantize_q8_1_rows(const float *x, int8_t *aq, float *ad, int64_t total_blocks, void *stream);

/* Replaces mmvq_iq*_f32 (iq_dequant.cu:388-423; launcher iq/mod.rs:819-843):
 * y[m, n_out] = sum_g w_d * a_d * int_dot.  Preconditions as the reference:
 * 1 <= m <= 16, n_in % 256 == 0. */
int vmx_iq_mmvq_f32(int ggml_type, const void *w, const int8_t *aq, const float *ad, float *y, int n_out,
                    int n_in, int m, void *stream);

/* Replaces iq_matmul / IqMatmulOp::cuda_fwd (iq/mod.rs:759-877): quantise x to
 * q8_1 into `workspace`, then MMVQ.  workspace >= vmx_iq_matmul_workspace_bytes. */
size_t vmx_iq_matmul_workspace_bytes(int n_in, int m);
int vmx_iq_matmul_f32(int ggml_type, const void *w, const float *x, float *y, int n_out, int n_in, int m,
                      void *workspace, size_t workspace_bytes, void *stream);

/* Parity aid: the integer sub-dots the MMVQ kernel accumulates,
 * out[(o*m + mi)*(n_in/32) + g] (int32), for bit-exact comparison with
 * cpu_dot_iq* (iq/mod.rs:556-658). */
int vmx_iq_subdots_i32(int ggml_type, const void *w, const int8_t *aq, int32_t *out, int n_out, int n_in,
                       int m, void *stream);

/* Load-time repack of an I-quant weight [n_out, n_in] (IQ3_XXS, IQ3_S, IQ2_S) into the order a gfx950 wave consumes it:
 * the same bytes (rows rounded up to a multiple of 8, no other padding), grouped as 8 rows x one K-slice of at most 16
 * super-blocks, every field a region of 16-byte lane entries, so that the decode kernel streams it with coalesced
 * 16-byte loads.  The reference keeps the raw GGUF bytes (GgufWeightLoader::load_linear, gguf/mod.rs:1612-1662); this is
 * the load_weights-time layout step of this library's IqLinear.  `packed` needs vmx_iq_packed_bytes (0 = type without a
 * packed form) and 16-byte alignment. */
size_t vmx_iq_packed_bytes(int ggml_type, int n_out, int n_in);
int vmx_iq_pack(int ggml_type, const void *raw, void *packed, int n_out, int n_in, void *stream);

/* mmvq_iq*_f32 (iq_dequant.cu:388-423) for ONE activation row over the packed form: same integer sub-dots, f32 sums in a
 * different order.  aq 16-byte aligned, ad 8-byte aligned. */
int vmx_iq_mmvq_packed_f32(int ggml_type, const void *packed, const int8_t *aq, const float *ad, float *y, int n_out, int n_in,
                           void *stream);

/* The MLP front half of a GGUF Gemma layer in one launch (gemma4_quantized.rs:1390-1400: gate_proj and up_proj through
 * IqLinear::forward, then gelu(gate) * up): act[i] = bf16(bf16(gelu_erf(bf16(gate_i))) * bf16(up_i)) with gate = W_gate x and
 * up = W_up x as vmx_iq_mmvq_packed_f32 computes them (same bits), plus q8_1(act) (act_q int8 [inter], act_d f32 [inter / 32])
 * for the down projection.  Both weights packed, of the same type; n_in <= 4096, inter % 64 == 0; act_q / act_d must not alias aq / ad. */
int vmx_iq_mmvq_packed_geglu_bf16(int ggml_type, const void *gate_packed, const void *up_packed, const int8_t *aq, const float *ad,
                                  uint16_t *act, int8_t *act_q, float *act_d, int inter, int n_in, void *stream);

/* Parity aid: the integer sub-dots of the packed kernel, out[o * (n_in/32) + g] (the layout of vmx_iq_subdots_i32 at m = 1). */
int vmx_iq_subdots_packed_i32(int ggml_type, const void *packed, const int8_t *aq, int32_t *out, int n_out, int n_in, void *stream);

/* mmvq_iq*_f32 for 2 .. 16 activation rows (the M <= 16 accumulators of iq_dequant.cu:388-423; batched decode) on the int8
 * matrix cores: a sub-block is decoded once into signed int8 bytes and v_mfma_i32_16x16x32_i8 forms its dot with all rows; the
 * integer sub-dots are the reference's bit for bit (vmx_iq_subdots_mma_i32: out[(o * m + mi) * (n_in / 32) + g], the layout of
 * vmx_iq_subdots_i32), the f32 sums run in another order.  The weight is the 16-row-tile form vmx_iq_mma_pack makes at load
 * time (IQ3_XXS, IQ3_S, IQ2_S; the same bytes, rows rounded up to a multiple of 16; vmx_iq_mma_packed_bytes = 0 for other
 * types).  packed 16-byte aligned, aq 8-byte aligned, aq [m, n_in], ad [m, n_in / 32], y [m, n_out]. */
size_t vmx_iq_mma_packed_bytes(int ggml_type, int n_out, int n_in);
int vmx_iq_mma_pack(int ggml_type, const void *raw, void *packed, int n_out, int n_in, void *stream);
int vmx_iq_mmvq_mma_f32(int ggml_type, const void *packed, const int8_t *aq, const float *ad, float *y, int n_out, int n_in, int m,
                        void *stream);
int vmx_iq_subdots_mma_i32(int ggml_type, const void *packed, const int8_t *aq, const float *ad, int32_t *out, int n_out, int n_in,
                           int m, void *stream);
/* Parity / measurement aid: rows from which the Gemma 4 executor's I-quant linears take the matrix-core form (default 3: up to two rows the per-row kernels are faster; 17 = never). */
int vmx_debug_iq_mma_min_rows(int rows);
/* Test hook (no counterpart in the reference): 0 = the Gemma executor's matrix-core gate|up launch writes its f32 projections and
 * the stand-alone GeGLU + q8_1 kernel follows, 1 (default) = both inside the launch's epilogue; the two give the same bits. */
int vmx_debug_gemma4_mma_geglu(int on);
/* Measurement / test aid: on != 0 sends the Gemma 4 executor's prompt-sized linears through the two-term bf16 split even for
 * the types that have the exact one-pass integer form (A/B runs and the test that compares the two). */
int vmx_debug_gemma4_split_always(int on);

/* Measurement aid: while buf != NULL, vmx_iq_mmvq_packed_f32 launches that work on whole 8-row units record per wave 8
 * shader-clock stamps (u64: entry, loads issued, tables built, after each of the four row pairs, end) at
 * buf[(workgroup * 16 + wave) * 10 ..]; buf must hold 160 u64 per workgroup. */
int vmx_debug_iq_packed_stamps(void *buf);

/* Replaces gather_dequant_q6k_f32 (crates/core/kernels/gather_dequant.cu:31-83):
 * out[i, :] = dequant(row ids[i]) of a Q6_K table [num_embeddings, dim];
 * out-of-range ids leave zeros. */
int vmx_gather_dequant_q6k_f32(const void *blocks, const int32_t *ids, float *out, int num_ids, int dim,
                               int num_embeddings, void *stream);

/* GGML legacy / K-quant tensors (GgufLinear -> candle QMatMul, crates/core/src/quantization/gguf/mod.rs:114-258;
 * the arithmetic is third-party (candle-kernels 0.10.2), so these follow the public ggml block specs and parity is
 * tolerance-only).  Supported type ids: 2 Q4_0, 3 Q4_1, 6 Q5_0, 7 Q5_1, 8 Q8_0, 10 Q2_K, 11 Q3_K, 12 Q4_K, 13 Q5_K, 14 Q6_K. */
int vmx_ggml_type_size(int ggml_type);    /* bytes per block (I-quant ids included), -1 if unknown */
int vmx_ggml_block_elems(int ggml_type);  /* 32 or 256 */
/* qt.dequantize (gguf/mod.rs:284): n_blocks blocks -> f32. */
int vmx_kquant_dequantize_f32(int ggml_type, const void *blocks, float *out, int64_t n_blocks, void *stream);
/* qmm.forward for small m (gguf/mod.rs:197): y[m, n_out] = x[m, n_in] @ dequant(W)^T, f32 in/out, 1 <= m <= 8. */
int vmx_kquant_gemv_f32(int ggml_type, const void *w, const float *x, float *y, int n_out, int n_in, int m, void *stream);
/* The same product for any m (the reference's QMatMul leaves the vector kernels past 8 rows and multiplies the dequantised
 * dense weight with the f32 activations; QuantizedLinear / GgufWeightLoader, crates/core/src/quantization/gguf/mod.rs): weight
 * and activations are each split into two bf16 terms and multiplied on the bf16 matrix cores (hi.hi + lo.hi + hi.lo), so the
 * result is the f32 GEMM to about 2^-16 relative per product.  workspace >= vmx_kquant_gemm_workspace_bytes. */
size_t vmx_kquant_gemm_workspace_bytes(int m, int n_out, int n_in);
int vmx_kquant_gemm_f32(int ggml_type, const void *w, const float *x, float *y, int m, int n_out, int n_in, void *workspace,
                        size_t workspace_bytes, void *stream);

/* The same product for 1..16 rows of bf16 activations over a Q3_K table (the tied lm_head of a GGUF Gemma at several concurrent
 * sequences: the reference casts the hidden state to f32 and runs qmm.forward, gguf/mod.rs:197): y[m, n_out] f32 =
 * x[m, n_in] @ dequant(W)^T with the table streamed once for all rows; the integer part of a weight times the bf16 activation is
 * exact on the matrix cores and d * scale multiplies the f32 group sums.  n_in a multiple of 256, at most 4096. */
int vmx_q3k_rows16_bf16(const void *w, const uint16_t *x, float *y, int n_out, int n_in, int m, void *stream);

/* ------------------------------------------------------- AWQ / GPTQ int4 */

/* Replaces awq_to_gptq_qweight_int4 (crates/core/kernels/marlin_gemm.cu:491;
 * CPU form awq_marlin.rs:170-240): AWQ [K, N/8] -> GPTQ-sequential [K/8, N]. */
int vmx_awq_to_gptq_qweight(const uint32_t *awq, uint32_t *out, int K, int N, void *stream);

/* Replaces awq_marlin_repack_int4 used on qzeros (marlin_gemm.cu:431;
 * awq_marlin.rs:126-151): per-word nibble reorder, shape preserved. */
int vmx_awq_repack_nibbles(const uint32_t *in, uint32_t *out, int64_t n_words, void *stream);

/* Load-time repack into the gfx950 tile stream consumed by vmx_int4_gemm_bf16.
 * Takes the tensors exactly as the reference's MarlinLinear holds them after
 * AwqMarlinLinear::load_weights (awq_marlin.rs:580-667) minus the final
 * transpose: qweight GPTQ-sequential [K/8, N]; scales bf16 [K/group, N];
 * qzeros [K/group, N/8] sequential nibbles, or NULL for the symmetric GPTQ
 * zero point 8 (marlin_gemm.cu:43-48).  Needs K % 128 == 0, N % 16 == 0,
 * group % 128 == 0 (or group == K).  `packed` needs vmx_int4_packed_bytes. */
size_t vmx_int4_packed_bytes(int K, int N, int group);
int vmx_int4_pack(const uint32_t *qweight, const uint16_t *scales_bf16, const uint32_t *qzeros, void *packed,
                  int K, int N, int group, void *stream);

/* Replaces the awq_gemv_int4_*_bf16 family and marlin_gemm_int4(_zp)_bf16 for
 * small M (crates/core/kernels/awq_gemv.cu:46-665; dispatcher
 * marlin_cuda.rs:337-640): y[m,n] = bf16(sum_k x[m,k]*(nib-zp)*scale + bias),
 * f32 accumulation.  Any M >= 1: rows are processed 16 at a time (one MFMA row tile), so prefill-sized M re-streams
 * the weights M/16 times with the same arithmetic as decode (the reference instead switches to dequant + cuBLAS for
 * M > 8, marlin_cuda.rs:893).  workspace >= vmx_int4_gemm_workspace_bytes (split-K partial sums); bias bf16 [N] or NULL. */
size_t vmx_int4_gemm_workspace_bytes(int M, int N, int K);
int vmx_int4_gemm_bf16(uint16_t *y, const uint16_t *x, const void *packed, const uint16_t *bias, int M, int N,
                       int K, int group, void *workspace, size_t workspace_bytes, void *stream);

/* Replaces awq_marlin_dequant_int4_bf16 (crates/core/kernels/awq_marlin_dequant.cu:35-69):
 * tile stream -> dense bf16 [K, N] (prefill path and parity aid). */
int vmx_int4_dequant_bf16(const void *packed, uint16_t *dense, int K, int N, int group, void *stream);

/* ------------------------------------------------------------------------ EXL3 */

/* Replaces reconstruct_kernel<K,cb> (crates/core/kernels/exl3/reconstruct.cu; exl3_reconstruct, exl3_cuda.rs:940-1040):
 * trellis i16 [K/16, N/16, 16*bits] -> dense fp16 [K, N].  codebook 0 default, 1 mcg, 2 mul1 (exl3.rs:90-152). */
int vmx_exl3_reconstruct_f16(const void *trellis, uint16_t *dense, int K, int N, int bits, int codebook, void *stream);

/* Replaces had_hf_r_128_kernel<pre,post> (crates/core/kernels/exl3/hadamard.cu; had_r_128_fp16, exl3_cuda.rs:1044-1230):
 * per 128-column block y = H128 v / sqrt(128) in f32, rounded to fp16; mode 0 none, 1 pre-scale (fp16 multiply by
 * scale[col] first), 2 post-scale (fp16 multiply after the rounding). */
int vmx_exl3_had_r_128_f16(uint16_t *y, const uint16_t *x, const uint16_t *scale, int rows, int cols, int mode, void *stream);

/* Replaces exl3_gemm / Exl3GemmInplaceOp (exl3_cuda.rs:271-416,645-887; kernels exl3_gemm_decode_kernel / exl3_gemm_kernel):
 * y = had128(had128(x * suh) @ decode(trellis)) * svh, x/y fp16, suh fp16 [K] / svh fp16 [N] (NULL = ones).
 * K % 128 == 0, N % 128 == 0, any M >= 1 (16 rows per pass).  f32 accumulation in a fixed order: unlike the
 * reference's stream-K chain the result does not depend on the number of compute units. */
size_t vmx_exl3_gemm_workspace_bytes(int M, int K, int N);
int vmx_exl3_gemm_f16(uint16_t *y, const uint16_t *x, const void *trellis, const uint16_t *suh, const uint16_t *svh, int M,
                      int K, int N, int bits, int codebook, void *workspace, size_t workspace_bytes, void *stream);

/* ----------------------------------------------------------------- FP8 weights */

/* Replaces fp8_gemm_bf16_{naive,tiled,with_bias} (crates/core/kernels/fp8_gemm.cu:38-218; launcher
 * fp8_cuda.rs:519-727) for small M: y[m,n] = bf16(sum_k x[m,k] * (e4m3(W[n,k]) * scale) + bias[n]);
 * W u8 [N, K] OCP e4m3fn, scale f32 [1] or [N] (per_channel_scale), M >= 1, K % 16 == 0.  Up to 16 rows stream the weights once
 * through a dot-product kernel; more rows run on the bf16 matrix cores over weights converted fp8 -> bf16 in registers. */
int vmx_fp8_gemm_bf16(uint16_t *y, const uint16_t *x, const void *weight, const float *scale, const uint16_t *bias,
                      int M, int N, int K, int per_channel_scale, void *stream);

/* Replaces fp8_dequant_weights_bf16 / fp8_dequant_bf16 (fp8_gemm.cu:222-243, fp8_quant.cu:239-262):
 * out[r,c] = bf16(e4m3(in[r,c]) * scale[r or 0]). */
int vmx_fp8_dequant_bf16(uint16_t *out, const void *in, const float *scale, int64_t rows, int64_t cols,
                         int per_row_scale, void *stream);

/* Replaces fp8_static_quant_bf16, fp8_dynamic_per_token_quant_bf16 and fp8_quantize_input_bf16
 * (fp8_quant.cu:89-112,190-237; fp8_gemm.cu:246-330): out = e4m3(x / scale), round-to-nearest-even,
 * saturating at +-448.  dynamic_per_token: scale = max(absmax/448, 1e-12) per row, written to
 * scales_out; otherwise scale_in[token] (per_token_scale) or scale_in[0]. */
int vmx_fp8_quant_bf16(void *out, float *scales_out, const uint16_t *in, const float *scale_in, int tokens, int hidden,
                       int dynamic_per_token, int per_token_scale, void *stream);

/* Replaces fp8_scaled_mm_bf16 / fp8_scaled_mm_f32 (crates/core/kernels/fp8_gemm.cu:332-466), the W8A8 GEMM:
 * out[m,n] = sum_k (e4m3(A[m,k]) * a_scale) * (e4m3(B[n,k]) * b_scale) + bias[n]; A u8 [M, K], B u8 [N, K];
 * a_scale_mode 0: a_scale[0], 1: a_scale[m]; b_scale_mode 0: b_scale[0], 1: b_scale[n]; bias bf16 [N] or null (bf16 output only);
 * out bf16 [M, N], or f32 when out_f32.  Runs on v_mfma_f32_16x16x32_fp8_fp8 with the scales applied to the f32 sum. K % 16 == 0. */
int vmx_fp8_scaled_mm(void *out, int out_f32, const void *a, const void *b, const float *a_scale, const float *b_scale,
                      const uint16_t *bias, int M, int N, int K, int a_scale_mode, int b_scale_mode, void *stream);

/* GPTQ 8-bit checkpoints.  vmx_gptq_dequant_int8_bf16 replaces gptq_dequant_int8_bf16 (crates/core/kernels/gptq_dequant.cu:128-170):
 * qweight u32 [K/4, N] (four k per word, low byte first), scales bf16 [groups, N], qzeros u32 [groups, ceil(N/4)];
 * out[k][n] = bf16((q - z) * s), or out[n][k] with transposed != 0.  group_size <= 0: one group.
 * vmx_gptq_int8_gemm_bf16 is gptq_gemm for such weights (quantization/gptq_cuda.rs:322-360): y = bf16(bf16(x @ W) + bias), through a
 * caller-owned bf16 [N, K] workspace and the bf16 matrix-core GEMM; K % 16 == 0. */
int vmx_gptq_dequant_int8_bf16(uint16_t *out, const uint32_t *qweight, const uint16_t *scales, const uint32_t *qzeros, int K, int N,
                               int group_size, int transposed, void *stream);
int vmx_gptq_int8_gemm_bf16(uint16_t *y, const uint16_t *x, const uint32_t *qweight, const uint16_t *scales, const uint32_t *qzeros,
                            const uint16_t *bias, int M, int N, int K, int group_size, uint16_t *workspace, void *stream);

/* Block-scaled FP8 weights (Fp8Linear::dequantize_block_weights and the block-quantized branch of forward,
 * crates/core/src/quantization/fp8.rs:305-342, :417-425), with the semantics that function documents:
 * W_bf16[n,k] = bf16(e4m3(W[n,k]) * scale[n / block_n][k / block_k]), scale f32 [ceil(N/block_n), ceil(K/block_k)];
 * y = bf16(bf16(x @ W_bf16^T) + bias).  vmx_fp8_block_gemm_bf16 dequantises into `workspace` (bf16 [N, K]) and runs the bf16
 * matrix-core GEMM; K % 16 == 0. */
int vmx_fp8_dequant_block_bf16(uint16_t *out, const void *in, const float *scale, int64_t rows, int64_t cols, int block_n, int block_k,
                               void *stream);
int vmx_fp8_block_gemm_bf16(uint16_t *y, const uint16_t *x, const void *weight, const float *block_scale, const uint16_t *bias, int M,
                            int N, int K, int block_n, int block_k, uint16_t *workspace, void *stream);

/* Replaces fused_rmsnorm_fp8 (crates/core/kernels/layernorm_quant.cu:33-134): per row, normed = x * rsqrt(mean(x^2) + eps) * w in
 * f32, scales[row] = max(absmax(normed) / 448, 1e-12), out = e4m3(normed / scale) saturating: the activation operand and a_scale
 * (mode 1) of vmx_fp8_scaled_mm. */
int vmx_fused_rmsnorm_fp8(void *out, float *scales, const uint16_t *input, const uint16_t *weight, int rows, int hidden, float eps,
                          void *stream);

/* ----------------------------------------------------------- glue operators */

/* Replaces rms_norm_bf16 / fused_add_rms_norm_bf16 (crates/core/kernels/layernorm.cu:59-129,232-270;
 * launchers cuda_kernels.rs rms_norm_cuda).  residual may be NULL (plain norm); when given it
 * is updated in place with bf16(residual + x) first.  weight_plus_one selects Gemma's (1+w). */
int vmx_rms_norm_bf16(uint16_t *out, const uint16_t *x, uint16_t *residual, const uint16_t *weight, float eps,
                      int tokens, int hidden, int weight_plus_one, void *stream);

/* Replaces rotary_embedding_neox_bf16 (crates/core/kernels/rope.cu:94-114): in place on
 * q [tokens, heads*head_size], k [tokens, kv_heads*head_size] (k may be NULL). */
int vmx_rope_neox_bf16(const int32_t *positions, uint16_t *q, uint16_t *k, const float *cos_sin_cache,
                       int tokens, int rot_dim, int head_size, int num_heads, int num_kv_heads, void *stream);

/* Replaces rotary_embedding_gptj_bf16 (crates/core/kernels/rope.cu:119-141; pairing rule :34-38): as vmx_rope_neox_bf16 but
 * the pair of angle i is elements (2i, 2i + 1) of a head (GPT-J, CodeGen); same cos|sin table.  q, k 4-byte aligned. */
int vmx_rope_gptj_bf16(const int32_t *positions, uint16_t *q, uint16_t *k, const float *cos_sin_cache,
                       int tokens, int rot_dim, int head_size, int num_heads, int num_kv_heads, void *stream);

/* Replaces fused_qknorm_rope_bf16 (crates/core/kernels/qknorm_rope.cu:139-196): per-head RMSNorm (eps 1e-6) then NeoX
 * RoPE with ONE rounding at the store; q/k laid out [batch, heads, seq, head_dim]; cos/sin f32 [max_pos, head_dim/2];
 * positions i32 [batch, seq]. */
int vmx_fused_qknorm_rope_bf16(uint16_t *out_q, uint16_t *out_k, const uint16_t *in_q, const uint16_t *in_k,
                               const uint16_t *q_weight, const uint16_t *k_weight, const float *cos_cache,
                               const float *sin_cache, const int32_t *positions, int batch, int num_q_heads,
                               int num_kv_heads, int seq_len, int head_dim, void *stream);

/* Replace add_bf16 and gelu_and_mul_bf16 / gelu_tanh_and_mul_bf16 (crates/core/kernels/activations.cu:16-31,37-77,96-110);
 * gate and up are separate tensors here (the reference's packed [2d] row is gate = in, up = in + d). */
int vmx_add_bf16(uint16_t *out, const uint16_t *a, const uint16_t *b, int64_t n, void *stream);
int vmx_gelu_and_mul_bf16(uint16_t *out, const uint16_t *gate, const uint16_t *up, int64_t n, int tanh_approx, void *stream);

/* Replaces silu_and_mul_separate_bf16 (crates/core/kernels/activations.cu:148-164). */
int vmx_silu_and_mul_bf16(uint16_t *out, const uint16_t *gate, const uint16_t *up, int64_t n, void *stream);

/* Replaces argmax_bf16 (crates/core/kernels/sampling.cu:47-94): ids i32 [rows].  Among equal
 * maxima the smallest index wins (the reference leaves ties to its shuffle order).
 * workspace >= vmx_argmax_workspace_bytes (per-slice partial results). */
size_t vmx_argmax_workspace_bytes(int rows, int vocab);
int vmx_argmax_bf16(int32_t *out_ids, const uint16_t *logits, int rows, int vocab, void *workspace,
                    size_t workspace_bytes, void *stream);
/* The same over fp16 logits: the dtype-generic argmax of sampling/gpu.rs (greedy path) on a model that runs in fp16. */
int vmx_argmax_f16(int32_t *out_ids, const uint16_t *logits, int rows, int vocab, void *workspace,
                   size_t workspace_bytes, void *stream);

/* Replace temperature_scale_bf16, softmax_to_probs(_f32) and top_k_top_p_sample(_per_seq)
 * (crates/core/kernels/sampling.cu:102-190,197-293,351-520; callers sampling/gpu.rs:834-1094).
 * probs f32 [rows, vocab]; rand_vals f32 [rows] uniform in [0,1); top_k 0 = off, top_p 1.0 = off; the
 * per-sequence arrays (device, may be NULL) override the scalars.  Same thresholds as the reference
 * (k-th largest distinct value; distinct values accumulated from the top until top_p * sum); the draw is the
 * first index in vocabulary order whose running filtered sum exceeds rand * sum. */
int vmx_temperature_scale_bf16(uint16_t *logits, float inv_temperature, int rows, int vocab, void *stream);
int vmx_softmax_to_probs(float *probs, const void *logits, int logits_dtype, int rows, int vocab, void *stream);
int vmx_top_k_top_p_sample(int32_t *out_ids, const float *probs, const float *rand_vals, const int32_t *top_k_per_seq,
                           const float *top_p_per_seq, int top_k, float top_p, int rows, int vocab, void *stream);
/* Replaces temperature_scale_f32_per_seq (crates/core/kernels/sampling.cu:529-543): f32 logits [rows, vocab] in place, row r
 * times inv_temps[r] (device f32 [rows]). */
int vmx_temperature_scale_f32_per_seq(float *logits, const float *inv_temps, int rows, int vocab, void *stream);
/* Replaces multinomial_sample_no_filter_parallel (sampling.cu:566-631): the top_k = 0, top_p = 1 fast path.  Same chunking
 * (256 contiguous chunks of ceil(vocab / 256)) and the same order of f32 additions, so the drawn token is the reference's for
 * the same probs and rand_vals; a draw that overruns the last chunk gives vocab - 1. */
int vmx_multinomial_sample_no_filter(int32_t *out_ids, const float *probs, const float *rand_vals, int rows, int vocab, void *stream);

/* KVScales (kv_cache/quantization.rs:80-277) as device-resident state of vmx_kv_scales_state_bytes() = 32 bytes whose first two
 * f32 are k_scale and v_scale, so `state` and `(float *)state + 1` are the k_scale / v_scale arguments of every entry point here.
 * vmx_kv_scales_init = KVScales::new (1, 1, calibrated = 0), from_values / set (calibrated = 1) and reset.
 * vmx_kv_scales_calibrate_if_needed = KVScales::calibrate_if_needed (:200-213) on the bf16 K / V about to be written: a no-op once
 * calibrated, else scale = max(absmax * headroom / 448, 1e-12) (compute_scale_with_headroom :272-277; the reference's default
 * headroom is 1.5, :108) and calibrated = 1.  Runs on the stream, no host round trip; headroom must be positive and finite. */
size_t vmx_kv_scales_state_bytes(void);
int vmx_kv_scales_init(void *state, float k_scale, float v_scale, int calibrated, void *stream);
int vmx_kv_scales_calibrate_if_needed(void *state, const uint16_t *k, int64_t nk, const uint16_t *v, int64_t nv, float headroom,
                                      void *stream);

/* Replaces reshape_and_cache_bf16 and the FP8 quantise branch of CacheEngine::write_batch
 * (crates/core/kernels/cache_ops.cu:28-62; kv_cache/cache_engine.rs:196-247):
 * caches [blocks, block_size, kv_heads, head_dim]; kv_dtype VMX_KV_AUTO (bf16 copy),
 * VMX_KV_FP8_E4M3 or VMX_KV_INT8 (k_scale/v_scale: device f32[1]; INT8: byte = round(clamp(x / scale, +-127)) + 128,
 * kv_cache/quantization.rs:330-345). */
int vmx_reshape_and_cache(const uint16_t *key, const uint16_t *value, void *k_cache, void *v_cache,
                          const int32_t *slot_mapping, int tokens, int kv_heads, int head_dim, int block_size,
                          int kv_dtype, const float *k_scale, const float *v_scale, void *stream);

/* Replaces reshape_and_cache_hnd_bf16 / _fp16 (crates/core/kernels/cache_ops.cu:106-176): the same scatter into caches laid
 * out [blocks, kv_heads, block_size, head_dim] (16-bit elements of either type; head_dim % 8 == 0, 16-byte aligned pointers). */
int vmx_reshape_and_cache_hnd(const uint16_t *key, const uint16_t *value, void *k_cache, void *v_cache,
                              const int32_t *slot_mapping, int tokens, int kv_heads, int head_dim, int block_size, void *stream);

/* Replaces copy_blocks_bf16 (crates/core/kernels/cache_ops.cu:183-202; copy-on-write / preemption): for each pair
 * (src, dst) = block_mapping[2i], block_mapping[2i + 1] (device i32) copy block src to block dst in both caches.
 * bytes_per_block = block_size * kv_heads * head_dim * element bytes (any cache dtype or layout), a multiple of 16. */
int vmx_copy_blocks(void *k_cache, void *v_cache, const int32_t *block_mapping, int num_pairs, int64_t bytes_per_block, void *stream);

/* Replaces paged_attention_v2_* + paged_attention_v2_reduce_* (crates/core/kernels/paged_attention.cu:558-954;
 * launcher signature cuda_kernels.rs:641-658): decode attention for q [seqs, heads, head_dim] bf16 over the
 * paged cache.  window = 0 means full causal.  Any num_heads / num_kv_heads ratio up to 16.  workspace >= vmx_paged_attention_workspace_bytes. */
size_t vmx_paged_attention_workspace_bytes(int num_seqs, int num_heads, int head_dim, int max_seq_len);
int vmx_paged_attention_bf16(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                             const int32_t *block_tables, const int32_t *seq_lens, float scale, int num_seqs,
                             int num_heads, int num_kv_heads, int max_blocks_per_seq, int max_seq_len,
                             int head_dim, int block_size, int window, int kv_dtype, const float *k_scale,
                             const float *v_scale, void *workspace, size_t workspace_bytes, void *stream);
/* AttentionBackend::batched_decode_attention_with_lse (crates/core/src/layers/attention/backend.rs:99-137): the same decode
 * attention, also returning lse [seqs, heads] f32 = ln sum_i exp(scale * q . k_i) over each head's (windowed) context: what a
 * decode-context-parallel group exchanges to merge the ranks' partial outputs (distributed/context_parallel.rs). */
int vmx_paged_attention_bf16_with_lse(uint16_t *out, float *lse, const uint16_t *q, const void *k_cache, const void *v_cache,
                                      const int32_t *block_tables, const int32_t *seq_lens, float scale, int num_seqs,
                                      int num_heads, int num_kv_heads, int max_blocks_per_seq, int max_seq_len, int head_dim,
                                      int block_size, int window, int kv_dtype, const float *k_scale, const float *v_scale,
                                      void *workspace, size_t workspace_bytes, void *stream);
/* The local half of lse_correct_and_reduce (crates/core/src/distributed/context_parallel.rs:281-315): with the ranks'
 * statistics gathered as lse_all [ranks][rows] (rows = seqs * heads), out [rows, head_dim] f32 = attn_out (bf16) *
 * exp(lse_all[rank] - logsumexp_r lse_all[r]); the caller's all-reduce (sum) over the group gives the merged attention. */
int vmx_lse_correct_f32(float *out, const uint16_t *attn_out, const float *lse_all, int ranks, int rank, int64_t rows,
                        int head_dim, void *stream);
/* The fp16 variants of the same kernels (paged_attention_v2_f16 / _reduce_f16, crates/core/kernels/paged_attention.cu;
 * load_kv_to_f32<__half> :59-61): q, both caches and out are fp16; no quantised cache in this form. */
int vmx_paged_attention_f16(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                            const int32_t *block_tables, const int32_t *seq_lens, float scale, int num_seqs, int num_heads,
                            int num_kv_heads, int max_blocks_per_seq, int max_seq_len, int head_dim, int block_size, int window,
                            void *workspace, size_t workspace_bytes, void *stream);

/* No counterpart in the reference: vmx_paged_attention_bf16 for callers that cannot allocate or size a workspace (code
 * under hipGraph capture, tight serving loops, harnesses that forbid hipMalloc).  The split-KV partials live in a static
 * per-GPU device array of vmx_paged_attention_static_max_bytes() = 32 MiB that the library resolves in vmx_init(device);
 * a geometry whose vmx_paged_attention_workspace_bytes exceeds that is rejected with VMX_EINVAL.  Calls on one GPU share
 * the array: calls that overlap on one GPU (two streams, two threads) race; calls on one stream are ordered. */
size_t vmx_paged_attention_static_max_bytes(void);
int vmx_paged_attention_bf16_static(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                                    const int32_t *block_tables, const int32_t *seq_lens, float scale, int num_seqs,
                                    int num_heads, int num_kv_heads, int max_blocks_per_seq, int max_seq_len, int head_dim,
                                    int block_size, int window, int kv_dtype, const float *k_scale, const float *v_scale,
                                    void *stream);

/* Replaces AttentionBackend::prefill_attention (crates/core/src/layers/attention/backend.rs:60-71; semantics of the naive
 * backend, naive.rs:38-83; the reference's GPU build goes through the third-party FlashInfer wrapper): causal attention of
 * q_len new tokens at positions seqlen_offset .. seqlen_offset + q_len - 1 over the paged cache, which must already hold
 * their K/V rows (vmx_reshape_and_cache first, as the reference writes before it reads).  q [q_len, heads, head_dim] bf16,
 * out [q_len, heads * head_dim] bf16, block_table i32 [blocks of this sequence]; window = 0 means full causal, else each
 * token sees the last `window` positions including its own.  No workspace; hipGraph-capturable. */
int vmx_prefill_attention_bf16(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                               const int32_t *block_table, int q_len, int seqlen_offset, float scale, int num_heads,
                               int num_kv_heads, int head_dim, int block_size, int window, int kv_dtype, const float *k_scale,
                               const float *v_scale, void *stream);
/* The same for a model that runs in fp16 (EXL3 checkpoints, crates/core/src/quantization/weight_loader.rs:453-461): q, the
 * cache and out hold fp16; fp16 forms of the matrix instructions, softmax weights rounded to fp16. */
int vmx_prefill_attention_f16(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache,
                              const int32_t *block_table, int q_len, int seqlen_offset, float scale, int num_heads,
                              int num_kv_heads, int head_dim, int block_size, int window, void *stream);

/* Replaces the cuBLAS half_matmul_pooled used for the (tied) lm_head on the decode path
 * (crates/core/src/models/qwen3_quantized.rs:1258-1302; cuda_kernels.rs half_matmul_pooled):
 * y[m, n] = bf16(sum_k x[m,k] * W[n,k] + bias[n]), W bf16 [N, K] row-major, 1 <= M <= 16.
 * The reference's arithmetic lives in a third-party library, so parity is tolerance-only. */
int vmx_dense_gemv_bf16(uint16_t *y, const uint16_t *x, const uint16_t *W, const uint16_t *bias, int M, int N,
                        int K, void *stream);
/* The same product over fp16 x, W, bias and y (the lm_head of a model that runs in fp16; same call site). */
int vmx_dense_gemv_f16(uint16_t *y, const uint16_t *x, const uint16_t *W, const uint16_t *bias, int M, int N,
                       int K, void *stream);

/* Replaces embedding_lookup_bf16 (crates/core/kernels/activations.cu embedding_lookup_*):
 * out[t, :] = table[ids[t], :]; out-of-range ids give a zero row. */
int vmx_embedding_lookup_bf16(uint16_t *out, const uint16_t *table, const int32_t *ids, int tokens, int hidden,
                              int vocab, void *stream);

/* ---------------------------------------------------------- mixture of experts */

typedef struct vmx_comm vmx_comm;   /* device communicator, declared below */

/* Replaces topk_softmax_{f32,bf16}_warp / _generic (crates/core/kernels/topk_softmax.cu:63-311):
 * softmax over experts (expf, max-subtracted), k rounds of argmax (ties -> smaller index), optional
 * renormalisation by the selected sum.  round_probs_bf16 rounds the probabilities to bf16 before the
 * selection, as the quantized Mixtral layer's bf16 softmax does (models/mixtral_quantized.rs:150-170).
 * logits f32 or bf16 [tokens, num_experts<=256]; weights f32 [tokens,k]; ids i32 [tokens,k]. */
int vmx_topk_softmax(const void *logits, int logits_dtype, float *topk_weights, int32_t *topk_ids, int tokens,
                     int num_experts, int k, int renormalize, int round_probs_bf16, void *stream);

/* Replaces moe_align_block_size(_small)_kernel + moe_sort_tokens_kernel (crates/core/kernels/fused_moe_align.cu:34-240):
 * group the flattened (token*k+slot) indices by expert, each group padded to a multiple of block_size with
 * `numel`; expert_ids[block] names each block's expert; total_tokens_post_pad[0] = padded length.
 * Within an expert the indices ascend (the reference's atomic order is unspecified). */
int vmx_moe_align_block_size(const int32_t *topk_ids, int32_t *sorted_token_ids, int32_t *expert_ids,
                             int32_t *total_tokens_post_pad, int numel, int num_experts, int block_size,
                             int sorted_capacity, void *stream);

/* Replaces QuantizedMoELayer::forward's expert loop (models/mixtral_quantized.rs:138-195) for int4 experts:
 * out[t] = sum_j bf16(w2_e(silu(w1_e x_t) * w3_e x_t) * weight[t,j]), e = ids[t,j], accumulated in bf16 in slot
 * order.  From 12 (token, slot) pairs on the pairs are sorted by expert on the device (vmx_moe_align_block_size) and an
 * expert's weights are streamed once per block of its rows (token-grouped GEMM on the matrix cores).  w13_packed / w2_packed: DEVICE arrays of num_experts pointers to tile streams (vmx_int4_pack) of the
 * merged gate|up [hidden -> 2*inter] and the down [inter -> hidden] projections.  No host read of the routing. */
size_t vmx_moe_int4_workspace_bytes(int tokens, int k, int hidden, int inter);
int vmx_moe_int4_forward(uint16_t *out, const uint16_t *x, const float *topk_weights, const int32_t *topk_ids,
                         const void *const *w13_packed, const void *const *w2_packed, int tokens, int k, int num_experts,
                         int hidden, int inter, int group, void *workspace, size_t workspace_bytes, void *stream);

/* Expert parallelism with token dispatch -- EPMoELayer::forward (crates/core/src/moe/ep_layer.rs:180-210) over
 * TokenDispatcher::dispatch / combine (moe/token_dispatch.rs:113-231): every rank brings its OWN tokens (data-parallel
 * attention) and holds num_experts / world whole experts (ExpertPlacement::Linear, or RoundRobin with round_robin = 1;
 * moe/expert_map.rs:100-115,167-172); w13_local / w2_local: device arrays of the rank's num_experts / world tile streams in
 * local order.  A (token, slot) pair's row goes to its expert's rank, the expert's bf16 output row comes back, the source
 * sums the slots with the routing weights: out equals vmx_moe_int4_forward on the same tokens with every expert resident.
 * Routing and counts stay on the device (vmx_comm_all_to_all_v_dev): no host read, hipGraph-capturable.  cap_rows: the most
 * rows a rank can receive in one call (all ranks' tokens * k at worst); the communicator needs the peer-mapped transport with
 * exchange_bytes >= tokens * k * hidden * 2 of the largest rank. */
size_t vmx_moe_ep_workspace_bytes(int tokens, int k, int hidden, int inter, int cap_rows);
int vmx_moe_ep_int4_forward(uint16_t *out, const uint16_t *x, const float *topk_weights, const int32_t *topk_ids,
                            const void *const *w13_local, const void *const *w2_local, int tokens, int k, int num_experts,
                            int hidden, int inter, int group, int w13_glu, int round_robin, vmx_comm *comm, int cap_rows,
                            void *workspace, size_t workspace_bytes, void *stream);

/* Measurement / parity aid: vmx_moe_int4_forward streams an expert once per (token, slot) pair for a few pairs, once per block
 * of 16 sorted rows (small-M kernel) from 12 pairs and once per block of 64 rows (64 x 128-tile matrix-core kernel, the
 * grouped GEMM of crates/core/kernels/fused_moe_gemm.cu:72-545) from 96 pairs; rows = 0 / 16 / 64 forces one form, -1 restores
 * the choice by size. */
int vmx_debug_moe_block_rows(int rows);

/* ------------------------------------------------- device communicator (tensor / expert parallel group) */

/* RCCL over xGMI, resolved with dlopen like the reference's NCCL binding (crates/core/src/distributed/nccl.rs:479-560), behind
 * the operations of DeviceCommunicator (distributed/communicator.rs:29-94).  One process per GPU: rank 0 calls
 * vmx_comm_unique_id, the 128 bytes travel to the other ranks out of band (the launcher's TCP exchange in the reference,
 * torch.distributed in bench.py), then every rank calls vmx_comm_create.  Collectives run on the caller's stream and can be
 * captured into a hipGraph.  Element types and reductions use RCCL's numbering. */
#define VMX_COMM_I8 0
#define VMX_COMM_U8 1
#define VMX_COMM_I32 2
#define VMX_COMM_U32 3
#define VMX_COMM_I64 4
#define VMX_COMM_U64 5
#define VMX_COMM_F16 6
#define VMX_COMM_F32 7
#define VMX_COMM_F64 8
#define VMX_COMM_BF16 9
#define VMX_COMM_SUM 0   /* ReduceOp::Sum, Product, Max, Min, Average (communicator.rs:12-24) */
#define VMX_COMM_PROD 1
#define VMX_COMM_MAX 2
#define VMX_COMM_MIN 3
#define VMX_COMM_AVG 4
int vmx_comm_unique_id(void *out128);
int vmx_comm_create(const void *id128, int rank, int world, vmx_comm **out);
void vmx_comm_destroy(vmx_comm *c);
int vmx_comm_rank(const vmx_comm *c);
int vmx_comm_world(const vmx_comm *c);
/* all_reduce (communicator.rs:36), in place.  The two typed forms are the ones a decode step makes: Sum of the row-parallel
 * linear outputs (parallel_layers.rs:296) and Max of the 64-bit argmax keys of the vocabulary-parallel head. */
int vmx_comm_all_reduce(vmx_comm *c, void *buf, size_t count, int dtype, int op, void *stream);
int vmx_comm_all_reduce_sum_bf16(vmx_comm *c, uint16_t *buf, size_t count, void *stream);
int vmx_comm_all_reduce_sum_f16(vmx_comm *c, uint16_t *buf, size_t count, void *stream);   /* fp16 rows: models that run in fp16 */
int vmx_comm_all_reduce_max_u64(vmx_comm *c, unsigned long long *buf, size_t count, void *stream);
/* all_gather along dimension 0 (communicator.rs:44; ColumnParallelLinear's gather_output, parallel_layers.rs:152):
 * out [world][bytes_per_rank]. */
int vmx_comm_all_gather(vmx_comm *c, const void *in, void *out, size_t bytes_per_rank, void *stream);
/* reduce_scatter(Sum) along dimension 0 (communicator.rs:51): in [world][count_per_rank], out [count_per_rank] = this rank's
 * chunk of the sum; dtype VMX_COMM_BF16 or VMX_COMM_F32. */
int vmx_comm_reduce_scatter_sum(vmx_comm *c, const void *in, void *out, size_t count_per_rank, int dtype, void *stream);
/* broadcast from `root`, in place (communicator.rs:56; the step header of tensor_parallel.rs:91-165). */
int vmx_comm_broadcast(vmx_comm *c, void *buf, size_t bytes, int root, void *stream);
/* point-to-point (communicator.rs:59-62; RCCL side only). */
int vmx_comm_send(vmx_comm *c, const void *buf, size_t bytes, int dst, void *stream);
int vmx_comm_recv(vmx_comm *c, void *buf, size_t bytes, int src, void *stream);
/* barrier (communicator.rs:65): the stream continues once every rank's stream has arrived. */
int vmx_comm_barrier(vmx_comm *c, void *stream);
/* all_to_all / all_to_all_v (communicator.rs:67-93; grouped send/recv of nccl.rs:876-1007): segment r of `in` goes to rank r,
 * segment r of `out` comes from rank r; counts in elements of elem_bytes, host arrays of `world` entries. */
int vmx_comm_all_to_all(vmx_comm *c, const void *in, void *out, size_t bytes_per_rank, void *stream);
int vmx_comm_all_to_all_v(vmx_comm *c, const void *in, const size_t *send_counts, void *out, const size_t *recv_counts,
                          size_t elem_bytes, void *stream);
/* all_to_all_v with the counts in device memory: send_rows[world] (int32) is read by the kernel, recv_rows[world] is written,
 * received rows are packed in rank order (at most out_capacity_rows).  It folds the count exchange of the expert dispatch
 * (moe/token_dispatch.rs:306-327) into the transfer, so routing never visits the host.  Peer-mapped transport only. */
int vmx_comm_all_to_all_v_dev(vmx_comm *c, const void *in, const int32_t *send_rows, void *out, int32_t *recv_rows,
                              size_t row_bytes, size_t out_capacity_rows, void *stream);
/* Collectives over peer-mapped memory for the latency-bound messages of a decode step (design reference: the one-shot kernel of
 * crates/core/kernels/custom_allreduce.cu:264-484; meant for the 8 KiB x 72 per token row-parallel sums).  Set-up, once per
 * communicator and on every rank: vmx_comm_p2p_export allocates this rank's staging (two all-reduce slots of max_bytes <= 256
 * KiB, two exchange slots of exchange_bytes, fine-grained device memory -- VMX_EUNSUPPORTED when the device refuses it) and
 * flag words and returns two hipIpc handles (128 bytes); the caller all-gathers them in rank order; vmx_comm_p2p_import maps
 * the peers.  All ranks must agree that the set-up succeeded (else every rank calls vmx_comm_p2p_disable) because the
 * transport of a call is chosen from that state and the message size alone: all-reduce messages of at most max_bytes
 * (count % 8 == 0; the buffer must be 16-byte aligned), all-gather / reduce-scatter / broadcast payloads of at most
 * exchange_bytes (and at most 256 KiB when RCCL is available), barriers, and all_to_all(_v) of a communicator without RCCL go
 * over it: one kernel, ranks added in rank order (identical bits everywhere), hipGraph-replayable (call counters live in
 * device memory).  A flag wait gives up after the time-out (2 s; vmx_comm_p2p_set_timeout_ms): the rank's results become NaN,
 * its peers are told, and every later call on the communicator -- and on an executor it is attached to -- returns VMX_ELAUNCH
 * (vmx_comm_error reads the code without synchronising; vmx_comm_p2p_timeouts counts the waits that gave up).
 * vmx_comm_create_local makes a communicator without RCCL (at most 8 ranks). */
int vmx_comm_create_local(int rank, int world, vmx_comm **out);
int vmx_comm_p2p_export(vmx_comm *c, size_t max_bytes, size_t exchange_bytes, void *handles_out128);
int vmx_comm_p2p_import(vmx_comm *c, const void *all_handles);
int vmx_comm_p2p_disable(vmx_comm *c);
int vmx_comm_p2p_ready(const vmx_comm *c);
int vmx_comm_p2p_set_timeout_ms(vmx_comm *c, int ms);
int vmx_comm_p2p_timeouts(vmx_comm *c, int *out);
int vmx_comm_error(const vmx_comm *c, int *code_out);

/* ------------------------------------------------------- whole-token decoder */

/* Engine-facing surface of the quantized Qwen3-family decoder: the C form of
 * ModelForward::forward_decode_batch for QuantizedQwen3ForCausalLM
 * (crates/core/src/engine/model_forward.rs:22-138, models/qwen3_quantized.rs:557-1302).
 * create/load calls allocate and may synchronise; vmx_qwen3_decode is a fixed launch
 * sequence (no allocation, no host reads) and may be captured into a hipGraph. */
typedef struct vmx_qwen3_config {
    int hidden_size, num_layers, num_heads, num_kv_heads, head_dim, intermediate_size, vocab_size;
    float rms_norm_eps;
    int max_position;        /* rows of the cos/sin cache */
    int group_size;          /* AWQ/GPTQ group size (multiple of 128) */
    int qk_norm;             /* 1: per-head RMSNorm on q and k (Qwen3) */
    int tie_word_embeddings; /* 1: lm_head shares the embedding table */
    int kv_cache_dtype;      /* VMX_KV_AUTO (bf16) or VMX_KV_FP8_E4M3 */
    int block_size, num_blocks; /* paged KV cache geometry, per layer */
    int max_batch;           /* 1..16 */
    int max_seq_len;         /* sizes the split-KV workspace */
    int tp_size, tp_rank;    /* tensor parallelism: 0/1 = off.  With tp_size > 1 num_heads, num_kv_heads and
                                intermediate_size are THIS RANK's shares, vocab_size stays global (the lm_head is
                                split by rows, vocab_size / tp_size per rank) and a communicator must be attached */
    int max_prefill_tokens;  /* rows one vmx_qwen3_prefill call may carry (0 = no prefill entry point; at most 8192) */
    int num_experts, experts_per_tok; /* > 0: the MLP of every layer is a mixture of int4 experts (Mixtral: 8 / 2); intermediate_size
                                         is then the expert width.  0 = dense MLP */
} vmx_qwen3_config;

typedef struct vmx_qwen3 vmx_qwen3;

/* tensor ids for vmx_qwen3_load_tensor */
#define VMX_T_EMBED 0       /* bf16 [vocab, hidden] */
#define VMX_T_LM_HEAD 1     /* bf16 [vocab, hidden] (untied models) */
#define VMX_T_FINAL_NORM 2  /* bf16 [hidden] */
#define VMX_T_COS_SIN 3     /* f32 [max_position, head_dim]: cos[hd/2] | sin[hd/2] per row */
#define VMX_T_INPUT_LN 4    /* per layer: bf16 [hidden] */
#define VMX_T_POST_LN 5
#define VMX_T_Q_NORM 6      /* bf16 [head_dim] */
#define VMX_T_K_NORM 7
#define VMX_T_KV_SCALES 8   /* f32 [2] = {k_scale, v_scale} (FP8 KV cache) */
#define VMX_T_ROUTER 9      /* per layer, MoE models: bf16 [num_experts, hidden] (the dense `gate`, mixtral_quantized.rs:118-119) */

/* linear slots for vmx_qwen3_load_awq_linear */
#define VMX_L_Q 0
#define VMX_L_K 1
#define VMX_L_V 2
#define VMX_L_O 3
#define VMX_L_GATE 4
#define VMX_L_UP 5
#define VMX_L_DOWN 6
/* Gemma 4 per-layer-input linears: per layer the gate [P, hidden] and projection [hidden, P]; once per model
 * per_layer_model_projection [layers * P, hidden] (its `layer` argument is ignored) */
#define VMX_L_PLE_GATE 7
#define VMX_L_PLE_PROJ 8
#define VMX_L_PLE_MODEL_PROJ 9

int vmx_qwen3_create(const vmx_qwen3_config *cfg, vmx_qwen3 **out);
void vmx_qwen3_destroy(vmx_qwen3 *m);

/* Copy one non-quantized tensor (device or host pointer) into the model. */
int vmx_qwen3_load_tensor(vmx_qwen3 *m, int which, int layer, const void *src, size_t bytes, void *stream);

/* Load one AWQ projection as the checkpoint stores it (QuantizedWeightLoader::load_linear for AWQ,
 * crates/core/src/quantization/weight_loader.rs:597-714): qweight u32 [K, N/8], qzeros u32 [K/g, N/8]
 * (both AWQ-interleaved), scales f16 [K/g, N].  Runs the reference's load-time pipeline
 * (awq_marlin.rs:580-667) on the device and repacks into the tile stream; q|k|v and gate|up are
 * merged into one linear each once all their parts have arrived. */
int vmx_qwen3_load_awq_linear(vmx_qwen3 *m, int layer, int slot, const uint32_t *qweight, const uint32_t *qzeros,
                              const uint16_t *scales_f16, void *stream);
/* One projection of a 4-bit GPTQ checkpoint (GptqLinear / MarlinLinear::load_weights, quantization/gptq.rs:430-460,
 * marlin.rs:473-560): qweight u32 [in/8, out] (eight k per word), scales f16 [in/group, out], qzeros u32 [in/group, out/8] or NULL.
 * With qzeros: w = (nib - z) * s, z as stored (gptq_dequant.cu:48-87); NULL: w = (nib - 8) * s (marlin_gemm.cu:43-48).
 * g_idx is not an argument: as in the reference's kernels the group of row k is k / group_size. */
int vmx_qwen3_load_gptq_linear(vmx_qwen3 *m, int layer, int slot, const uint32_t *qweight, const uint32_t *qzeros,
                               const uint16_t *scales_f16, void *stream);

/* Load one expert of a mixture-of-experts layer in GPTQ form (QuantizedMoELayer / MarlinLinear, models/mixtral_quantized.rs:92-130;
 * w = (nib - 8) * scale as marlin_gemm.cu:43-48): w1 = gate, w3 = up (qweight u32 [hidden/8, inter], scales f16 [hidden/g, inter]),
 * w2 = down (qweight u32 [inter/8, hidden], scales f16 [inter/g, hidden]); device or host pointers.  Attention projections of such
 * a model are loaded with vmx_qwen3_load_awq_linear (slots Q, K, V, O). */
int vmx_qwen3_load_gptq_expert(vmx_qwen3 *m, int layer, int expert, const uint32_t *w1_qweight, const uint16_t *w1_scales_f16,
                               const uint32_t *w3_qweight, const uint16_t *w3_scales_f16, const uint32_t *w2_qweight,
                               const uint16_t *w2_scales_f16, void *stream);

/* Load one projection of an UNQUANTIZED checkpoint: bf16 [out, in], the dense Linear of the reference's plain model
 * (crates/core/src/models/qwen3.rs; BASELINE's "Qwen3-0.6B bf16" configuration).  When all seven slots of a layer were
 * loaded this way and no quantized form was, the layer runs on dense bf16 GEMVs (f32 accumulate, bf16 output). */
int vmx_qwen3_load_dense_linear(vmx_qwen3 *m, int layer, int slot, const uint16_t *weight_bf16, size_t bytes, void *stream);

/* Load one projection as a GGUF I-quant tensor (IqLinear, crates/core/src/quantization/gguf/mod.rs:262-377;
 * GgufWeightLoader::load_linear :1612-1662): raw block bytes of the [out, in] weight, ggml_type one of the I-quant ids.
 * A layer runs either on its int4 linears or, when all seven projections were loaded this way, on I-quant MMVQ
 * (activations quantised to q8_1 per linear input). */
int vmx_qwen3_load_iq_linear(vmx_qwen3 *m, int layer, int slot, int ggml_type, const void *blocks, size_t bytes,
                             void *stream);

/* Load one projection as an EXL3 trellis-coded tensor (Exl3Linear, crates/core/src/quantization/exl3.rs:204-286;
 * Exl3WeightLoader::load_linear, weight_loader.rs:469-559): trellis i16 [in/16][out/16][16*bits] and the fp16 vectors
 * suh [in], svh [out] as the checkpoint stores them; codebook 0 default, 1 mcg, 2 mul1.  When all seven projections of a
 * layer were loaded this way the layer runs y = bf16(had(had(fp16(x) * suh) @ W) * svh) per projection (activations cast
 * to fp16 for the linear and back, exl3.rs:222-228,272-277). */
int vmx_qwen3_load_exl3_linear(vmx_qwen3 *m, int layer, int slot, const void *trellis, const uint16_t *suh, const uint16_t *svh,
                               int in_features, int out_features, int bits, int codebook, void *stream);

/* One decode step: input_ids/positions/slot_mapping/seq_lens i32 [batch], block_tables i32
 * [batch, max_blocks_per_seq] (all device).  Writes greedy token ids to out_ids (device i32 [batch],
 * may be NULL) and, if logits_out != NULL, bf16 logits [batch, vocab] there. */
int vmx_qwen3_decode(vmx_qwen3 *m, const int32_t *input_ids, const int32_t *positions, const int32_t *slot_mapping,
                     const int32_t *block_tables, const int32_t *seq_lens, int batch, int max_blocks_per_seq,
                     int max_seq_len, int32_t *out_ids, uint16_t *logits_out, void *stream);

/* ModelForward::forward for one sequence (crates/core/src/engine/model_forward.rs:22-64; QuantizedQwen3ForCausalLM::forward,
 * models/qwen3_quantized.rs:1090-1256): q_len <= max_prefill_tokens prompt tokens at positions seqlen_offset ..
 * seqlen_offset + q_len - 1; slot_mapping i32 [q_len], block_table i32 [blocks of the sequence] (device).  K/V rows are
 * written to the paged cache, attention is causal over the cache (vmx_prefill_attention_bf16).  Only the last row goes
 * through the lm_head: out_id_last (device i32 [1], may be NULL) and logits_last (bf16 [vocab], may be NULL). */
int vmx_qwen3_prefill(vmx_qwen3 *m, const int32_t *input_ids, int seqlen_offset, const int32_t *slot_mapping,
                      const int32_t *block_table, int q_len, int32_t *out_id_last, uint16_t *logits_last, void *stream);

/* Device-side step bookkeeping so consecutive steps replay from one graph: input_ids <- next_ids,
 * positions += 1, seq_lens = positions + 1, slot_mapping from block_tables (what the reference's
 * engine computes on the host per step: crates/core/src/engine/helpers.rs execute_batched_decode). */
int vmx_decode_advance(int32_t *input_ids, const int32_t *next_ids, int32_t *positions, int32_t *slot_mapping,
                       int32_t *seq_lens, const int32_t *block_tables, int batch, int max_blocks_per_seq,
                       int block_size, void *stream);

/* Measurement aid: launch only the int4 linears of one decode step (4 per layer, in step order, on the
 * model's own buffers and weights).  Replayed back to back between two HIP events on its stream this
 * gives the kernel's average launch duration; _info returns the launch count and the summed
 * algorithmic bytes of one such pass. */
int vmx_qwen3_linears_only(vmx_qwen3 *m, int batch, void *stream);
/* Restrict both measurement calls to some of a layer's linears: mask bit 0 q|k|v, 1 o_proj, 2 gate|up, 3 down_proj (15 = all). */
int vmx_qwen3_linears_select(vmx_qwen3 *m, int mask);
int vmx_qwen3_linears_info(vmx_qwen3 *m, int batch, int *launches, double *alg_bytes);

/* Measurement aid: stream `bytes` from p with the int4 linear's access shape (upw 1 KiB units per wave,
 * all requested up front) and discard them; gives the HBM floor for a launch of that size. */
int vmx_debug_stream_read(const void *p, size_t bytes, int upw, void *sink, void *stream);

/* Measurement aid: out[i] = XCD id that workgroup i of an n_wgs-workgroup launch ran on (how the dispatcher
 * spreads consecutive workgroup ids over the 8 dies, each with its own L2). */
int vmx_debug_xcc_ids(int *out, int n_wgs, int wg_threads, void *stream);

/* Self-check of the kernels' register-only cross-lane exchanges (DPP row operations, v_permlane16/32_swap) against the LDS
 * crossbar forms they replace: out64[lane] = number of helpers whose result differs in any bit on that lane (all zero when
 * they agree).  No counterpart in the reference: test infrastructure of this library. */
int vmx_debug_lane_ops_check(int *out64, unsigned seed, void *stream);

/* Measurement / parity aid: the executors run q/k/v post-processing + attention + merge as ONE launch for contexts of up to
 * 1024 tokens and head_dim <= 128, as three launches otherwise; enable = 0 forces the three-launch form (default on, or off
 * when the environment has VMX_NO_FUSED_ATTN=1). */
int vmx_debug_set_fused_attention(int enable);

/* Rows of a decode call normally are different sequences with disjoint cache blocks (execute_batched_decode,
 * engine/helpers.rs:1076).  on != 0: the rows are consecutive tokens of one sequence over a shared block table (a prompt fed
 * through the decode path); the executor then writes every row's K/V before any row's attention starts. */
int vmx_qwen3_set_rows_share_cache(vmx_qwen3 *m, int on);

/* The 16-bit type the model runs in: VMX_BF16 (default) or VMX_F16.  The reference's Exl3WeightLoader forces fp16 on the
 * activation path whatever dtype was configured -- embedding, norms and lm_head of published EXL3 checkpoints are fp16 too
 * (crates/core/src/quantization/weight_loader.rs:453-461), and Exl3Linear::forward casts to fp16 and back on any other
 * dtype (exl3.rs:222-277).  With VMX_F16 every 16-bit tensor handed to vmx_qwen3_load_tensor (embedding, lm_head, norm
 * weights), the logits and the KV cache hold fp16, and every rounding point of the bf16 form rounds to fp16.  Call before
 * loading tensors; needs kv_cache_dtype VMX_KV_AUTO, a dense MLP, and every projection loaded as an EXL3 linear. */
int vmx_qwen3_set_activation_dtype(vmx_qwen3 *m, int dtype);

/* Attach the tensor-parallel group of a model created with tp_size > 1 (not owned by the model). */
int vmx_qwen3_set_comm(vmx_qwen3 *m, vmx_comm *comm);

/* Parity aid: read back an internal activation of the last step (0 = residual stream h,
 * 1 = final normed x, 2 = logits), bf16. */
int vmx_qwen3_debug_read(vmx_qwen3 *m, int which, void *dst, size_t bytes, void *stream);
/* Parity aid: copy one layer's whole K (which = 0) or V (1) cache [blocks, block_size, kv_heads, head_dim] to buf
 * (write = 0) or from it (write = 1); device or host pointer.  Lets a test start a sequence at an arbitrary length. */
int vmx_qwen3_debug_kv(vmx_qwen3 *m, int layer, int which, int write, void *buf, size_t bytes, void *stream);


/* ---------------------------------------------------- whole-token decoder, Gemma 4 (GGUF) */

/* Engine-facing surface of the quantized Gemma 4 decoder: the C form of ModelForward::forward_decode_batch for
 * QuantizedGemma4ForCausalLM on the paged-decode path (crates/core/src/models/gemma4_quantized.rs: attention
 * :1311-1424, GeGLU MLP :1468-1473, sandwich-norm layer :1743-1774, embedding scale / final norm / soft-cap
 * :2217-2279) with the honest per-layer KV geometry of KVCacheManager::new_heterogeneous
 * (crates/server/src/main.rs:44-93): sliding layers kv_heads_sliding x head_dim_sliding with a window, full layers
 * kv_heads_full x head_dim_full, full causal.  Every projection is a GGUF I-quant tensor (IqLinear); the token
 * embedding (tied lm_head) is bf16 or a GGML K-quant table.  Not covered: KV-shared layers, PLE inputs, MoE blocks,
 * attention-logit soft-capping (the reference leaves the paged path for those, :1282-1299). */
typedef struct vmx_gemma4_config {
    int hidden_size, num_layers, num_heads, intermediate_size, vocab_size;
    float rms_norm_eps;
    int head_dim_sliding, head_dim_full;      /* 256 / 512 on the 12B model */
    int kv_heads_sliding, kv_heads_full;      /* 8 / 1 */
    int sliding_window;                       /* tokens a sliding layer attends to */
    int rot_dim_sliding, rot_dim_full;        /* rotated leading dims (full rotary / proportional partial rotary) */
    int max_position;                         /* rows of the two cos/sin tables */
    int kv_cache_dtype;                       /* VMX_KV_AUTO (bf16) or VMX_KV_FP8_E4M3 */
    int block_size, num_blocks;               /* paged KV geometry, the same block ids in every layer's cache */
    int max_batch, max_seq_len;
    float final_logit_softcap;                /* 0 = off */
    int embed_ggml_type;                      /* 0 = bf16 table, else the GGML type id of token_embd (11 Q3_K, 14 Q6_K, ...) */
    /* per-layer inputs of the E2B / E4B variants (gemma4_quantized.rs:1663-1683, :2143-2192); 0 = the model has none */
    int hidden_size_per_layer_input;          /* P: 0 or a multiple of 256, at most 1024 */
    int vocab_size_per_layer_input;           /* rows of embed_tokens_per_layer [vocab_pl, num_layers * P] */
    int ple_embed_ggml_type;                  /* 0 = bf16 table, else a GGML legacy / K-quant type id */
    /* tensor parallelism (0 or 1 = off).  With tp_size > 1, num_heads, kv_heads_sliding and intermediate_size are THIS RANK'S shares
     * (q/k/v/gate/up split by output rows, o/down by input columns in whole 256-blocks); kv_heads_full is the rank's share, or the
     * whole count when the model has fewer full-attention KV heads than ranks (those K/V projections are then replicated);
     * hidden_size and vocab_size stay global, vocab_size % tp_size == 0 (the tied lm_head is vocabulary-parallel over the rank's
     * rows of the full table); per-layer-input tensors are replicated.  Needs vmx_gemma4_set_comm before the first step. */
    int tp_size, tp_rank;
    /* rows per pass of vmx_gemma4_prefill.  0 (or <= 16): passes of 16 rows through the q8_1 MMVQ.  More: every pass of more than
     * 16 rows runs the reference's large-M linear (weight dequantised once per pass, unquantised activations, IqLinear::forward_inner
     * gguf/mod.rs:358-365) on the matrix cores; costs activation buffers for that many rows plus one dequantised projection. */
    int max_prefill_tokens;
} vmx_gemma4_config;

typedef struct vmx_gemma4 vmx_gemma4;

/* layer_flags[i]: bit 0 = sliding-attention layer, bit 1 = V reuses the K projection (attention_k_eq_v, :863-870) */
#define VMX_G4_SLIDING 1
#define VMX_G4_K_EQ_V 2
/* KV-shared layer (gemma4_quantized.rs:441-469, :805-826, :1332-1337): no k/v projections, q only; attention reads the cache
 * of the last earlier non-shared layer of the same attention type.  Only a suffix of the stack may carry it. */
#define VMX_G4_KV_SHARED 4
/* this layer's MLP is twice intermediate_size wide (use_double_wide_mlp on KV-shared layers, gemma4_quantized.rs:514-525) */
#define VMX_G4_DOUBLE_WIDE_MLP 8

/* tensor ids for vmx_gemma4_load_tensor */
#define VMX_G4T_EMBED 0          /* bf16 [vocab, hidden] or raw GGML blocks of token_embd */
#define VMX_G4T_FINAL_NORM 1     /* bf16 [hidden] */
#define VMX_G4T_COS_SIN_SLIDING 2 /* f32 [max_position, rot_dim_sliding]: cos | sin halves per row */
#define VMX_G4T_COS_SIN_FULL 3   /* f32 [max_position, rot_dim_full] */
#define VMX_G4T_INPUT_LN 4       /* per layer, bf16 [hidden] */
#define VMX_G4T_POST_ATTN_LN 5
#define VMX_G4T_PRE_FFN_LN 6
#define VMX_G4T_POST_FFN_LN 7
#define VMX_G4T_Q_NORM 8         /* bf16 [head_dim of the layer] */
#define VMX_G4T_K_NORM 9
#define VMX_G4T_LAYER_SCALAR 10  /* bf16 [1] */
#define VMX_G4T_KV_SCALES 11     /* f32 [2] = {k_scale, v_scale} */
#define VMX_G4T_PLE_EMBED 12     /* embed_tokens_per_layer: bf16 [vocab_pl, layers * P] or raw GGML blocks */
#define VMX_G4T_PLE_PROJ_NORM 13 /* per_layer_projection_norm, bf16 [P] */
#define VMX_G4T_POST_PLE_NORM 14 /* per layer: post_per_layer_input_norm, bf16 [hidden] */

int vmx_gemma4_create(const vmx_gemma4_config *cfg, const int *layer_flags, vmx_gemma4 **out);
void vmx_gemma4_destroy(vmx_gemma4 *m);
int vmx_gemma4_load_tensor(vmx_gemma4 *m, int which, int layer, const void *src, size_t bytes, void *stream);
/* slot: VMX_L_Q .. VMX_L_DOWN; raw I-quant blocks of the [out, in] weight (GgufWeightLoader::load_linear, gguf/mod.rs:1612-1662) */
int vmx_gemma4_load_iq_linear(vmx_gemma4 *m, int layer, int slot, int ggml_type, const void *blocks, size_t bytes, void *stream);
/* One decode step; arguments as vmx_qwen3_decode.  Fixed launch sequence, hipGraph-capturable. */
int vmx_gemma4_decode(vmx_gemma4 *m, const int32_t *input_ids, const int32_t *positions, const int32_t *slot_mapping,
                      const int32_t *block_tables, const int32_t *seq_lens, int batch, int max_blocks_per_seq, int max_seq_len,
                      int32_t *out_ids, uint16_t *logits_out, void *stream);
/* Measurement aid: launches and summed algorithmic bytes (weights + q8_1 activations + f32 outputs) of the MMVQ calls of one step,
 * and a pass that runs only those launches on the model's own buffers. */
int vmx_gemma4_mmvq_info(vmx_gemma4 *m, int batch, int *launches, double *alg_bytes);
/* ModelForward::forward for one sequence (engine/model_forward.rs:22-64; models/gemma4_quantized.rs forward): q_len prompt
 * tokens at positions seqlen_offset .. seqlen_offset + q_len - 1 in passes of 16 rows; slot_mapping i32 [q_len], block_table
 * i32 [blocks of the sequence] (device).  K/V rows go to the paged cache, attention is causal over the cache (sliding layers
 * inside their window); only the last row goes through the lm_head: out_id_last (i32 [1]) / logits_last (bf16 [vocab]), either
 * may be NULL. */
int vmx_gemma4_prefill(vmx_gemma4 *m, const int32_t *input_ids, int seqlen_offset, const int32_t *slot_mapping,
                       const int32_t *block_table, int q_len, int32_t *out_id_last, uint16_t *logits_last, void *stream);
int vmx_gemma4_mmvq_only(vmx_gemma4 *m, int batch, void *stream);
/* Measurement aid (no reference counterpart): which projections vmx_gemma4_mmvq_only launches and vmx_gemma4_mmvq_info counts
 * (bit 0 q|k|v, 1 o_proj, 2 gate|up, 3 down_proj; default 15 = all), as vmx_qwen3_linears_select. */
int vmx_gemma4_mmvq_select(vmx_gemma4 *m, int mask);
/* Parity aid: 0 = residual stream h, 1 = final normed x (bf16 [batch, hidden]). */
int vmx_gemma4_debug_read(vmx_gemma4 *m, int which, void *dst, size_t bytes, void *stream);
/* Parity aid: as vmx_qwen3_debug_kv, for a layer's cache in its own geometry (sliding or full). */
/* KVScales::reset on every layer's FP8 cache (kv_cache/quantization.rs:231-239): scales back to 1, not calibrated; the next K/V
 * write of each layer (a decode step, or a prompt chunk of at most 16 tokens) pins them from the bf16 K / V it writes with
 * `headroom` (the reference's default is 1.5), as KVScales::calibrate_if_needed does.  After vmx_gemma4_create the scales are 1
 * and count as calibrated; VMX_G4T_KV_SCALES loads explicit (calibrated) scales. */
/* Attach the tensor-parallel communicator (not owned; rank / world must match the config).  Row-parallel outputs (o_proj, down_proj)
 * are rounded to bf16 per rank and all-reduced, greedy ids come from a vocabulary-parallel argmax; logits_out then holds this rank's
 * [rows, vocab_size / tp_size] slice. */
int vmx_gemma4_set_comm(vmx_gemma4 *m, vmx_comm *comm);
int vmx_gemma4_kv_scales_reset(vmx_gemma4 *m, float headroom, void *stream);
/* the scales layer `layer` attends with (its target's for a KV-shared layer) and whether they are pinned; waits for the stream */
int vmx_gemma4_kv_scales_get(vmx_gemma4 *m, int layer, float *k_scale, float *v_scale, int *calibrated, void *stream);
int vmx_gemma4_debug_kv(vmx_gemma4 *m, int layer, int which, int write, void *buf, size_t bytes, void *stream);


/* ------------------------------------------------------------------ GGUF container (host only) */

/* ---- quantization detection and the safetensors container (host only) ----------------------------------------------
 * Replaces crates/core/src/quantization/detection.rs:82-314 (detect_from_directory / detect_from_config_json /
 * detect_from_quantize_config): config.json's "quantization_config" wins (quant_method, bits, group_size, desc_act,
 * activation_scheme), then quantize_config.json (GPTQ form: bits + group_size, "gptq_marlin" -> Marlin; AWQ form: w_bit,
 * q_group_size), then *.gguf files in the directory.  Fields the files do not state are -1 / "". */
#define VMX_QM_NONE 0
#define VMX_QM_AWQ 1
#define VMX_QM_GPTQ 2
#define VMX_QM_MARLIN 3        /* gptq_marlin */
#define VMX_QM_AWQ_MARLIN 4
#define VMX_QM_EXL3 5
#define VMX_QM_FP8 6
#define VMX_QM_GGUF 7
#define VMX_QM_OTHER 8         /* a method the reference names that has no kernels here (method_name says which) */
typedef struct vmx_quant_info {
    int method;                /* VMX_QM_* */
    int bits, group_size;      /* -1 when absent */
    int desc_act;              /* -1 absent, 0 false, 1 true */
    char method_name[32];      /* the quant_method string as found ("awq", "gptq_marlin", ...) */
    char activation_scheme[32];
} vmx_quant_info;
/* Either text may be NULL (file absent); unparsable text counts as absent, as in the reference. */
int vmx_quant_detect_json(const char *config_json, const char *quantize_config_json, vmx_quant_info *out);
int vmx_quant_detect_dir(const char *model_dir, vmx_quant_info *out);

/* The safetensors file the reference's weight loader reads "<prefix>.qweight / .qzeros / .scales / .trellis ..." from
 * (crates/core/src/quantization/weight_loader.rs:470-559,597-714): u64 LE header length, JSON header, tensor bytes.
 * Header lengths, offsets and shape x dtype sizes are validated against the file; tensor_info strings stay valid until
 * close.  read_tensor copies one tensor's bytes to host memory (the caller uploads them with the executor's load calls). */
typedef struct vmx_safetensors vmx_safetensors;
typedef struct vmx_safetensors_tensor {
    const char *name, *dtype;  /* dtype as spelled in the file: "I32", "F16", "BF16", "U8", ... */
    int element_bytes, n_dims;
    int64_t shape[8];
    uint64_t file_offset, n_bytes;
} vmx_safetensors_tensor;
int vmx_safetensors_open(const char *path, vmx_safetensors **out);
void vmx_safetensors_close(vmx_safetensors *h);
int vmx_safetensors_tensor_count(const vmx_safetensors *h);
int vmx_safetensors_find(const vmx_safetensors *h, const char *name);              /* index in header order, -1 if absent */
int vmx_safetensors_tensor_info(const vmx_safetensors *h, int index, vmx_safetensors_tensor *out);
int vmx_safetensors_read_tensor(const vmx_safetensors *h, int index, void *dst_host, size_t dst_bytes);
const char *vmx_safetensors_metadata(const vmx_safetensors *h, const char *key);   /* "__metadata__" string entry or NULL */

/* Replaces GgufHeader::read / read_tensor_bytes (crates/core/src/quantization/gguf/header.rs:221-321) and the HF ->
 * llama.cpp name map of the GGUF weight loader (gguf/mod.rs:1416-1479).  GGUF v1-3; tensor shapes come back in the logical
 * orientation (on-disk dims reversed: a 2-D weight is [out, in]); n_bytes = elements / block * type_size; tensor data
 * starts at tensor_data_offset (aligned to general.alignment, default 32) + offset.  Unknown GGML type ids, a bad magic
 * and truncated data are errors (VMX_EINVAL / VMX_EUNSUPPORTED with the reference's wording). */
typedef struct vmx_gguf vmx_gguf;
typedef struct vmx_gguf_tensor {
    const char *name;    /* owned by the handle */
    int n_dims;
    uint64_t shape[8];
    int ggml_type;
    uint64_t offset, n_bytes;
} vmx_gguf_tensor;

int vmx_gguf_open(const char *path, vmx_gguf **out);
int vmx_gguf_parse(const void *data, size_t bytes, vmx_gguf **out);   /* header bytes already in memory */
void vmx_gguf_close(vmx_gguf *g);
int vmx_gguf_version(const vmx_gguf *g);
uint64_t vmx_gguf_tensor_data_offset(const vmx_gguf *g);
int vmx_gguf_tensor_count(const vmx_gguf *g);
int vmx_gguf_metadata_count(const vmx_gguf *g);
int vmx_gguf_find_tensor(const vmx_gguf *g, const char *name);        /* index in file order, -1 if absent */
int vmx_gguf_tensor_info(const vmx_gguf *g, int index, vmx_gguf_tensor *out);
int vmx_gguf_read_tensor(const vmx_gguf *g, int index, void *dst_host, size_t dst_bytes);
const char *vmx_gguf_metadata_key(const vmx_gguf *g, int index);
/* value-type id 0..12 of a key (9 = array: *array_len, *elem_type filled), -1 if absent */
int vmx_gguf_metadata_type(const vmx_gguf *g, const char *key, uint64_t *array_len, int *elem_type);
/* array_index < 0: the value itself, else that element of an array value */
int vmx_gguf_metadata_int(const vmx_gguf *g, const char *key, int array_index, int64_t *out);
int vmx_gguf_metadata_float(const vmx_gguf *g, const char *key, int array_index, double *out);
const char *vmx_gguf_metadata_string(const vmx_gguf *g, const char *key, int array_index);
/* "model.layers.3.self_attn.q_proj" -> "blk.3.attn_q"; candidate `which` in priority order; returns the length, 0 if none */
int vmx_gguf_llama_cpp_name(const char *hf_prefix, int which, char *out, size_t out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* VMX_H */

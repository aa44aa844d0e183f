// Internal C++ launch interfaces shared between translation units of libvmx (not part of the C ABI).
#pragma once
#include "vmx_internal.h"

// int4_kernels.hip
int vmx_int4_splits(int N, int K);
int vmx_int4_gemm_grouped(const void *const *packed_table, const int32_t *expert_ids, int num_experts, const uint16_t *x,
                          int x_row_div, float *partial, int pairs, int N, int K, int group, hipStream_t st);
int vmx_int4_gemm_partial(const void *packed, const uint16_t *x, float *partial, int M, int N, int K, int group, hipStream_t st);
// merged gate|up linear with the gated activation in the GEMM epilogue (pack with glu = 1; only when _glu_fusable)
int vmx_int4_pack_cols(const uint32_t *qweight, const uint16_t *scales_bf16, const uint32_t *qzeros, void *packed, int K, int N,
                       int group, int glu, void *stream);
bool vmx_int4_glu_fusable(int N, int K);
int vmx_int4_gemm_grouped_glu(const void *const *packed_table, const int32_t *expert_ids, int num_experts, const uint16_t *x,
                              int x_row_div, uint16_t *act, int pairs, int N, int K, int group, hipStream_t st);
// token-grouped matrix-core form: pairs sorted by expert in blocks of 64 rows (vmx_moe_align_block_size); act != NULL = gated epilogue
// block_rows 64: the 64 x 128-tile kernel (prompts); 16: the small-M kernel with gathered rows (decode batches)
int vmx_int4_gemm_token_grouped(const void *const *packed_table, const int32_t *sorted, const int32_t *block_expert,
                                const int32_t *total_post_pad, int max_blocks, int block_rows, int num_experts, const uint16_t *x,
                                int x_row_div, float *partial, uint16_t *act, int pairs, int N, int K, int group, hipStream_t st);
bool vmx_int4_token_grouped16_ok(int N, int K);
// moe_kernels.hip: vmx_moe_int4_forward with the experts' gate|up tensors packed for the fused activation (w13_glu != 0)
int vmx_moe_int4_forward_glu(uint16_t *out, const uint16_t *x, const float *topk_weights, const int32_t *topk_ids,
                             const void *const *w13_packed, const void *const *w2_packed, int tokens, int k, int num_experts, int hidden,
                             int inter, int group, void *workspace, size_t workspace_bytes, int w13_glu, hipStream_t st,
                             uint16_t *h_resid = nullptr, float *ssq = nullptr);   // h_resid: also h = bf16(h + out) and its tile sums of squares
int vmx_int4_gemm_glu(const void *packed, const uint16_t *x, uint16_t *act, int M, int N, int K, int group, hipStream_t st);
// decode rows (M <= 4): RMSNorm of the residual stream in the prologue (q|k|v, gate|up) and the residual add in the epilogue
// (o_proj, down_proj), so that a dense int4 layer is five launches
int vmx_int4_norm_splits(int M, int N, int K, int glu);   // split-K slices of the _norm forms; 0 = shape not supported
bool vmx_int4_resid_fusable(int M, int N, int K);
// ssq [M][hidden/16] f32: sums of squares of the 16-column tiles of h, written by _resid and read by the _norm forms
int vmx_int4_gemm_partial_norm(const void *packed, const uint16_t *h, const uint16_t *norm_w, float eps, const float *ssq, float *partial,
                               int M, int N, int K, int group, hipStream_t st);
int vmx_int4_gemm_glu_norm(const void *packed, const uint16_t *h, const uint16_t *norm_w, float eps, const float *ssq, uint16_t *act, int M,
                           int N, int K, int group, hipStream_t st);
int vmx_int4_gemm_resid(const void *packed, const uint16_t *x, uint16_t *h, float *ssq, int M, int N, int K, int group, hipStream_t st);
int vmx_awq_convert_into(const uint32_t *awq_qweight, const uint32_t *awq_qzeros, uint32_t *qweight_out, uint32_t *qzeros_out, int K,
                         int N, int group, int Ntot, int col0, hipStream_t st);

struct VmxAttnParts;
int vmx_exl3_forward_multi_f32(const uint16_t *x, int x_bf16, int nseg, const void *const *trellis, const uint16_t *const *suh,
                               const int *n_out, const uint16_t *svh_merged, float *y32, int ld, int M, int K, int bits, int codebook,
                               void *workspace, size_t workspace_bytes, hipStream_t st, const float **partial_out = nullptr,
                               int *splits_out = nullptr,   // y32 == NULL: GEMM only, the consumer transforms *partial_out
                               const uint16_t *norm_w = nullptr, const float *blk_ssq = nullptr, float eps = 0.f,
                               const struct VmxAttnParts *attn_parts = nullptr);
// attn_parts: the input rows are the merge of these decode-attention partitions (128-wide heads, at most 4 partitions), x unused
// norm_w: x is the residual stream and the linear's input is its RMSNorm; blk_ssq [M][K / 128] = sums of squares of x's 128-column blocks

// attention_kernels.hip: q/k/v post-processing + short-context decode attention + merge in one launch
bool vmx_fused_decode_attention_ok(int head_dim, int qpk, int rot_dim, int max_seq_len, int window);
int vmx_fused_decode_attention(uint16_t *out, const float *qkv_partial, int splits, void *k_cache, void *v_cache,
                               const uint16_t *q_norm_w, const uint16_t *k_norm_w, float eps, const float *cos_sin,
                               const int32_t *positions, const int32_t *slots, const int32_t *block_tables, const int32_t *seq_lens,
                               float scale, int num_seqs, int num_heads, int num_kv_heads, int head_dim, int rot_dim,
                               int max_blocks_per_seq, int block_size, int window, int kv_dtype, const float *k_scale,
                               const float *v_scale, int8_t *aq, float *ad, int v_norm, hipStream_t st,   // v_norm: weightless RMSNorm on v
                               const uint16_t *had_svh = nullptr);   // EXL3: qkv_partial is the trellis GEMM before its output transform

// exl3_kernels.hip: Exl3Linear::forward for the whole-token executor; workspace = vmx_exl3_gemm_workspace_bytes(M, K, N)
int vmx_exl3_forward_f32(const uint16_t *x, int x_bf16, const void *trellis, const uint16_t *suh, const uint16_t *svh, float *y32,
                         int ld, int M, int K, int N, int bits, int codebook, void *workspace, size_t workspace_bytes, hipStream_t st);

// prefill_attn.hip: prompt attention on 32 x 32 matrix-core tiles (64- / 128-wide heads; kv: 0 bf16, 1 FP8 e4m3, 3 fp16 cache with
// fp16 queries and output); VMX_EUNSUPPORTED (no message) when the geometry is not covered
int vmx_prefill_attention_tiled(uint16_t *out, const uint16_t *q, const void *k_cache, const void *v_cache, const int32_t *block_table,
                                int q_len, int seqlen_offset, float scale, int num_heads, int num_kv_heads, int head_dim, int block_size,
                                int window, int kv, const float *k_scale, const float *v_scale, hipStream_t st);

// attention_kernels.hip
struct VmxAttnParts {
    float *tmp_out, *exp_sums, *max_logits;  // [S][H][P][hd], [S][H][P], [S][H][P]
    int max_parts, part_tokens;
};
int vmx_paged_attention_parts(const uint16_t *q, const void *k_cache, const void *v_cache, const int32_t *block_tables,
                              const int32_t *seq_lens, float scale, int num_seqs, int num_heads, int num_kv_heads,
                              int max_blocks_per_seq, int max_seq_len, int head_dim, int block_size, int window, int kv_dtype,
                              const float *k_scale, const float *v_scale, void *workspace, VmxAttnParts *out, hipStream_t st);
// aq/ad (optional): also emit the q8_1 form of the output rows [seqs][heads * head_dim] for an I-quant consumer
#define VMX_KV_F16_INTERNAL 3   // vmx_paged_attention_parts only: fp16 queries over an fp16 cache
int vmx_paged_attention_merge(uint16_t *out, const VmxAttnParts *parts, const int32_t *seq_lens, int num_seqs, int num_heads,
                              int head_dim, int window, hipStream_t st, int8_t *aq = nullptr, float *ad = nullptr, int out_f16 = 0);

// attention_kernels.hip: the same work as vmx_fused_decode_attention spread over P workgroups per KV head; the P partition
// results stay in `workspace` (described by *out) for a consumer that merges them (vmx_int4_gemm_resid_attn)
int vmx_split_decode_attention_parts(int head_dim, int qpk, int rot_dim, int max_seq_len, int window);   // 0 = not covered
int vmx_split_decode_attention_parts_wide(int head_dim, int qpk, int rot_dim, int max_seq_len, int window);   // + 256-wide heads, qpk <= 2
size_t vmx_split_decode_attention_workspace_bytes(int num_seqs, int num_heads, int head_dim);
int vmx_split_decode_attention(VmxAttnParts *out, void *workspace, const float *qkv_partial, int splits, void *k_cache, void *v_cache,
                               const uint16_t *q_norm_w, const uint16_t *k_norm_w, float eps, const float *cos_sin,
                               const int32_t *positions, const int32_t *slots, const int32_t *block_tables, const int32_t *seq_lens,
                               float scale, int num_seqs, int num_heads, int num_kv_heads, int head_dim, int rot_dim,
                               int max_blocks_per_seq, int max_seq_len, int block_size, int window, int kv_dtype, const float *k_scale,
                               const float *v_scale, int v_norm, hipStream_t st, const uint16_t *had_svh = nullptr,   // as above
                               int long_parts = 0);   // > 0: that many partitions (vmx_split_decode_attention_long_parts), merged by vmx_paged_attention_merge
int vmx_split_decode_attention_long_parts(int head_dim, int qpk, int rot_dim, int max_seq_len, int window, int num_kv_heads,
                                          int num_seqs);   // 0 = not covered; at most 64
size_t vmx_split_decode_attention_long_workspace_bytes(int num_seqs, int num_heads, int head_dim, int parts);
// int4_kernels.hip: vmx_int4_gemm_resid whose activations are the merge of those partitions (K = heads * head_dim)
int vmx_int4_gemm_resid_attn(const void *packed, const VmxAttnParts *parts, int head_dim, uint16_t *h, float *ssq, int M, int N, int K,
                             int group, hipStream_t st);

// comm.hip: VMX_ELAUNCH (and the message) when the communicator's peer-mapped transport has failed; a host read, no device call
int vmx_comm_check_health(const vmx_comm *c, const char *what);

// glue_kernels.hip
int vmx_argmax_keys(unsigned long long *keys, const uint16_t *logits, int rows, int vocab, uint32_t index_offset, void *workspace,
                    hipStream_t st, int f16 = 0);   // f16: the logits are fp16
int vmx_argmax_keys_to_ids(const unsigned long long *keys, int32_t *out_ids, int rows, hipStream_t st);

// iq_kernels.hip
int vmx_quantize_q8_1_bf16(const uint16_t *x, int8_t *aq, float *ad, int64_t total_blocks, hipStream_t st);
int vmx_iq_mmvq_ld(int ggml_type, const void *w, const int8_t *aq, const float *ad, float *y, int n_out, int n_in, int m, int ldy,
                   hipStream_t st);
struct VmxMmvqSeg {
    const void *w;      // raw I-quant blocks [n_out, n_in]
    int ggml_type, n_out, y_off;
};
int vmx_iq_mmvq_multi(const VmxMmvqSeg *segs, int nseg, const int8_t *aq, const float *ad, float *y, int n_in, int m, int ldy,
                      hipStream_t st);

// iq_packed.hip: the same MMVQ over the load-time repacked form (vmx_iq_pack), one activation row; segs[i].w = packed weights
bool vmx_iq_packed_supported(int ggml_type);
bool vmx_iq_packed_preferred(const VmxMmvqSeg *segs_packed, int nseg, int n_in, int m);   // segs[i].w = packed copy or NULL
int vmx_iq_mmvq_packed_multi(const VmxMmvqSeg *segs, int nseg, const int8_t *aq, const float *ad, float *y, int n_in, hipStream_t st);
// gate|up MMVQ + GeGLU + q8_1 of the result in one launch (packed tensors of one type, one activation row)
bool vmx_iq_packed_geglu_ok(const VmxMmvqSeg *gate_packed, const VmxMmvqSeg *up_packed, int n_in, int m);
// ... and with Gemma's post-attention / pre-feedforward norms and the residual add in the prologue (o: f32 output of o_proj)
bool vmx_iq_packed_norm_geglu_ok(const VmxMmvqSeg *gate_packed, const VmxMmvqSeg *up_packed, int n_in, int m);
int vmx_iq_mmvq_packed_norm_geglu(const VmxMmvqSeg *gate_packed, const VmxMmvqSeg *up_packed, const float *o, const uint16_t *h_in,
                                  uint16_t *h_out, const uint16_t *w_post, const uint16_t *w_next, float eps, uint16_t *act, int8_t *aq_out,
                                  float *ad_out, int n_in, hipStream_t st);
// up to three packed tensors behind the norms of a layer boundary (post-feedforward norm of `o` + residual [* scalar] + the next
// input norm), computed in the prologue like the form above; y + segs[i].y_off receive the f32 outputs
bool vmx_iq_packed_norm_multi_ok(const VmxMmvqSeg *segs_packed, int nseg, int n_in, int m);
int vmx_iq_mmvq_packed_norm_multi(const VmxMmvqSeg *segs_packed, int nseg, const float *o, const uint16_t *h_in, uint16_t *h_out,
                                  const uint16_t *w_post, const uint16_t *scalar, const uint16_t *w_next, float eps, float *y, int n_in,
                                  hipStream_t st);
int vmx_iq_mmvq_packed_geglu(const VmxMmvqSeg *gate_packed, const VmxMmvqSeg *up_packed, const int8_t *aq, const float *ad, uint16_t *act,
                             int8_t *aq_out, float *ad_out, int n_in, hipStream_t st);

// iq_mma.hip: the same MMVQ for 2 .. 16 activation rows on the int8 matrix cores over the 16-row-tile form (vmx_iq_mma_pack)
bool vmx_iq_mma_supported(int ggml_type);
bool vmx_iq_mma_geglu_ok(const VmxMmvqSeg *gate, const VmxMmvqSeg *up, int n_in, int m);
int vmx_iq_mmvq_mma_geglu(const VmxMmvqSeg *gate, const VmxMmvqSeg *up, const int8_t *aq, const float *ad, uint16_t *act, int8_t *aq_out,
                          float *ad_out, int n_in, int m, hipStream_t st);
int vmx_iq_mmvq_mma_multi(const VmxMmvqSeg *segs, int nseg, const int8_t *aq, const float *ad, float *y, int n_in, int m, int ldy, hipStream_t st);

// kquant_kernels.hip
int vmx_kquant_gather_scaled_bf16(int ggml_type, const void *table, const int32_t *ids, uint16_t *out, int tokens, int dim, int vocab,
                                  float scale, hipStream_t st);

// dense_kernels.hip
// split_gemm.hip: prompt-sized I-quant linear = dequantise once (bf16 hi/lo halves in `workspace`), then C = A @ W^T in f32
size_t vmx_iq_gemm_workspace(int N, int K);
int vmx_iq_dequantize_split_bf16(int ggml_type, const void *blocks, uint16_t *hi, uint16_t *lo, int64_t n_blocks, hipStream_t st);
int vmx_iq_dequant_split(int ggml_type, const void *blocks, void *workspace, int N, int K, hipStream_t st);
int vmx_split_gemm_f32(const uint16_t *A, const void *workspace, float *C, int M, int N, int K, int ldc, hipStream_t st);
// the exact one-pass form (weights = f32 super-block scale x fp16 integer; IQ2_XS, IQ2_S, IQ3_XXS, IQ3_S)
bool vmx_iq_int_form_ok(int ggml_type);
int vmx_iq_dequantize_int_f16(int ggml_type, const void *blocks, uint16_t *Q, float *Dt, int rows, int nsb, hipStream_t st);
size_t vmx_iq_gemm_int_workspace(int M, int N, int K);
int vmx_iq_dequant_int(int ggml_type, const void *blocks, void *workspace, int N, int K, hipStream_t st);
int vmx_bf16_rows_to_f16(const uint16_t *A, uint16_t *a16, int64_t n, hipStream_t st);
int vmx_int_gemm_f32(const uint16_t *a16, const void *workspace, float *C, int M, int N, int K, int ldc, hipStream_t st);
int vmx_dense_gemv_launch(uint16_t *y, const uint16_t *x, const uint16_t *W, const uint16_t *bias, int M, int N, int K, hipStream_t st,
                          int f16 = 0);   // f16: x, W, bias and y hold fp16 (same f32 accumulation, fp16 matrix / dot instructions)
// bf16 matrix-core GEMM y = bf16(bf16(x @ W^T) + bias), W [N, K] bf16 (fp8_kernels.hip)
int vmx_bf16_gemm_nt(uint16_t *y, const uint16_t *x, const uint16_t *w, const uint16_t *bias, int M, int N, int K, hipStream_t st);

// Q3_K table x 2..16 bf16 rows on the matrix cores (kquant_kernels.hip); f32 out [m][n_out]
bool vmx_q3k_gemm16_ok(int n_out, int n_in, int m);
int vmx_q3k_gemm16_bf16(const void *w, const uint16_t *x, float *y, int n_out, int n_in, int m, hipStream_t st);

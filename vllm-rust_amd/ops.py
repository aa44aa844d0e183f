"""Thin torch-tensor front-ends over the C ABI (one per entry point).

Every function takes CUDA(HIP) tensors, validates only what ctypes cannot, and
launches on torch's current stream.  Shape/dtype preconditions are enforced by
the library itself and surface as VmxError with the library's message.
"""
from __future__ import annotations

import torch

from ._lib import check, lib

__all__ = [
    "init", "iq_type_size", "iq_dequantize", "quantize_q8_1", "iq_mmvq", "iq_matmul", "iq_gemm", "iq_subdots", "iq_pack", "iq_mmvq_packed", "iq_mmvq_packed_geglu", "iq_subdots_packed", "iq_mma_pack", "iq_mmvq_mma", "iq_subdots_mma",
    "gather_dequant_q6k", "awq_to_gptq_qweight", "awq_repack_nibbles", "int4_pack", "int4_gemm", "int4_dequant",
    "rms_norm", "rope_neox", "rope_gptj", "reshape_and_cache_hnd", "copy_blocks", "silu_and_mul", "argmax", "KVScales", "reshape_and_cache", "paged_attention", "paged_attention_f16", "paged_attention_static", "paged_attention_with_lse", "lse_correct", "dense_gemv",
    "embedding_lookup", "fp8_gemm", "fp8_dequant", "fp8_quant", "fp8_scaled_mm", "fp8_dequant_block", "fp8_block_gemm", "fused_rmsnorm_fp8",
    "topk_softmax", "moe_align_block_size", "moe_int4_forward", "moe_ep_int4_forward",
]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def init(device: int = 0) -> None:
    check(lib.vmx_init(device))
    torch.cuda.set_device(device)


def iq_type_size(ggml_type: int) -> int:
    return lib.vmx_iq_type_size(ggml_type)


def iq_dequantize(ggml_type: int, blocks: torch.Tensor, n_blocks: int) -> torch.Tensor:
    out = torch.empty(n_blocks * 256, dtype=torch.float32, device=blocks.device)
    check(lib.vmx_iq_dequantize_f32(ggml_type, _ptr(blocks), _ptr(out), n_blocks, _stream()))
    return out


def quantize_q8_1(x: torch.Tensor):
    assert x.dtype == torch.float32 and x.is_contiguous()
    m, n_in = x.shape
    aq = torch.empty((m, n_in), dtype=torch.int8, device=x.device)
    ad = torch.empty((m, n_in // 32), dtype=torch.float32, device=x.device)
    check(lib.vmx_quantize_q8_1_rows(_ptr(x), _ptr(aq), _ptr(ad), m * (n_in // 32), _stream()))
    return aq, ad


def iq_mmvq(ggml_type: int, w: torch.Tensor, aq: torch.Tensor, ad: torch.Tensor, n_out: int, n_in: int, out=None):
    m = aq.shape[0]
    y = out if out is not None else torch.empty((m, n_out), dtype=torch.float32, device=w.device)
    check(lib.vmx_iq_mmvq_f32(ggml_type, _ptr(w), _ptr(aq), _ptr(ad), _ptr(y), n_out, n_in, m, _stream()))
    return y


def iq_pack(ggml_type: int, raw: torch.Tensor, n_out: int, n_in: int) -> torch.Tensor:
    """Load-time repack of raw I-quant blocks [n_out, n_in] (uint8 CUDA tensor) into the wave-ordered stream of iq_mmvq_packed."""
    need = lib.vmx_iq_packed_bytes(ggml_type, n_out, n_in)
    packed = torch.empty(max(need, 16), dtype=torch.uint8, device=raw.device)
    check(lib.vmx_iq_pack(ggml_type, _ptr(raw), _ptr(packed), n_out, n_in, _stream()))
    return packed


def iq_mmvq_packed(ggml_type: int, packed: torch.Tensor, aq: torch.Tensor, ad: torch.Tensor, n_out: int, n_in: int, out=None):
    """One activation row (aq int8 [1, n_in], ad f32 [1, n_in / 32]) against packed weights -> f32 [1, n_out]."""
    y = out if out is not None else torch.empty((1, n_out), dtype=torch.float32, device=packed.device)
    check(lib.vmx_iq_mmvq_packed_f32(ggml_type, _ptr(packed), _ptr(aq), _ptr(ad), _ptr(y), n_out, n_in, _stream()))
    return y


def iq_mmvq_packed_geglu(ggml_type: int, gate_packed: torch.Tensor, up_packed: torch.Tensor, aq: torch.Tensor, ad: torch.Tensor,
                         inter: int, n_in: int):
    """gate|up q8_1 MMVQ + GeGLU + q8_1 of the result in one launch -> (act bf16 bits [inter], act_q int8 [inter], act_d f32 [inter/32])."""
    dev = gate_packed.device
    act = torch.empty(inter, dtype=torch.int16, device=dev)
    act_q = torch.empty(inter, dtype=torch.int8, device=dev)
    act_d = torch.empty(inter // 32, dtype=torch.float32, device=dev)
    check(lib.vmx_iq_mmvq_packed_geglu_bf16(ggml_type, _ptr(gate_packed), _ptr(up_packed), _ptr(aq), _ptr(ad), _ptr(act), _ptr(act_q),
                                            _ptr(act_d), inter, n_in, _stream()))
    return act, act_q, act_d


def iq_subdots_packed(ggml_type: int, packed: torch.Tensor, aq: torch.Tensor, n_out: int, n_in: int) -> torch.Tensor:
    out = torch.zeros((n_out, 1, n_in // 32), dtype=torch.int32, device=packed.device)
    check(lib.vmx_iq_subdots_packed_i32(ggml_type, _ptr(packed), _ptr(aq), _ptr(out), n_out, n_in, _stream()))
    return out


def iq_mma_pack(ggml_type: int, raw: torch.Tensor, n_out: int, n_in: int) -> torch.Tensor:
    """Load-time repack of raw I-quant blocks [n_out, n_in] into 16-row tiles for the matrix-core MMVQ (iq_mmvq_mma)."""
    need = lib.vmx_iq_mma_packed_bytes(ggml_type, n_out, n_in)
    packed = torch.empty(max(need, 16), dtype=torch.uint8, device=raw.device)
    check(lib.vmx_iq_mma_pack(ggml_type, _ptr(raw), _ptr(packed), n_out, n_in, _stream()))
    return packed


def iq_mmvq_mma(ggml_type: int, packed: torch.Tensor, aq: torch.Tensor, ad: torch.Tensor, n_out: int, n_in: int):
    """1 .. 16 activation rows (aq int8 [m, n_in], ad f32 [m, n_in / 32]) against 16-row-tile weights -> f32 [m, n_out]."""
    m = aq.shape[0]
    y = torch.empty((m, n_out), dtype=torch.float32, device=packed.device)
    check(lib.vmx_iq_mmvq_mma_f32(ggml_type, _ptr(packed), _ptr(aq), _ptr(ad), _ptr(y), n_out, n_in, m, _stream()))
    return y


def iq_subdots_mma(ggml_type: int, packed: torch.Tensor, aq: torch.Tensor, ad: torch.Tensor, n_out: int, n_in: int) -> torch.Tensor:
    m = aq.shape[0]
    out = torch.zeros((n_out, m, n_in // 32), dtype=torch.int32, device=packed.device)
    check(lib.vmx_iq_subdots_mma_i32(ggml_type, _ptr(packed), _ptr(aq), _ptr(ad), _ptr(out), n_out, n_in, m, _stream()))
    return out


def iq_matmul(ggml_type: int, w: torch.Tensor, x: torch.Tensor, n_out: int, n_in: int, workspace=None):
    m = x.shape[0]
    need = lib.vmx_iq_matmul_workspace_bytes(n_in, max(m, 1))
    ws = workspace if workspace is not None else torch.empty(max(need, 1), dtype=torch.uint8, device=w.device)
    y = torch.empty((m, n_out), dtype=torch.float32, device=w.device)
    check(lib.vmx_iq_matmul_f32(ggml_type, _ptr(w), _ptr(x), _ptr(y), n_out, n_in, m, _ptr(ws), ws.numel(), _stream()))
    return y


def iq_gemm(ggml_type: int, blocks: torch.Tensor, x: torch.Tensor, n_out: int) -> torch.Tensor:
    """Prompt-sized I-quant linear (more than 16 rows): x bf16 [m, n_in] times the dequantised weight, f32 [m, n_out]."""
    m, n_in = x.shape
    need = lib.vmx_iq_gemm_workspace_bytes(n_out, n_in)
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    y = torch.empty((m, n_out), dtype=torch.float32, device=x.device)
    check(lib.vmx_iq_gemm_f32(ggml_type, _ptr(blocks), _ptr(x), _ptr(y), m, n_out, n_in, _ptr(ws), need, _stream()))
    return y


def iq_subdots(ggml_type: int, w: torch.Tensor, aq: torch.Tensor, n_out: int, n_in: int) -> torch.Tensor:
    m = aq.shape[0]
    out = torch.empty((n_out, m, n_in // 32), dtype=torch.int32, device=w.device)
    check(lib.vmx_iq_subdots_i32(ggml_type, _ptr(w), _ptr(aq), _ptr(out), n_out, n_in, m, _stream()))
    return out


def gather_dequant_q6k(blocks: torch.Tensor, ids: torch.Tensor, dim: int, num_embeddings: int) -> torch.Tensor:
    out = torch.empty((ids.numel(), dim), dtype=torch.float32, device=blocks.device)
    check(lib.vmx_gather_dequant_q6k_f32(_ptr(blocks), _ptr(ids), _ptr(out), ids.numel(), dim, num_embeddings, _stream()))
    return out


def awq_to_gptq_qweight(awq: torch.Tensor, K: int, N: int) -> torch.Tensor:
    out = torch.empty((K // 8, N), dtype=torch.int32, device=awq.device)
    check(lib.vmx_awq_to_gptq_qweight(_ptr(awq), _ptr(out), K, N, _stream()))
    return out


def awq_repack_nibbles(words: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(words)
    check(lib.vmx_awq_repack_nibbles(_ptr(words), _ptr(out), words.numel(), _stream()))
    return out


def int4_pack(qweight: torch.Tensor, scales_bf16: torch.Tensor, qzeros, K: int, N: int, group: int) -> torch.Tensor:
    nbytes = lib.vmx_int4_packed_bytes(K, N, group)
    packed = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=qweight.device)
    check(lib.vmx_int4_pack(_ptr(qweight), _ptr(scales_bf16), _ptr(qzeros), _ptr(packed), K, N, group, _stream()))
    return packed


def int4_gemm(x_bf16: torch.Tensor, packed: torch.Tensor, bias, N: int, K: int, group: int, workspace=None, out=None):
    M = x_bf16.shape[0]
    need = lib.vmx_int4_gemm_workspace_bytes(max(M, 1), N, K)
    ws = workspace if workspace is not None else torch.empty(max(need, 1), dtype=torch.uint8, device=packed.device)
    y = out if out is not None else torch.empty((M, N), dtype=torch.bfloat16, device=packed.device)
    check(lib.vmx_int4_gemm_bf16(_ptr(y), _ptr(x_bf16), _ptr(packed), _ptr(bias), M, N, K, group, _ptr(ws), ws.numel(),
                                 _stream()))
    return y


def int4_dequant(packed: torch.Tensor, K: int, N: int, group: int) -> torch.Tensor:
    dense = torch.empty((K, N), dtype=torch.bfloat16, device=packed.device)
    check(lib.vmx_int4_dequant_bf16(_ptr(packed), _ptr(dense), K, N, group, _stream()))
    return dense


def rms_norm(x: torch.Tensor, weight, eps: float, residual=None, weight_plus_one: bool = False) -> torch.Tensor:
    """x bf16 [tokens, hidden]; with `residual` it is updated in place (fused add)."""
    out = torch.empty_like(x)
    tokens, hidden = x.shape
    check(lib.vmx_rms_norm_bf16(_ptr(out), _ptr(x), _ptr(residual), _ptr(weight), eps, tokens, hidden, int(weight_plus_one),
                                _stream()))
    return out


def rope_neox(positions: torch.Tensor, q: torch.Tensor, k, cos_sin: torch.Tensor, rot_dim: int, head_size: int,
              num_heads: int, num_kv_heads: int) -> None:
    check(lib.vmx_rope_neox_bf16(_ptr(positions), _ptr(q), _ptr(k), _ptr(cos_sin), positions.numel(), rot_dim, head_size,
                                 num_heads, num_kv_heads, _stream()))


def rope_gptj(positions: torch.Tensor, q: torch.Tensor, k, cos_sin: torch.Tensor, rot_dim: int, head_size: int,
              num_heads: int, num_kv_heads: int) -> None:
    check(lib.vmx_rope_gptj_bf16(_ptr(positions), _ptr(q), _ptr(k), _ptr(cos_sin), positions.numel(), rot_dim, head_size,
                                 num_heads, num_kv_heads, _stream()))


def silu_and_mul(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(gate)
    check(lib.vmx_silu_and_mul_bf16(_ptr(out), _ptr(gate), _ptr(up), gate.numel(), _stream()))
    return out


def argmax(logits: torch.Tensor) -> torch.Tensor:
    rows, vocab = logits.shape
    ws = torch.empty(max(lib.vmx_argmax_workspace_bytes(rows, vocab), 8), dtype=torch.uint8, device=logits.device)
    out = torch.empty(rows, dtype=torch.int32, device=logits.device)
    check(lib.vmx_argmax_bf16(_ptr(out), _ptr(logits), rows, vocab, _ptr(ws), ws.numel(), _stream()))
    return out


class KVScales:
    """Device-resident per-tensor FP8 KV scales with the reference's first-write calibration (KVScales,
    crates/core/src/kv_cache/quantization.rs:80-240).  `k_scale` / `v_scale` are one-element f32 views into the state that the
    cache-write and attention entry points take as their scale pointers."""
    DEFAULT_HEADROOM = 1.5          # DEFAULT_KV_HEADROOM_FACTOR, :108

    def __init__(self, device="cuda", k_scale: float = 1.0, v_scale: float = 1.0, calibrated: bool = False):
        self.state = torch.zeros(lib.vmx_kv_scales_state_bytes() // 4, dtype=torch.float32, device=device)
        self.headroom = self.DEFAULT_HEADROOM
        self.set(k_scale, v_scale, calibrated)

    def set(self, k_scale: float, v_scale: float, calibrated: bool = True) -> None:
        check(lib.vmx_kv_scales_init(_ptr(self.state), k_scale, v_scale, int(calibrated), _stream()))

    def reset(self) -> None:
        self.set(1.0, 1.0, False)

    def calibrate_if_needed(self, k: torch.Tensor, v: torch.Tensor) -> None:
        check(lib.vmx_kv_scales_calibrate_if_needed(_ptr(self.state), _ptr(k), k.numel(), _ptr(v), v.numel(), self.headroom, _stream()))

    @property
    def k_scale(self) -> torch.Tensor:
        return self.state[0:1]

    @property
    def v_scale(self) -> torch.Tensor:
        return self.state[1:2]

    @property
    def is_calibrated(self) -> bool:
        return bool(self.state[2:3].view(torch.int32).item())


def reshape_and_cache(key, value, k_cache, v_cache, slots, kv_heads: int, head_dim: int, block_size: int,
                      kv_dtype: int = 0, k_scale=None, v_scale=None) -> None:
    check(lib.vmx_reshape_and_cache(_ptr(key), _ptr(value), _ptr(k_cache), _ptr(v_cache), _ptr(slots), slots.numel(),
                                    kv_heads, head_dim, block_size, kv_dtype, _ptr(k_scale), _ptr(v_scale), _stream()))


def paged_attention_f16(q, k_cache, v_cache, block_tables, seq_lens, scale: float, num_heads: int, num_kv_heads: int, head_dim: int,
                        block_size: int, max_seq_len: int, window: int = 0, workspace=None) -> torch.Tensor:
    """fp16 queries over an fp16 paged cache -> fp16 [seqs, heads * head_dim]."""
    num_seqs = seq_lens.numel()
    need = lib.vmx_paged_attention_workspace_bytes(num_seqs, num_heads, head_dim, max_seq_len)
    if workspace is None:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=q.device)
    out = torch.empty((num_seqs, num_heads * head_dim), dtype=torch.float16, device=q.device)
    check(lib.vmx_paged_attention_f16(_ptr(out), _ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(block_tables), _ptr(seq_lens), scale, num_seqs,
                                      num_heads, num_kv_heads, block_tables.shape[1], max_seq_len, head_dim, block_size, window,
                                      _ptr(workspace), workspace.numel(), _stream()))
    return out


def reshape_and_cache_hnd(key, value, k_cache, v_cache, slots, kv_heads: int, head_dim: int, block_size: int) -> None:
    """Scatter into caches laid out [blocks, kv_heads, block_size, head_dim] (16-bit elements)."""
    check(lib.vmx_reshape_and_cache_hnd(_ptr(key), _ptr(value), _ptr(k_cache), _ptr(v_cache), _ptr(slots), slots.numel(), kv_heads,
                                        head_dim, block_size, _stream()))


def copy_blocks(k_cache, v_cache, block_mapping: torch.Tensor, bytes_per_block: int) -> None:
    """block_mapping int32 [pairs, 2] = (src, dst) block indices; copies whole blocks inside both caches."""
    check(lib.vmx_copy_blocks(_ptr(k_cache), _ptr(v_cache), _ptr(block_mapping), block_mapping.numel() // 2, int(bytes_per_block), _stream()))


def paged_attention(q, k_cache, v_cache, block_tables, seq_lens, scale: float, num_heads: int, num_kv_heads: int,
                    head_dim: int, block_size: int, max_seq_len: int, window: int = 0, kv_dtype: int = 0, k_scale=None,
                    v_scale=None, workspace=None) -> torch.Tensor:
    num_seqs = seq_lens.numel()
    need = lib.vmx_paged_attention_workspace_bytes(num_seqs, num_heads, head_dim, max_seq_len)
    ws = workspace if workspace is not None else torch.empty(max(need, 8), dtype=torch.uint8, device=q.device)
    out = torch.zeros((num_seqs, num_heads * head_dim), dtype=torch.bfloat16, device=q.device)
    check(lib.vmx_paged_attention_bf16(_ptr(out), _ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(block_tables), _ptr(seq_lens),
                                       scale, num_seqs, num_heads, num_kv_heads, block_tables.shape[1], max_seq_len, head_dim,
                                       block_size, window, kv_dtype, _ptr(k_scale), _ptr(v_scale), _ptr(ws), ws.numel(),
                                       _stream()))
    return out


def paged_attention_with_lse(q, k_cache, v_cache, block_tables, seq_lens, scale: float, num_heads: int, num_kv_heads: int,
                             head_dim: int, block_size: int, max_seq_len: int, window: int = 0, kv_dtype: int = 0, k_scale=None,
                             v_scale=None, workspace=None):
    """Decode attention plus lse f32 [seqs, heads] = ln sum_i exp(scale * q . k_i) per head (-inf for an empty context)."""
    num_seqs = seq_lens.numel()
    need = lib.vmx_paged_attention_workspace_bytes(num_seqs, num_heads, head_dim, max_seq_len)
    ws = workspace if workspace is not None else torch.empty(max(need, 8), dtype=torch.uint8, device=q.device)
    out = torch.zeros((num_seqs, num_heads * head_dim), dtype=torch.bfloat16, device=q.device)
    lse = torch.empty((num_seqs, num_heads), dtype=torch.float32, device=q.device)
    check(lib.vmx_paged_attention_bf16_with_lse(_ptr(out), _ptr(lse), _ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(block_tables),
                                                _ptr(seq_lens), scale, num_seqs, num_heads, num_kv_heads, block_tables.shape[1],
                                                max_seq_len, head_dim, block_size, window, kv_dtype, _ptr(k_scale), _ptr(v_scale),
                                                _ptr(ws), ws.numel(), _stream()))
    return out, lse


def lse_correct(attn_out: torch.Tensor, lse_all: torch.Tensor, rank: int, head_dim: int) -> torch.Tensor:
    """This rank's share of the context-parallel merge: attn_out bf16 [rows * head_dim], lse_all f32 [ranks, rows] -> f32
    [rows, head_dim] = attn_out * exp(lse_all[rank] - logsumexp over ranks); summing the shares over the ranks merges them."""
    ranks, rows = lse_all.shape[0], lse_all[0].numel()
    out = torch.empty((rows, head_dim), dtype=torch.float32, device=attn_out.device)
    check(lib.vmx_lse_correct_f32(_ptr(out), _ptr(attn_out), _ptr(lse_all), ranks, rank, rows, head_dim, _stream()))
    return out


def paged_attention_static(q, k_cache, v_cache, block_tables, seq_lens, scale: float, num_heads: int, num_kv_heads: int,
                           head_dim: int, block_size: int, max_seq_len: int, window: int = 0, kv_dtype: int = 0, k_scale=None,
                           v_scale=None, out=None) -> torch.Tensor:
    """vmx_paged_attention_bf16_static: split-KV partials in the library's static per-GPU array (no workspace, no allocation
    when `out` is given)."""
    num_seqs = seq_lens.numel()
    if out is None:
        out = torch.zeros((num_seqs, num_heads * head_dim), dtype=torch.bfloat16, device=q.device)
    check(lib.vmx_paged_attention_bf16_static(_ptr(out), _ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(block_tables), _ptr(seq_lens),
                                              scale, num_seqs, num_heads, num_kv_heads, block_tables.shape[1], max_seq_len, head_dim,
                                              block_size, window, kv_dtype, _ptr(k_scale), _ptr(v_scale), _stream()))
    return out


def dense_gemv(x: torch.Tensor, W: torch.Tensor, bias=None) -> torch.Tensor:
    """x bf16 [M, K], W bf16 [N, K] -> bf16 [M, N]."""
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    check(lib.vmx_dense_gemv_bf16(_ptr(y), _ptr(x), _ptr(W), _ptr(bias), M, N, K, _stream()))
    return y


def embedding_lookup(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    vocab, hidden = table.shape
    out = torch.empty((ids.numel(), hidden), dtype=torch.bfloat16, device=table.device)
    check(lib.vmx_embedding_lookup_bf16(_ptr(out), _ptr(table), _ptr(ids), ids.numel(), hidden, vocab, _stream()))
    return out


def fp8_gemm(x: torch.Tensor, W_u8: torch.Tensor, scale: torch.Tensor, bias=None) -> torch.Tensor:
    """x bf16 [M, K], W e4m3 bytes [N, K], scale f32 [1] or [N] -> bf16 [M, N]."""
    M, K = x.shape
    N = W_u8.shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    check(lib.vmx_fp8_gemm_bf16(_ptr(y), _ptr(x), _ptr(W_u8), _ptr(scale), _ptr(bias), M, N, K, int(scale.numel() > 1), _stream()))
    return y


def fp8_scaled_mm(a_u8: torch.Tensor, b_u8: torch.Tensor, a_scale: torch.Tensor, b_scale: torch.Tensor, bias=None,
                  out_dtype=torch.bfloat16) -> torch.Tensor:
    """W8A8: a e4m3 bytes [M, K], b e4m3 bytes [N, K]; a_scale f32 [1] or [M], b_scale f32 [1] or [N]; -> bf16 or f32 [M, N]."""
    M, K = a_u8.shape
    N = b_u8.shape[0]
    out = torch.empty((M, N), dtype=out_dtype, device=a_u8.device)
    check(lib.vmx_fp8_scaled_mm(_ptr(out), int(out_dtype == torch.float32), _ptr(a_u8), _ptr(b_u8), _ptr(a_scale), _ptr(b_scale), _ptr(bias),
                                M, N, K, int(a_scale.numel() > 1), int(b_scale.numel() > 1), _stream()))
    return out


def fp8_dequant_block(q_u8: torch.Tensor, block_scale: torch.Tensor, block_n: int, block_k: int) -> torch.Tensor:
    rows, cols = q_u8.shape
    out = torch.empty((rows, cols), dtype=torch.bfloat16, device=q_u8.device)
    check(lib.vmx_fp8_dequant_block_bf16(_ptr(out), _ptr(q_u8), _ptr(block_scale), rows, cols, block_n, block_k, _stream()))
    return out


def fp8_block_gemm(x: torch.Tensor, W_u8: torch.Tensor, block_scale: torch.Tensor, block_n: int, block_k: int, bias=None) -> torch.Tensor:
    """Block-scaled FP8 linear: x bf16 [M, K], W e4m3 bytes [N, K], block_scale f32 [ceil(N/block_n), ceil(K/block_k)]."""
    M, K = x.shape
    N = W_u8.shape[0]
    ws = torch.empty((N, K), dtype=torch.bfloat16, device=x.device)
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    check(lib.vmx_fp8_block_gemm_bf16(_ptr(y), _ptr(x), _ptr(W_u8), _ptr(block_scale), _ptr(bias), M, N, K, block_n, block_k, _ptr(ws),
                                      _stream()))
    return y


def fused_rmsnorm_fp8(x: torch.Tensor, weight: torch.Tensor, eps: float):
    """(e4m3 bytes [rows, hidden], f32 scales [rows]) of rms_norm(x) * weight with a dynamic per-row scale."""
    rows, hidden = x.shape
    out = torch.empty((rows, hidden), dtype=torch.uint8, device=x.device)
    scales = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib.vmx_fused_rmsnorm_fp8(_ptr(out), _ptr(scales), _ptr(x), _ptr(weight), rows, hidden, eps, _stream()))
    return out, scales


def fp8_dequant(q_u8: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    rows, cols = q_u8.shape
    out = torch.empty((rows, cols), dtype=torch.bfloat16, device=q_u8.device)
    check(lib.vmx_fp8_dequant_bf16(_ptr(out), _ptr(q_u8), _ptr(scale), rows, cols, int(scale.numel() > 1), _stream()))
    return out


def fp8_quant(x: torch.Tensor, scale=None):
    """Dynamic per-token when scale is None, else static (scale f32 [1] or [tokens])."""
    T, H = x.shape
    out = torch.empty((T, H), dtype=torch.uint8, device=x.device)
    scales = torch.empty(T, dtype=torch.float32, device=x.device)
    check(lib.vmx_fp8_quant_bf16(_ptr(out), _ptr(scales), _ptr(x), _ptr(scale), T, H, int(scale is None),
                                 int(scale is not None and scale.numel() > 1), _stream()))
    return out, scales


def topk_softmax(logits: torch.Tensor, k: int, renormalize: bool = True, round_probs_bf16: bool = False):
    T, E = logits.shape
    dt = 0 if logits.dtype == torch.float32 else 1
    w = torch.empty((T, k), dtype=torch.float32, device=logits.device)
    ids = torch.empty((T, k), dtype=torch.int32, device=logits.device)
    check(lib.vmx_topk_softmax(_ptr(logits), dt, _ptr(w), _ptr(ids), T, E, k, int(renormalize), int(round_probs_bf16), _stream()))
    return w, ids


def moe_align_block_size(topk_ids: torch.Tensor, num_experts: int, block_size: int):
    numel = topk_ids.numel()
    cap = numel + num_experts * (block_size - 1)
    sorted_ids = torch.empty(cap, dtype=torch.int32, device=topk_ids.device)
    expert_ids = torch.full(((cap + block_size - 1) // block_size,), -1, dtype=torch.int32, device=topk_ids.device)
    total = torch.zeros(1, dtype=torch.int32, device=topk_ids.device)
    check(lib.vmx_moe_align_block_size(_ptr(topk_ids), _ptr(sorted_ids), _ptr(expert_ids), _ptr(total), numel, num_experts,
                                       block_size, cap, _stream()))
    return sorted_ids, expert_ids, total


def moe_int4_forward(x, topk_weights, topk_ids, w13_table, w2_table, num_experts: int, hidden: int, inter: int, group: int):
    """w13_table / w2_table: int64 CUDA tensors holding the experts' packed-buffer device pointers."""
    T, k = topk_ids.shape
    ws = torch.empty(lib.vmx_moe_int4_workspace_bytes(T, k, hidden, inter), dtype=torch.uint8, device=x.device)
    out = torch.empty((T, hidden), dtype=torch.bfloat16, device=x.device)
    check(lib.vmx_moe_int4_forward(_ptr(out), _ptr(x), _ptr(topk_weights), _ptr(topk_ids), _ptr(w13_table), _ptr(w2_table), T, k,
                                   num_experts, hidden, inter, group, _ptr(ws), ws.numel(), _stream()))
    return out


def moe_ep_int4_forward(x, topk_weights, topk_ids, w13_local, w2_local, num_experts: int, hidden: int, inter: int, group: int, comm,
                        cap_rows: int, round_robin: bool = False, w13_glu: bool = False):
    """Expert-parallel MoE layer with token dispatch: this rank's tokens x [T, hidden], the rank's experts in w13_local /
    w2_local (int64 CUDA tensors of packed-buffer pointers, local order); comm: a Communicator with the exchange transport."""
    T, k = topk_ids.shape
    ws = torch.empty(lib.vmx_moe_ep_workspace_bytes(T, k, hidden, inter, cap_rows), dtype=torch.uint8, device=x.device)
    out = torch.empty((T, hidden), dtype=torch.bfloat16, device=x.device)
    check(lib.vmx_moe_ep_int4_forward(_ptr(out), _ptr(x), _ptr(topk_weights), _ptr(topk_ids), _ptr(w13_local), _ptr(w2_local), T, k,
                                      num_experts, hidden, inter, group, int(w13_glu), int(round_robin), comm.handle, int(cap_rows),
                                      _ptr(ws), ws.numel(), _stream()))
    return out

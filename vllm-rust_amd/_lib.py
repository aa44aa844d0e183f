"""ctypes binding of lib/libvmx.so (the C ABI declared in include/vmx.h).

The library is the product: if it is missing this module raises at import and
nothing in the package falls back to another implementation.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

# torch ships its own HIP runtime; it must be the one already resident when libvmx.so (linked against
# libamdhip64 by soname) is loaded, otherwise the process ends up with a runtime that sees no device.
import torch  # noqa: F401  (plumbing: device memory, streams)

_PKG = Path(__file__).resolve().parent
import os

# VMX_LIB_VARIANT=name loads lib/libvmx_name.so instead: A/B builds of one kernel for measurement (development aid)
LIB_PATH = _PKG / "lib" / ("libvmx_%s.so" % os.environ["VMX_LIB_VARIANT"] if os.environ.get("VMX_LIB_VARIANT") else "libvmx.so")

if not LIB_PATH.exists():
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C {_PKG / 'csrc'}` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no fallback path"
    )

lib = C.CDLL(str(LIB_PATH))

_i, _i64, _f, _p, _sz = C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_size_t

_SIGS = {
    "vmx_init": (_i, [_i]),
    "vmx_abi_version": (_i, []),
    "vmx_last_error": (C.c_char_p, []),
    "vmx_iq_type_size": (_i, [_i]),
    "vmx_iq_dequantize_f32": (_i, [_i, _p, _p, _i64, _p]),
    "vmx_iq_gemm_workspace_bytes": (_sz, [_i, _i]),
    "vmx_iq_gemm_f32": (_i, [_i, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "vmx_quantize_q8_1_rows": (_i, [_p, _p, _p, _i64, _p]),
    "vmx_iq_mmvq_f32": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "vmx_iq_matmul_workspace_bytes": (_sz, [_i, _i]),
    "vmx_iq_matmul_f32": (_i, [_i, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "vmx_iq_subdots_i32": (_i, [_i, _p, _p, _p, _i, _i, _i, _p]),
    "vmx_iq_packed_bytes": (_sz, [_i, _i, _i]),
    "vmx_iq_pack": (_i, [_i, _p, _p, _i, _i, _p]),
    "vmx_iq_mmvq_packed_f32": (_i, [_i, _p, _p, _p, _p, _i, _i, _p]),
    "vmx_iq_mmvq_packed_geglu_bf16": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "vmx_debug_iq_packed_stamps": (_i, [_p]),
    "vmx_iq_mma_packed_bytes": (_sz, [_i, _i, _i]),
    "vmx_iq_mma_pack": (_i, [_i, _p, _p, _i, _i, _p]),
    "vmx_iq_mmvq_mma_f32": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "vmx_iq_subdots_mma_i32": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "vmx_debug_iq_mma_min_rows": (_i, [_i]),
    "vmx_debug_gemma4_mma_geglu": (_i, [_i]),
    "vmx_debug_gemma4_split_always": (_i, [_i]),
    "vmx_iq_subdots_packed_i32": (_i, [_i, _p, _p, _p, _i, _i, _p]),
    "vmx_gather_dequant_q6k_f32": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "vmx_ggml_type_size": (_i, [_i]),
    "vmx_ggml_block_elems": (_i, [_i]),
    "vmx_kquant_dequantize_f32": (_i, [_i, _p, _p, _i64, _p]),
    "vmx_kquant_gemv_f32": (_i, [_i, _p, _p, _p, _i, _i, _i, _p]),
    "vmx_q3k_rows16_bf16": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "vmx_kquant_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "vmx_kquant_gemm_f32": (_i, [_i, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "vmx_awq_to_gptq_qweight": (_i, [_p, _p, _i, _i, _p]),
    "vmx_awq_repack_nibbles": (_i, [_p, _p, _i64, _p]),
    "vmx_int4_packed_bytes": (_sz, [_i, _i, _i]),
    "vmx_int4_pack": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "vmx_int4_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "vmx_int4_gemm_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "vmx_int4_dequant_bf16": (_i, [_p, _p, _i, _i, _i, _p]),
    "vmx_exl3_reconstruct_f16": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vmx_exl3_had_r_128_f16": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "vmx_exl3_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "vmx_exl3_gemm_f16": (_i, [_p] * 5 + [_i] * 5 + [_p, _sz, _p]),
    "vmx_fp8_gemm_bf16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vmx_fp8_dequant_bf16": (_i, [_p, _p, _p, _i64, _i64, _i, _p]),
    "vmx_fp8_quant_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vmx_fp8_scaled_mm": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vmx_fp8_dequant_block_bf16": (_i, [_p, _p, _p, _i64, _i64, _i, _i, _p]),
    "vmx_fp8_block_gemm_bf16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "vmx_fused_rmsnorm_fp8": (_i, [_p, _p, _p, _p, _i, _i, _f, _p]),
    "vmx_topk_softmax": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vmx_moe_align_block_size": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vmx_moe_int4_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "vmx_moe_int4_forward": (_i, [_p] * 6 + [_i] * 6 + [_p, _sz, _p]),
    "vmx_debug_moe_block_rows": (_i, [_i]),
    "vmx_moe_ep_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "vmx_moe_ep_int4_forward": (_i, [_p] * 6 + [_i] * 8 + [_p, _i, _p, _sz, _p]),
    "vmx_fused_qknorm_rope_bf16": (_i, [_p] * 9 + [_i] * 5 + [_p]),
    "vmx_add_bf16": (_i, [_p, _p, _p, _i64, _p]),
    "vmx_gelu_and_mul_bf16": (_i, [_p, _p, _p, _i64, _i, _p]),
    "vmx_temperature_scale_bf16": (_i, [_p, _f, _i, _i, _p]),
    "vmx_softmax_to_probs": (_i, [_p, _p, _i, _i, _i, _p]),
    "vmx_top_k_top_p_sample": (_i, [_p, _p, _p, _p, _p, _i, _f, _i, _i, _p]),
    "vmx_temperature_scale_f32_per_seq": (_i, [_p, _p, _i, _i, _p]),
    "vmx_multinomial_sample_no_filter": (_i, [_p, _p, _p, _i, _i, _p]),
    "vmx_rms_norm_bf16": (_i, [_p, _p, _p, _p, _f, _i, _i, _i, _p]),
    "vmx_rope_neox_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vmx_rope_gptj_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vmx_reshape_and_cache_hnd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vmx_gptq_dequant_int8_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vmx_gptq_int8_gemm_bf16": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "vmx_copy_blocks": (_i, [_p, _p, _p, _i, C.c_int64, _p]),
    "vmx_silu_and_mul_bf16": (_i, [_p, _p, _p, _i64, _p]),
    "vmx_argmax_workspace_bytes": (_sz, [_i, _i]),
    "vmx_argmax_bf16": (_i, [_p, _p, _i, _i, _p, _sz, _p]),
    "vmx_argmax_f16": (_i, [_p, _p, _i, _i, _p, _sz, _p]),
    "vmx_kv_scales_state_bytes": (_sz, []),
    "vmx_kv_scales_init": (_i, [_p, _f, _f, _i, _p]),
    "vmx_kv_scales_calibrate_if_needed": (_i, [_p, _p, _i64, _p, _i64, _f, _p]),
    "vmx_reshape_and_cache": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "vmx_paged_attention_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "vmx_paged_attention_bf16": (_i, [_p] * 6 + [_f] + [_i] * 9 + [_p, _p, _p, _sz, _p]),
    "vmx_paged_attention_bf16_with_lse": (_i, [_p] * 7 + [_f] + [_i] * 9 + [_p, _p, _p, _sz, _p]),
    "vmx_lse_correct_f32": (_i, [_p, _p, _p, _i, _i, _i64, _i, _p]),
    "vmx_paged_attention_f16": (_i, [_p] * 6 + [_f] + [_i] * 8 + [_p, _sz, _p]),
    "vmx_paged_attention_static_max_bytes": (_sz, []),
    "vmx_paged_attention_bf16_static": (_i, [_p] * 6 + [_f] + [_i] * 9 + [_p, _p, _p]),
    "vmx_dense_gemv_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "vmx_dense_gemv_f16": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "vmx_embedding_lookup_bf16": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "vmx_comm_unique_id": (_i, [_p]),
    "vmx_comm_create": (_i, [_p, _i, _i, _p]),
    "vmx_comm_destroy": (None, [_p]),
    "vmx_comm_rank": (_i, [_p]),
    "vmx_comm_world": (_i, [_p]),
    "vmx_comm_all_reduce_sum_bf16": (_i, [_p, _p, _sz, _p]),
    "vmx_comm_all_reduce_sum_f16": (_i, [_p, _p, _sz, _p]),
    "vmx_comm_all_reduce_max_u64": (_i, [_p, _p, _sz, _p]),
    "vmx_comm_all_reduce": (_i, [_p, _p, _sz, _i, _i, _p]),
    "vmx_comm_all_gather": (_i, [_p, _p, _p, _sz, _p]),
    "vmx_comm_reduce_scatter_sum": (_i, [_p, _p, _p, _sz, _i, _p]),
    "vmx_comm_broadcast": (_i, [_p, _p, _sz, _i, _p]),
    "vmx_comm_send": (_i, [_p, _p, _sz, _i, _p]),
    "vmx_comm_recv": (_i, [_p, _p, _sz, _i, _p]),
    "vmx_comm_barrier": (_i, [_p, _p]),
    "vmx_comm_all_to_all": (_i, [_p, _p, _p, _sz, _p]),
    "vmx_comm_all_to_all_v": (_i, [_p, _p, _p, _p, _p, _sz, _p]),
    "vmx_comm_all_to_all_v_dev": (_i, [_p, _p, _p, _p, _p, _sz, _sz, _p]),
    "vmx_comm_create_local": (_i, [_i, _i, _p]),
    "vmx_comm_p2p_export": (_i, [_p, _sz, _sz, _p]),
    "vmx_comm_p2p_import": (_i, [_p, _p]),
    "vmx_comm_p2p_disable": (_i, [_p]),
    "vmx_comm_p2p_ready": (_i, [_p]),
    "vmx_comm_p2p_set_timeout_ms": (_i, [_p, _i]),
    "vmx_comm_p2p_timeouts": (_i, [_p, _p]),
    "vmx_comm_error": (_i, [_p, _p]),
    "vmx_qwen3_set_comm": (_i, [_p, _p]),
    "vmx_qwen3_create": (_i, [_p, _p]),
    "vmx_qwen3_destroy": (None, [_p]),
    "vmx_qwen3_load_tensor": (_i, [_p, _i, _i, _p, _sz, _p]),
    "vmx_qwen3_load_awq_linear": (_i, [_p, _i, _i, _p, _p, _p, _p]),
    "vmx_qwen3_load_gptq_linear": (_i, [_p, _i, _i, _p, _p, _p, _p]),
    "vmx_qwen3_load_iq_linear": (_i, [_p, _i, _i, _i, _p, _sz, _p]),
    "vmx_qwen3_load_dense_linear": (_i, [_p, _i, _i, _p, _sz, _p]),
    "vmx_qwen3_set_rows_share_cache": (_i, [_p, _i]),
    "vmx_qwen3_set_activation_dtype": (_i, [_p, _i]),
    "vmx_qwen3_load_exl3_linear": (_i, [_p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vmx_qwen3_decode": (_i, [_p] * 6 + [_i, _i, _i, _p, _p, _p]),
    "vmx_debug_set_fused_attention": (_i, [_i]),
    "vmx_qwen3_debug_read": (_i, [_p, _i, _p, _sz, _p]),
    "vmx_qwen3_debug_kv": (_i, [_p, _i, _i, _i, _p, _sz, _p]),
    "vmx_debug_stream_read": (_i, [_p, _sz, _i, _p, _p]),
    "vmx_debug_xcc_ids": (_i, [_p, _i, _i, _p]),
    "vmx_debug_lane_ops_check": (_i, [_p, C.c_uint, _p]),
    "vmx_decode_advance": (_i, [_p] * 6 + [_i, _i, _i, _p]),
    "vmx_qwen3_load_gptq_expert": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "vmx_qwen3_prefill": (_i, [_p, _p, _i, _p, _p, _i, _p, _p, _p]),
    "vmx_qwen3_linears_only": (_i, [_p, _i, _p]),
    "vmx_qwen3_linears_select": (_i, [_p, _i]),
    "vmx_qwen3_linears_info": (_i, [_p, _i, _p, _p]),
    "vmx_prefill_attention_bf16": (_i, [_p, _p, _p, _p, _p, _i, _i, _f, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "vmx_prefill_attention_f16": (_i, [_p, _p, _p, _p, _p, _i, _i, _f, _i, _i, _i, _i, _i, _p]),
    "vmx_quant_detect_json": (_i, [C.c_char_p, C.c_char_p, _p]),
    "vmx_quant_detect_dir": (_i, [C.c_char_p, _p]),
    "vmx_safetensors_open": (_i, [C.c_char_p, _p]),
    "vmx_safetensors_close": (None, [_p]),
    "vmx_safetensors_tensor_count": (_i, [_p]),
    "vmx_safetensors_find": (_i, [_p, C.c_char_p]),
    "vmx_safetensors_tensor_info": (_i, [_p, _i, _p]),
    "vmx_safetensors_read_tensor": (_i, [_p, _i, _p, _sz]),
    "vmx_safetensors_metadata": (C.c_char_p, [_p, C.c_char_p]),
    "vmx_gguf_open": (_i, [C.c_char_p, _p]),
    "vmx_gguf_parse": (_i, [_p, _sz, _p]),
    "vmx_gguf_close": (None, [_p]),
    "vmx_gguf_version": (_i, [_p]),
    "vmx_gguf_tensor_data_offset": (C.c_uint64, [_p]),
    "vmx_gguf_tensor_count": (_i, [_p]),
    "vmx_gguf_metadata_count": (_i, [_p]),
    "vmx_gguf_find_tensor": (_i, [_p, C.c_char_p]),
    "vmx_gguf_tensor_info": (_i, [_p, _i, _p]),
    "vmx_gguf_read_tensor": (_i, [_p, _i, _p, _sz]),
    "vmx_gguf_metadata_key": (C.c_char_p, [_p, _i]),
    "vmx_gguf_metadata_type": (_i, [_p, C.c_char_p, _p, _p]),
    "vmx_gguf_metadata_int": (_i, [_p, C.c_char_p, _i, _p]),
    "vmx_gguf_metadata_float": (_i, [_p, C.c_char_p, _i, _p]),
    "vmx_gguf_metadata_string": (C.c_char_p, [_p, C.c_char_p, _i]),
    "vmx_gguf_llama_cpp_name": (_i, [C.c_char_p, _i, C.c_char_p, _sz]),
    "vmx_gemma4_create": (_i, [_p, _p, _p]),
    "vmx_gemma4_destroy": (None, [_p]),
    "vmx_gemma4_load_tensor": (_i, [_p, _i, _i, _p, _sz, _p]),
    "vmx_gemma4_load_iq_linear": (_i, [_p, _i, _i, _i, _p, _sz, _p]),
    "vmx_gemma4_decode": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "vmx_gemma4_mmvq_info": (_i, [_p, _i, _p, _p]),
    "vmx_gemma4_prefill": (_i, [_p, _p, _i, _p, _p, _i, _p, _p, _p]),
    "vmx_gemma4_mmvq_only": (_i, [_p, _i, _p]),
    "vmx_gemma4_mmvq_select": (_i, [_p, _i]),
    "vmx_gemma4_debug_read": (_i, [_p, _i, _p, _sz, _p]),
    "vmx_gemma4_set_comm": (_i, [_p, _p]),
    "vmx_gemma4_kv_scales_reset": (_i, [_p, _f, _p]),
    "vmx_gemma4_kv_scales_get": (_i, [_p, _i, _p, _p, _p, _p]),
    "vmx_gemma4_debug_kv": (_i, [_p, _i, _i, _i, _p, _sz, _p]),
}


class Qwen3Config(C.Structure):
    """struct vmx_qwen3_config (include/vmx.h)."""
    _fields_ = [
        ("hidden_size", _i), ("num_layers", _i), ("num_heads", _i), ("num_kv_heads", _i), ("head_dim", _i),
        ("intermediate_size", _i), ("vocab_size", _i), ("rms_norm_eps", _f), ("max_position", _i), ("group_size", _i),
        ("qk_norm", _i), ("tie_word_embeddings", _i), ("kv_cache_dtype", _i), ("block_size", _i), ("num_blocks", _i),
        ("max_batch", _i), ("max_seq_len", _i), ("tp_size", _i), ("tp_rank", _i), ("max_prefill_tokens", _i),
        ("num_experts", _i), ("experts_per_tok", _i),
    ]


class GgufTensor(C.Structure):
    """struct vmx_gguf_tensor (include/vmx.h)."""
    _fields_ = [("name", C.c_char_p), ("n_dims", _i), ("shape", C.c_uint64 * 8), ("ggml_type", _i), ("offset", C.c_uint64),
                ("n_bytes", C.c_uint64)]


class Gemma4Config(C.Structure):
    """struct vmx_gemma4_config (include/vmx.h)."""
    _fields_ = [
        ("hidden_size", _i), ("num_layers", _i), ("num_heads", _i), ("intermediate_size", _i), ("vocab_size", _i),
        ("rms_norm_eps", _f), ("head_dim_sliding", _i), ("head_dim_full", _i), ("kv_heads_sliding", _i), ("kv_heads_full", _i),
        ("sliding_window", _i), ("rot_dim_sliding", _i), ("rot_dim_full", _i), ("max_position", _i), ("kv_cache_dtype", _i),
        ("block_size", _i), ("num_blocks", _i), ("max_batch", _i), ("max_seq_len", _i), ("final_logit_softcap", _f),
        ("embed_ggml_type", _i), ("hidden_size_per_layer_input", _i), ("vocab_size_per_layer_input", _i), ("ple_embed_ggml_type", _i), ("tp_size", _i), ("tp_rank", _i), ("max_prefill_tokens", _i),
    ]


def _bind():
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args


_bind()


class VmxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vmx error {code}: {msg}")
        self.code = code
        self.msg = msg


def check(rc: int) -> None:
    if rc != 0:
        raise VmxError(rc, lib.vmx_last_error().decode("utf-8", "replace"))


def declared_symbols():
    """Names bound above; tests compare this with the prototypes in include/vmx.h."""
    return sorted(_SIGS)

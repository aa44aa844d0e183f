"""ctypes front-end of the CPU oracle (oracle/_build/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg.  Nothing under vllm_rust_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liborc.so"

IQ_TYPES = {"iq2_xs": 17, "iq3_xxs": 18, "iq3_s": 21, "iq2_s": 22, "iq4_xs": 23}
IQ_TYPE_SIZE = {17: 74, 22: 82, 18: 98, 21: 110, 23: 136}


def build() -> Path:
    """Compile the oracle with gcc (no GPU toolchain involved)."""
    subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return _LIB_PATH


def _load() -> C.CDLL:
    if not _LIB_PATH.exists():
        build()
    return C.CDLL(str(_LIB_PATH))


_lib = _load()


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


for _name, _res, _args in [
    ("orc_iq_type_size", C.c_int, [C.c_int]),
    ("orc_iq_dequant", C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_iq_subdots", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    ("orc_quantize_q8_1", None, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    ("orc_iq_mmvq_q8", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    ("orc_iq_mmvq_q8_mt", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    ("orc_awq_to_gptq_u32", C.c_uint32, [C.c_uint32]),
    ("orc_repack_awq_nibbles", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_awq_to_gptq_qweight", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    ("orc_gptq_dequant_int8", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    ("orc_int4_dequant", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    ("orc_int4_gemv_f32", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    ("orc_int4_gemv_bf16", C.c_int, [C.c_void_p] * 5 + [C.c_int64] * 4 + [C.c_void_p]),
    ("orc_f32_to_bf16_array", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_bf16_to_f32_array", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_f32_to_f16_array", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_f16_to_f32_array", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_rms_norm_bf16", None, [C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    ("orc_fused_add_rms_norm_bf16", None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int64, C.c_void_p]),
    ("orc_rope_neox_bf16", None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("orc_silu_and_mul_bf16", None, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_argmax_bf16", None, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    ("orc_fp8_e4m3_decode", C.c_float, [C.c_uint8]),
    ("orc_fp8_e4m3_encode_rne", C.c_uint8, [C.c_float]),
    ("orc_fp8_e4m3_encode_trunc", C.c_uint8, [C.c_float]),
    ("orc_kv_fp8_quant", C.c_uint8, [C.c_uint16, C.c_float]),
    ("orc_rope_gptj_bf16", None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("orc_reshape_and_cache_hnd", None, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_int]),
    ("orc_copy_blocks", None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    ("orc_reshape_and_cache", None, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]),
    ("orc_paged_attention_v2", None, [C.c_void_p] * 5 + [C.c_float, C.c_int64] + [C.c_int] * 8 + [C.c_float, C.c_float, C.c_void_p]),
    ("orc_paged_attention_lse", None, [C.c_void_p] * 4 + [C.c_float, C.c_int64] + [C.c_int] * 7 + [C.c_float, C.c_void_p]),
    ("orc_fp8_e5m2_decode", C.c_float, [C.c_uint8]),
    ("orc_fp8_e5m2_encode_rne", C.c_uint8, [C.c_float]),
    ("orc_q6k_dequant", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_q4_0_dequant", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_q8_0_dequant", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_q4_k_dequant", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_q3_k_dequant", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_q4_1_dequant", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_q5_0_dequant", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_q5_1_dequant", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_q2_k_dequant", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_q5_k_dequant", None, [C.c_void_p, C.c_int64, C.c_void_p]),
    ("orc_exl3_decode", C.c_uint16, [C.c_uint32, C.c_int]),
    ("orc_exl3_reconstruct", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    ("orc_exl3_had_fp16", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    ("orc_exl3_had_f32_to_fp16", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    ("orc_exl3_gemm", C.c_int, [C.c_void_p] * 4 + [C.c_int64] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 4),
    ("orc_fp8_gemm_bf16", None, [C.c_void_p] * 4 + [C.c_int64] * 3 + [C.c_int, C.c_void_p]),
    ("orc_fp8_quant_bf16", None, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("orc_fp8_scaled_mm", None, [C.c_void_p] * 5 + [C.c_int64] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    ("orc_fp8_dequant_block_bf16", None, [C.c_void_p, C.c_void_p] + [C.c_int64] * 4 + [C.c_void_p]),
    ("orc_fused_rmsnorm_fp8", None, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    ("orc_fp8_dequant_bf16", None, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    ("orc_dense_gemv_bf16", None, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
]:
    _f = getattr(_lib, _name)
    _f.restype = _res
    _f.argtypes = _args


# ---------------------------------------------------------------- conversions
def f32_to_bf16(x) -> np.ndarray:
    x = _c(x, np.float32)
    out = np.empty(x.shape, np.uint16)
    _lib.orc_f32_to_bf16_array(_p(x), x.size, _p(out))
    return out


def bf16_to_f32(x) -> np.ndarray:
    x = _c(x, np.uint16)
    out = np.empty(x.shape, np.float32)
    _lib.orc_bf16_to_f32_array(_p(x), x.size, _p(out))
    return out


def f32_to_f16(x) -> np.ndarray:
    x = _c(x, np.float32)
    out = np.empty(x.shape, np.uint16)
    _lib.orc_f32_to_f16_array(_p(x), x.size, _p(out))
    return out


def f16_to_f32(x) -> np.ndarray:
    x = _c(x, np.uint16)
    out = np.empty(x.shape, np.float32)
    _lib.orc_f16_to_f32_array(_p(x), x.size, _p(out))
    return out


# ---------------------------------------------------------------- I-quants
def iq_dequant(ggml_type: int, data) -> np.ndarray:
    data = _c(data, np.uint8).reshape(-1)
    ts = IQ_TYPE_SIZE[ggml_type]
    assert data.size % ts == 0
    nb = data.size // ts
    out = np.empty(nb * 256, np.float32)
    rc = _lib.orc_iq_dequant(ggml_type, _p(data), nb, _p(out))
    assert rc == 0
    return out


def quantize_q8_1(x):
    x = _c(x, np.float32)
    m, n_in = x.shape
    assert n_in % 32 == 0
    aq = np.empty((m, n_in), np.int8)
    ad = np.empty((m, n_in // 32), np.float32)
    _lib.orc_quantize_q8_1(_p(x), m, n_in, _p(aq), _p(ad))
    return aq, ad


def iq_subdots(ggml_type: int, w, aq, n_out: int, n_in: int) -> np.ndarray:
    w = _c(w, np.uint8).reshape(-1)
    aq = _c(aq, np.int8)
    m = aq.shape[0]
    out = np.empty((n_out, m, n_in // 32), np.int32)
    rc = _lib.orc_iq_subdots(ggml_type, _p(w), _p(aq), n_out, n_in, m, _p(out))
    assert rc == 0
    return out


def iq_mmvq_q8(ggml_type: int, w, aq, ad, n_out: int, n_in: int, threads: bool = False) -> np.ndarray:
    w = _c(w, np.uint8).reshape(-1)
    aq = _c(aq, np.int8)
    ad = _c(ad, np.float32)
    m = aq.shape[0]
    y = np.empty((m, n_out), np.float32)
    fn = _lib.orc_iq_mmvq_q8_mt if threads else _lib.orc_iq_mmvq_q8
    rc = fn(ggml_type, _p(w), _p(aq), _p(ad), n_out, n_in, m, _p(y))
    assert rc == 0
    return y


def iq_matmul(ggml_type: int, w, x, n_out: int, n_in: int) -> np.ndarray:
    """q8_1_mmvq_cpu: quantise X to q8_1, integer dots, f32 accumulate."""
    aq, ad = quantize_q8_1(np.asarray(x, np.float32).reshape(-1, n_in))
    return iq_mmvq_q8(ggml_type, w, aq, ad, n_out, n_in)


# ---------------------------------------------------------------- int4 (AWQ / GPTQ)
def awq_to_gptq_u32(w: int) -> int:
    return int(_lib.orc_awq_to_gptq_u32(w & 0xFFFFFFFF))


def repack_awq_nibbles(words) -> np.ndarray:
    words = _c(words, np.uint32)
    out = np.empty_like(words)
    _lib.orc_repack_awq_nibbles(_p(words), words.size, _p(out))
    return out


def awq_to_gptq_qweight(awq, K: int, N: int) -> np.ndarray:
    awq = _c(awq, np.uint32)
    assert awq.shape == (K, N // 8)
    out = np.empty((K // 8, N), np.uint32)
    rc = _lib.orc_awq_to_gptq_qweight(_p(awq), K, N, _p(out))
    assert rc == 0
    return out


def int4_dequant(qweight, scales_f32, qzeros, K: int, N: int, group: int) -> np.ndarray:
    qweight = _c(qweight, np.uint32)
    scales_f32 = _c(scales_f32, np.float32)
    qz = _c(qzeros, np.uint32) if qzeros is not None else None
    w = np.empty((K, N), np.float32)
    rc = _lib.orc_int4_dequant(_p(qweight), _p(scales_f32), _p(qz) if qz is not None else None, K, N, group, _p(w))
    assert rc == 0
    return w


def int4_gemv_f32(x, qweight, scales_f32, qzeros, K: int, N: int, group: int) -> np.ndarray:
    x = _c(x, np.float32).reshape(-1, K)
    qweight = _c(qweight, np.uint32)
    scales_f32 = _c(scales_f32, np.float32)
    qz = _c(qzeros, np.uint32) if qzeros is not None else None
    y = np.empty((x.shape[0], N), np.float32)
    rc = _lib.orc_int4_gemv_f32(_p(x), _p(qweight), _p(scales_f32), _p(qz) if qz is not None else None,
                                x.shape[0], K, N, group, _p(y))
    assert rc == 0
    return y


def int4_gemv_bf16(x_bf16, qweight, scales_bf16, qzeros, bias_bf16, K: int, N: int, group: int) -> np.ndarray:
    x = _c(x_bf16, np.uint16).reshape(-1, K)
    qweight = _c(qweight, np.uint32)
    sc = _c(scales_bf16, np.uint16)
    qz = _c(qzeros, np.uint32) if qzeros is not None else None
    bs = _c(bias_bf16, np.uint16) if bias_bf16 is not None else None
    y = np.empty((x.shape[0], N), np.uint16)
    rc = _lib.orc_int4_gemv_bf16(_p(x), _p(qweight), _p(sc), _p(qz) if qz is not None else None,
                                 _p(bs) if bs is not None else None, x.shape[0], K, N, group, _p(y))
    assert rc == 0
    return y


# ---------------------------------------------------------------- glue ops
def rms_norm_bf16(x_bf16, w_bf16, eps: float, gemma_plus_one: bool = False) -> np.ndarray:
    x = _c(x_bf16, np.uint16)
    tokens, H = x.shape
    w = _c(w_bf16, np.uint16) if w_bf16 is not None else None
    out = np.empty_like(x)
    _lib.orc_rms_norm_bf16(_p(x), _p(w) if w is not None else None, eps, tokens, H, int(gemma_plus_one), _p(out))
    return out


def fused_add_rms_norm_bf16(x_bf16, residual_bf16, w_bf16, eps: float):
    x = _c(x_bf16, np.uint16)
    res = _c(residual_bf16, np.uint16).copy()
    tokens, H = x.shape
    w = _c(w_bf16, np.uint16)
    out = np.empty_like(x)
    _lib.orc_fused_add_rms_norm_bf16(_p(x), _p(res), _p(w), eps, tokens, H, _p(out))
    return out, res


def rope_neox_bf16(positions, q_bf16, k_bf16, cos_sin, rot_dim: int, head_size: int, n_heads: int, n_kv_heads: int):
    pos = _c(positions, np.int32)
    q = _c(q_bf16, np.uint16).copy()
    k = _c(k_bf16, np.uint16).copy() if k_bf16 is not None else None
    cs = _c(cos_sin, np.float32)
    _lib.orc_rope_neox_bf16(_p(pos), _p(q), _p(k) if k is not None else None, _p(cs), pos.size, rot_dim, head_size,
                            n_heads, n_kv_heads)
    return q, k


def gptq_dequant_int8(qweight, scales_bf16, qzeros, K: int, N: int, group: int) -> np.ndarray:
    """GPTQ 8-bit weights -> bf16 bits [K, N] (gptq_dequant.cu:128-170)."""
    qw, sc, qz = _c(qweight, np.uint32), _c(scales_bf16, np.uint16), _c(qzeros, np.uint32)
    out = np.empty((K, N), np.uint16)
    rc = _lib.orc_gptq_dequant_int8(_p(qw), _p(sc), _p(qz), K, N, group, _p(out))
    assert rc == 0
    return out


def rope_gptj_bf16(positions, q_bf16, k_bf16, cos_sin, rot_dim: int, head_size: int, n_heads: int, n_kv_heads: int):
    pos = _c(positions, np.int32)
    q = _c(q_bf16, np.uint16).copy()
    k = _c(k_bf16, np.uint16).copy() if k_bf16 is not None else None
    cs = _c(cos_sin, np.float32)
    _lib.orc_rope_gptj_bf16(_p(pos), _p(q), _p(k) if k is not None else None, _p(cs), pos.size, rot_dim, head_size,
                            n_heads, n_kv_heads)
    return q, k


def silu_and_mul_bf16(gate_bf16, up_bf16) -> np.ndarray:
    g = _c(gate_bf16, np.uint16)
    u = _c(up_bf16, np.uint16)
    out = np.empty_like(g)
    _lib.orc_silu_and_mul_bf16(_p(g), _p(u), g.size, _p(out))
    return out


def argmax_bf16(logits_bf16) -> np.ndarray:
    lg = _c(logits_bf16, np.uint16)
    rows, vocab = lg.shape
    out = np.empty(rows, np.int32)
    _lib.orc_argmax_bf16(_p(lg), rows, vocab, _p(out))
    return out


def fp8_e4m3_decode(b: int) -> float:
    return float(_lib.orc_fp8_e4m3_decode(b))


def fp8_e4m3_encode_rne(v: float) -> int:
    return int(_lib.orc_fp8_e4m3_encode_rne(v))


def fp8_e4m3_encode_trunc(v: float) -> int:
    return int(_lib.orc_fp8_e4m3_encode_trunc(v))


class KVScales:
    """KVScales of the reference (crates/core/src/kv_cache/quantization.rs:80-240): identity scales until the first write pins
    scale = max(absmax * headroom / 448, 1e-12) per tensor (compute_scale_with_headroom :272-277; candle's `x / 448.0` on an f32
    tensor is the affine multiply by f32(1 / 448)); pinned scales never change until reset()."""

    def __init__(self, k_scale: float = 1.0, v_scale: float = 1.0, calibrated: bool = False, headroom: float = 1.5):
        self.k_scale, self.v_scale, self.calibrated, self.headroom = np.float32(k_scale), np.float32(v_scale), calibrated, headroom

    def set_headroom_factor(self, factor: float) -> None:
        if not np.isfinite(factor) or factor <= 0.0:
            raise ValueError(f"KV scale headroom factor must be a positive finite value, got {factor}")
        self.headroom = factor

    @staticmethod
    def _scale(x_bf16, headroom: float) -> np.float32:
        amax = np.float32(np.abs(bf16_to_f32(_c(x_bf16, np.uint16))).max()) if np.size(x_bf16) else np.float32(0.0)
        return np.maximum(np.float32(amax * np.float32(headroom)) * np.float32(1.0 / 448.0), np.float32(1e-12))

    def calibrate_if_needed(self, k_bf16, v_bf16) -> None:
        if self.calibrated:
            return
        self.k_scale, self.v_scale, self.calibrated = self._scale(k_bf16, self.headroom), self._scale(v_bf16, self.headroom), True

    def set(self, k_scale: float, v_scale: float) -> None:
        self.k_scale, self.v_scale, self.calibrated = np.float32(k_scale), np.float32(v_scale), True

    def reset(self) -> None:
        self.k_scale, self.v_scale, self.calibrated = np.float32(1.0), np.float32(1.0), False


def reshape_and_cache_hnd(key_u16, value_u16, k_cache, v_cache, slots, kv_heads: int, head_dim: int, block_size: int) -> None:
    """In place on uint16 caches [blocks, kv_heads, block_size, head_dim]."""
    key, val, sl = _c(key_u16, np.uint16), _c(value_u16, np.uint16), _c(slots, np.int32)
    assert k_cache.flags.c_contiguous and v_cache.flags.c_contiguous and k_cache.dtype == np.uint16
    _lib.orc_reshape_and_cache_hnd(_p(key), _p(val), _p(k_cache), _p(v_cache), _p(sl), sl.size, kv_heads, head_dim, block_size)


def copy_blocks(k_cache, v_cache, mapping, bytes_per_block: int) -> None:
    """In place: block pairs (src, dst) of int32 [pairs, 2], applied in order."""
    mp = _c(mapping, np.int32)
    assert k_cache.flags.c_contiguous and v_cache.flags.c_contiguous
    _lib.orc_copy_blocks(_p(k_cache), _p(v_cache), _p(mp), mp.size // 2, bytes_per_block)


def reshape_and_cache(key_bf16, value_bf16, k_cache, v_cache, slots, kv_heads: int, head_dim: int, block_size: int,
                      kv_dtype: int = 0, k_scale: float = 1.0, v_scale: float = 1.0) -> None:
    key = _c(key_bf16, np.uint16)
    val = _c(value_bf16, np.uint16)
    sl = _c(slots, np.int32)
    assert k_cache.flags.c_contiguous and v_cache.flags.c_contiguous
    _lib.orc_reshape_and_cache(_p(key), _p(val), _p(k_cache), _p(v_cache), _p(sl), sl.size, kv_heads, head_dim,
                               block_size, kv_dtype, k_scale, v_scale)


def paged_attention_v2(q_bf16, k_cache, v_cache, block_tables, seq_lens, scale: float, num_heads: int,
                       num_kv_heads: int, head_dim: int, block_size: int, partition_size: int, window: int = 0,
                       kv_dtype: int = 0, k_scale: float = 1.0, v_scale: float = 1.0) -> np.ndarray:
    q = _c(q_bf16, np.uint16)
    bt = _c(block_tables, np.int32)
    sl = _c(seq_lens, np.int32)
    num_seqs = sl.size
    out = np.zeros((num_seqs, num_heads * head_dim), np.uint16)
    _lib.orc_paged_attention_v2(_p(q), _p(k_cache), _p(v_cache), _p(bt), _p(sl), scale, num_seqs, num_heads,
                                num_kv_heads, bt.shape[1], head_dim, block_size, partition_size, window, kv_dtype,
                                k_scale, v_scale, _p(out))
    return out


def paged_attention_lse(q_bf16, k_cache, block_tables, seq_lens, scale: float, num_heads: int, num_kv_heads: int, head_dim: int,
                        block_size: int, window: int = 0, kv_dtype: int = 0, k_scale: float = 1.0) -> np.ndarray:
    """float64 [seqs, heads]: ln sum exp(scale * q . k) per head over its (windowed) context."""
    q, bt, sl = _c(q_bf16, np.uint16), _c(block_tables, np.int32), _c(seq_lens, np.int32)
    out = np.empty((sl.size, num_heads), np.float64)
    _lib.orc_paged_attention_lse(_p(q), _p(k_cache), _p(bt), _p(sl), scale, sl.size, num_heads, num_kv_heads, bt.shape[1], head_dim,
                                 block_size, window, kv_dtype, k_scale, _p(out))
    return out


def fp8_e5m2_decode(b: int) -> float:
    return float(_lib.orc_fp8_e5m2_decode(b))


def fp8_e5m2_encode_rne(v: float) -> int:
    return int(_lib.orc_fp8_e5m2_encode_rne(v))


def q6k_dequant(data) -> np.ndarray:
    data = _c(data, np.uint8).reshape(-1)
    assert data.size % 210 == 0
    nb = data.size // 210
    out = np.empty(nb * 256, np.float32)
    _lib.orc_q6k_dequant(_p(data), nb, _p(out))
    return out


# ---------------------------------------------------------------- FP8 weight path
def fp8_gemm_bf16(x_bf16, W_u8, scale, bias_bf16, per_channel: bool) -> np.ndarray:
    x = _c(x_bf16, np.uint16); W = _c(W_u8, np.uint8); sc = _c(scale, np.float32)
    M, K = x.shape; N = W.shape[0]
    b = _c(bias_bf16, np.uint16) if bias_bf16 is not None else None
    y = np.empty((M, N), np.uint16)
    _lib.orc_fp8_gemm_bf16(_p(x), _p(W), _p(sc), _p(b) if b is not None else None, M, N, K, int(per_channel), _p(y))
    return y


def dense_gemv_bf16(x_bf16, W_bf16) -> np.ndarray:
    """y[m, n] = bf16(sum_k x[m, k] * W[n, k]) (f32 sequential accumulate): the unquantized Linear / lm_head on decode rows."""
    x = _c(x_bf16, np.uint16); W = _c(W_bf16, np.uint16)
    M, K = x.shape; N = W.shape[0]
    assert W.shape[1] == K
    y = np.empty((M, N), np.uint16)
    _lib.orc_dense_gemv_bf16(_p(x), _p(W), M, N, K, _p(y))
    return y


def set_threads(n: int) -> int:
    """OpenMP threads the oracle's parallel loops use from now on (libgomp of this process); returns the previous maximum."""
    gomp = C.CDLL("libgomp.so.1")
    prev = gomp.omp_get_max_threads()
    gomp.omp_set_num_threads(int(n))
    return prev


def fp8_quant_bf16(x_bf16, scale=None, dynamic: bool = True, per_token: bool = False):
    x = _c(x_bf16, np.uint16)
    T, H = x.shape
    out = np.empty((T, H), np.uint8)
    scales = np.zeros(T, np.float32)
    sc = _c(scale, np.float32) if scale is not None else None
    _lib.orc_fp8_quant_bf16(_p(x), _p(sc) if sc is not None else None, T, H, int(dynamic), int(per_token), _p(out), _p(scales))
    return out, scales


def fp8_scaled_mm(a_u8, b_u8, a_scale, b_scale, bias_bf16=None, a_per_row: bool = False, b_per_col: bool = False,
                  out_f32: bool = False) -> np.ndarray:
    A, B = _c(a_u8, np.uint8), _c(b_u8, np.uint8)
    sa, sb = _c(a_scale, np.float32).reshape(-1), _c(b_scale, np.float32).reshape(-1)
    b = _c(bias_bf16, np.uint16) if bias_bf16 is not None else None
    (M, K), N = A.shape, B.shape[0]
    out = np.zeros((M, N), np.float32 if out_f32 else np.uint16)
    _lib.orc_fp8_scaled_mm(_p(A), _p(B), _p(sa), _p(sb), _p(b) if b is not None else None, M, N, K, int(a_per_row), int(b_per_col),
                           int(out_f32), _p(out))
    return out


def fp8_dequant_block_bf16(q_u8, block_scale, block_n: int, block_k: int) -> np.ndarray:
    q, sc = _c(q_u8, np.uint8), _c(block_scale, np.float32)
    out = np.zeros(q.shape, np.uint16)
    _lib.orc_fp8_dequant_block_bf16(_p(q), _p(sc), q.shape[0], q.shape[1], block_n, block_k, _p(out))
    return out


def fused_rmsnorm_fp8(x_bf16, w_bf16, eps: float):
    x, w = _c(x_bf16, np.uint16), _c(w_bf16, np.uint16)
    out, scales = np.zeros(x.shape, np.uint8), np.zeros(x.shape[0], np.float32)
    _lib.orc_fused_rmsnorm_fp8(_p(x), _p(w), x.shape[0], x.shape[1], eps, _p(out), _p(scales))
    return out, scales


def fp8_dequant_bf16(q_u8, scale, per_row: bool) -> np.ndarray:
    q = _c(q_u8, np.uint8); sc = _c(scale, np.float32)
    out = np.empty(q.shape, np.uint16)
    _lib.orc_fp8_dequant_bf16(_p(q), _p(sc), q.shape[0], q.shape[1], int(per_row), _p(out))
    return out


# ---------------------------------------------------------------- GGML legacy / K-quants
GGML_BLOCK = {2: (18, 32), 3: (20, 32), 6: (22, 32), 7: (24, 32), 8: (34, 32), 10: (84, 256), 11: (110, 256), 12: (144, 256),
              13: (176, 256), 14: (210, 256)}


def ggml_dequant(ggml_type: int, data) -> np.ndarray:
    ts, qk = GGML_BLOCK[ggml_type]
    data = _c(data, np.uint8).reshape(-1)
    assert data.size % ts == 0
    nb = data.size // ts
    out = np.empty(nb * qk, np.float32)
    fn = {2: _lib.orc_q4_0_dequant, 3: _lib.orc_q4_1_dequant, 6: _lib.orc_q5_0_dequant, 7: _lib.orc_q5_1_dequant,
          8: _lib.orc_q8_0_dequant, 10: _lib.orc_q2_k_dequant, 11: _lib.orc_q3_k_dequant, 12: _lib.orc_q4_k_dequant,
          13: _lib.orc_q5_k_dequant, 14: _lib.orc_q6k_dequant}[ggml_type]
    fn(_p(data), nb, _p(out))
    return out


# ---------------------------------------------------------------- EXL3
def exl3_decode(w: int, cb: int = 0) -> int:
    return int(_lib.orc_exl3_decode(w & 0xFFFF, cb))


def exl3_reconstruct(trellis_u16, K: int, N: int, bits: int, cb: int = 0) -> np.ndarray:
    t = _c(trellis_u16, np.uint16)
    assert t.size == (K // 16) * (N // 16) * 16 * bits
    out = np.empty((K, N), np.uint16)
    assert _lib.orc_exl3_reconstruct(_p(t), K, N, bits, cb, _p(out)) == 0
    return out


def exl3_had_fp16(x_f16, scale_f16=None, mode: int = 0) -> np.ndarray:
    x = _c(x_f16, np.uint16)
    rows, cols = x.shape
    sc = _c(scale_f16, np.uint16) if scale_f16 is not None else None
    y = np.empty_like(x)
    assert _lib.orc_exl3_had_fp16(_p(x), _p(sc) if sc is not None else None, rows, cols, mode, _p(y)) == 0
    return y


def exl3_gemm(x_f16, trellis_u16, suh_f16, svh_f16, K: int, N: int, bits: int, cb: int = 0) -> np.ndarray:
    x = _c(x_f16, np.uint16).reshape(-1, K)
    M = x.shape[0]
    t = _c(trellis_u16, np.uint16)
    su = _c(suh_f16, np.uint16) if suh_f16 is not None else None
    sv = _c(svh_f16, np.uint16) if svh_f16 is not None else None
    y = np.empty((M, N), np.uint16)
    w = np.empty((K, N), np.uint16); a = np.empty((M, K), np.uint16); c = np.empty((M, N), np.float32)
    rc = _lib.orc_exl3_gemm(_p(x), _p(t), _p(su) if su is not None else None, _p(sv) if sv is not None else None, M, K, N, bits, cb,
                            _p(y), _p(w), _p(a), _p(c))
    assert rc == 0
    return y

"""CPU restatement of the reference's GGUF header parser, test infrastructure only (never imported by the product).

Follows GgufHeader::read (crates/core/src/quantization/gguf/header.rs:221-321): magic, version (v1: u32 counts and
lengths, v2/v3: u64), metadata key/values (value types 0..12, arrays recursive), tensor table with the on-disk dims
reversed, type classification (header.rs:325-362), data offset aligned to general.alignment (default 32).
Parity unpinned: the reference's own tests read a model file that is not in the tree; the facts they assert
(dim reversal, n_bytes = elements / 256 * type_size, I-quant classification) are asserted on synthetic files."""
import struct

# GGML type id -> (bytes per block, elements per block): candle-native, plain integers, I-quants
TYPE_GEOMETRY = {0: (4, 1), 1: (2, 1), 2: (18, 32), 3: (20, 32), 6: (22, 32), 7: (24, 32), 8: (34, 32), 9: (36, 32), 10: (84, 256),
                 11: (110, 256), 12: (144, 256), 13: (176, 256), 14: (210, 256), 15: (292, 256), 30: (2, 1), 24: (1, 1), 25: (2, 1),
                 26: (4, 1), 27: (8, 1), 17: (74, 256), 18: (98, 256), 21: (110, 256), 22: (82, 256), 23: (136, 256)}


class GgufError(ValueError):
    pass


class _R:
    def __init__(self, b):
        self.b, self.p = b, 0

    def take(self, fmt):
        n = struct.calcsize(fmt)
        if self.p + n > len(self.b):
            raise GgufError("unexpected end of data")
        v = struct.unpack_from(fmt, self.b, self.p)[0]
        self.p += n
        return v

    def count(self, is64):
        return self.take("<Q") if is64 else self.take("<I")

    def string(self, is64):
        n = self.count(is64)
        if self.p + n > len(self.b):
            raise GgufError("unexpected end of data")
        s = self.b[self.p:self.p + n].decode("utf-8")
        self.p += n
        return s

    def value(self, t, is64):
        if t == 0: return self.take("<B")
        if t == 1: return self.take("<b")
        if t == 2: return self.take("<H")
        if t == 3: return self.take("<h")
        if t == 4: return self.take("<I")
        if t == 5: return self.take("<i")
        if t == 6: return self.take("<f")
        if t == 7:
            v = self.take("<B")
            if v > 1:
                raise GgufError(f"unexpected bool value {v}")
            return bool(v)
        if t == 8: return self.string(is64)
        if t == 9:
            et = self.take("<I")
            n = self.count(is64)
            return [self.value(et, is64) for _ in range(n)]
        if t == 10: return self.take("<Q")
        if t == 11: return self.take("<q")
        if t == 12: return self.take("<d")
        raise GgufError(f"unrecognized metadata value-type {t}")


def parse(data: bytes):
    """-> dict(version, metadata {key: value}, tensors [(name, shape, ggml_type, offset, n_bytes)], tensor_data_offset)"""
    r = _R(data)
    if data[:4] != b"GGUF":
        raise GgufError("not a GGUF file")
    r.p = 4
    version = r.take("<I")
    is64 = version >= 2
    n_tensors, n_kv = r.count(is64), r.count(is64)
    meta, types = {}, {}
    for _ in range(n_kv):
        k = r.string(is64)
        t = r.take("<I")
        meta[k] = r.value(t, is64)
        types[k] = t
    tensors = []
    for _ in range(n_tensors):
        name = r.string(is64)
        nd = r.take("<I")
        dims = [r.count(is64) for _ in range(nd)][::-1]
        tid = r.take("<I")
        off = r.take("<Q")
        if tid not in TYPE_GEOMETRY:
            raise GgufError(f"GGUF tensor '{name}': unsupported GGML type id {tid}")
        ts, bs = TYPE_GEOMETRY[tid]
        n = 1
        for d in dims:
            n *= d
        tensors.append((name, dims, tid, off, n // bs * ts))
    align = 32
    a = meta.get("general.alignment")
    if a is not None and types["general.alignment"] in (0, 2, 4):
        align = a
    elif a is not None and types["general.alignment"] in (1, 3, 5) and a >= 0:
        align = a
    return {"version": version, "metadata": meta, "meta_types": types, "tensors": tensors,
            "tensor_data_offset": (r.p + align - 1) // align * align}


LLAMA_CPP_SUBS = {"self_attn.q_proj": ["attn_q"], "self_attn.k_proj": ["attn_k"], "self_attn.v_proj": ["attn_v"],
                  "self_attn.o_proj": ["attn_output"], "self_attn.q_norm": ["attn_q_norm"], "self_attn.k_norm": ["attn_k_norm"],
                  "mlp.gate_proj": ["ffn_gate"], "mlp.up_proj": ["ffn_up"], "mlp.down_proj": ["ffn_down"],
                  "per_layer_input_gate": ["inp_gate"], "per_layer_projection": ["proj"], "post_per_layer_input_norm": ["post_norm"],
                  "layer_scalar": ["layer_output_scale"], "input_layernorm": ["attn_norm"],
                  "post_attention_layernorm": ["post_attention_norm", "ffn_norm"], "pre_feedforward_layernorm": ["ffn_norm"],
                  "post_feedforward_layernorm": ["post_ffw_norm"]}
LLAMA_CPP_TOP = {"model.embed_tokens": "token_embd", "model.norm": "output_norm", "lm_head": "output", "model.rope_freqs": "rope_freqs",
                 "model.embed_tokens_per_layer": "per_layer_token_embd", "model.per_layer_model_projection": "per_layer_model_proj",
                 "model.per_layer_projection_norm": "per_layer_proj_norm"}


def llama_cpp_names(prefix: str):
    """to_llama_cpp_prefixes / to_llama_cpp_top_level (gguf/mod.rs:1416-1495)."""
    if prefix in LLAMA_CPP_TOP:
        return [LLAMA_CPP_TOP[prefix]]
    if not prefix.startswith("model.layers."):
        return []
    rest = prefix[len("model.layers."):]
    if "." not in rest:
        return []
    idx, sub = rest.split(".", 1)
    if not idx.isdigit() or not idx.isascii():
        return []
    return [f"blk.{idx}.{s}" for s in LLAMA_CPP_SUBS.get(sub, [])]

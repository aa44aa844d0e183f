"""Host-side driver of the quantized Gemma 4 decoder (libvmx `vmx_gemma4_*`).

Mirrors the engine-facing surface of the reference's QuantizedGemma4ForCausalLM on its paged-decode path
(crates/core/src/models/gemma4_quantized.rs:1311-1424, 1743-1774, 2217-2279): GGUF I-quant projections, per-layer
KV geometry (sliding layers with a window, full-attention layers with one wide KV head), sandwich norms, GeGLU,
layer scalars, final-logit soft-cap, tied (optionally K-quant) token embedding.  All arithmetic is in the library.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
from typing import List, Optional

import numpy as np
import torch

from ._lib import Gemma4Config, check, lib
from .qwen3 import rope_cos_sin_cache

(T_EMBED, T_FINAL_NORM, T_COS_SIN_SLIDING, T_COS_SIN_FULL, T_INPUT_LN, T_POST_ATTN_LN, T_PRE_FFN_LN, T_POST_FFN_LN, T_Q_NORM,
 T_K_NORM, T_LAYER_SCALAR, T_KV_SCALES, T_PLE_EMBED, T_PLE_PROJ_NORM, T_POST_PLE_NORM) = range(15)
L_Q, L_K, L_V, L_O, L_GATE, L_UP, L_DOWN, L_PLE_GATE, L_PLE_PROJ, L_PLE_MODEL_PROJ = range(10)
G4_SLIDING, G4_K_EQ_V, G4_KV_SHARED, G4_DOUBLE_WIDE_MLP = 1, 2, 4, 8

IQ_TYPE_SIZE = {17: 74, 22: 82, 18: 98, 21: 110, 23: 136}
GGML_TYPE_SIZE = {2: (18, 32), 3: (20, 32), 6: (22, 32), 7: (24, 32), 8: (34, 32), 10: (84, 256), 11: (110, 256), 12: (144, 256), 13: (176, 256), 14: (210, 256)}
# projection types of gemma-4-12b-it-UD-IQ3_XXS (crates/core/src/quantization/gguf/header.rs:390-399): q, k, v, o, gate, up, down
GEMMA4_12B_IQ3_MIX = {L_Q: 18, L_K: 22, L_V: 18, L_O: 21, L_GATE: 18, L_UP: 18, L_DOWN: 21}


@dataclasses.dataclass
class Gemma4Spec:
    hidden_size: int
    num_layers: int
    num_heads: int
    intermediate_size: int
    vocab_size: int
    head_dim_sliding: int = 256
    head_dim_full: int = 512
    kv_heads_sliding: int = 8
    kv_heads_full: int = 1
    sliding_window: int = 1024
    full_every: int = 6                      # layer i is a full-attention layer when (i + 1) % full_every == 0
    layer_is_sliding: Optional[List[bool]] = None
    k_eq_v_on_full: bool = False
    num_kv_shared_layers: int = 0            # the last n layers read an earlier layer's KV cache (gemma4_quantized.rs:441-469)
    use_double_wide_mlp: bool = False        # KV-shared layers carry a twice-as-wide MLP (gemma4_quantized.rs:514-525)
    rope_theta_sliding: float = 10000.0
    rope_theta_full: float = 1e6
    partial_rotary_factor_full: float = 0.25
    rms_norm_eps: float = 1e-6
    max_position: int = 4096
    kv_cache_dtype: int = 1                  # FP8 e4m3
    block_size: int = 16
    num_blocks: int = 64
    max_batch: int = 1
    max_seq_len: int = 1024
    final_logit_softcap: float = 30.0
    embed_ggml_type: int = 0                 # 0 bf16, 11 Q3_K (the 12B file), 14 Q6_K
    hidden_size_per_layer_input: int = 0     # P of the per-layer inputs (E2B / E4B variants), 0 = none
    vocab_size_per_layer_input: int = 0
    ple_embed_ggml_type: int = 0             # embed_tokens_per_layer: 0 bf16, else a K-quant type id
    max_prefill_tokens: int = 0              # rows per prefill pass; > 16 switches prompt passes to the dequant + GEMM linear
    tp_size: int = 1                         # with tp_size > 1 the head counts and intermediate_size are this rank's shares
    tp_rank: int = 0

    @staticmethod
    def gemma4_12b(**kw):
        return Gemma4Spec(3840, 48, 16, 15360, 262144, **kw)

    def sliding(self, i: int) -> bool:
        if self.layer_is_sliding is not None:
            return bool(self.layer_is_sliding[i])
        return (i + 1) % self.full_every != 0

    def kv_shared(self, i: int) -> bool:
        return self.num_kv_shared_layers > 0 and i >= max(self.num_layers - self.num_kv_shared_layers, 0)

    def kv_target(self, i: int) -> Optional[int]:
        """Layer whose cache layer i attends over: the last layer before the shared suffix with i's attention type
        (kv_sharing_target_layer, gemma4_quantized.rs:441-469); None for a layer that owns its cache."""
        if not self.kv_shared(i):
            return None
        first_shared = max(self.num_layers - self.num_kv_shared_layers, 0)
        for cand in range(first_shared - 1, -1, -1):
            if self.sliding(cand) == self.sliding(i):
                return cand
        return None

    def layer_intermediate(self, i: int) -> int:
        return self.intermediate_size * (2 if self.use_double_wide_mlp and self.kv_shared(i) else 1)

    def layer_flags(self) -> List[int]:
        out = []
        for i in range(self.num_layers):
            f = G4_SLIDING if self.sliding(i) else 0
            if self.kv_shared(i):
                f |= G4_KV_SHARED | (G4_DOUBLE_WIDE_MLP if self.use_double_wide_mlp else 0)
            elif not self.sliding(i) and self.k_eq_v_on_full:
                f |= G4_K_EQ_V
            out.append(f)
        return out

    def rot_dim_full(self) -> int:
        r = int(self.head_dim_full * self.partial_rotary_factor_full)
        return r - r % 2

    def layer_geom(self, i: int):
        """(head_dim, kv_heads, rot_dim, window, k_eq_v) of layer i."""
        if self.sliding(i):
            return self.head_dim_sliding, self.kv_heads_sliding, self.head_dim_sliding, self.sliding_window, False
        return self.head_dim_full, self.kv_heads_full, self.rot_dim_full(), 0, self.k_eq_v_on_full and not self.kv_shared(i)

    def linear_shapes(self, i: int):
        hd, kvh, _, _, keqv = self.layer_geom(i)
        H, I, nq, nkv = self.hidden_size, self.layer_intermediate(i), self.num_heads * hd, kvh * hd
        sh = {L_Q: (H, nq), L_K: (H, nkv), L_V: (H, nkv), L_O: (nq, H), L_GATE: (H, I), L_UP: (H, I), L_DOWN: (I, H)}
        if keqv or self.kv_shared(i):
            del sh[L_V]
        if self.kv_shared(i):
            del sh[L_K]
        if self.hidden_size_per_layer_input > 0:
            sh[L_PLE_GATE] = (H, self.hidden_size_per_layer_input)
            sh[L_PLE_PROJ] = (self.hidden_size_per_layer_input, H)
        return sh


def shard_gemma4_spec(spec: Gemma4Spec, tp_rank: int, tp_size: int) -> Gemma4Spec:
    """Per-rank geometry: query heads, sliding KV heads and the MLP width divide; a full-attention layer with fewer KV heads than
    ranks keeps them all (its K/V projections are replicated); hidden size and vocabulary stay global.  Row-parallel inputs
    (o_proj, down_proj) must be whole 256-element blocks per rank."""
    if tp_size == 1:
        return dataclasses.replace(spec, tp_size=1, tp_rank=0)
    for name in ("num_heads", "kv_heads_sliding", "intermediate_size", "vocab_size"):
        if getattr(spec, name) % tp_size:
            raise ValueError(f"{name}={getattr(spec, name)} is not divisible by tp_size={tp_size}")
    if spec.kv_heads_full >= tp_size and spec.kv_heads_full % tp_size:
        raise ValueError(f"kv_heads_full={spec.kv_heads_full} is not divisible by tp_size={tp_size}")
    for k in (spec.num_heads // tp_size * spec.head_dim_sliding, spec.num_heads // tp_size * spec.head_dim_full,
              spec.intermediate_size // tp_size):
        if k % 256:
            raise ValueError(f"a row-parallel shard of {k} inputs is not a whole number of 256-element blocks")
    kvf = spec.kv_heads_full // tp_size if spec.kv_heads_full >= tp_size else spec.kv_heads_full
    return dataclasses.replace(spec, num_heads=spec.num_heads // tp_size, kv_heads_sliding=spec.kv_heads_sliding // tp_size,
                               kv_heads_full=kvf, intermediate_size=spec.intermediate_size // tp_size, tp_size=tp_size, tp_rank=tp_rank)


def shard_iq_blocks(raw: torch.Tensor, slot: int, n_out: int, n_in: int, type_size: int, tp_rank: int, tp_size: int,
                    replicate: bool = False) -> torch.Tensor:
    """This rank's part of an I-quant weight stored as [n_out][n_in / 256] blocks of type_size bytes: output rows for
    q/k/v/gate/up (column-parallel), 256-element input blocks for o/down (row-parallel); `replicate`: the whole tensor."""
    if replicate or tp_size == 1:
        return raw
    blocks = raw.reshape(n_out, n_in // 256, type_size)
    if slot in (L_O, L_DOWN):
        per = (n_in // 256) // tp_size
        return blocks[:, tp_rank * per:(tp_rank + 1) * per].contiguous().reshape(-1)
    per = n_out // tp_size
    return blocks[tp_rank * per:(tp_rank + 1) * per].contiguous().reshape(-1)


def proportional_cos_sin_cache(head_dim: int, rot_dim: int, max_position: int, theta: float) -> np.ndarray:
    """cos|sin table of RotaryEmbedding::new_gemma4_proportional (crates/core/src/layers/rotary.rs:128-174): the exponent's
    denominator is the full head_dim, only rot_dim leading dims rotate; values rounded to bf16 (model dtype)."""
    i = np.arange(0, rot_dim, 2, dtype=np.float32)
    inv_freq = (np.float32(1.0) / np.power(np.float32(theta), i / np.float32(head_dim))).astype(np.float32)
    t = np.arange(max_position, dtype=np.float32)[:, None]
    freqs = (t * inv_freq[None, :]).astype(np.float32)

    def bf16_round(a):
        u = a.astype(np.float32).view(np.uint32)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.view(np.float32)

    return np.concatenate([bf16_round(np.cos(freqs)), bf16_round(np.sin(freqs))], axis=1).astype(np.float32)


class QuantizedGemma4ForCausalLM:
    def __init__(self, spec: Gemma4Spec, device: int = 0):
        self.spec = spec
        self.device = torch.device("cuda", device)
        s = spec
        cfg = Gemma4Config(s.hidden_size, s.num_layers, s.num_heads, s.intermediate_size, s.vocab_size, s.rms_norm_eps,
                           s.head_dim_sliding, s.head_dim_full, s.kv_heads_sliding, s.kv_heads_full, s.sliding_window,
                           s.head_dim_sliding, s.rot_dim_full(), s.max_position, s.kv_cache_dtype, s.block_size, s.num_blocks,
                           s.max_batch, s.max_seq_len, s.final_logit_softcap, s.embed_ggml_type, s.hidden_size_per_layer_input,
                           s.vocab_size_per_layer_input, s.ple_embed_ggml_type, s.tp_size if s.tp_size > 1 else 0, s.tp_rank, s.max_prefill_tokens)
        self._comm = None
        flags = (C.c_int * s.num_layers)(*s.layer_flags())
        self._h = C.c_void_p()
        check(lib.vmx_gemma4_create(C.byref(cfg), flags, C.byref(self._h)))
        self.cos_sin_sliding = rope_cos_sin_cache(s.head_dim_sliding, s.max_position, s.rope_theta_sliding)
        self.cos_sin_full = proportional_cos_sin_cache(s.head_dim_full, s.rot_dim_full(), s.max_position, s.rope_theta_full)
        self.load_tensor(T_COS_SIN_SLIDING, 0, torch.from_numpy(self.cos_sin_sliding).to(self.device))
        self.load_tensor(T_COS_SIN_FULL, 0, torch.from_numpy(self.cos_sin_full).to(self.device))

    def close(self):
        if self._h:
            lib.vmx_gemma4_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def load_tensor(self, which: int, layer: int, t: torch.Tensor) -> None:
        t = t.contiguous()
        check(lib.vmx_gemma4_load_tensor(self._h, which, layer, t.data_ptr(), t.numel() * t.element_size(), self._stream()))
        torch.cuda.current_stream().synchronize()

    def load_iq_linear(self, layer: int, slot: int, ggml_type: int, blocks: torch.Tensor) -> None:
        blocks = blocks.contiguous()
        check(lib.vmx_gemma4_load_iq_linear(self._h, layer, slot, ggml_type, blocks.data_ptr(), blocks.numel(), self._stream()))
        torch.cuda.current_stream().synchronize()

    def set_comm(self, comm) -> None:
        """Attach a vllm_rust_amd.parallel.Communicator (kept alive by the model)."""
        self._comm = comm
        check(lib.vmx_gemma4_set_comm(self._h, comm.handle if comm is not None else None))

    def load_synthetic(self, slot_types=None, seed: int = 0, keep_host_copy: bool = False, full_spec: Optional[Gemma4Spec] = None):
        """Seeded random-init weights of the architecture: valid random I-quant blocks per projection (slot_types: ggml type
        per slot, default the 12B UD-IQ3_XXS mix), bf16 norms and layer scalars, bf16 or K-quant token embedding.
        Super-block scales are sized so activations stay O(1).  Returns the host copy when keep_host_copy (tiny models)."""
        # full_spec: this model is one tensor-parallel rank of it (self.spec = shard_gemma4_spec(full_spec, rank, world)); every
        # rank draws the full tensors from the same generator and keeps its part, so the ranks together hold the single-GPU model
        s = full_spec if full_spec is not None else self.spec
        rank, world = (self.spec.tp_rank, self.spec.tp_size) if full_spec is not None else (0, 1)
        slot_types = slot_types or GEMMA4_12B_IQ3_MIX
        g = torch.Generator(device=self.device).manual_seed(seed)
        ck = {}

        def keep(name, t):
            if keep_host_copy:
                ck[name] = t.cpu() if isinstance(t, torch.Tensor) else t

        def random_kquant_table(ggml_type, rows, cols):
            ts, qk = GGML_TYPE_SIZE[ggml_type]
            nb = rows * (cols // qk)
            tab = torch.randint(0, 256, (nb, ts), dtype=torch.uint8, device=self.device, generator=g)
            # valid small f16 super-scale(s): positions of the f16 fields per type (d; Q4_K also dmin)
            dpos = {2: [0], 8: [0], 11: [108], 12: [0, 2], 14: [208]}[ggml_type]
            mag = {2: 0.01, 8: 0.0006, 11: 0.0008, 12: 0.0001, 14: 0.00004}[ggml_type]
            for pos in dpos:
                d = ((torch.rand(nb, device=self.device, generator=g) * 0.5 + 0.75) * mag).to(torch.float16).view(torch.int16)
                tab[:, pos] = (d & 0xFF).to(torch.uint8)
                tab[:, pos + 1] = ((d >> 8) & 0xFF).to(torch.uint8)
            return tab.reshape(-1)

        def random_iq_linear(li, slot, K, N, gain=1.0):
            t = slot_types.get(slot, 18)
            ts = IQ_TYPE_SIZE[t]
            nb = N * (K // 256)
            raw = torch.randint(0, 256, (nb, ts), dtype=torch.uint8, device=self.device, generator=g)
            d = (torch.rand(nb, device=self.device, generator=g) * 0.5 + 0.75) * (gain / (120.0 * math.sqrt(K)))
            if t == 23:
                d = d * 6.0
            dbits = d.to(torch.float16).view(torch.int16)
            raw[:, 0] = (dbits & 0xFF).to(torch.uint8)
            raw[:, 1] = ((dbits >> 8) & 0xFF).to(torch.uint8)
            # replicated under tensor parallelism: the per-layer-input linears, and K/V of layers with fewer KV heads than ranks
            whole = slot >= L_PLE_GATE or (slot in (L_K, L_V) and s.layer_geom(li)[1] < world)
            mine = shard_iq_blocks(raw.reshape(-1), slot, N, K, ts, rank, world, replicate=whole)
            self.load_iq_linear(li, slot, t, mine)
            keep(f"{li}.{slot}.iq", (t, raw.reshape(-1).cpu()) if keep_host_copy else None)

        if s.embed_ggml_type == 0:
            emb = (torch.randn((s.vocab_size, s.hidden_size), device=self.device, generator=g) * 0.05).to(torch.bfloat16)
        else:
            emb = random_kquant_table(s.embed_ggml_type, s.vocab_size, s.hidden_size)
        self.load_tensor(T_EMBED, 0, emb); keep("embed", emb)
        del emb
        fn = (1.0 + 0.1 * torch.randn(s.hidden_size, device=self.device, generator=g)).to(torch.bfloat16)
        self.load_tensor(T_FINAL_NORM, 0, fn); keep("final_norm", fn)
        for li in range(s.num_layers):
            hd = s.layer_geom(li)[0]
            for which, n, name in ((T_INPUT_LN, s.hidden_size, "input_ln"), (T_POST_ATTN_LN, s.hidden_size, "post_attn_ln"),
                                   (T_PRE_FFN_LN, s.hidden_size, "pre_ffn_ln"), (T_POST_FFN_LN, s.hidden_size, "post_ffn_ln"),
                                   (T_Q_NORM, hd, "q_norm"), (T_K_NORM, hd, "k_norm")):
                w = (1.0 + 0.1 * torch.randn(n, device=self.device, generator=g)).to(torch.bfloat16)
                self.load_tensor(which, li, w); keep(f"{li}.{name}", w)
            sc = (0.75 + 0.5 * torch.rand(1, device=self.device, generator=g)).to(torch.bfloat16)
            self.load_tensor(T_LAYER_SCALAR, li, sc); keep(f"{li}.scalar", sc)
            if s.hidden_size_per_layer_input > 0:
                w = (1.0 + 0.1 * torch.randn(s.hidden_size, device=self.device, generator=g)).to(torch.bfloat16)
                self.load_tensor(T_POST_PLE_NORM, li, w); keep(f"{li}.post_ple_norm", w)
            for slot, (K, N) in s.linear_shapes(li).items():
                random_iq_linear(li, slot, K, N)
        if s.hidden_size_per_layer_input > 0:
            P, LP = s.hidden_size_per_layer_input, s.hidden_size_per_layer_input * s.num_layers
            random_iq_linear(0, L_PLE_MODEL_PROJ, s.hidden_size, LP, gain=math.sqrt(s.hidden_size))
            w = (1.0 + 0.1 * torch.randn(P, device=self.device, generator=g)).to(torch.bfloat16)
            self.load_tensor(T_PLE_PROJ_NORM, 0, w); keep("ple_proj_norm", w)
            if s.ple_embed_ggml_type == 0:
                pe = (torch.randn((s.vocab_size_per_layer_input, LP), device=self.device, generator=g) * (1.0 / math.sqrt(P))).to(torch.bfloat16)
            else:
                pe = random_kquant_table(s.ple_embed_ggml_type, s.vocab_size_per_layer_input, LP)
            self.load_tensor(T_PLE_EMBED, 0, pe); keep("ple_embed", pe)
        return ck

    # the per-layer-input tensors: blk.N.inp_gate / proj / post_norm and the top-level per_layer_token_embd / per_layer_model_proj /
    # per_layer_proj_norm (to_llama_cpp_prefixes, crates/core/src/quantization/gguf/mod.rs:1421-1424,1450-1453)
    GGUF_SLOT_NAMES = {L_Q: "self_attn.q_proj", L_K: "self_attn.k_proj", L_V: "self_attn.v_proj", L_O: "self_attn.o_proj",
                       L_GATE: "mlp.gate_proj", L_UP: "mlp.up_proj", L_DOWN: "mlp.down_proj",
                       L_PLE_GATE: "per_layer_input_gate", L_PLE_PROJ: "per_layer_projection"}
    GGUF_NORM_NAMES = {T_INPUT_LN: "input_layernorm", T_POST_ATTN_LN: "post_attention_layernorm", T_PRE_FFN_LN: "pre_feedforward_layernorm",
                       T_POST_FFN_LN: "post_feedforward_layernorm", T_Q_NORM: "self_attn.q_norm", T_K_NORM: "self_attn.k_norm",
                       T_LAYER_SCALAR: "layer_scalar"}

    def load_gguf(self, path: str) -> int:
        """Load every weight of the model from a GGUF file through the library's container parser: projections as raw I-quant
        blocks (GgufWeightLoader::load_linear, crates/core/src/quantization/gguf/mod.rs:1612-1662), F32/F16/BF16 norm vectors
        and layer scalars cast to bf16, token_embd as bf16 or as its raw K-quant blocks.  Tensor names are resolved with the
        HF -> llama.cpp map in priority order (gguf/mod.rs:1416-1479).  Returns the number of tensors loaded."""
        from ._lib import GgufTensor
        s = self.spec
        h = C.c_void_p()
        check(lib.vmx_gguf_open(str(path).encode(), C.byref(h)))
        try:
            def find(prefix):
                for which in range(3):
                    buf = C.create_string_buffer(128)
                    if lib.vmx_gguf_llama_cpp_name(prefix.encode(), which, buf, 128) == 0:
                        break
                    idx = lib.vmx_gguf_find_tensor(h, buf.value + b".weight")
                    if idx >= 0:
                        return idx
                return -1

            def read(idx):
                t = GgufTensor()
                check(lib.vmx_gguf_tensor_info(h, idx, C.byref(t)))
                raw = torch.empty(int(t.n_bytes), dtype=torch.uint8)
                check(lib.vmx_gguf_read_tensor(h, idx, raw.data_ptr(), raw.numel()))
                return t, raw

            def dense_bf16(idx):
                t, raw = read(idx)
                if t.ggml_type == 0:
                    return raw.view(torch.float32).to(torch.bfloat16)
                if t.ggml_type == 1:
                    return raw.view(torch.float16).to(torch.bfloat16)
                if t.ggml_type == 30:
                    return raw.view(torch.bfloat16)
                raise ValueError(f"{t.name.decode()}: expected a dense tensor, got GGML type {t.ggml_type}")

            n = 0
            idx = find("model.embed_tokens")
            if idx < 0:
                raise KeyError("token_embd.weight not found")
            t, raw = read(idx)
            if s.embed_ggml_type == 0:
                self.load_tensor(T_EMBED, 0, dense_bf16(idx).to(self.device))
            else:
                if t.ggml_type != s.embed_ggml_type:
                    raise ValueError(f"token_embd is GGML type {t.ggml_type}, the model was created for {s.embed_ggml_type}")
                self.load_tensor(T_EMBED, 0, raw.to(self.device))
            n += 1
            self.load_tensor(T_FINAL_NORM, 0, dense_bf16(find("model.norm")).to(self.device)); n += 1
            for li in range(s.num_layers):
                for which, name in self.GGUF_NORM_NAMES.items():
                    idx = find(f"model.layers.{li}.{name}")
                    if idx < 0:
                        if which == T_LAYER_SCALAR:
                            continue   # identity scalar stays
                        raise KeyError(f"layer {li}: {name} not found")
                    self.load_tensor(which, li, dense_bf16(idx).reshape(-1).to(self.device)); n += 1
                for slot, (K, N) in s.linear_shapes(li).items():
                    idx = find(f"model.layers.{li}.{self.GGUF_SLOT_NAMES[slot]}")
                    if idx < 0:
                        raise KeyError(f"layer {li}: {self.GGUF_SLOT_NAMES[slot]} not found")
                    t, raw = read(idx)
                    if list(t.shape[:t.n_dims]) != [N, K]:
                        raise ValueError(f"{t.name.decode()}: shape {list(t.shape[:t.n_dims])}, expected {[N, K]}")
                    self.load_iq_linear(li, slot, t.ggml_type, raw.to(self.device)); n += 1
                if s.hidden_size_per_layer_input > 0:
                    idx = find(f"model.layers.{li}.post_per_layer_input_norm")
                    if idx < 0:
                        raise KeyError(f"layer {li}: post_per_layer_input_norm not found")
                    self.load_tensor(T_POST_PLE_NORM, li, dense_bf16(idx).reshape(-1).to(self.device)); n += 1
            if s.hidden_size_per_layer_input > 0:
                LP = s.hidden_size_per_layer_input * s.num_layers
                idx = find("model.per_layer_model_projection")
                if idx < 0:
                    raise KeyError("per_layer_model_proj.weight not found")
                t, raw = read(idx)
                if list(t.shape[:t.n_dims]) != [LP, s.hidden_size]:
                    raise ValueError(f"{t.name.decode()}: shape {list(t.shape[:t.n_dims])}, expected {[LP, s.hidden_size]}")
                self.load_iq_linear(0, L_PLE_MODEL_PROJ, t.ggml_type, raw.to(self.device)); n += 1
                idx = find("model.per_layer_projection_norm")
                if idx < 0:
                    raise KeyError("per_layer_proj_norm.weight not found")
                self.load_tensor(T_PLE_PROJ_NORM, 0, dense_bf16(idx).reshape(-1).to(self.device)); n += 1
                idx = find("model.embed_tokens_per_layer")
                if idx < 0:
                    raise KeyError("per_layer_token_embd.weight not found")
                t, raw = read(idx)
                if list(t.shape[:t.n_dims]) != [s.vocab_size_per_layer_input, LP]:
                    raise ValueError(f"{t.name.decode()}: shape {list(t.shape[:t.n_dims])}, expected {[s.vocab_size_per_layer_input, LP]}")
                if s.ple_embed_ggml_type == 0:
                    self.load_tensor(T_PLE_EMBED, 0, dense_bf16(idx).to(self.device))
                else:
                    if t.ggml_type != s.ple_embed_ggml_type:
                        raise ValueError(f"per_layer_token_embd is GGML type {t.ggml_type}, the model was created for {s.ple_embed_ggml_type}")
                    self.load_tensor(T_PLE_EMBED, 0, raw.to(self.device))
                n += 1
            return n
        finally:
            lib.vmx_gguf_close(h)

    def iq_weight_bytes(self, slot_types=None) -> int:
        slot_types = slot_types or GEMMA4_12B_IQ3_MIX
        s = self.spec
        return sum(N * (K // 256) * IQ_TYPE_SIZE[slot_types[slot]] for li in range(s.num_layers)
                   for slot, (K, N) in s.linear_shapes(li).items())

    def forward_decode_batch(self, input_ids, positions, slot_mapping, block_tables, seq_lens, max_seq_len=None, out_ids=None,
                             logits=None):
        """All arguments are int32 CUDA tensors; returns out_ids (int32 [B])."""
        B = input_ids.numel()
        if out_ids is None:
            out_ids = torch.empty(B, dtype=torch.int32, device=self.device)
        msl = int(max_seq_len if max_seq_len is not None else self.spec.max_seq_len)
        check(lib.vmx_gemma4_decode(self._h, input_ids.data_ptr(), positions.data_ptr(), slot_mapping.data_ptr(),
                                    block_tables.data_ptr(), seq_lens.data_ptr(), B, block_tables.shape[1], msl, out_ids.data_ptr(),
                                    None if logits is None else logits.data_ptr(), self._stream()))
        return out_ids

    def advance(self, input_ids, next_ids, positions, slot_mapping, seq_lens, block_tables) -> None:
        check(lib.vmx_decode_advance(input_ids.data_ptr(), next_ids.data_ptr(), positions.data_ptr(), slot_mapping.data_ptr(),
                                     seq_lens.data_ptr(), block_tables.data_ptr(), input_ids.numel(), block_tables.shape[1],
                                     self.spec.block_size, self._stream()))

    def prefill(self, input_ids, seqlen_offset: int, block_table, out_id=None, logits_last=None):
        """Native prompt path (vmx_gemma4_prefill): all of input_ids in passes of 16 rows; returns the greedy id after the last
        prompt token (int32 [1]); logits_last: optional bf16 [vocab] for that row."""
        s = self.spec
        S = input_ids.numel()
        bt = block_table.reshape(-1).to(torch.int32).contiguous()
        pos = torch.arange(seqlen_offset, seqlen_offset + S, dtype=torch.int64, device=self.device)
        slots = (bt[(pos // s.block_size)].to(torch.int64) * s.block_size + pos % s.block_size).to(torch.int32).contiguous()
        if out_id is None:
            out_id = torch.empty(1, dtype=torch.int32, device=self.device)
        check(lib.vmx_gemma4_prefill(self._h, input_ids.contiguous().data_ptr(), int(seqlen_offset), slots.data_ptr(), bt.data_ptr(), S,
                                     out_id.data_ptr(), None if logits_last is None else logits_last.data_ptr(), self._stream()))
        return out_id

    def mmvq_only(self, batch: int = 1) -> None:
        check(lib.vmx_gemma4_mmvq_only(self._h, batch, self._stream()))

    def mmvq_select(self, mask: int = 15) -> None:
        """Which projections mmvq_only launches and mmvq_info counts: 1 q|k|v, 2 o_proj, 4 gate|up, 8 down_proj."""
        check(lib.vmx_gemma4_mmvq_select(self._h, int(mask)))

    def mmvq_info(self, batch: int = 1):
        n, by = C.c_int(), C.c_double()
        check(lib.vmx_gemma4_mmvq_info(self._h, batch, C.byref(n), C.byref(by)))
        return n.value, by.value

    def kv_scales_reset(self, headroom: float = 1.5) -> None:
        """KVScales::reset on every layer: the next K/V write of each layer pins its FP8 scales from what it writes."""
        check(lib.vmx_gemma4_kv_scales_reset(self._h, headroom, self._stream()))

    def kv_scales(self, layer: int):
        """(k_scale, v_scale, calibrated) that layer `layer` attends with."""
        k, v, cal = C.c_float(), C.c_float(), C.c_int()
        check(lib.vmx_gemma4_kv_scales_get(self._h, layer, C.byref(k), C.byref(v), C.byref(cal), self._stream()))
        return k.value, v.value, bool(cal.value)

    def debug_kv(self, layer: int, which: int, t: torch.Tensor, write: bool) -> None:
        """Parity aid: copy a layer's whole K (which = 0) / V (1) cache into `t` or from it (contiguous CUDA tensor of the cache's size)."""
        check(lib.vmx_gemma4_debug_kv(self._h, layer, which, int(write), t.data_ptr(), t.numel() * t.element_size(), self._stream()))
        torch.cuda.current_stream().synchronize()

    def debug_read(self, which: int, shape) -> torch.Tensor:
        t = torch.empty(shape, dtype=torch.bfloat16, device=self.device)
        check(lib.vmx_gemma4_debug_read(self._h, which, t.data_ptr(), t.numel() * 2, self._stream()))
        torch.cuda.current_stream().synchronize()
        return t

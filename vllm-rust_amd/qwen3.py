"""Host-side driver of the quantized Qwen3 decoder (libvmx `vmx_qwen3_*`).

Mirrors the engine-facing surface of the reference's QuantizedQwen3ForCausalLM
(crates/core/src/models/qwen3_quantized.rs:557-1302; ModelForward::forward_decode_batch,
engine/model_forward.rs:66-76): weights are handed over per projection in AWQ checkpoint form,
a decode step takes token ids + per-sequence positions/slots/block tables and returns greedy
token ids (and optionally logits).  All arithmetic happens in the library.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math

import numpy as np
import torch

from ._lib import Qwen3Config, check, lib

T_EMBED, T_LM_HEAD, T_FINAL_NORM, T_COS_SIN, T_INPUT_LN, T_POST_LN, T_Q_NORM, T_K_NORM, T_KV_SCALES, T_ROUTER = range(10)
L_Q, L_K, L_V, L_O, L_GATE, L_UP, L_DOWN = range(7)


@dataclasses.dataclass
class Qwen3Spec:
    hidden_size: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    head_dim: int
    intermediate_size: int
    vocab_size: int
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    max_position: int = 4096
    group_size: int = 128
    qk_norm: bool = True
    tie_word_embeddings: bool = False
    kv_cache_dtype: int = 0
    block_size: int = 16
    num_blocks: int = 64
    max_batch: int = 1
    max_seq_len: int = 1024
    tp_size: int = 1      # with tp_size > 1 the head counts and intermediate_size below are this rank's shares
    tp_rank: int = 0
    max_prefill_tokens: int = 0   # rows per native prefill call (0: prompts go through the decode path in max_batch chunks)
    num_experts: int = 0          # > 0: mixture-of-experts MLP (Mixtral: 8 experts, 2 per token); intermediate_size = expert width
    experts_per_tok: int = 0
    act_dtype: str = "bf16"       # "f16": the model runs in fp16 (EXL3 checkpoints, weight_loader.rs:453-461): fp16 embedding /
                                  # norm / lm_head tensors, fp16 activations, logits and KV cache

    @staticmethod
    def qwen3_8b(**kw):
        return Qwen3Spec(4096, 36, 32, 8, 128, 12288, 151936, **kw)

    @staticmethod
    def mixtral_8x7b(**kw):
        """Mixtral-8x7B geometry (models/mixtral_quantized.rs): Llama-style attention without q/k norms, 8 experts top-2."""
        return Qwen3Spec(4096, 32, 32, 8, 128, 14336, 32000, rms_norm_eps=1e-5, rope_theta=1e6, qk_norm=False, num_experts=8,
                         experts_per_tok=2, **kw)

    @staticmethod
    def qwen3_0_6b(**kw):
        return Qwen3Spec(1024, 28, 16, 8, 128, 3072, 151936, tie_word_embeddings=True, **kw)


def rope_cos_sin_cache(head_dim: int, max_position: int, theta: float, dtype: str = "bf16") -> np.ndarray:
    """cos|sin table as the reference builds it (layers/rotary.rs:84-103): f32 inv_freq and
    angles, cos/sin rounded to the model dtype (bf16, or fp16 with dtype "f16") and widened back to f32."""
    i = np.arange(0, head_dim, 2, dtype=np.float32)
    inv_freq = (np.float32(1.0) / np.power(np.float32(theta), i / np.float32(head_dim))).astype(np.float32)
    t = np.arange(max_position, dtype=np.float32)[:, None]
    freqs = (t * inv_freq[None, :]).astype(np.float32)

    def bf16_round(a):
        u = a.astype(np.float32).view(np.uint32)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.view(np.float32)

    if dtype == "f16":
        return np.concatenate([np.cos(freqs).astype(np.float16), np.sin(freqs).astype(np.float16)], axis=1).astype(np.float32)
    return np.concatenate([bf16_round(np.cos(freqs)), bf16_round(np.sin(freqs))], axis=1).astype(np.float32)


class QuantizedQwen3ForCausalLM:
    def __init__(self, spec: Qwen3Spec, device: int = 0):
        self.spec = spec
        self.device = torch.device("cuda", device)
        cfg = Qwen3Config(spec.hidden_size, spec.num_layers, spec.num_heads, spec.num_kv_heads, spec.head_dim,
                          spec.intermediate_size, spec.vocab_size, spec.rms_norm_eps, spec.max_position, spec.group_size,
                          int(spec.qk_norm), int(spec.tie_word_embeddings), spec.kv_cache_dtype, spec.block_size,
                          spec.num_blocks, spec.max_batch, spec.max_seq_len, spec.tp_size, spec.tp_rank, spec.max_prefill_tokens,
                          spec.num_experts, spec.experts_per_tok)
        self._h = C.c_void_p()
        check(lib.vmx_qwen3_create(C.byref(cfg), C.byref(self._h)))
        assert spec.act_dtype in ("bf16", "f16")
        self.act_torch_dtype = torch.float16 if spec.act_dtype == "f16" else torch.bfloat16
        if spec.act_dtype == "f16":
            check(lib.vmx_qwen3_set_activation_dtype(self._h, 2))   # VMX_F16
        cs = torch.from_numpy(rope_cos_sin_cache(spec.head_dim, spec.max_position, spec.rope_theta, spec.act_dtype)).to(self.device)
        self.load_tensor(T_COS_SIN, 0, cs)

    def set_comm(self, comm) -> None:
        """Attach a vllm_rust_amd.parallel.Communicator (kept alive by the model)."""
        self._comm = comm
        check(lib.vmx_qwen3_set_comm(self._h, comm.handle if comm is not None else None))

    def close(self):
        if self._h:
            lib.vmx_qwen3_destroy(self._h)
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
        check(lib.vmx_qwen3_load_tensor(self._h, which, layer, t.data_ptr(), t.numel() * t.element_size(), self._stream()))
        torch.cuda.current_stream().synchronize()  # `t` may be a temporary

    def load_awq_linear(self, layer: int, slot: int, qweight: torch.Tensor, qzeros: torch.Tensor, scales_f16: torch.Tensor):
        assert qweight.dtype == torch.int32 and qzeros.dtype == torch.int32 and scales_f16.dtype == torch.float16
        check(lib.vmx_qwen3_load_awq_linear(self._h, layer, slot, qweight.data_ptr(), qzeros.data_ptr(), scales_f16.data_ptr(),
                                            self._stream()))
        torch.cuda.current_stream().synchronize()

    def load_gptq_linear(self, layer: int, slot: int, qweight: torch.Tensor, qzeros, scales_f16: torch.Tensor):
        """qweight int32 [in/8, out], qzeros int32 [in/group, out/8] or None (symmetric zero point 8), scales f16 [in/group, out]."""
        assert qweight.dtype == torch.int32 and scales_f16.dtype == torch.float16 and (qzeros is None or qzeros.dtype == torch.int32)
        qweight, scales_f16 = qweight.contiguous(), scales_f16.contiguous()
        qzeros = qzeros.contiguous() if qzeros is not None else None
        check(lib.vmx_qwen3_load_gptq_linear(self._h, layer, slot, qweight.data_ptr(), qzeros.data_ptr() if qzeros is not None else None,
                                             scales_f16.data_ptr(), self._stream()))
        torch.cuda.current_stream().synchronize()

    def load_iq_linear(self, layer: int, slot: int, ggml_type: int, blocks: torch.Tensor) -> None:
        """blocks: uint8 tensor (device or host) with the raw GGUF bytes of the [out, in] weight."""
        blocks = blocks.contiguous()
        check(lib.vmx_qwen3_load_iq_linear(self._h, layer, slot, ggml_type, blocks.data_ptr(), blocks.numel(), self._stream()))
        torch.cuda.current_stream().synchronize()

    def load_dense_linear(self, layer: int, slot: int, weight: torch.Tensor) -> None:
        """weight: bf16 [out, in] (device or host): one projection of an unquantized checkpoint."""
        assert weight.dtype == torch.bfloat16 and weight.dim() == 2
        weight = weight.contiguous()
        check(lib.vmx_qwen3_load_dense_linear(self._h, layer, slot, weight.data_ptr(), weight.numel() * 2, self._stream()))
        torch.cuda.current_stream().synchronize()

    def load_exl3_linear(self, layer: int, slot: int, trellis: torch.Tensor, suh: torch.Tensor, svh: torch.Tensor, codebook: int = 0) -> None:
        """trellis: int16 [in/16, out/16, 16 * bits]; suh f16 [in]; svh f16 [out]  (the tensors of an EXL3 checkpoint entry)."""
        assert trellis.dtype == torch.int16 and trellis.dim() == 3 and suh.dtype == torch.float16 and svh.dtype == torch.float16
        K, N, bits = trellis.shape[0] * 16, trellis.shape[1] * 16, trellis.shape[2] // 16
        trellis, suh, svh = trellis.contiguous(), suh.contiguous(), svh.contiguous()
        check(lib.vmx_qwen3_load_exl3_linear(self._h, layer, slot, trellis.data_ptr(), suh.data_ptr(), svh.data_ptr(), K, N, bits, codebook,
                                             self._stream()))
        torch.cuda.current_stream().synchronize()

    def load_gptq_expert(self, layer: int, expert: int, w1, w3, w2) -> None:
        """w1 / w3 / w2: (qweight int32 [K/8, N], scales f16 [K/g, N]) of the gate, up and down projections of one expert."""
        for qw, sc in (w1, w3, w2):
            assert qw.dtype == torch.int32 and sc.dtype == torch.float16 and qw.is_contiguous() and sc.is_contiguous()
        check(lib.vmx_qwen3_load_gptq_expert(self._h, layer, expert, w1[0].data_ptr(), w1[1].data_ptr(), w3[0].data_ptr(), w3[1].data_ptr(),
                                             w2[0].data_ptr(), w2[1].data_ptr(), self._stream()))
        torch.cuda.current_stream().synchronize()

    def load_synthetic_moe(self, seed: int = 0, keep_host_copy: bool = False):
        """Random-init Mixtral-style checkpoint: AWQ attention projections, bf16 router, GPTQ (symmetric) experts."""
        s = self.spec
        ck = self._load_synthetic_dense(seed, keep_host_copy)
        g = torch.Generator(device=self.device).manual_seed(seed + 1)

        def keep(name, t):
            if keep_host_copy:
                ck[name] = t.cpu()

        H, I = s.hidden_size, s.intermediate_size
        for li in range(s.num_layers):
            for slot, (K, N) in self.linear_shapes().items():
                if slot not in (L_Q, L_K, L_V, L_O):
                    continue
                qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=self.device, generator=g)
                qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // s.group_size, N // 8), dtype=torch.int32, device=self.device, generator=g)
                sc = ((torch.rand((K // s.group_size, N), device=self.device, generator=g) * 0.5 + 0.75) * (1.0 / (6.5 * math.sqrt(K)))).to(torch.float16)
                self.load_awq_linear(li, slot, qw, qz, sc)
                keep(f"{li}.{slot}.qweight", qw); keep(f"{li}.{slot}.qzeros", qz); keep(f"{li}.{slot}.scales", sc)
            router = (torch.randn((s.num_experts, H), device=self.device, generator=g) * 0.3).to(torch.bfloat16)
            self.load_tensor(T_ROUTER, li, router); keep(f"{li}.router", router)
            for e in range(s.num_experts):
                def mk(K, N):
                    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int32, device=self.device, generator=g)
                    sc = ((torch.rand((K // s.group_size, N), device=self.device, generator=g) * 0.5 + 0.75) * (1.0 / (4.6 * math.sqrt(K)))).to(torch.float16)
                    return qw, sc
                w1, w3, w2 = mk(H, I), mk(H, I), mk(I, H)
                self.load_gptq_expert(li, e, w1, w3, w2)
                for nm, (qw, sc) in (("w1", w1), ("w3", w3), ("w2", w2)):
                    keep(f"{li}.e{e}.{nm}.qweight", qw); keep(f"{li}.e{e}.{nm}.scales", sc)
        return ck

    def linear_shapes(self):
        s = self.spec
        nq, nkv = s.num_heads * s.head_dim, s.num_kv_heads * s.head_dim
        return {L_Q: (s.hidden_size, nq), L_K: (s.hidden_size, nkv), L_V: (s.hidden_size, nkv), L_O: (nq, s.hidden_size),
                L_GATE: (s.hidden_size, s.intermediate_size), L_UP: (s.hidden_size, s.intermediate_size),
                L_DOWN: (s.intermediate_size, s.hidden_size)}

    IQ_TYPE_SIZE = {17: 74, 22: 82, 18: 98, 21: 110, 23: 136}

    def load_synthetic_iq(self, slot_types, seed: int = 0, keep_host_copy: bool = False):
        """Random valid I-quant blocks for every projection (slot_types: ggml type id per slot, e.g. the
        Gemma-4-12B UD-IQ3_XXS mix), bf16 norms/embeddings as load_synthetic.  Super-block scales are sized so
        that activations stay O(1)."""
        s = self.spec
        ck = self._load_synthetic_dense(seed, keep_host_copy)
        g = torch.Generator(device=self.device).manual_seed(seed + 1)
        for li in range(s.num_layers):
            for slot, (K, N) in self.linear_shapes().items():
                t = slot_types[slot]
                ts = self.IQ_TYPE_SIZE[t]
                nb = N * (K // 256)
                raw = torch.randint(0, 256, (nb, ts), dtype=torch.uint8, device=self.device, generator=g)
                # f16 super-scale d: grid magnitudes are O(10-40) times a sub-scale O(1-8): aim for ~1/sqrt(K) weights
                d = (torch.rand(nb, device=self.device, generator=g) * 0.5 + 0.75) * (1.0 / (120.0 * math.sqrt(K)))
                if t == 23:
                    d = d * 6.0   # IQ4_XS: values +-127 times (ls - 32)
                dbits = d.to(torch.float16).view(torch.int16)
                raw[:, 0] = (dbits & 0xFF).to(torch.uint8)
                raw[:, 1] = ((dbits >> 8) & 0xFF).to(torch.uint8)
                self.load_iq_linear(li, slot, t, raw.reshape(-1))
                if keep_host_copy:
                    ck[f"{li}.{slot}.iq"] = (t, raw.reshape(-1).cpu())
        return ck

    def load_synthetic_tp(self, full_spec, seed: int = 0):
        """This rank's shard of ONE seeded AWQ checkpoint of `full_spec` (self.spec = shard_spec(full_spec, rank, world)): every
        rank draws the same full tensors from the same generator and keeps its slice -- q/k/v/gate/up by output columns, o/down
        by input rows (whole quantisation groups), lm_head by vocabulary rows; embedding and norms whole -- so a TP run computes
        the same model as the single-GPU run with this seed (load_synthetic(seed)), up to the bf16 all-reduce."""
        from .parallel import local_experts, shard_awq_column, shard_awq_row
        s, f = self.spec, full_spec
        r, w = s.tp_rank, s.tp_size
        g = torch.Generator(device=self.device).manual_seed(seed)
        emb = (torch.randn((f.vocab_size, f.hidden_size), device=self.device, generator=g) * 0.5).to(torch.bfloat16)
        self.load_tensor(T_EMBED, 0, emb)
        if not f.tie_word_embeddings:
            head = (torch.randn((f.vocab_size, f.hidden_size), device=self.device, generator=g) * 0.05).to(torch.bfloat16)
            rows = f.vocab_size // w
            self.load_tensor(T_LM_HEAD, 0, head[r * rows:(r + 1) * rows].contiguous())
            del head
        del emb
        self.load_tensor(T_FINAL_NORM, 0, (1.0 + 0.1 * torch.randn(f.hidden_size, device=self.device, generator=g)).to(torch.bfloat16))
        for li in range(f.num_layers):
            for which, n in ((T_INPUT_LN, f.hidden_size), (T_POST_LN, f.hidden_size), (T_Q_NORM, f.head_dim), (T_K_NORM, f.head_dim)):
                self.load_tensor(which, li, (1.0 + 0.1 * torch.randn(n, device=self.device, generator=g)).to(torch.bfloat16))
        g = torch.Generator(device=self.device).manual_seed(seed + 1)
        H, I = f.hidden_size, f.intermediate_size
        nq, nkv = f.num_heads * f.head_dim, f.num_kv_heads * f.head_dim
        full_shapes = {0: (H, nq), 1: (H, nkv), 2: (H, nkv), 3: (nq, H), 4: (H, I), 5: (H, I), 6: (I, H)}
        if f.num_experts > 0:   # the draw order of load_synthetic_moe: attention projections, router, then every expert
            full_shapes = {k: v for k, v in full_shapes.items() if k <= 3}
            mine = local_experts(f.num_experts, r, w)
        for li in range(f.num_layers):
            for slot, (K, N) in full_shapes.items():
                qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=self.device, generator=g)
                qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // f.group_size, N // 8), dtype=torch.int32, device=self.device, generator=g)
                sc = ((torch.rand((K // f.group_size, N), device=self.device, generator=g) * 0.5 + 0.75) * (1.0 / (6.5 * math.sqrt(K)))).to(torch.float16)
                if slot in (3, 6):
                    qw, qz, sc = shard_awq_row(qw, qz, sc, r, w, f.group_size)
                else:
                    qw, qz, sc = shard_awq_column(qw, qz, sc, r, w)
                self.load_awq_linear(li, slot, qw, qz, sc)
            if f.num_experts > 0:
                router = (torch.randn((f.num_experts, H), device=self.device, generator=g) * 0.3).to(torch.bfloat16)
                self.load_tensor(T_ROUTER, li, router)
                for e in range(f.num_experts):
                    def mk(K, N):
                        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int32, device=self.device, generator=g)
                        sc = ((torch.rand((K // f.group_size, N), device=self.device, generator=g) * 0.5 + 0.75) * (1.0 / (4.6 * math.sqrt(K)))).to(torch.float16)
                        return qw, sc
                    w1, w3, w2 = mk(H, I), mk(H, I), mk(I, H)   # drawn on every rank to keep the stream aligned
                    if e in mine:
                        self.load_gptq_expert(li, e, w1, w3, w2)

    def load_synthetic_bf16(self, seed: int = 0, keep_host_copy: bool = False):
        """Seeded random-init UNQUANTIZED checkpoint of the architecture: every projection a bf16 [out, in] matrix with
        ~1/sqrt(in) entries, bf16 norms and embeddings as load_synthetic."""
        s = self.spec
        ck = self._load_synthetic_dense(seed, keep_host_copy)
        g = torch.Generator(device=self.device).manual_seed(seed + 2)
        for li in range(s.num_layers):
            for slot, (K, N) in self.linear_shapes().items():
                w = (torch.randn((N, K), device=self.device, generator=g) * (1.0 / math.sqrt(K))).to(torch.bfloat16)
                self.load_dense_linear(li, slot, w)
                if keep_host_copy:
                    ck[f"{li}.{slot}.dense"] = w.cpu()
        return ck

    def load_synthetic_exl3(self, bits: int = 4, codebook: int = 0, seed: int = 0, keep_host_copy: bool = False):
        """Random trellis words for every projection (EXL3 at `bits` bpw), random sign vectors; svh carries the scale that
        keeps activations O(1) (decoded weights have unit-order magnitude).  bf16 norms/embeddings as load_synthetic."""
        s = self.spec
        ck = self._load_synthetic_dense(seed, keep_host_copy)
        g = torch.Generator(device=self.device).manual_seed(seed + 1)
        for li in range(s.num_layers):
            for slot, (K, N) in self.linear_shapes().items():
                tr = torch.randint(-32768, 32767, (K // 16, N // 16, 16 * bits), dtype=torch.int16, device=self.device, generator=g)
                suh = (torch.randint(0, 2, (K,), device=self.device, generator=g) * 2 - 1).to(torch.float16)
                svh = ((torch.randint(0, 2, (N,), device=self.device, generator=g) * 2 - 1).to(torch.float32)
                       * (torch.rand(N, device=self.device, generator=g) * 0.5 + 0.75) / math.sqrt(K)).to(torch.float16)
                self.load_exl3_linear(li, slot, tr, suh, svh, codebook)
                if keep_host_copy:
                    ck[f"{li}.{slot}.exl3"] = (bits, codebook, tr.cpu(), suh.cpu(), svh.cpu())
        return ck

    def load_synthetic(self, seed: int = 0, keep_host_copy: bool = False):
        """Random-init weights of the model's architecture in AWQ checkpoint form (seeded).
        Scales are sized so activations stay O(1) through the stack.  Returns the host copy of the
        checkpoint when keep_host_copy (tiny test models only)."""
        s = self.spec
        ck = self._load_synthetic_dense(seed, keep_host_copy)
        g = torch.Generator(device=self.device).manual_seed(seed + 1)

        def keep(name, t):
            if keep_host_copy:
                ck[name] = t.cpu()

        for li in range(s.num_layers):
            for slot, (K, N) in self.linear_shapes().items():
                qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=self.device, generator=g)
                qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // s.group_size, N // 8), dtype=torch.int32, device=self.device,
                                   generator=g)
                # uniform int4 minus zero point has std ~6.5; aim for unit-variance outputs
                sc = (torch.rand((K // s.group_size, N), device=self.device, generator=g) * 0.5 + 0.75) * (1.0 / (6.5 * math.sqrt(K)))
                sc = sc.to(torch.float16)
                self.load_awq_linear(li, slot, qw, qz, sc)
                keep(f"{li}.{slot}.qweight", qw); keep(f"{li}.{slot}.qzeros", qz); keep(f"{li}.{slot}.scales", sc)
        return ck

    def _load_synthetic_dense(self, seed: int, keep_host_copy: bool):
        """Embedding, lm_head and norm weights (in the model's 16-bit dtype)."""
        s = self.spec
        g = torch.Generator(device=self.device).manual_seed(seed)
        ck = {}

        def keep(name, t):
            if keep_host_copy:
                ck[name] = t.cpu()

        emb = (torch.randn((s.vocab_size, s.hidden_size), device=self.device, generator=g) * 0.5).to(self.act_torch_dtype)
        self.load_tensor(T_EMBED, 0, emb); keep("embed", emb)
        if not s.tie_word_embeddings:
            rows = s.vocab_size // s.tp_size if s.tp_size > 1 else s.vocab_size
            head = (torch.randn((rows, s.hidden_size), device=self.device, generator=g) * 0.05).to(self.act_torch_dtype)
            self.load_tensor(T_LM_HEAD, 0, head); keep("lm_head", head)
            del head
        del emb
        fn = (1.0 + 0.1 * torch.randn(s.hidden_size, device=self.device, generator=g)).to(self.act_torch_dtype)
        self.load_tensor(T_FINAL_NORM, 0, fn); keep("final_norm", fn)
        for li in range(s.num_layers):
            for which, n, name in ((T_INPUT_LN, s.hidden_size, "input_ln"), (T_POST_LN, s.hidden_size, "post_ln"),
                                   (T_Q_NORM, s.head_dim, "q_norm"), (T_K_NORM, s.head_dim, "k_norm")):
                w = (1.0 + 0.1 * torch.randn(n, device=self.device, generator=g)).to(self.act_torch_dtype)
                self.load_tensor(which, li, w); keep(f"{li}.{name}", w)
        return ck

    def forward_decode_batch(self, input_ids, positions, slot_mapping, block_tables, seq_lens, max_seq_len=None,
                             out_ids=None, logits=None):
        """All arguments are int32 CUDA tensors; returns out_ids (int32 [B])."""
        B = input_ids.numel()
        if out_ids is None:
            out_ids = torch.empty(B, dtype=torch.int32, device=self.device)
        msl = int(max_seq_len if max_seq_len is not None else self.spec.max_seq_len)
        check(lib.vmx_qwen3_decode(self._h, input_ids.data_ptr(), positions.data_ptr(), slot_mapping.data_ptr(),
                                   block_tables.data_ptr(), seq_lens.data_ptr(), B, block_tables.shape[1], msl,
                                   out_ids.data_ptr(), None if logits is None else logits.data_ptr(), self._stream()))
        return out_ids

    def prefill(self, input_ids: torch.Tensor, seqlen_offset: int, block_table: torch.Tensor, out_id=None, logits_last=None):
        """Native prompt path (vmx_qwen3_prefill): all of input_ids (at most max_prefill_tokens) in one pass; returns the greedy
        id after the last prompt token (int32 [1]); logits_last: optional bf16 [vocab] for that row."""
        s = self.spec
        S = input_ids.numel()
        bt = block_table.reshape(-1).to(torch.int32).contiguous()
        pos = torch.arange(seqlen_offset, seqlen_offset + S, dtype=torch.int64, device=self.device)
        slots = (bt[(pos // s.block_size)].to(torch.int64) * s.block_size + pos % s.block_size).to(torch.int32).contiguous()
        if out_id is None:
            out_id = torch.empty(1, dtype=torch.int32, device=self.device)
        check(lib.vmx_qwen3_prefill(self._h, input_ids.contiguous().data_ptr(), int(seqlen_offset), slots.data_ptr(), bt.data_ptr(), S,
                                    out_id.data_ptr(), None if logits_last is None else logits_last.data_ptr(), self._stream()))
        return out_id

    def forward(self, input_ids: torch.Tensor, seqlen_offset: int, block_table: torch.Tensor, return_logits: bool = True):
        """ModelForward::forward for one sequence (engine/model_forward.rs:22-64): a prompt (chunk) of S tokens at
        positions offset..offset+S-1.  Runs through the decode executor max_batch tokens at a time: token i is a
        pseudo-sequence of length offset+i+1 over the same block table, its K/V row is written before the chunk's
        attention runs, so the result is causal attention over everything up to and including itself.
        Returns (greedy ids int32 [S], logits bf16 [S, vocab] or None)."""
        s = self.spec
        S = input_ids.numel()
        bt1 = block_table.reshape(1, -1).to(torch.int32)
        ids_out = torch.empty(S, dtype=torch.int32, device=self.device)
        logits = torch.empty((S, s.vocab_size), dtype=self.act_torch_dtype, device=self.device) if return_logits else None
        check(lib.vmx_qwen3_set_rows_share_cache(self._h, 1))
        try:
            self._forward_chunks(input_ids, seqlen_offset, bt1, ids_out, logits)
        finally:
            check(lib.vmx_qwen3_set_rows_share_cache(self._h, 0))
        return ids_out, logits

    def _forward_chunks(self, input_ids, seqlen_offset, bt1, ids_out, logits):
        s = self.spec
        S = input_ids.numel()
        for c0 in range(0, S, s.max_batch):
            n = min(s.max_batch, S - c0)
            pos = torch.arange(seqlen_offset + c0, seqlen_offset + c0 + n, dtype=torch.int32, device=self.device)
            slots = (bt1[0, (pos // s.block_size).long()] * s.block_size + pos % s.block_size).to(torch.int32)
            bt = bt1.expand(n, -1).contiguous()
            self.forward_decode_batch(input_ids[c0:c0 + n].contiguous(), pos, slots, bt, (pos + 1).contiguous(),
                                      out_ids=ids_out[c0:c0 + n], logits=None if logits is None else logits[c0:c0 + n])

    def debug_kv(self, layer: int, which: int, t: torch.Tensor, write: bool) -> None:
        """Parity aid: copy a layer's whole K (which = 0) / V (1) cache into `t` or from it (contiguous CUDA tensor of the cache's size)."""
        check(lib.vmx_qwen3_debug_kv(self._h, layer, which, int(write), t.data_ptr(), t.numel() * t.element_size(), self._stream()))
        torch.cuda.current_stream().synchronize()

    def debug_read(self, which: int, shape) -> torch.Tensor:
        t = torch.empty(shape, dtype=self.act_torch_dtype, device=self.device)
        check(lib.vmx_qwen3_debug_read(self._h, which, t.data_ptr(), t.numel() * 2, self._stream()))
        torch.cuda.current_stream().synchronize()
        return t

    def advance(self, input_ids, next_ids, positions, slot_mapping, seq_lens, block_tables) -> None:
        check(lib.vmx_decode_advance(input_ids.data_ptr(), next_ids.data_ptr(), positions.data_ptr(), slot_mapping.data_ptr(),
                                     seq_lens.data_ptr(), block_tables.data_ptr(), input_ids.numel(), block_tables.shape[1],
                                     self.spec.block_size, self._stream()))

    def linears_only(self, batch: int = 1) -> None:
        check(lib.vmx_qwen3_linears_only(self._h, batch, self._stream()))

    def linears_select(self, mask: int = 15) -> None:
        """Which linears linears_only / linears_info cover: bit 0 q|k|v, 1 o_proj, 2 gate|up, 3 down_proj."""
        check(lib.vmx_qwen3_linears_select(self._h, int(mask)))

    def linears_info(self, batch: int = 1):
        n, by = C.c_int(), C.c_double()
        check(lib.vmx_qwen3_linears_info(self._h, batch, C.byref(n), C.byref(by)))
        return n.value, by.value

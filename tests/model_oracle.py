"""CPU restatement of the reference's quantized Qwen3 decode step, composed from the oracle's
operators in the reference's order (models/qwen3_quantized.rs:694-927 attention, :532-552 MLP,
:1036-1088 layer, lm_head :1258-1302).  Test infrastructure only."""
import numpy as np

from oracle import oracle as orc


def _np(t):
    import torch
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    if t.dtype == torch.float16:
        return t.view(torch.int16).numpy().view(np.uint16)
    if t.dtype == torch.int32:
        return t.numpy().view(np.uint32)
    return t.numpy()


def add_bf16(a_u16, b_u16):
    return orc.f32_to_bf16(orc.bf16_to_f32(a_u16) + orc.bf16_to_f32(b_u16))


class OracleQwen3:
    def __init__(self, spec, ck, cos_sin):
        self.s = spec
        self.cos_sin = cos_sin
        self.ck = {k: _np(v) for k, v in ck.items() if not isinstance(v, tuple)}
        s = spec
        self.lin = {}
        self.iq = {}
        for li in range(s.num_layers):
            for slot in range(7):
                if f"{li}.{slot}.iq" in ck:
                    t, raw = ck[f"{li}.{slot}.iq"]
                    self.iq[(li, slot)] = (t, raw.numpy())
                    continue
                qw, qz, sc = (self.ck[f"{li}.{slot}.{n}"] for n in ("qweight", "qzeros", "scales"))
                K, N = qw.shape[0], qw.shape[1] * 8
                gq = orc.awq_to_gptq_qweight(qw, K, N)                       # awq_marlin.rs:622
                gz = orc.repack_awq_nibbles(qz)                              # :642
                sb = orc.f32_to_bf16(orc.f16_to_f32(sc))                     # :649-656
                self.lin[(li, slot)] = (gq, gz, sb, K, N)
        kv_shape = (s.num_blocks, s.block_size, s.num_kv_heads, s.head_dim)
        kv_np = np.uint8 if s.kv_cache_dtype == 1 else np.uint16
        self.k_cache = [np.zeros(kv_shape, kv_np) for _ in range(s.num_layers)]
        self.v_cache = [np.zeros(kv_shape, kv_np) for _ in range(s.num_layers)]
        self.kv_scales = (1.0, 1.0)

    def linear(self, li, slot, x_u16):
        if (li, slot) in self.iq:   # IqLinear::forward_inner (gguf/mod.rs:329-377): f32 cast, q8_1 MMVQ, cast back
            t, raw = self.iq[(li, slot)]
            x = orc.bf16_to_f32(x_u16)
            n_in = x.shape[1]
            n_out = raw.size // (n_in // 256 * orc.IQ_TYPE_SIZE[t])
            return orc.f32_to_bf16(orc.iq_matmul(t, raw, x, n_out, n_in))
        gq, gz, sb, K, N = self.lin[(li, slot)]
        return orc.int4_gemv_bf16(x_u16, gq, sb, gz, None, K, N, self.s.group_size)

    def decode(self, ids, positions, slots, block_tables, seq_lens):
        s = self.s
        B = len(ids)
        h = self.ck["embed"][ids]                                           # [B, H] bf16
        for li in range(s.num_layers):
            x = orc.rms_norm_bf16(h, self.ck[f"{li}.input_ln"], s.rms_norm_eps)
            q, k, v = self.linear(li, 0, x), self.linear(li, 1, x), self.linear(li, 2, x)
            if s.qk_norm:
                q = orc.rms_norm_bf16(q.reshape(B * s.num_heads, s.head_dim), self.ck[f"{li}.q_norm"], s.rms_norm_eps).reshape(B, -1)
                k = orc.rms_norm_bf16(k.reshape(B * s.num_kv_heads, s.head_dim), self.ck[f"{li}.k_norm"], s.rms_norm_eps).reshape(B, -1)
            q, k = orc.rope_neox_bf16(positions, q, k, self.cos_sin, s.head_dim, s.head_dim, s.num_heads, s.num_kv_heads)
            orc.reshape_and_cache(k, v, self.k_cache[li], self.v_cache[li], slots, s.num_kv_heads, s.head_dim, s.block_size,
                                  s.kv_cache_dtype, self.kv_scales[0], self.kv_scales[1])
            psize = 64 if int(seq_lens.max()) <= 4096 else 256               # cuda_kernels.rs:722-728
            attn = orc.paged_attention_v2(q, self.k_cache[li], self.v_cache[li], block_tables, seq_lens,
                                          float(1.0 / np.sqrt(s.head_dim)), s.num_heads, s.num_kv_heads, s.head_dim,
                                          s.block_size, psize, 0, s.kv_cache_dtype, self.kv_scales[0], self.kv_scales[1])
            h = add_bf16(self.linear(li, 3, attn), h)
            x = orc.rms_norm_bf16(h, self.ck[f"{li}.post_ln"], s.rms_norm_eps)
            a = orc.silu_and_mul_bf16(self.linear(li, 4, x), self.linear(li, 5, x))
            h = add_bf16(h, self.linear(li, 6, a))
        x = orc.rms_norm_bf16(h, self.ck["final_norm"], s.rms_norm_eps)
        head = self.ck["embed"] if s.tie_word_embeddings else self.ck["lm_head"]
        logits = orc.f32_to_bf16((orc.bf16_to_f32(x).astype(np.float64) @ orc.bf16_to_f32(head).astype(np.float64).T).astype(np.float32))
        return logits, h, x

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
        self.ex = {}
        self.dense = {}
        for li in range(s.num_layers):
            for slot in range(7):
                if f"{li}.{slot}.dense" in ck:   # unquantized Linear: bf16 [out, in]
                    self.dense[(li, slot)] = self.ck[f"{li}.{slot}.dense"]
                    continue
                if f"{li}.{slot}.exl3" in ck:
                    import torch
                    bits, cb, tr, suh, svh = ck[f"{li}.{slot}.exl3"]
                    self.ex[(li, slot)] = (bits, cb, tr.numpy().view(np.uint16).reshape(-1), tr.shape[0] * 16, tr.shape[1] * 16,
                                           suh.view(torch.int16).numpy().view(np.uint16), svh.view(torch.int16).numpy().view(np.uint16))
                    continue
                if f"{li}.{slot}.iq" in ck:
                    t, raw = ck[f"{li}.{slot}.iq"]
                    self.iq[(li, slot)] = (t, raw.numpy())
                    continue
                if f"{li}.{slot}.qweight" not in self.ck:
                    continue   # mixture-of-experts model: no dense gate / up / down
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
        if (li, slot) in self.dense:   # plain Linear of models/qwen3.rs: bf16 matmul, f32 accumulate, bf16 result
            return orc.dense_gemv_bf16(x_u16, self.dense[(li, slot)])
        if (li, slot) in self.ex:   # Exl3Linear::forward (exl3.rs:216-286): cast to fp16, exl3_gemm, cast back to bf16
            bits, cb, tr, K, N, suh, svh = self.ex[(li, slot)]
            x16 = orc.f32_to_f16(orc.bf16_to_f32(x_u16))
            return orc.f32_to_bf16(orc.f16_to_f32(orc.exl3_gemm(x16, tr, suh, svh, K, N, bits, cb)))
        if (li, slot) in self.iq:   # IqLinear::forward_inner (gguf/mod.rs:329-377): f32 cast, q8_1 MMVQ, cast back
            t, raw = self.iq[(li, slot)]
            x = orc.bf16_to_f32(x_u16)
            n_in = x.shape[1]
            n_out = raw.size // (n_in // 256 * orc.IQ_TYPE_SIZE[t])
            return orc.f32_to_bf16(orc.iq_matmul(t, raw, x, n_out, n_in))
        gq, gz, sb, K, N = self.lin[(li, slot)]
        return orc.int4_gemv_bf16(x_u16, gq, sb, gz, None, K, N, self.s.group_size)

    def moe(self, li, x_u16):
        """QuantizedMoELayer::forward (models/mixtral_quantized.rs:138-195): dense router in bf16, softmax rounded to bf16,
        top-k (ties -> smaller index), renormalise, out = sum_j bf16(expert_j(x) * bf16(w_j)) accumulated in bf16."""
        s = self.s
        T, H, I, g = x_u16.shape[0], s.hidden_size, s.intermediate_size, s.group_size
        xf = orc.bf16_to_f32(x_u16).astype(np.float64)
        logits = orc.bf16_to_f32(orc.f32_to_bf16((xf @ orc.bf16_to_f32(self.ck[f"{li}.router"]).astype(np.float64).T).astype(np.float32)))
        e = np.exp(logits - logits.max(-1, keepdims=True)).astype(np.float32)
        p = orc.bf16_to_f32(orc.f32_to_bf16((e / e.sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)))
        ids = np.argsort(-p, axis=-1, kind="stable")[:, :s.experts_per_tok]
        w = np.take_along_axis(p, ids, -1)
        w = (w / w.sum(-1, keepdims=True)).astype(np.float32)
        out = np.zeros((T, H), np.uint16)

        def ex(e, nm):
            return self.ck[f"{li}.e{e}.{nm}.qweight"], orc.f32_to_bf16(orc.f16_to_f32(self.ck[f"{li}.e{e}.{nm}.scales"]))
        for t in range(T):
            acc = np.zeros(H, np.float32)
            for j in range(s.experts_per_tok):
                w1, w3, w2 = ex(ids[t, j], "w1"), ex(ids[t, j], "w3"), ex(ids[t, j], "w2")
                gate = orc.int4_gemv_bf16(x_u16[t:t + 1], w1[0], w1[1], None, None, H, I, g)
                up = orc.int4_gemv_bf16(x_u16[t:t + 1], w3[0], w3[1], None, None, H, I, g)
                hid = orc.silu_and_mul_bf16(gate, up)
                eo = orc.bf16_to_f32(orc.int4_gemv_bf16(hid, w2[0], w2[1], None, None, I, H, g))[0]
                ws = orc.bf16_to_f32(orc.f32_to_bf16(np.float32(w[t, j])))
                acc = orc.bf16_to_f32(orc.f32_to_bf16(acc + orc.bf16_to_f32(orc.f32_to_bf16(eo * ws))))
            out[t] = orc.f32_to_bf16(acc)
        return out

    # ---- a model that runs in fp16 (EXL3 checkpoints: weight_loader.rs:453-461): the same operator order with fp16 tensors.
    # The reference's own vector kernels are bf16-only, so on this path its norms / rotary / activation go through the
    # third-party tensor library (absent here): f32 arithmetic with one rounding to fp16 per operator is assumed -- parity unpinned.
    @staticmethod
    def _f(u16):
        return u16.view(np.float16).astype(np.float32)

    @staticmethod
    def _h(f32):
        return np.asarray(f32, np.float32).astype(np.float16).view(np.uint16)

    def _rms_f16(self, x_u16, w_u16, eps):
        x = self._f(x_u16)
        inv = (np.float32(1.0) / np.sqrt((x.astype(np.float64) ** 2).mean(-1, keepdims=True) + eps)).astype(np.float32)
        return self._h(x * inv * self._f(w_u16)[None, :])

    def decode_f16(self, ids, positions, slots, block_tables, seq_lens):
        s = self.s
        B, hd, Hq, Hkv = len(ids), s.head_dim, s.num_heads, s.num_kv_heads
        f, hh = self._f, self._h
        h = self.ck["embed"][ids]                                           # [B, H] fp16 bits
        cs = self.cos_sin                                                   # f32 [max_pos, hd], values already rounded to fp16
        for li in range(s.num_layers):
            x = self._rms_f16(h, self.ck[f"{li}.input_ln"], s.rms_norm_eps)
            q, k, v = (self.linear_f16(li, sl, x) for sl in (0, 1, 2))
            if s.qk_norm:
                q = self._rms_f16(q.reshape(B * Hq, hd), self.ck[f"{li}.q_norm"], s.rms_norm_eps).reshape(B, -1)
                k = self._rms_f16(k.reshape(B * Hkv, hd), self.ck[f"{li}.k_norm"], s.rms_norm_eps).reshape(B, -1)

            def rope(t_u16, nh):
                t = f(t_u16).reshape(B, nh, hd)
                c, sn = cs[positions][:, None, :hd // 2], cs[positions][:, None, hd // 2:]
                a, b = t[..., :hd // 2], t[..., hd // 2:]
                return hh(np.concatenate([a * c - b * sn, b * c + a * sn], -1)).reshape(B, nh * hd)
            q, k = rope(q, Hq), rope(k, Hkv)
            for b in range(B):
                if slots[b] >= 0:
                    self.k_cache[li].reshape(-1, Hkv, hd)[slots[b]] = k[b].reshape(Hkv, hd)
                    self.v_cache[li].reshape(-1, Hkv, hd)[slots[b]] = v[b].reshape(Hkv, hd)
            attn = np.zeros((B, Hq * hd), np.uint16)
            for b in range(B):
                n = int(seq_lens[b])
                if n == 0:
                    continue
                rows = np.array([block_tables[b, t // s.block_size] * s.block_size + t % s.block_size for t in range(n)])
                K = f(self.k_cache[li].reshape(-1, Hkv, hd)[rows]).astype(np.float64)    # [n, Hkv, hd]
                V = f(self.v_cache[li].reshape(-1, Hkv, hd)[rows]).astype(np.float64)
                qb = f(q[b]).reshape(Hq, hd).astype(np.float64)
                for hq in range(Hq):
                    kv = hq // (Hq // Hkv)
                    lg = (K[:, kv, :] @ qb[hq]) / np.sqrt(hd)
                    p = np.exp(lg - lg.max())
                    attn[b, hq * hd:(hq + 1) * hd] = hh(((p / p.sum()) @ V[:, kv, :]).astype(np.float32))
            h = hh(f(self.linear_f16(li, 3, attn)) + f(h))
            x = self._rms_f16(h, self.ck[f"{li}.post_ln"], s.rms_norm_eps)
            g, u = f(self.linear_f16(li, 4, x)), f(self.linear_f16(li, 5, x))
            a = hh((g / (np.float32(1.0) + np.exp(-g))) * u)
            h = hh(f(h) + f(self.linear_f16(li, 6, a)))
        x = self._rms_f16(h, self.ck["final_norm"], s.rms_norm_eps)
        if getattr(self, "_head64", None) is None:
            head = self.ck["embed"] if s.tie_word_embeddings else self.ck["lm_head"]
            self._head64 = f(head).astype(np.float64).T.copy()
        logits = hh((f(x).astype(np.float64) @ self._head64).astype(np.float32))
        return logits, h, x

    def linear_f16(self, li, slot, x_u16):
        """Exl3Linear::forward on fp16 rows (exl3.rs:216-286): no casts either side."""
        bits, cb, tr, K, N, suh, svh = self.ex[(li, slot)]
        return orc.exl3_gemm(x_u16, tr, suh, svh, K, N, bits, cb)

    def decode(self, ids, positions, slots, block_tables, seq_lens):
        s = self.s
        if getattr(s, "act_dtype", "bf16") == "f16":
            return self.decode_f16(ids, positions, slots, block_tables, seq_lens)
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
            if getattr(s, "num_experts", 0) > 0:
                h = add_bf16(h, self.moe(li, x))
                continue
            a = orc.silu_and_mul_bf16(self.linear(li, 4, x), self.linear(li, 5, x))
            h = add_bf16(h, self.linear(li, 6, a))
        x = orc.rms_norm_bf16(h, self.ck["final_norm"], s.rms_norm_eps)
        if getattr(self, "_head64", None) is None:   # widened once: a 150k-row head is 1.2 GB in f64
            head = self.ck["embed"] if s.tie_word_embeddings else self.ck["lm_head"]
            self._head64 = orc.bf16_to_f32(head).astype(np.float64).T.copy()
        logits = orc.f32_to_bf16((orc.bf16_to_f32(x).astype(np.float64) @ self._head64).astype(np.float32))
        return logits, h, x


def mul_bf16(a_u16, b_u16):
    return orc.f32_to_bf16(orc.bf16_to_f32(a_u16) * orc.bf16_to_f32(b_u16))


def _rbf(x_f32):
    return orc.bf16_to_f32(orc.f32_to_bf16(np.asarray(x_f32, np.float32)))


class OracleGemma4:
    """CPU restatement of the reference's quantized Gemma 4 decode step on its paged path, composed from the oracle's
    operators in the reference's order (models/gemma4_quantized.rs: attention :1311-1424, MLP :1468-1473, layer
    :1743-1774, embedding scale / final norm / tied lm_head / soft-cap :2217-2279)."""

    def __init__(self, spec, ck, cos_sin_sliding, cos_sin_full):
        self.s = spec
        self.cs = (cos_sin_sliding, cos_sin_full)
        self.ck = {k: (_np(v) if not isinstance(v, tuple) else v) for k, v in ck.items()}
        s = spec
        kv_np = np.uint8 if s.kv_cache_dtype == 1 else np.uint16
        self.k_cache, self.v_cache = [], []
        for li in range(s.num_layers):
            hd, kvh, _, _, _ = s.layer_geom(li)
            self.k_cache.append(np.zeros((s.num_blocks, s.block_size, kvh, hd), kv_np))
            self.v_cache.append(np.zeros((s.num_blocks, s.block_size, kvh, hd), kv_np))
        # per-layer KVScales: identity and pinned unless reset_kv_scales() arms the first-write calibration
        self.kvs = [orc.KVScales(calibrated=True) for _ in range(s.num_layers)]
        if s.embed_ggml_type == 0:
            self.embed_f32 = orc.bf16_to_f32(self.ck["embed"])
        else:   # dequantised table (QuantizedEmbedding / QMatMul over the same blocks)
            self.embed_f32 = orc.ggml_dequant(s.embed_ggml_type, self.ck["embed"]).reshape(s.vocab_size, s.hidden_size)

    def linear(self, li, slot, x_u16):
        """IqLinear::forward_inner (gguf/mod.rs:329-377): q8_1 MMVQ up to 16 rows; beyond that the weight is dequantised to dense
        f32 and multiplied with the unquantised activations (float64 accumulate here: summation order is the only freedom)."""
        t, raw = self.ck[f"{li}.{slot}.iq"]
        raw = raw.numpy() if hasattr(raw, "numpy") else raw
        x = orc.bf16_to_f32(x_u16)
        n_in = x.shape[1]
        n_out = raw.size // (n_in // 256 * orc.IQ_TYPE_SIZE[t])
        if x.shape[0] > 16:
            w = orc.iq_dequant(t, raw).reshape(n_out, n_in).astype(np.float64)
            return orc.f32_to_bf16((x.astype(np.float64) @ w.T).astype(np.float32))
        return orc.f32_to_bf16(orc.iq_matmul(t, raw, x, n_out, n_in))

    def reset_kv_scales(self, headroom=1.5):
        for kv in self.kvs:
            kv.reset()
            kv.set_headroom_factor(headroom)

    def per_layer_inputs(self, ids, h0):
        """compute_per_layer_inputs (gemma4_quantized.rs:2143-2192): bf16 [B, layers, P], or None without per-layer inputs."""
        s = self.s
        P, L, B = s.hidden_size_per_layer_input, s.num_layers, len(ids)
        if P == 0:
            return None

        def scalar(v, shape):
            return np.broadcast_to(orc.f32_to_bf16(np.float32(v)), shape)

        if s.ple_embed_ggml_type == 0:
            table = orc.bf16_to_f32(self.ck["ple_embed"]).reshape(s.vocab_size_per_layer_input, L * P)
        else:
            table = orc.ggml_dequant(s.ple_embed_ggml_type, self.ck["ple_embed"]).reshape(s.vocab_size_per_layer_input, L * P)
        emb = mul_bf16(orc.f32_to_bf16(np.ascontiguousarray(table[ids])), scalar(np.sqrt(np.float64(P)), (B, L * P)))
        proj = mul_bf16(self.linear(0, 9, h0), scalar(np.float64(s.hidden_size) ** -0.5, (B, L * P)))
        normed = orc.rms_norm_bf16(proj.reshape(B * L, P), self.ck["ple_proj_norm"], s.rms_norm_eps).reshape(B, L * P)
        return mul_bf16(add_bf16(normed, emb), scalar(np.float64(2.0) ** -0.5, (B, L * P))).reshape(B, L, P)

    def decode(self, ids, positions, slots, block_tables, seq_lens):
        from scipy.special import erf
        s = self.s
        B = len(ids)
        norm_bf16 = orc.f32_to_bf16(np.float32(np.sqrt(np.float32(s.hidden_size))))        # affine factor rounded to bf16
        h = mul_bf16(orc.f32_to_bf16(self.embed_f32[ids]), np.broadcast_to(norm_bf16, (B, s.hidden_size)))
        pli = self.per_layer_inputs(ids, h)
        for li in range(s.num_layers):
            hd, kvh, rot, window, keqv = s.layer_geom(li)
            sliding = s.sliding(li)
            x = orc.rms_norm_bf16(h, self.ck[f"{li}.input_ln"], s.rms_norm_eps)
            q = self.linear(li, 0, x)
            q = orc.rms_norm_bf16(q.reshape(B * s.num_heads, hd), self.ck[f"{li}.q_norm"], s.rms_norm_eps).reshape(B, -1)
            cl = s.kv_target(li)      # a KV-shared layer rotates q only and reads its target's cache (:1332-1337, :1718)
            if cl is None:
                cl = li
                k = self.linear(li, 1, x)
                v = k if keqv else self.linear(li, 2, x)
                ones = orc.f32_to_bf16(np.ones(hd, np.float32))
                v = orc.rms_norm_bf16(v.reshape(B * kvh, hd), ones, s.rms_norm_eps).reshape(B, -1)
                k = orc.rms_norm_bf16(k.reshape(B * kvh, hd), self.ck[f"{li}.k_norm"], s.rms_norm_eps).reshape(B, -1)
                q, k = orc.rope_neox_bf16(positions, q, k, self.cs[0 if sliding else 1], rot, hd, s.num_heads, kvh)
                if s.kv_cache_dtype == 1:
                    self.kvs[li].calibrate_if_needed(k, v)      # cache_engine write: first write pins the scales
                orc.reshape_and_cache(k, v, self.k_cache[li], self.v_cache[li], slots, kvh, hd, s.block_size, s.kv_cache_dtype,
                                      float(self.kvs[li].k_scale), float(self.kvs[li].v_scale))
            else:
                q, _ = orc.rope_neox_bf16(positions, q, np.zeros((B, kvh * hd), np.uint16), self.cs[0 if sliding else 1], rot, hd,
                                          s.num_heads, kvh)
            psize = 64 if int(seq_lens.max()) <= 4096 else 256
            attn = orc.paged_attention_v2(q, self.k_cache[cl], self.v_cache[cl], block_tables, seq_lens, 1.0, s.num_heads, kvh, hd,
                                          s.block_size, psize, window, s.kv_cache_dtype, float(self.kvs[cl].k_scale),
                                          float(self.kvs[cl].v_scale))
            o = self.linear(li, 3, attn)
            h = add_bf16(orc.rms_norm_bf16(o, self.ck[f"{li}.post_attn_ln"], s.rms_norm_eps), h)
            x = orc.rms_norm_bf16(h, self.ck[f"{li}.pre_ffn_ln"], s.rms_norm_eps)
            g = orc.bf16_to_f32(self.linear(li, 4, x)).astype(np.float64)
            ge = orc.f32_to_bf16((0.5 * g * (1.0 + erf(g / np.sqrt(2.0)))).astype(np.float32))      # gelu_erf, rounded to bf16
            a = mul_bf16(ge, self.linear(li, 5, x))
            d = self.linear(li, 6, a)
            h = add_bf16(orc.rms_norm_bf16(d, self.ck[f"{li}.post_ffn_ln"], s.rms_norm_eps), h)
            if pli is not None:     # apply_ple (gemma4_quantized.rs:1663-1683); gelu_tanh in f32, one rounding (parity unpinned)
                g = orc.bf16_to_f32(self.linear(li, 7, h)).astype(np.float64)
                ge = orc.f32_to_bf16((0.5 * g * (1.0 + np.tanh(0.7978845608028654 * g * (1.0 + 0.044715 * g * g)))).astype(np.float32))
                contrib = self.linear(li, 8, mul_bf16(ge, np.ascontiguousarray(pli[:, li, :])))
                h = add_bf16(h, orc.rms_norm_bf16(contrib, self.ck[f"{li}.post_ple_norm"], s.rms_norm_eps))
            h = mul_bf16(h, np.broadcast_to(self.ck[f"{li}.scalar"].reshape(1, 1), h.shape))
        x = orc.rms_norm_bf16(h, self.ck["final_norm"], s.rms_norm_eps)
        logits = (orc.bf16_to_f32(x).astype(np.float64) @ self.embed_f32.astype(np.float64).T).astype(np.float32)
        logits = _rbf(logits)
        cap = s.final_logit_softcap
        if cap > 0:
            inv = _rbf(np.float32(1.0 / cap))
            logits = _rbf(_rbf(np.tanh(_rbf(logits * inv))) * np.float32(cap))
        return orc.f32_to_bf16(logits), h, x

"""Quantized Gemma 4 decode executor against the composed CPU oracle (parity unpinned: the reference holds no golden
vectors for the model level; the operators underneath are pinned individually)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(spec, steps, B, slot_types=None, seed=3, calibrate_headroom=None):
    import torch
    import vllm_rust_amd as vmx
    from oracle import oracle as orc
    from tests.model_oracle import OracleGemma4
    from vllm_rust_amd.gemma4 import QuantizedGemma4ForCausalLM
    vmx.init(0)
    model = QuantizedGemma4ForCausalLM(spec)
    ck = model.load_synthetic(slot_types=slot_types, seed=seed, keep_host_copy=True)
    ref = OracleGemma4(spec, ck, model.cos_sin_sliding, model.cos_sin_full)
    if calibrate_headroom is not None:      # FP8 scales pinned by the first write, on both sides
        model.kv_scales_reset(calibrate_headroom)
        ref.reset_kv_scales(calibrate_headroom)
        assert model.kv_scales(0) == (1.0, 1.0, False)
    nblk = spec.num_blocks // B
    bt = np.arange(B * nblk, dtype=np.int32).reshape(B, nblk)
    ids = np.array([7, 123, 55, 900][:B], np.int32)
    worst = 0.0
    for p in range(steps):
        pos = np.full(B, p, np.int32)
        slots = (bt[np.arange(B), p // spec.block_size] * spec.block_size + p % spec.block_size).astype(np.int32)
        sl = pos + 1
        logits = torch.empty((B, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
        out = model.forward_decode_batch(*(torch.from_numpy(a).cuda() for a in (ids, pos, slots, bt, sl)), logits=logits)
        torch.cuda.synchronize()
        exp, h_exp, _ = ref.decode(ids, pos, slots, bt, sl)
        got = orc.bf16_to_f32(logits.cpu().view(torch.int16).numpy().view(np.uint16))
        expf = orc.bf16_to_f32(exp)
        rel = float(np.sqrt(((got - expf) ** 2).sum() / (expf ** 2).sum()))
        hg = orc.bf16_to_f32(model.debug_read(0, (B, spec.hidden_size)).cpu().view(torch.int16).numpy().view(np.uint16))
        he = orc.bf16_to_f32(h_exp)
        relh = float(np.sqrt(((hg - he) ** 2).sum() / (he ** 2).sum()))
        assert np.isfinite(got).all()
        assert relh <= 2e-2, f"step {p}: residual stream differs from the CPU oracle (rel L2 {relh})"
        assert rel <= 2e-2, f"step {p}: logits differ from the CPU oracle (rel L2 {rel})"
        worst = max(worst, rel)
        if calibrate_headroom is not None:
            for li in range(spec.num_layers):      # pinned at step 0 from that step's K / V; never moved afterwards
                ks, vs, cal = model.kv_scales(li)
                cl = spec.kv_target(li) if spec.kv_target(li) is not None else li
                assert cal and abs(ks / float(ref.kvs[cl].k_scale) - 1) < 0.03 and abs(vs / float(ref.kvs[cl].v_scale) - 1) < 0.03, (p, li, ks, vs)
        # greedy id: the oracle's argmax, or a logit within one bf16 step of the oracle's maximum
        o = out.cpu().numpy()
        for b in range(B):
            assert expf[b, o[b]] >= expf[b].max() - abs(expf[b].max()) * 2 ** -6
        ids = o.astype(np.int32)
    model.close()
    return worst


def test_gemma4_decode_matches_oracle_fp8_kv_window():
    """3 layers (sliding, sliding, full), window 4 < steps so the sliding window is exercised, FP8 KV, soft-cap, bf16 embedding."""
    from vllm_rust_amd.gemma4 import Gemma4Spec
    spec = Gemma4Spec(256, 3, 4, 512, 1000, head_dim_sliding=64, head_dim_full=128, kv_heads_sliding=2, kv_heads_full=1,
                      sliding_window=4, full_every=3, max_position=64, kv_cache_dtype=1, num_blocks=8, max_batch=2, max_seq_len=32)
    _run(spec, steps=7, B=2)


def test_gemma4_decode_k_eq_v_q3k_embedding_bf16_kv():
    """Full layers reuse K as V; token embedding / tied lm_head is a Q3_K table; bf16 KV; all five I-quant types in use."""
    from vllm_rust_amd.gemma4 import Gemma4Spec
    spec = Gemma4Spec(256, 2, 4, 512, 512, head_dim_sliding=64, head_dim_full=128, kv_heads_sliding=2, kv_heads_full=1,
                      sliding_window=16, full_every=2, k_eq_v_on_full=True, max_position=64, kv_cache_dtype=0, num_blocks=4,
                      max_batch=1, max_seq_len=32, embed_ggml_type=11, final_logit_softcap=0.0)
    _run(spec, steps=4, B=1, slot_types={0: 17, 1: 22, 2: 18, 3: 21, 4: 23, 5: 18, 6: 21})


def test_gemma4_kv_shared_layers_double_wide_mlp():
    """6 layers (s, s, f, s, s, f), the last three KV-shared: layers 3 and 4 read layer 1's cache, layer 5 reads layer 2's, none
    of them has k/v projections, and their MLPs are twice as wide (num_kv_shared_layers + use_double_wide_mlp)."""
    from vllm_rust_amd.gemma4 import Gemma4Spec
    spec = Gemma4Spec(256, 6, 4, 512, 1000, head_dim_sliding=64, head_dim_full=128, kv_heads_sliding=2, kv_heads_full=1,
                      sliding_window=4, full_every=3, num_kv_shared_layers=3, use_double_wide_mlp=True, max_position=64,
                      kv_cache_dtype=1, num_blocks=8, max_batch=2, max_seq_len=32)
    assert [spec.kv_target(i) for i in range(6)] == [None, None, None, 1, 1, 2]
    assert sorted(spec.linear_shapes(4)) == [0, 3, 4, 5, 6] and spec.linear_shapes(4)[6] == (1024, 256)
    _run(spec, steps=7, B=2)


@pytest.mark.parametrize("ple_type", [0, 14])
def test_gemma4_per_layer_inputs(ple_type):
    """Per-layer inputs (E2B / E4B variants): embed_tokens_per_layer (bf16 or Q6_K) + per_layer_model_projection + norm once per
    step, and per layer the gelu_tanh gate, projection and post norm before the layer scalar; with a KV-shared last layer."""
    from vllm_rust_amd.gemma4 import Gemma4Spec
    spec = Gemma4Spec(256, 4, 4, 512, 1000, head_dim_sliding=64, head_dim_full=128, kv_heads_sliding=2, kv_heads_full=1,
                      sliding_window=4, full_every=3, num_kv_shared_layers=1, hidden_size_per_layer_input=256,
                      vocab_size_per_layer_input=1000, ple_embed_ggml_type=ple_type, max_position=64, kv_cache_dtype=0, num_blocks=8,
                      max_batch=2, max_seq_len=32)
    _run(spec, steps=5, B=2)


def test_gemma4_fp8_kv_first_write_calibration():
    """FP8 KV scales start unpinned; the first decode step's K / V pin them at absmax * 1.5 / 448 per layer and tensor
    (KVScales::calibrate_if_needed), later steps reuse them; logits and residual stream against the oracle doing the same.
    An uncalibrated cache refuses a prompt chunk of more than one 16-row pass."""
    import torch
    from vllm_rust_amd._lib import VmxError
    from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM
    spec = Gemma4Spec(256, 4, 4, 512, 1000, head_dim_sliding=64, head_dim_full=128, kv_heads_sliding=2, kv_heads_full=1,
                      sliding_window=4, full_every=3, num_kv_shared_layers=1, max_position=64, kv_cache_dtype=1, num_blocks=8,
                      max_batch=2, max_seq_len=32)
    _run(spec, steps=6, B=2, calibrate_headroom=1.5)
    model = QuantizedGemma4ForCausalLM(spec)
    try:
        model.load_synthetic(seed=1)
        model.kv_scales_reset()
        bt = torch.arange(2, dtype=torch.int32, device="cuda")
        with pytest.raises(VmxError, match="not calibrated"):
            model.prefill(torch.arange(17, dtype=torch.int32, device="cuda"), 0, bt)
        model.prefill(torch.arange(16, dtype=torch.int32, device="cuda"), 0, bt)        # one pass: calibrates
        assert model.kv_scales(0)[2] and model.kv_scales(0)[0] != 1.0
        model.prefill(torch.arange(1, dtype=torch.int32, device="cuda") + 16, 16, bt)
    finally:
        model.close()


def test_gemma4_kv_shared_flags_must_be_a_suffix_with_a_target():
    import ctypes as C
    from vllm_rust_amd._lib import Gemma4Config, lib
    cfg = Gemma4Config(256, 3, 4, 512, 100, 1e-6, 64, 128, 2, 1, 4, 64, 32, 64, 0, 16, 4, 1, 32, 0.0, 0)
    h = C.c_void_p()
    assert lib.vmx_gemma4_create(C.byref(cfg), (C.c_int * 3)(1, 1 | 4, 1), C.byref(h)) == -1     # shared layer in the middle
    assert b"KV-shared" in lib.vmx_last_error()
    assert lib.vmx_gemma4_create(C.byref(cfg), (C.c_int * 3)(1, 1, 0 | 4), C.byref(h)) == -1     # full-attention, no earlier full layer
    assert b"no earlier" in lib.vmx_last_error()


@pytest.mark.parametrize("kv_dtype,k_eq_v,shared", [(0, False, 0), (1, False, 0), (0, True, 0), (0, False, 1)])
def test_gemma4_native_prefill_matches_token_by_token_oracle(kv_dtype, k_eq_v, shared):
    """vmx_gemma4_prefill on a 37-token prompt (passes of 16, 16 and 5 rows; sliding window 8 < prompt, so windowed causal
    attention inside and across passes) followed by decode steps: last-token logits and every decoded step against the CPU
    oracle, which walks the same tokens one at a time.  Also a second prefill call that continues at seqlen_offset > 0."""
    import torch
    import vllm_rust_amd as vmx
    from oracle import oracle as orc
    from tests.model_oracle import OracleGemma4
    from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM
    vmx.init(0)
    spec = Gemma4Spec(256, 3 + shared, 4, 512, 1000, head_dim_sliding=64, head_dim_full=128, kv_heads_sliding=2, kv_heads_full=1,
                      sliding_window=8, full_every=3, k_eq_v_on_full=k_eq_v, num_kv_shared_layers=shared, max_position=128,
                      kv_cache_dtype=kv_dtype, num_blocks=8, max_batch=1, max_seq_len=96)
    model = QuantizedGemma4ForCausalLM(spec)
    try:
        ck = model.load_synthetic(seed=5, keep_host_copy=True)
        # Gemma attends with scale 1 on RMS-normed q and k: with norm weights near 1 the logits have a standard deviation of
        # sqrt(head_dim) = 8..11 and the softmax over a few dozen tokens is close to one-hot, so a one-ulp difference between
        # two near-tied tokens moves the whole output (measured: 1.2 % relative error at 5 tokens, 13 % at 37, with the same
        # kernels).  Norm weights of head_dim^-1/4 give logits of order 1, as trained checkpoints have; the error is then the
        # same ~2 % (q8_1 noise of a 256-wide model) at every prompt length.
        from vllm_rust_amd.gemma4 import T_K_NORM, T_Q_NORM
        gq = torch.Generator(device="cuda").manual_seed(9)
        for li in range(spec.num_layers):
            hd = spec.head_dim_full if (li + 1) % 3 == 0 else spec.head_dim_sliding
            for which, name in ((T_Q_NORM, "q_norm"), (T_K_NORM, "k_norm")):
                w = ((1.0 + 0.1 * torch.randn(hd, device="cuda", generator=gq)) * (1.0 / (hd ** 0.25))).to(torch.bfloat16)
                model.load_tensor(which, li, w)
                ck[f"{li}.{name}"] = w.cpu()
        ref = OracleGemma4(spec, ck, model.cos_sin_sliding, model.cos_sin_full)
        bs = spec.block_size
        bt = np.asarray([[5, 2, 7, 0, 3, 1]], np.int32)
        P1, P2 = 37, 9
        prompt = np.random.default_rng(1).integers(0, spec.vocab_size, P1 + P2).astype(np.int32)
        bar = 3.5e-2 if kv_dtype == 0 else 5e-2   # measured 0.022-0.030 (bf16 KV), 0.028-0.033 (FP8 KV)

        def slot(p):
            return np.asarray([bt[0, p // bs] * bs + p % bs], np.int32)

        def oracle_walk(p0, p1):
            lg = None
            for p in range(p0, p1):
                lg, _, _ = ref.decode(prompt[p:p + 1], np.asarray([p], np.int32), slot(p), bt, np.asarray([p + 1], np.int32))
            return orc.bf16_to_f32(lg)[0]

        def rel(a, b):
            return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))

        bt_t = torch.from_numpy(bt).cuda()
        lg = torch.empty(spec.vocab_size, dtype=torch.bfloat16, device="cuda")
        nxt = model.prefill(torch.from_numpy(prompt[:P1]).cuda(), 0, bt_t[0], logits_last=lg)
        torch.cuda.synchronize()
        exp = oracle_walk(0, P1)
        got = orc.bf16_to_f32(lg.cpu().view(torch.int16).numpy().view(np.uint16))
        r1 = rel(got, exp)
        assert np.isfinite(got).all() and r1 <= bar, r1
        assert exp[int(nxt.item())] >= exp.max() - abs(exp.max()) * 2 ** -6
        # continue the same sequence with a second chunk at an offset, then two decode steps
        model.prefill(torch.from_numpy(prompt[P1:]).cuda(), P1, bt_t[0], logits_last=lg)
        torch.cuda.synchronize()
        exp = oracle_walk(P1, P1 + P2)
        r2 = rel(orc.bf16_to_f32(lg.cpu().view(torch.int16).numpy().view(np.uint16)), exp)
        assert r2 <= bar, r2
        ids = np.asarray([int(np.argmax(exp))], np.int32)
        for step in range(2):
            p = P1 + P2 + step
            lgb = torch.empty((1, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
            out = model.forward_decode_batch(*(torch.from_numpy(a).cuda() for a in (ids, np.asarray([p], np.int32), slot(p), bt,
                                                                                   np.asarray([p + 1], np.int32))), logits=lgb)
            torch.cuda.synchronize()
            e, _, _ = ref.decode(ids, np.asarray([p], np.int32), slot(p), bt, np.asarray([p + 1], np.int32))
            r3 = rel(orc.bf16_to_f32(lgb.cpu().view(torch.int16).numpy().view(np.uint16)), orc.bf16_to_f32(e))
            assert r3 <= bar, (step, r3)
            ids = out.cpu().numpy().astype(np.int32)
        print(f"prefill rel-L2 {r1:.4f} / {r2:.4f}")
    finally:
        model.close()


@pytest.mark.parametrize("kv_dtype", [0, 1])
def test_gemma4_prefill_large_pass_dense_linears(kv_dtype):
    """max_prefill_tokens = 64: a 37-token chunk and a 20-token continuation each run as ONE pass whose linears take the reference's
    large-M form (weight dequantised to f32, unquantised activations) instead of the q8_1 MMVQ; with FP8 KV the first write
    calibrates the scales on the whole chunk, as the reference's cache engine does.  Oracle: the same chunk as one batch of rows
    of one sequence (all K/V rows written, then each row attends up to its own position), linears dense beyond 16 rows."""
    import torch
    import vllm_rust_amd as vmx
    from oracle import oracle as orc
    from tests.model_oracle import OracleGemma4
    from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM, T_K_NORM, T_Q_NORM
    vmx.init(0)
    spec = Gemma4Spec(256, 4, 4, 512, 1000, head_dim_sliding=64, head_dim_full=128, kv_heads_sliding=2, kv_heads_full=1,
                      sliding_window=8, full_every=3, num_kv_shared_layers=1, max_position=128, kv_cache_dtype=kv_dtype, num_blocks=8,
                      max_batch=1, max_seq_len=96, max_prefill_tokens=64)
    model = QuantizedGemma4ForCausalLM(spec)
    try:
        ck = model.load_synthetic(seed=5, keep_host_copy=True)
        gq = torch.Generator(device="cuda").manual_seed(9)      # q/k norm weights of head_dim^-1/4: logits of order 1 (see above)
        for li in range(spec.num_layers):
            hd = spec.layer_geom(li)[0]
            for which, name in ((T_Q_NORM, "q_norm"), (T_K_NORM, "k_norm")):
                w = ((1.0 + 0.1 * torch.randn(hd, device="cuda", generator=gq)) * (1.0 / (hd ** 0.25))).to(torch.bfloat16)
                model.load_tensor(which, li, w)
                ck[f"{li}.{name}"] = w.cpu()
        ref = OracleGemma4(spec, ck, model.cos_sin_sliding, model.cos_sin_full)
        if kv_dtype == 1:
            model.kv_scales_reset(1.5)
            ref.reset_kv_scales(1.5)
        bs = spec.block_size
        bt = np.asarray([[5, 2, 7, 0, 3, 1]], np.int32)
        P1, P2 = 37, 20
        prompt = np.random.default_rng(1).integers(0, spec.vocab_size, P1 + P2).astype(np.int32)

        def chunk(p0, p1):
            pos = np.arange(p0, p1, dtype=np.int32)
            slots = (bt[0, pos // bs] * bs + pos % bs).astype(np.int32)
            lg, _, _ = ref.decode(prompt[p0:p1], pos, slots, np.repeat(bt, p1 - p0, axis=0), pos + 1)
            return orc.bf16_to_f32(lg)[-1]

        def rel(a, b):
            return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))

        bt_t = torch.from_numpy(bt).cuda()
        lg = torch.empty(spec.vocab_size, dtype=torch.bfloat16, device="cuda")
        nxt = model.prefill(torch.from_numpy(prompt[:P1]).cuda(), 0, bt_t[0], logits_last=lg)
        torch.cuda.synchronize()
        exp = chunk(0, P1)
        r1 = rel(orc.bf16_to_f32(lg.cpu().view(torch.int16).numpy().view(np.uint16)), exp)
        bar = 2e-2 if kv_dtype == 0 else 4e-2     # measured 0.012 / 0.015 (bf16 KV); FP8 KV adds its e4m3 rounding of every K / V
        assert r1 <= bar, r1
        assert exp[int(nxt.item())] >= exp.max() - abs(exp.max()) * 2 ** -6
        if kv_dtype == 1:      # pinned from the whole 37-row chunk on both sides
            for li in range(spec.num_layers):
                ks, vs, cal = model.kv_scales(li)
                cl = spec.kv_target(li) if spec.kv_target(li) is not None else li
                assert cal and abs(ks / float(ref.kvs[cl].k_scale) - 1) < 0.02 and abs(vs / float(ref.kvs[cl].v_scale) - 1) < 0.02
        model.prefill(torch.from_numpy(prompt[P1:]).cuda(), P1, bt_t[0], logits_last=lg)
        torch.cuda.synchronize()
        exp = chunk(P1, P1 + P2)
        r2 = rel(orc.bf16_to_f32(lg.cpu().view(torch.int16).numpy().view(np.uint16)), exp)
        assert r2 <= bar, r2
        print(f"dense-linear prefill rel-L2 {r1:.4f} / {r2:.4f}")
    finally:
        model.close()


def test_gemma4_rejects_bad_config():
    import ctypes as C
    from vllm_rust_amd._lib import Gemma4Config, lib
    cfg = Gemma4Config(250, 1, 4, 512, 100, 1e-6, 64, 128, 2, 1, 4, 64, 32, 64, 0, 16, 4, 1, 32, 0.0, 0)
    h = C.c_void_p()
    flags = (C.c_int * 1)(1)
    assert lib.vmx_gemma4_create(C.byref(cfg), flags, C.byref(h)) == -1
    assert b"hidden_size" in lib.vmx_last_error()


@pytest.mark.parametrize("ple", [False, True])
def test_gemma4_load_from_gguf_file_equals_direct_load(tmp_path, ple):
    """Write the synthetic checkpoint as a GGUF v3 file (llama.cpp tensor names, on-disk dims reversed, F32 norms, Q3_K
    token_embd), load it through the library's container parser and compare a decode step with the directly loaded model.
    ple: the per-layer-input variant (E2B / E4B) with its tensors under the names to_llama_cpp_prefixes gives them
    (quantization/gguf/mod.rs:1421-1424,1450-1453): blk.N.inp_gate / proj / post_norm, per_layer_token_embd (Q6_K),
    per_layer_model_proj, per_layer_proj_norm."""
    import torch
    import vllm_rust_amd as vmx
    from tests.test_gguf_host import build_gguf
    from oracle import gguf_oracle as go
    from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM
    vmx.init(0)
    spec = Gemma4Spec(256, 2, 4, 512, 512, head_dim_sliding=64, head_dim_full=128, kv_heads_sliding=2, kv_heads_full=1,
                      sliding_window=16, full_every=2, max_position=64, kv_cache_dtype=1, num_blocks=4, max_batch=1, max_seq_len=32,
                      embed_ggml_type=11, **(dict(hidden_size_per_layer_input=256, vocab_size_per_layer_input=512,
                                                  ple_embed_ggml_type=14) if ple else {}))
    a = QuantizedGemma4ForCausalLM(spec)
    ck = a.load_synthetic(seed=5, keep_host_copy=True)
    # checkpoint -> GGUF tensors
    names = {0: "attn_q", 1: "attn_k", 2: "attn_v", 3: "attn_output", 4: "ffn_gate", 5: "ffn_up", 6: "ffn_down", 7: "inp_gate", 8: "proj"}
    norms = {"input_ln": "attn_norm", "post_attn_ln": "post_attention_norm", "pre_ffn_ln": "ffn_norm", "post_ffn_ln": "post_ffw_norm",
             "q_norm": "attn_q_norm", "k_norm": "attn_k_norm", "scalar": "layer_output_scale"}
    tensors, payload = [], {}
    tensors.append(("token_embd.weight", [spec.hidden_size, spec.vocab_size], 11)); payload["token_embd.weight"] = ck["embed"].numpy().tobytes()
    tensors.append(("output_norm.weight", [spec.hidden_size], 0)); payload["output_norm.weight"] = ck["final_norm"].float().numpy().tobytes()
    for li in range(spec.num_layers):
        for k, g in norms.items():
            w = ck[f"{li}.{k}"].float().numpy()
            tensors.append((f"blk.{li}.{g}.weight", [w.size], 0)); payload[f"blk.{li}.{g}.weight"] = w.tobytes()
        for slot, (K, N) in spec.linear_shapes(li).items():
            t, raw = ck[f"{li}.{slot}.iq"]
            tensors.append((f"blk.{li}.{names[slot]}.weight", [K, N], t)); payload[f"blk.{li}.{names[slot]}.weight"] = raw.numpy().tobytes()
        if ple:
            w = ck[f"{li}.post_ple_norm"].float().numpy()
            tensors.append((f"blk.{li}.post_norm.weight", [w.size], 0)); payload[f"blk.{li}.post_norm.weight"] = w.tobytes()
    extra = 0
    if ple:
        LP = spec.hidden_size_per_layer_input * spec.num_layers
        t, raw = ck["0.9.iq"]
        tensors.append(("per_layer_model_proj.weight", [spec.hidden_size, LP], t)); payload["per_layer_model_proj.weight"] = raw.numpy().tobytes()
        w = ck["ple_proj_norm"].float().numpy()
        tensors.append(("per_layer_proj_norm.weight", [w.size], 0)); payload["per_layer_proj_norm.weight"] = w.tobytes()
        tensors.append(("per_layer_token_embd.weight", [LP, spec.vocab_size_per_layer_input], 14))
        payload["per_layer_token_embd.weight"] = ck["ple_embed"].numpy().tobytes()
        extra = spec.num_layers + 3
    data, blobs = build_gguf(3, [("general.architecture", 8, "gemma4")], tensors)
    hdr = go.parse(data)
    buf = bytearray(data)
    for name, shape, tid, off, nbytes in hdr["tensors"]:
        assert len(payload[name]) == nbytes, name
        buf[hdr["tensor_data_offset"] + off: hdr["tensor_data_offset"] + off + nbytes] = payload[name]
    path = tmp_path / "tiny-gemma4.gguf"
    path.write_bytes(bytes(buf))
    b = QuantizedGemma4ForCausalLM(spec)
    assert b.load_gguf(path) == 2 + spec.num_layers * 7 + sum(len(spec.linear_shapes(li)) for li in range(spec.num_layers)) + extra
    ids = torch.tensor([17], dtype=torch.int32, device="cuda"); pos = torch.zeros(1, dtype=torch.int32, device="cuda")
    slots = torch.zeros(1, dtype=torch.int32, device="cuda"); bt = torch.zeros((1, 4), dtype=torch.int32, device="cuda")
    sl = torch.ones(1, dtype=torch.int32, device="cuda")
    la = torch.empty((1, spec.vocab_size), dtype=torch.bfloat16, device="cuda"); lb = torch.empty_like(la)
    a.forward_decode_batch(ids, pos, slots, bt, sl, logits=la)
    b.forward_decode_batch(ids, pos, slots, bt, sl, logits=lb)
    torch.cuda.synchronize()
    assert torch.equal(la.view(torch.int16), lb.view(torch.int16))
    a.close(); b.close()

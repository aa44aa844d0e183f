"""Whole-token executors at the TRUE layer geometries of the benchmark models (2-layer slices, vocabulary cut to 8192)
against the composed CPU oracle: the kernel instantiations a Qwen3-8B / Qwen3-0.6B / Mixtral-8x7B / Gemma-4-12B decode step
actually launches (128-wide heads with 4 or 2 query heads per KV head, hidden 4096 / 3840, the gated-activation epilogue at
4096 -> 12288, 256- and 512-wide Gemma heads, Q3_K lm_head), at sequence lengths where the new token crosses a 16-token
tile, a cache block, the 1024-token switch between the one-launch and the three-launch attention, and the sliding window.

The KV caches are pre-filled with the same random rows on both sides (vmx_*_debug_kv), so a sequence can start at any length
without replaying a thousand oracle steps.  Parity is unpinned in the sense of the operator tests below it (the reference holds
no model-level goldens); bars are the decoder tests' relative-L2 2e-2 on the residual stream and the logits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VOCAB = 8192


@pytest.fixture(scope="module")
def gpu():
    import torch
    import vllm_rust_amd as vmx
    assert torch.cuda.is_available()
    vmx.init(0)
    return vmx, torch


def _u16(torch, t):
    return t.cpu().view(torch.int16).numpy().view(np.uint16)


def _rel(a, b):
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


def _act_f32(spec, u16):
    """16-bit activation bits of the model's dtype (bf16, or fp16 for spec.act_dtype == "f16") as f32."""
    from oracle import oracle as orc
    return u16.view(np.float16).astype(np.float32) if getattr(spec, "act_dtype", "bf16") == "f16" else orc.bf16_to_f32(u16)


def _fill_kv(torch, model, ref, num_layers, shapes, kv_dtype, seed, k_std=1.0):
    """Same random cache rows in the model and in the oracle; shapes[li] = (blocks, block_size, kv_heads, head_dim).
    k_std: standard deviation of the bf16 K rows (Gemma attends with scale 1.0 over 256 / 512 dims: unit-variance keys would
    make every softmax row one-hot and the comparison a coin flip between near-tied tokens)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    for li in range(num_layers):
        for which, cache in ((0, ref.k_cache), (1, ref.v_cache)):
            if kv_dtype == 0:
                t = (torch.randn(shapes[li], device="cuda", generator=g) * (k_std if which == 0 else 1.0)).to(
                    getattr(model, "act_torch_dtype", torch.bfloat16))
                host = _u16(torch, t)
            else:   # finite e4m3 bytes of moderate magnitude (exponent field <= 5: |x| < 0.5, so that softmax rows stay spread), random sign
                mag = torch.randint(0, 0x30, shapes[li], device="cuda", generator=g, dtype=torch.int32)
                sgn = torch.randint(0, 2, shapes[li], device="cuda", generator=g, dtype=torch.int32) << 7
                t = (mag | sgn).to(torch.uint8)
                host = t.cpu().numpy()
            model.debug_kv(li, which, t.contiguous(), True)
            cache[li][...] = host


def _decode_and_compare(torch, model, ref, spec, start_lens, steps, first_ids, bar=2e-2, fused_switch=None):
    """Sequences b start with start_lens[b] tokens already in the (pre-filled) cache; `steps` decode steps."""
    from oracle import oracle as orc
    from vllm_rust_amd._lib import lib
    B = len(start_lens)
    bs = spec.block_size
    mb = (max(start_lens) + steps + bs - 1) // bs + 1
    assert B * mb <= spec.num_blocks
    rng = np.random.default_rng(5)
    bt = rng.permutation(spec.num_blocks)[:B * mb].reshape(B, mb).astype(np.int32)   # scattered, disjoint block tables
    ids = np.asarray(first_ids, np.int32)
    pos = np.asarray(start_lens, np.int32)
    worst = 0.0
    for step in range(steps):
        slots = (bt[np.arange(B), pos // bs] * bs + pos % bs).astype(np.int32)
        sl = pos + 1
        msl = int(sl.max())
        if fused_switch is not None:
            lib.vmx_debug_set_fused_attention(int(fused_switch(step)))
        logits = torch.empty((B, spec.vocab_size), dtype=getattr(model, "act_torch_dtype", torch.bfloat16), device="cuda")
        out = model.forward_decode_batch(*(torch.from_numpy(a).cuda() for a in (ids, pos, slots, bt, sl)), max_seq_len=msl, logits=logits)
        torch.cuda.synchronize()
        exp, exp_h, _ = ref.decode(ids, pos, slots, bt, sl)
        got = _act_f32(spec, _u16(torch, logits)); expf = _act_f32(spec, exp)
        got_h = _act_f32(spec, _u16(torch, model.debug_read(0, (B, spec.hidden_size)))); exp_hf = _act_f32(spec, exp_h)
        assert np.isfinite(got).all()
        rh, rl = _rel(got_h, exp_hf), _rel(got, expf)
        print(f"step {step} seq_lens {sl.tolist()} rel-L2 h {rh:.4f} logits {rl:.4f}")
        assert rh <= bar and rl <= bar, (step, sl.tolist(), rh, rl)
        worst = max(worst, rl)
        o = out.cpu().numpy()
        for b in range(B):   # greedy id: the oracle's argmax, or a logit within one bf16 step of the oracle's maximum
            assert expf[b, o[b]] >= expf[b].max() - abs(expf[b].max()) * 2 ** -6, (step, b)
        ids = o.astype(np.int32)
        pos = pos + 1
    lib.vmx_debug_set_fused_attention(1)
    return worst


def _qwen3_case(gpu, spec, loader, start_lens, steps, first_ids, fused_switch=None, bar=2e-2, **loader_kw):
    from tests.model_oracle import OracleQwen3
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, rope_cos_sin_cache
    vmx, torch = gpu
    model = QuantizedQwen3ForCausalLM(spec)
    try:
        ck = getattr(model, loader)(seed=13, keep_host_copy=True, **loader_kw)
        ref = OracleQwen3(spec, ck, rope_cos_sin_cache(spec.head_dim, spec.max_position, spec.rope_theta, spec.act_dtype))
        shape = (spec.num_blocks, spec.block_size, spec.num_kv_heads, spec.head_dim)
        _fill_kv(torch, model, ref, spec.num_layers, [shape] * spec.num_layers, spec.kv_cache_dtype, seed=99)
        return _decode_and_compare(torch, model, ref, spec, start_lens, steps, first_ids, bar, fused_switch)
    finally:
        model.close()


def _spec_8b(**kw):
    from vllm_rust_amd.qwen3 import Qwen3Spec
    base = dict(hidden_size=4096, num_layers=2, num_heads=32, num_kv_heads=8, head_dim=128, intermediate_size=12288, vocab_size=VOCAB,
                max_position=2048, num_blocks=320, max_batch=4, max_seq_len=2048)
    base.update(kw)
    return Qwen3Spec(**base)


def test_qwen3_8b_awq_tile_block_and_attention_switch(gpu):
    """Qwen3-8B AWQ layers, B = 4: new tokens at positions 14..17 (first tile boundary = first block boundary), 31..34,
    250.. and 1022..1025: seq_len 1023 and 1024 run the one-launch attention <128, 4>, 1025 and 1026 the split-KV path."""
    _qwen3_case(gpu, _spec_8b(), "load_synthetic", start_lens=[14, 31, 250, 1022], steps=4, first_ids=[3, 999, 4097, 8191])


def test_qwen3_8b_awq_three_launch_attention_forced(gpu):
    """The same geometry with the one-launch attention switched off: qkv_post_kernel + partition + merge at 128-wide heads,
    short and long contexts in one batch; B = 1 afterwards (the batch-1 kernels of the headline number)."""
    _qwen3_case(gpu, _spec_8b(), "load_synthetic", start_lens=[0, 15, 700, 1023], steps=2, first_ids=[1, 2, 3, 4],
                fused_switch=lambda step: False)
    _qwen3_case(gpu, _spec_8b(max_batch=1), "load_synthetic", start_lens=[300], steps=3, first_ids=[77])


def test_qwen3_8b_awq_five_launch_layers(gpu):
    """One to four decode rows take the five-launch layer (RMSNorm in the prologue of q|k|v and gate|up, residual add in the
    epilogue of o_proj and down_proj); three layers, so that the first (norm kernel after the embedding), a middle and the
    last layer (final norm kernel) are all covered; B = 2 with one row in each attention regime."""
    _qwen3_case(gpu, _spec_8b(max_batch=1, num_layers=3), "load_synthetic", start_lens=[41], steps=3, first_ids=[5])
    _qwen3_case(gpu, _spec_8b(max_batch=2, num_layers=3), "load_synthetic", start_lens=[7, 1023], steps=3, first_ids=[11, 4242])
    # three and four rows: the same layer with wider row staging (rows 2 and 3 of the matrix tile, four-row residual epilogue)
    _qwen3_case(gpu, _spec_8b(max_batch=3, num_layers=3), "load_synthetic", start_lens=[0, 300, 1020], steps=3, first_ids=[9, 8, 7])
    _qwen3_case(gpu, _spec_8b(max_batch=4, num_layers=3), "load_synthetic", start_lens=[15, 16, 511, 700], steps=3, first_ids=[1, 22, 333, 4444])


def test_qwen3_8b_awq_fp8_kv_batch16(gpu):
    """16 concurrent sequences (looped activation staging in the int4 GEMM, matrix-core lm_head) over an FP8 cache."""
    spec = _spec_8b(max_batch=16, kv_cache_dtype=1, num_blocks=16 * 12)
    from vllm_rust_amd.qwen3 import T_KV_SCALES
    from tests.model_oracle import OracleQwen3
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, rope_cos_sin_cache
    vmx, torch = gpu
    model = QuantizedQwen3ForCausalLM(spec)
    try:
        ck = model.load_synthetic(seed=13, keep_host_copy=True)
        ref = OracleQwen3(spec, ck, rope_cos_sin_cache(spec.head_dim, spec.max_position, spec.rope_theta))
        scales = torch.tensor([0.5, 0.75], dtype=torch.float32, device="cuda")
        for li in range(spec.num_layers):
            model.load_tensor(T_KV_SCALES, li, scales)
        ref.kv_scales = (0.5, 0.75)
        shape = (spec.num_blocks, spec.block_size, spec.num_kv_heads, spec.head_dim)
        _fill_kv(torch, model, ref, spec.num_layers, [shape] * spec.num_layers, 1, seed=7)
        starts = [0, 1, 15, 16, 17, 31, 32, 47, 63, 64, 100, 111, 127, 128, 150, 160]
        _decode_and_compare(torch, model, ref, spec, starts, 2, list(range(100, 116)), bar=3e-2)
    finally:
        model.close()


def test_qwen3_8b_exl3_4bpw(gpu):
    """Qwen3-8B geometry with every projection an EXL3 4.0 bpw linear (multi-tensor q|k|v and gate|up launches, consumer-side
    output transform at hidden 4096 / 12288)."""
    _qwen3_case(gpu, _spec_8b(max_batch=2), "load_synthetic_exl3", start_lens=[15, 1023], steps=2, first_ids=[5, 6], bar=3e-2, bits=4)


def test_qwen3_8b_exl3_decode_rows(gpu):
    """One and four decode rows of the EXL3 layer at short contexts: residual kernel with block sums + RMSNorm in the
    prologue of q|k|v and gate|up, output transform of q|k|v inside the spread attention, partitions merged in the prologue of
    o_proj (several (row, head) items per wave from two rows on); three layers at one row, so that the first (norm kernel
    after the embedding), a middle and the last layer (final norm kernel) are covered."""
    _qwen3_case(gpu, _spec_8b(max_batch=1, num_layers=3), "load_synthetic_exl3", start_lens=[41], steps=2, first_ids=[5], bar=3e-2, bits=4)
    _qwen3_case(gpu, _spec_8b(max_batch=4), "load_synthetic_exl3", start_lens=[0, 16, 511, 1020], steps=1, first_ids=[1, 22, 333, 4444],
                bar=3e-2, bits=4)


def test_qwen3_8b_exl3_fp16_activations(gpu):
    """The same EXL3 geometry run in fp16 end to end (what the reference does with an EXL3 checkpoint: fp16 embedding, norms,
    lm_head, activations and cache): the fused one-row layer at a short context (three layers), four rows across the attention
    regimes (spread attention merged by o_proj, the many-partition form past 512 tokens), and two rows at 15 / 1023 tokens.
    The oracle is the composed fp16 model of tests/model_oracle.py (parity unpinned: the reference's fp16 norms / rotary /
    activation run in its third-party tensor library)."""
    _qwen3_case(gpu, _spec_8b(max_batch=1, num_layers=3, act_dtype="f16"), "load_synthetic_exl3", start_lens=[41], steps=2, first_ids=[5],
                bar=2e-2, bits=4)
    _qwen3_case(gpu, _spec_8b(max_batch=4, act_dtype="f16"), "load_synthetic_exl3", start_lens=[0, 16, 511, 1020], steps=1,
                first_ids=[1, 22, 333, 4444], bar=2e-2, bits=4)
    _qwen3_case(gpu, _spec_8b(max_batch=2, act_dtype="f16"), "load_synthetic_exl3", start_lens=[15, 1023], steps=2, first_ids=[5, 6],
                bar=2e-2, bits=4)


def test_qwen3_8b_exl3_fp16_unfused_rows(gpu):
    """fp16 activations on the launch sequences without the decode-row fusions: eight rows (per-projection EXL3 launches with
    their own Hadamard kernels, q/k/v post-processing, one-launch or partitioned attention + merge, SiLU kernel) and the
    three-launch attention forced at two rows."""
    _qwen3_case(gpu, _spec_8b(max_batch=8, num_blocks=8 * 40, act_dtype="f16"), "load_synthetic_exl3",
                start_lens=[0, 1, 15, 16, 100, 300, 511, 600], steps=1, first_ids=list(range(50, 58)), bar=2e-2, bits=4)
    _qwen3_case(gpu, _spec_8b(max_batch=2, act_dtype="f16"), "load_synthetic_exl3", start_lens=[33, 700], steps=1, first_ids=[5, 6],
                fused_switch=lambda step: False, bar=2e-2, bits=4)


def test_mixtral_8x7b_moe_layers(gpu):
    """Mixtral-8x7B layers (no q/k norms, 8 GPTQ experts top-2 at 4096 -> 14336 with the grouped gated epilogue)."""
    from vllm_rust_amd.qwen3 import Qwen3Spec
    spec = Qwen3Spec(4096, 2, 32, 8, 128, 14336, VOCAB, rms_norm_eps=1e-5, rope_theta=1e6, qk_norm=False, num_experts=8, experts_per_tok=2,
                     max_position=2048, num_blocks=160, max_batch=2, max_seq_len=2048)
    _qwen3_case(gpu, spec, "load_synthetic_moe", start_lens=[15, 1024], steps=2, first_ids=[11, 12], bar=3e-2)
    # contexts of at most 1024 tokens on one to four rows: norm in the q|k|v prologue, attention partitions merged by o_proj, the
    # layer's residual add inside the expert combine; three layers so that a middle layer starts from the combine's tile sums
    spec3 = Qwen3Spec(4096, 3, 32, 8, 128, 14336, VOCAB, rms_norm_eps=1e-5, rope_theta=1e6, qk_norm=False, num_experts=8, experts_per_tok=2,
                      max_position=2048, num_blocks=200, max_batch=4, max_seq_len=1024)
    _qwen3_case(gpu, spec3, "load_synthetic_moe", start_lens=[300], steps=3, first_ids=[5], bar=3e-2)
    _qwen3_case(gpu, spec3, "load_synthetic_moe", start_lens=[0, 15, 16, 700], steps=2, first_ids=[1, 2, 3, 4], bar=3e-2)
    # eight rows = sixteen (token, slot) pairs: the experts' weights streamed once per 16-row block of their pairs (ring form)
    spec8 = Qwen3Spec(4096, 2, 32, 8, 128, 14336, VOCAB, rms_norm_eps=1e-5, rope_theta=1e6, qk_norm=False, num_experts=8, experts_per_tok=2,
                      max_position=2048, num_blocks=400, max_batch=8, max_seq_len=1024)
    _qwen3_case(gpu, spec8, "load_synthetic_moe", start_lens=[0, 15, 16, 700, 33, 250, 511, 64], steps=2,
                first_ids=[1, 2, 3, 4, 5, 6, 7, 8], bar=3e-2)


def test_qwen3_0_6b_tied(gpu):
    """Qwen3-0.6B layers: hidden 1024, 16 / 8 heads x 128 (two query heads per KV head: <128, 2>), tied lm_head."""
    from vllm_rust_amd.qwen3 import Qwen3Spec
    spec = Qwen3Spec(1024, 2, 16, 8, 128, 3072, VOCAB, tie_word_embeddings=True, max_position=2048, num_blocks=320, max_batch=4,
                     max_seq_len=2048)
    _qwen3_case(gpu, spec, "load_synthetic", start_lens=[0, 15, 255, 1023], steps=3, first_ids=[9, 8, 7, 6])


@pytest.mark.parametrize("B,kv_dtype", [(1, 0), (4, 0), (1, 1), (4, 1), (6, 0), (13, 1)])
def test_gemma4_12b_layers(gpu, B, kv_dtype):
    """Gemma-4-12B layers: hidden 3840, 16 query heads, one sliding layer (8 x 256, window 1024) and one full layer
    (1 x 512, proportional partial rotary), intermediate 15360, the UD-IQ3_XXS type mix, Q3_K token embedding / tied lm_head,
    bf16 and FP8 KV cache.  Sequence lengths on both sides of the sliding window.  Bars: 2e-2 with the bf16 cache; 5e-2 with the
    FP8 cache (the decoder tests' FP8 bar): a new token's K/V row is rounded to 3 mantissa bits on both sides, so a 1-ulp bf16
    difference in front of that rounding flips an e4m3 step (6 %) on a few per cent of the row's elements."""
    from tests.model_oracle import OracleGemma4
    from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM
    vmx, torch = gpu
    spec = Gemma4Spec(3840, 2, 16, 15360, VOCAB, full_every=2, max_position=2048, num_blocks=max(320, 96 * B), max_batch=B, max_seq_len=2048,
                      embed_ggml_type=11, kv_cache_dtype=kv_dtype)
    model = QuantizedGemma4ForCausalLM(spec)
    try:
        ck = model.load_synthetic(seed=21, keep_host_copy=True)
        ref = OracleGemma4(spec, ck, model.cos_sin_sliding, model.cos_sin_full)
        shapes = []
        for li in range(spec.num_layers):
            hd, kvh, _, _, _ = spec.layer_geom(li)
            shapes.append((spec.num_blocks, spec.block_size, kvh, hd))
        _fill_kv(torch, model, ref, spec.num_layers, shapes, spec.kv_cache_dtype, seed=3, k_std=0.08)
        # 6 and 13 rows: the matrix-core MMVQ and the sliding-layer attention with 4 / 2 workgroups per KV head
        starts = [1023] if B == 1 else [15, 300, 1022, 1100, 0, 63, 512, 1400, 257, 1024, 31, 800, 1200][:B]
        first = [17, 900, 4000, 8000, 5, 77, 1234, 4321, 2, 3000, 650, 7000, 42][:B]
        _decode_and_compare(torch, model, ref, spec, starts, 3 if B <= 4 else 2, first, bar=2e-2 if kv_dtype == 0 else 5e-2)
    finally:
        model.close()


def test_gemma4_matrix_core_geglu_epilogue_same_bits(gpu):
    """gate|up on the matrix cores with the GeGLU + q8_1 epilogue against the same launch followed by the stand-alone GeGLU
    kernel: same K-slice order and the same roundings, so logits and the residual stream must agree bit for bit (7 rows on the
    Gemma-4-12B layer geometry; the oracle comparison of the fused form is test_gemma4_12b_layers at 6 and 13 rows)."""
    from vllm_rust_amd._lib import lib
    from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM
    vmx, torch = gpu
    B = 7
    spec = Gemma4Spec(3840, 2, 16, 15360, VOCAB, full_every=2, max_position=2048, num_blocks=96 * B, max_batch=B, max_seq_len=2048,
                      embed_ggml_type=11, kv_cache_dtype=0)
    model = QuantizedGemma4ForCausalLM(spec)
    try:
        model.load_synthetic(seed=22)
        g = torch.Generator(device="cuda").manual_seed(4)
        for li in range(spec.num_layers):
            hd, kvh, _, _, _ = spec.layer_geom(li)
            for which in (0, 1):
                t = (torch.randn((spec.num_blocks, spec.block_size, kvh, hd), device="cuda", generator=g) * (0.08 if which == 0 else 1.0)).to(torch.bfloat16)
                model.debug_kv(li, which, t.contiguous(), True)
        bs = spec.block_size
        pos = np.asarray([15, 300, 1022, 1100, 0, 63, 512], np.int32)
        mb = (int(pos.max()) + 1 + bs - 1) // bs + 1
        bt = np.random.default_rng(6).permutation(spec.num_blocks)[:B * mb].reshape(B, mb).astype(np.int32)
        ids = np.asarray([17, 900, 4000, 8000, 5, 77, 1234], np.int32)
        slots = (bt[np.arange(B), pos // bs] * bs + pos % bs).astype(np.int32)
        sl = pos + 1
        res = []
        for on in (1, 0):
            assert lib.vmx_debug_gemma4_mma_geglu(on) == 0
            logits = torch.empty((B, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
            model.forward_decode_batch(*(torch.from_numpy(a).cuda() for a in (ids, pos, slots, bt, sl)), max_seq_len=int(sl.max()), logits=logits)
            torch.cuda.synchronize()
            res.append((logits.clone(), model.debug_read(0, (B, spec.hidden_size)).clone()))
        assert torch.equal(res[0][0].view(torch.int16), res[1][0].view(torch.int16))
        assert torch.equal(res[0][1].view(torch.int16), res[1][1].view(torch.int16))
        assert bool(torch.isfinite(res[0][0].float()).all())
    finally:
        lib.vmx_debug_gemma4_mma_geglu(1)
        model.close()

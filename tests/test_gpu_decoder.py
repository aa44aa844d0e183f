"""End-to-end parity of the whole-token decoder against the composed CPU oracle on a tiny
Qwen3-shaped model with random AWQ weights, plus hipGraph capture/replay equivalence."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import vllm_rust_amd as vmx
    assert torch.cuda.is_available()
    vmx.init(0)
    return vmx, torch


def _tiny_spec(**kw):
    from vllm_rust_amd.qwen3 import Qwen3Spec
    base = dict(hidden_size=256, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64, intermediate_size=512,
                vocab_size=1000, max_position=512, num_blocks=32, max_batch=4, max_seq_len=300)
    base.update(kw)
    return Qwen3Spec(**base)


def _u16(torch, t):
    return t.cpu().view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize("tied", [False, True])
def test_decode_steps_match_oracle(gpu, tied):
    from oracle import oracle as orc
    from tests.model_oracle import OracleQwen3
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, rope_cos_sin_cache
    vmx, torch = gpu
    spec = _tiny_spec(tie_word_embeddings=tied)
    model = QuantizedQwen3ForCausalLM(spec)
    ck = model.load_synthetic(seed=3, keep_host_copy=True)
    ref = OracleQwen3(spec, ck, rope_cos_sin_cache(spec.head_dim, spec.max_position, spec.rope_theta))
    rng = np.random.default_rng(0)
    B, MB = 3, 8
    bt = np.stack([rng.permutation(spec.num_blocks)[:MB] for _ in range(B)]).astype(np.int32)
    # make block tables disjoint across sequences
    bt = np.arange(B * MB, dtype=np.int32).reshape(B, MB)[:, ::-1].copy()
    ids = rng.integers(0, spec.vocab_size, size=B).astype(np.int32)
    for step in range(6):
        pos = np.full(B, step, np.int32)
        slots = (bt[np.arange(B), pos // spec.block_size] * spec.block_size + pos % spec.block_size).astype(np.int32)
        seq_lens = pos + 1
        logits = torch.empty((B, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
        out = model.forward_decode_batch(torch.from_numpy(ids).cuda(), torch.from_numpy(pos).cuda(), torch.from_numpy(slots).cuda(),
                                         torch.from_numpy(bt).cuda(), torch.from_numpy(seq_lens).cuda(), logits=logits)
        torch.cuda.synchronize()
        ref_logits, ref_h, _ = ref.decode(ids, pos, slots, bt, seq_lens)
        got = orc.bf16_to_f32(_u16(torch, logits)); exp = orc.bf16_to_f32(ref_logits)
        got_h = orc.bf16_to_f32(_u16(torch, model.debug_read(0, (B, spec.hidden_size)))); exp_h = orc.bf16_to_f32(ref_h)
        # bf16 rounding noise compounds through the layers: compare with a relative-L2 bound
        rel_h = np.sqrt(((got_h - exp_h) ** 2).sum() / (exp_h ** 2).sum())
        rel_l = np.sqrt(((got - exp) ** 2).sum() / (exp ** 2).sum())
        assert rel_h <= 2e-2 and rel_l <= 2e-2, (step, rel_h, rel_l)
        # greedy token: equal whenever the oracle's top-2 margin exceeds the observed logit error
        o = out.cpu().numpy()
        for b in range(B):
            srt = np.sort(exp[b])[::-1]
            if srt[0] - srt[1] > 4 * np.abs(got[b] - exp[b]).max():
                assert o[b] == int(np.argmax(exp[b]))
            assert got[b, o[b]] == got[b].max()
        ids = o.astype(np.int32) % spec.vocab_size
    model.close()


def test_graph_replay_equals_eager(gpu):
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM
    vmx, torch = gpu
    spec = _tiny_spec()
    model = QuantizedQwen3ForCausalLM(spec)
    model.load_synthetic(seed=5)
    B, MB = 2, 8
    bt = torch.arange(B * MB, dtype=torch.int32, device="cuda").reshape(B, MB)
    ids = torch.tensor([5, 77], dtype=torch.int32, device="cuda")
    pos = torch.zeros(B, dtype=torch.int32, device="cuda")
    slots = torch.empty(B, dtype=torch.int32, device="cuda")
    seq_lens = torch.empty(B, dtype=torch.int32, device="cuda")
    out = torch.empty(B, dtype=torch.int32, device="cuda")
    logits = torch.empty((B, spec.vocab_size), dtype=torch.bfloat16, device="cuda")

    def set_step(p):
        pos.fill_(p)
        slots.copy_(bt[:, p // spec.block_size] * spec.block_size + p % spec.block_size)
        seq_lens.fill_(p + 1)

    # eager reference run
    eager = []
    cur = ids.clone()
    for p in range(5):
        set_step(p)
        model.forward_decode_batch(cur, pos, slots, bt, seq_lens, out_ids=out, logits=logits)
        torch.cuda.synchronize()
        eager.append((out.clone(), logits.clone()))
        cur = out.clone()
    # captured run over the same inputs (the KV cache is rewritten with identical values)
    cur = ids.clone()
    g = torch.cuda.CUDAGraph()
    set_step(0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        model.forward_decode_batch(cur, pos, slots, bt, seq_lens, out_ids=out, logits=logits)  # warm-up on the side stream
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        model.forward_decode_batch(cur, pos, slots, bt, seq_lens, out_ids=out, logits=logits)
    for p in range(5):
        set_step(p)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager[p][0]) and torch.equal(logits, eager[p][1]), p
        cur.copy_(out)
    model.close()


def test_decode_preconditions(gpu):
    from vllm_rust_amd._lib import VmxError
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
    vmx, torch = gpu
    with pytest.raises(VmxError, match="max_batch"):
        QuantizedQwen3ForCausalLM(_tiny_spec(max_batch=17))
    with pytest.raises(VmxError, match="multiple of 128"):
        QuantizedQwen3ForCausalLM(_tiny_spec(hidden_size=200))
    m = QuantizedQwen3ForCausalLM(_tiny_spec())
    z = torch.zeros(4, dtype=torch.int32, device="cuda")
    with pytest.raises(VmxError, match="not fully loaded"):
        m.forward_decode_batch(z[:1], z[:1], z[:1], z[:1].reshape(1, 1), z[:1])
    m.close()


def test_dense_gemv_and_embedding(gpu):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(8)
    # rows 5..16 with K % 128 == 0 run on the matrix cores with activations staged 1024 k at a time: several passes, a short
    # last pass, ragged last 16-row tile of W, fewer tiles than waves; K % 128 != 0 stays on the vector kernel
    for M, N, K in [(1, 1000, 256), (3, 5003, 4096), (16, 777, 1024), (16, 5003, 4096), (5, 40, 1152), (11, 3000, 2176),
                    (8, 333, 1000), (4, 2048, 4096)]:
        x = orc.f32_to_bf16(rng.standard_normal((M, K)).astype(np.float32))
        W = orc.f32_to_bf16(rng.standard_normal((N, K)).astype(np.float32) * 0.05)
        xd = torch.from_numpy(x.view(np.int16)).cuda().view(torch.bfloat16)
        Wd = torch.from_numpy(W.view(np.int16)).cuda().view(torch.bfloat16)
        got = orc.bf16_to_f32(_u16(torch, vmx.dense_gemv(xd, Wd)))
        ref = orc.bf16_to_f32(x).astype(np.float64) @ orc.bf16_to_f32(W).astype(np.float64).T
        assert np.abs(got - ref).max() <= 2 ** -7 * np.abs(ref).max() + 1e-5
    table = torch.randn((50, 64), device="cuda").to(torch.bfloat16)
    ids = torch.tensor([3, 49, 50, -1, 0], dtype=torch.int32, device="cuda")
    out = vmx.embedding_lookup(table, ids)
    assert torch.equal(out[0], table[3]) and torch.equal(out[1], table[49]) and torch.equal(out[4], table[0])
    assert not out[2].any() and not out[3].any()


@pytest.mark.parametrize("kind", ["awq", "exl3"])
def test_tp_collective_path_world1_matches_plain_decode(gpu, kind):
    """A 1-rank RCCL communicator drives the row-parallel all-reduce and the vocab-parallel argmax path;
    its results must equal the plain path bit for bit, eagerly and from a captured graph.  (EXL3: the plain path lets the
    consumers of o / gate|up / down apply the output transform, the collective path runs it as a launch of its own.)"""
    from vllm_rust_amd.parallel import Communicator
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM
    vmx, torch = gpu
    spec = _tiny_spec()
    B, MB = 2, 8
    bt = torch.arange(B * MB, dtype=torch.int32, device="cuda").reshape(B, MB)
    outs = []
    for use_comm in (False, True):
        model = QuantizedQwen3ForCausalLM(spec)
        model.load_synthetic(seed=9) if kind == "awq" else model.load_synthetic_exl3(bits=4, seed=9)
        comm = None
        if use_comm:
            comm = Communicator(0, 1)
            model.set_comm(comm)
        ids = torch.tensor([5, 77], dtype=torch.int32, device="cuda")
        pos = torch.zeros(B, dtype=torch.int32, device="cuda")
        slots = (bt[:, 0] * spec.block_size).contiguous()
        sl = torch.ones(B, dtype=torch.int32, device="cuda")
        out = torch.empty(B, dtype=torch.int32, device="cuda")
        logits = torch.empty((B, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
        seq = []
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            model.forward_decode_batch(ids, pos, slots, bt, sl, out_ids=out, logits=logits)
            model.advance(ids, out, pos, slots, sl, bt)
            torch.cuda.synchronize()
            seq.append((out.clone(), logits.clone()))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                model.forward_decode_batch(ids, pos, slots, bt, sl, out_ids=out, logits=logits)
                model.advance(ids, out, pos, slots, sl, bt)
            for _ in range(3):
                g.replay()
                torch.cuda.synchronize()
                seq.append((out.clone(), logits.clone()))
        outs.append(seq)
        model.close()
        if comm:
            comm.close()
    for (o0, l0), (o1, l1) in zip(*outs):
        assert torch.equal(o0, o1) and torch.equal(l0, l1)


@pytest.mark.parametrize("kv_dtype", [0, 1])
def test_iq_decoder_matches_oracle(gpu, kv_dtype):
    """Every projection a GGUF I-quant (the Gemma-4-12B UD-IQ3_XXS type mix), optional FP8 KV cache."""
    from oracle import oracle as orc
    from tests.model_oracle import OracleQwen3
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, rope_cos_sin_cache, T_KV_SCALES
    vmx, torch = gpu
    spec = _tiny_spec(hidden_size=256, intermediate_size=512, kv_cache_dtype=kv_dtype)
    model = QuantizedQwen3ForCausalLM(spec)
    types = {0: 18, 1: 22, 2: 18, 3: 21, 4: 18, 5: 17, 6: 23}     # q,k,v,o,gate,up,down
    ck = model.load_synthetic_iq(types, seed=4, keep_host_copy=True)
    ref = OracleQwen3(spec, ck, rope_cos_sin_cache(spec.head_dim, spec.max_position, spec.rope_theta))
    if kv_dtype == 1:
        scales = torch.tensor([0.02, 0.03], dtype=torch.float32, device="cuda")
        for li in range(spec.num_layers):
            model.load_tensor(T_KV_SCALES, li, scales)
        ref.kv_scales = (0.02, 0.03)
    B, MB = 2, 8
    bt = np.arange(B * MB, dtype=np.int32).reshape(B, MB)
    ids = np.array([17, 900], np.int32)
    for step in range(4):
        pos = np.full(B, step, np.int32)
        slots = (bt[:, 0] * spec.block_size + pos).astype(np.int32)
        sl = pos + 1
        logits = torch.empty((B, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
        out = model.forward_decode_batch(*(torch.from_numpy(a).cuda() for a in (ids, pos, slots, bt, sl)), logits=logits)
        torch.cuda.synchronize()
        exp, exp_h, _ = ref.decode(ids, pos, slots, bt, sl)
        got = orc.bf16_to_f32(_u16(torch, logits)); exp = orc.bf16_to_f32(exp)
        rel = np.sqrt(((got - exp) ** 2).sum() / (exp ** 2).sum())
        assert rel <= (3e-2 if kv_dtype == 0 else 5e-2), (step, rel)
        ids = out.cpu().numpy().astype(np.int32)
    model.close()


@pytest.mark.parametrize("bits,cb", [(4, 0), (3, 1)])
def test_exl3_decoder_matches_oracle(gpu, bits, cb):
    """Every projection an EXL3 trellis-coded linear (bf16 model: activations cast to fp16 for the linear and back)."""
    from oracle import oracle as orc
    from tests.model_oracle import OracleQwen3
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, rope_cos_sin_cache
    vmx, torch = gpu
    spec = _tiny_spec(hidden_size=256, intermediate_size=512)
    model = QuantizedQwen3ForCausalLM(spec)
    ck = model.load_synthetic_exl3(bits=bits, codebook=cb, seed=6, keep_host_copy=True)
    ref = OracleQwen3(spec, ck, rope_cos_sin_cache(spec.head_dim, spec.max_position, spec.rope_theta))
    B, MB = 2, 8
    bt = np.arange(B * MB, dtype=np.int32).reshape(B, MB)
    ids = np.array([5, 731], np.int32)
    for step in range(4):
        pos = np.full(B, step, np.int32)
        slots = (bt[:, 0] * spec.block_size + pos).astype(np.int32)
        sl = pos + 1
        logits = torch.empty((B, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
        out = model.forward_decode_batch(*(torch.from_numpy(a).cuda() for a in (ids, pos, slots, bt, sl)), logits=logits)
        torch.cuda.synchronize()
        exp, exp_h, _ = ref.decode(ids, pos, slots, bt, sl)
        got = orc.bf16_to_f32(_u16(torch, logits)); exp = orc.bf16_to_f32(exp)
        rel = np.sqrt(((got - exp) ** 2).sum() / (exp ** 2).sum())
        assert rel <= 3e-2, (step, rel)
        ids = out.cpu().numpy().astype(np.int32)
    model.close()


def test_exl3_decoder_fp16_matches_oracle(gpu):
    """An EXL3 model run in fp16 end to end (vmx_qwen3_set_activation_dtype: fp16 embedding / norms / lm_head, fp16 activations,
    logits and cache) against the composed fp16 oracle; then the native prompt pass against the decode path on the same model."""
    from tests.model_oracle import OracleQwen3
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, rope_cos_sin_cache
    vmx, torch = gpu
    spec = _tiny_spec(hidden_size=256, intermediate_size=512, act_dtype="f16", max_prefill_tokens=64)
    model = QuantizedQwen3ForCausalLM(spec)
    ck = model.load_synthetic_exl3(bits=4, codebook=0, seed=6, keep_host_copy=True)
    ref = OracleQwen3(spec, ck, rope_cos_sin_cache(spec.head_dim, spec.max_position, spec.rope_theta, "f16"))
    B, MB = 2, 8
    bt = np.arange(B * MB, dtype=np.int32).reshape(B, MB)
    ids = np.array([5, 731], np.int32)
    f32 = lambda u: u.view(np.float16).astype(np.float32)
    for step in range(4):
        pos = np.full(B, step, np.int32)
        slots = (bt[:, 0] * spec.block_size + pos).astype(np.int32)
        sl = pos + 1
        logits = torch.empty((B, spec.vocab_size), dtype=torch.float16, device="cuda")
        out = model.forward_decode_batch(*(torch.from_numpy(a).cuda() for a in (ids, pos, slots, bt, sl)), logits=logits)
        torch.cuda.synchronize()
        exp, _, _ = ref.decode(ids, pos, slots, bt, sl)
        got = f32(_u16(torch, logits)); exp = f32(exp)
        assert np.isfinite(got).all()
        rel = np.sqrt(((got - exp) ** 2).sum() / (exp ** 2).sum())
        assert rel <= 1e-2, (step, rel)
        ids = out.cpu().numpy().astype(np.int32)
    model.close()
    # prompt pass (fp16 prefill attention, 16-row EXL3 passes) == the same tokens through the decode path
    prompt = torch.tensor([3, 500, 42, 42, 7, 999, 1, 250, 13, 77, 600, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17], dtype=torch.int32, device="cuda")
    btp = torch.tensor([9, 2, 5, 11], dtype=torch.int32, device="cuda")
    a = QuantizedQwen3ForCausalLM(spec); a.load_synthetic_exl3(bits=4, seed=21)
    lg_a = torch.empty(spec.vocab_size, dtype=torch.float16, device="cuda")
    id_a = a.prefill(prompt, 0, btp, logits_last=lg_a)
    b = QuantizedQwen3ForCausalLM(spec); b.load_synthetic_exl3(bits=4, seed=21)
    ids_b, lg_b = b.forward(prompt, 0, btp)
    torch.cuda.synchronize()
    ga, gb = lg_a.float().cpu().numpy(), lg_b[-1].float().cpu().numpy()
    assert np.sqrt(((ga - gb) ** 2).sum() / (gb ** 2).sum()) <= 1e-2
    assert gb[int(id_a[0])] >= gb.max() - abs(gb.max()) * 2 ** -8
    a.close(); b.close()


def test_prefill_forward_equals_token_by_token_decode(gpu):
    """forward() over a prompt (chunks of max_batch pseudo-sequences) must give the logits of feeding the same
    tokens one at a time: rows of the small-M linears and the per-sequence attention are independent."""
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM
    vmx, torch = gpu
    spec = _tiny_spec(max_batch=4, num_blocks=16)
    prompt = torch.tensor([3, 500, 42, 42, 7, 999, 1, 250, 13, 77, 600], dtype=torch.int32, device="cuda")
    bt = torch.tensor([9, 2, 5, 11], dtype=torch.int32, device="cuda")
    a = QuantizedQwen3ForCausalLM(spec); a.load_synthetic(seed=21)
    ids_a, logits_a = a.forward(prompt, 0, bt)
    torch.cuda.synchronize()
    b = QuantizedQwen3ForCausalLM(spec); b.load_synthetic(seed=21)
    for i in range(prompt.numel()):
        pos = torch.tensor([i], dtype=torch.int32, device="cuda")
        slot = (bt[i // spec.block_size] * spec.block_size + i % spec.block_size).reshape(1).to(torch.int32)
        lg = torch.empty((1, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
        o = b.forward_decode_batch(prompt[i:i + 1].contiguous(), pos, slot, bt.reshape(1, -1), pos + 1, logits=lg)
        torch.cuda.synchronize()
        assert torch.equal(lg[0], logits_a[i]) and int(o[0]) == int(ids_a[i]), i
    a.close(); b.close()


@pytest.mark.parametrize("iq", [False, True])
def test_native_prefill_matches_decode_path(gpu, iq):
    """vmx_qwen3_prefill (all prompt rows through the linears at once, MFMA prefill attention) against the same prompt fed
    through the decode path: identical KV-cache effect (the following decode step agrees) and last-row logits within the
    attention kernels' tolerance; a second chunk with history (seqlen_offset > 0) continues the same sequence."""
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM
    vmx, torch = gpu
    spec = _tiny_spec(max_batch=4, num_blocks=16)
    spec.max_prefill_tokens = 40
    prompt = torch.randint(0, spec.vocab_size, (37,), dtype=torch.int32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    bt = torch.tensor([9, 2, 5, 11], dtype=torch.int32, device="cuda")
    types = {0: 18, 1: 22, 2: 18, 3: 21, 4: 18, 5: 17, 6: 23}
    a = QuantizedQwen3ForCausalLM(spec)
    b = QuantizedQwen3ForCausalLM(spec)
    for mdl in (a, b):
        mdl.load_synthetic_iq(types, seed=21) if iq else mdl.load_synthetic(seed=21)
    # reference: decode path, prompt in chunks of max_batch pseudo-sequences
    _, logits_ref = a.forward(prompt, 0, bt)
    # native: 25 tokens, then the remaining 12 with history
    la = torch.empty(spec.vocab_size, dtype=torch.bfloat16, device="cuda")
    b.prefill(prompt[:25], 0, bt)
    nid = b.prefill(prompt[25:], 25, bt, logits_last=la)
    torch.cuda.synchronize()
    ref = logits_ref[-1].float(); got = la.float()
    rel = float(((got - ref) ** 2).sum().sqrt() / (ref ** 2).sum().sqrt())
    assert rel <= 2e-2, rel
    assert ref[int(nid[0])] >= ref.max() - abs(float(ref.max())) * 2 ** -6
    # the caches must now serve a decode step equally well in both models
    tok = torch.tensor([int(nid[0])], dtype=torch.int32, device="cuda")
    pos = torch.tensor([37], dtype=torch.int32, device="cuda")
    slot = (bt[37 // spec.block_size] * spec.block_size + 37 % spec.block_size).reshape(1).to(torch.int32)
    l1 = torch.empty((1, spec.vocab_size), dtype=torch.bfloat16, device="cuda"); l2 = torch.empty_like(l1)
    a.forward_decode_batch(tok, pos, slot, bt.reshape(1, -1), pos + 1, logits=l1)
    b.forward_decode_batch(tok, pos, slot, bt.reshape(1, -1), pos + 1, logits=l2)
    torch.cuda.synchronize()
    rel = float(((l1.float() - l2.float()) ** 2).sum().sqrt() / (l1.float() ** 2).sum().sqrt())
    assert rel <= 2e-2, rel
    # preconditions
    from vllm_rust_amd._lib import VmxError
    with pytest.raises(VmxError, match="max_prefill_tokens"):
        b.prefill(torch.zeros(41, dtype=torch.int32, device="cuda"), 0, bt)
    a.close(); b.close()


def test_mixtral_style_moe_decoder_matches_oracle(gpu):
    """Mixtral-shaped layers (attention without q/k norms, 4 GPTQ experts top-2 behind a dense router) through the executor
    against the composed CPU oracle; a prompt through the native prefill path must leave an equivalent cache."""
    from oracle import oracle as orc
    from tests.model_oracle import OracleQwen3
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, rope_cos_sin_cache
    vmx, torch = gpu
    spec = _tiny_spec(qk_norm=False, num_experts=4, experts_per_tok=2, max_batch=3, max_prefill_tokens=20, rope_theta=1e6)
    model = QuantizedQwen3ForCausalLM(spec)
    ck = model.load_synthetic_moe(seed=11, keep_host_copy=True)
    ref = OracleQwen3(spec, ck, rope_cos_sin_cache(spec.head_dim, spec.max_position, spec.rope_theta))
    B = 3
    bt = np.arange(B * 4, dtype=np.int32).reshape(B, 4)
    ids = np.array([7, 123, 999], np.int32)
    for p in range(3):
        pos = np.full(B, p, np.int32)
        slots = (bt[:, 0] * spec.block_size + p).astype(np.int32)
        sl = pos + 1
        logits = torch.empty((B, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
        out = model.forward_decode_batch(*(torch.from_numpy(a).cuda() for a in (ids, pos, slots, bt, sl)), logits=logits)
        torch.cuda.synchronize()
        exp, _, _ = ref.decode(ids, pos, slots, bt, sl)
        got = orc.bf16_to_f32(_u16(torch, logits)); exp = orc.bf16_to_f32(exp)
        rel = np.sqrt(((got - exp) ** 2).sum() / (exp ** 2).sum())
        assert rel <= 3e-2, (p, rel)
        ids = out.cpu().numpy().astype(np.int32)
    # native prefill of a 17-token prompt on a fresh block table, then one decode step, against the decode path
    prompt = torch.randint(0, spec.vocab_size, (17,), dtype=torch.int32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    bt2 = torch.tensor([20, 21], dtype=torch.int32, device="cuda"); bt3 = torch.tensor([24, 25], dtype=torch.int32, device="cuda")
    la = torch.empty(spec.vocab_size, dtype=torch.bfloat16, device="cuda")
    model.prefill(prompt, 0, bt2, logits_last=la)
    spec_chunk = spec.max_batch
    _, lref = model.forward(prompt, 0, bt3)
    torch.cuda.synchronize()
    a, b = la.float(), lref[-1].float()
    assert float(((a - b) ** 2).sum().sqrt() / (b ** 2).sum().sqrt()) <= 3e-2
    model.close()

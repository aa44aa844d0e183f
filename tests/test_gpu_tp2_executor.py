"""Two, four and eight tensor-parallel ranks of the whole-token executor as processes on the one GPU of the test box, joined by the
peer-mapped one-shot all-reduce (no RCCL): each rank loads its shard of ONE seeded checkpoint, rank 0 also runs the unsharded model in
the same process, and every decode step's greedy ids and logits are compared.  One case compares the sharded run with the CPU
oracle directly, at the true Qwen3-8B layer geometry with EXL3 4.0 bpw projections (two ranks, two layers).  Dense Qwen3: column / row-parallel linears, vocab-parallel
argmax.  Mixtral: tensor-parallel attention + expert-parallel MLP (each rank holds half of the experts; one all-reduce combines
them, which for two experts per token is the same single bf16 addition as on one GPU).  Gemma 4: I-quant projections split by
output rows / 256-element input blocks, the one full-attention KV head replicated, a KV-shared last layer, vocabulary-parallel tied
Q3_K lm_head.  EXL3 (BASELINE's TP configuration of Qwen3-8B EXL3): trellis tensors split by whole 128-column / 128-row Hadamard
blocks (parallel.py: shard_exl3_column / shard_exl3_row), row-parallel fp16 outputs summed through the same all-reduce."""
import numpy as np
import pytest

from helpers import make_exchange, spawn_ranks

pytestmark = pytest.mark.gpu


def _spec(moe, world=2):
    """Smallest geometry every rank of `world` can hold a legal shard of: one KV head and a 128-wide slice of o_proj's and
    down_proj's inputs per rank (the int4 group / the EXL3 Hadamard block)."""
    from vllm_rust_amd.qwen3 import Qwen3Spec
    if moe == "exl3_true":   # two layers of Qwen3-8B
        return Qwen3Spec(hidden_size=4096, num_layers=2, num_heads=32, num_kv_heads=8, head_dim=128, intermediate_size=12288, vocab_size=8192,
                         max_position=2048, num_blocks=64, max_batch=2, max_seq_len=512)
    kw = dict(hidden_size=256, num_layers=2, num_heads=max(4, 2 * world), num_kv_heads=max(2, world), head_dim=64,
              intermediate_size=max(512, 128 * world), vocab_size=1024, group_size=128, max_position=64, num_blocks=8, block_size=16,
              max_batch=2, max_seq_len=64)
    if moe in ("exl3", "exl3_f16"):   # every shard must hold whole 128-wide Hadamard blocks: 128-wide heads
        kw.update(head_dim=128, num_heads=max(4, world))
        if moe == "exl3_f16":   # the model runs in fp16 end to end: fp16 rows through the all-reduce
            kw.update(act_dtype="f16")
    elif moe:
        kw.update(num_experts=max(4, world), experts_per_tok=2, qk_norm=False)
    return Qwen3Spec(**kw)


def _load_exl3_shard(model, ck, full, rank, world):
    """This rank's slice of the EXL3 checkpoint `ck` (host copy of load_synthetic_exl3): q/k/v/gate/up by output columns, o/down by
    input rows; norms and the embedding whole, lm_head by vocabulary rows."""
    from vllm_rust_amd.parallel import shard_exl3_column, shard_exl3_row
    from vllm_rust_amd.qwen3 import T_EMBED, T_FINAL_NORM, T_INPUT_LN, T_K_NORM, T_LM_HEAD, T_POST_LN, T_Q_NORM
    VL = full.vocab_size // world
    model.load_tensor(T_EMBED, 0, ck["embed"].cuda())
    model.load_tensor(T_LM_HEAD, 0, ck["lm_head"][rank * VL:(rank + 1) * VL].contiguous().cuda())
    model.load_tensor(T_FINAL_NORM, 0, ck["final_norm"].cuda())
    for li in range(full.num_layers):
        for tid, key in ((T_INPUT_LN, "input_ln"), (T_POST_LN, "post_ln"), (T_Q_NORM, "q_norm"), (T_K_NORM, "k_norm")):
            model.load_tensor(tid, li, ck[f"{li}.{key}"].cuda())
        for slot in range(7):
            bits, cb, tr, suh, svh = ck[f"{li}.{slot}.exl3"]
            row_parallel = slot in (3, 6)   # o_proj, down_proj
            tr, suh, svh = (shard_exl3_row if row_parallel else shard_exl3_column)(tr, suh, svh, rank, world)
            model.load_exl3_linear(li, slot, tr.cuda(), suh.cuda(), svh.cuda(), cb)


def _gemma_spec(world=2):
    from vllm_rust_amd.gemma4 import Gemma4Spec
    # (sliding, sliding, full) x 2 with the last layer KV-shared; the single full-attention KV head is replicated on every rank
    return Gemma4Spec(256, 6, 4 * world, 256 * world, 1000, head_dim_sliding=64, head_dim_full=128, kv_heads_sliding=world, kv_heads_full=1, sliding_window=4,
                      full_every=3, num_kv_shared_layers=1, max_position=64, kv_cache_dtype=1, num_blocks=8, max_batch=2, max_seq_len=64,
                      embed_ggml_type=11)


def _worker(rank, world, conns, result_q, moe):
    try:
        import torch
        import vllm_rust_amd as vmx
        from vllm_rust_amd.parallel import LocalCommunicator, shard_spec
        from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM
        vmx.init(0)
        gemma = moe == "gemma"

        comm = LocalCommunicator(rank, world, make_exchange(rank, world, conns), max_bytes=64 * 1024)
        ref = None
        oracle = None
        if gemma:
            from vllm_rust_amd.gemma4 import QuantizedGemma4ForCausalLM, shard_gemma4_spec
            full = _gemma_spec(world)
            shard = QuantizedGemma4ForCausalLM(shard_gemma4_spec(full, rank, world))
            shard.load_synthetic(seed=7, full_spec=full)
            if rank == 0:
                ref = QuantizedGemma4ForCausalLM(full)
                ref.load_synthetic(seed=7)
            # Gemma attends with scale 1 on RMS-normed q and k: norm weights near 1 give logits of order sqrt(head_dim) and a
            # near one-hot softmax that turns a one-ulp difference into a different token's V row; weights of head_dim^-1/4
            # give logits of order 1, as trained checkpoints have
            from vllm_rust_amd.gemma4 import T_K_NORM, T_Q_NORM
            gq = torch.Generator().manual_seed(9)
            for li in range(full.num_layers):
                hd = full.layer_geom(li)[0]
                for which in (T_Q_NORM, T_K_NORM):
                    w = ((1.0 + 0.1 * torch.randn(hd, generator=gq)) * (1.0 / (hd ** 0.25))).to(torch.bfloat16).cuda()
                    for mdl in (shard, ref):
                        if mdl is not None:
                            mdl.load_tensor(which, li, w)
        elif moe in ("exl3", "exl3_f16", "exl3_true"):
            full = _spec(moe, world)
            whole = QuantizedQwen3ForCausalLM(full)   # every rank draws the one seeded checkpoint; rank 0 keeps it as the reference
            ck = whole.load_synthetic_exl3(bits=4, codebook=0, seed=7, keep_host_copy=True)
            shard = QuantizedQwen3ForCausalLM(shard_spec(full, rank, world))
            _load_exl3_shard(shard, ck, full, rank, world)
            if rank == 0 and moe in ("exl3", "exl3_f16"):
                ref = whole
            else:
                whole.close()
            if rank == 0 and moe == "exl3_true":   # the CPU oracle of the unsharded model, not another run of the library
                from tests.model_oracle import OracleQwen3
                from vllm_rust_amd.qwen3 import rope_cos_sin_cache
                oracle = OracleQwen3(full, ck, rope_cos_sin_cache(full.head_dim, full.max_position, full.rope_theta))
        else:
            full = _spec(moe, world)
            shard = QuantizedQwen3ForCausalLM(shard_spec(full, rank, world))
            shard.load_synthetic_tp(full, seed=7)
            if rank == 0:
                ref = QuantizedQwen3ForCausalLM(full)
                (ref.load_synthetic_moe if moe else ref.load_synthetic)(seed=7)
        shard.set_comm(comm)
        B, V, VL = 2, full.vocab_size, full.vocab_size // world
        nblk = full.num_blocks // B
        bt = torch.arange(B * nblk, dtype=torch.int32, device="cuda").reshape(B, nblk)
        ids = torch.tensor([5, 900], dtype=torch.int32, device="cuda")
        ref_ids = ids.clone()
        report = []
        recorded = []
        steps = 3 if moe == "exl3_true" else 6

        def compare(p, full_row, out_ref, lg_host, picked):
            a, b = lg_host, full_row[:, :VL]
            rel = float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
            near_max = all(full_row[i, picked[i]] >= full_row[i].max() - abs(full_row[i].max()) * 2 ** -5 for i in range(B))
            report.append((p, rel, near_max, picked.tolist(), out_ref))

        for p in range(steps):
            pos = torch.full((B,), p, dtype=torch.int32, device="cuda")
            slots = (bt[:, p // full.block_size] * full.block_size + p % full.block_size).to(torch.int32)
            sl = pos + 1
            adt = getattr(shard, "act_torch_dtype", torch.bfloat16)
            lg = torch.empty((B, VL), dtype=adt, device="cuda")
            out = shard.forward_decode_batch(ids, pos, slots, bt, sl, logits=lg)
            torch.cuda.synchronize()
            if ref is not None:
                lr = torch.empty((B, V), dtype=adt, device="cuda")
                out_ref = ref.forward_decode_batch(ref_ids, pos, slots, bt, sl, logits=lr)
                torch.cuda.synchronize()
                compare(p, lr.float().cpu().numpy(), out_ref.cpu().numpy().tolist(), lg.float().cpu().numpy(), out.cpu().numpy())
                ref_ids = out.clone()       # both models walk the sharded run's tokens, so every step is comparable
            elif oracle is not None:        # the CPU oracle is slow: it walks the recorded steps after the ranks are done
                recorded.append([t.cpu().numpy().astype(np.int32) for t in (ids, pos, slots, bt, sl)] + [lg.float().cpu().numpy(), out.cpu().numpy()])
            ids = out.clone()
        timeouts = comm.p2p_timeouts()
        if oracle is not None:
            from oracle import oracle as orc
            for p, (i_, pos_, slots_, bt_, sl_, lg_host, picked) in enumerate(recorded):
                exp, _, _ = oracle.decode(i_, pos_, slots_, bt_, sl_)
                full_row = orc.bf16_to_f32(exp)
                compare(p, full_row, full_row.argmax(1).tolist(), lg_host, picked)
        result_q.put((rank, (report, ids.cpu().numpy().tolist(), timeouts)))
        shard.close()
        if ref is not None:
            ref.close()
        comm.close()
    except Exception as e:   # report instead of hanging the parent
        import traceback
        result_q.put((rank, ([("exception", repr(e), traceback.format_exc())], None, -1)))


CASES = [(False, 2), (True, 2), ("gemma", 2), ("exl3", 2), (False, 4), (False, 8), (True, 8), ("gemma", 4), ("exl3", 4), ("exl3_true", 2), ("exl3_f16", 2)]


@pytest.mark.parametrize("moe,world", CASES, ids=[f"{'dense' if m is False else 'moe' if m is True else m}-tp{w}" for m, w in CASES])
def test_tp_executor_matches_single_gpu(moe, world):
    results = spawn_ranks(world, _worker, (moe,), timeout=400)
    for r, (report, _, timeouts) in results.items():
        assert not (report and report[0][0] == "exception"), report[0][1:]
        assert timeouts == 0, f"rank {r}: {timeouts} flag waits timed out"
    for r in range(1, world):
        assert results[r][1] == results[0][1]                 # the vocab-parallel argmax gives every rank the same tokens
    report = results[0][0]
    assert len(report) == (3 if moe == "exl3_true" else 6)
    for p, rel, near_max, picked, ref_picked in report:
        # the row-parallel o / down outputs are rounded to bf16 per rank before the sum: a bf16-ulp-level difference per layer.
        # On the 256-wide Gemma model that difference re-rolls the q8_1 activation rounding of every later projection: each run
        # sits about 2 % from the CPU oracle (tests/test_gpu_gemma4.py), so two runs sit about 3 % from each other.  The EXL3
        # case against the CPU oracle takes the bar of the single-GPU EXL3 tests at that geometry.
        bar = 6e-2 if moe == "gemma" else 3e-2 if moe == "exl3_true" else 2e-2
        assert rel <= bar, [(q, round(r, 4)) for q, r, *_ in report]
        assert near_max, (p, picked, ref_picked)

"""Mixtral-shaped three-layer model, four rows: per-row error against the oracle with the expert GEMM form forced (0 / 16)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
from vllm_rust_amd.qwen3 import Qwen3Spec, QuantizedQwen3ForCausalLM, rope_cos_sin_cache
from tests.model_oracle import OracleQwen3
from tests.test_gpu_true_geometry import _fill_kv, _u16, _act_f32, VOCAB
from oracle import oracle as orc
vmx.init(0)
spec = Qwen3Spec(4096, 3, 32, 8, 128, 14336, VOCAB, rms_norm_eps=1e-5, rope_theta=1e6, qk_norm=False, num_experts=8, experts_per_tok=2,
                 max_position=2048, num_blocks=200, max_batch=4, max_seq_len=1024)
res = {}
for rows in (0, 16):
    check(lib.vmx_debug_moe_block_rows(rows))
    model = QuantizedQwen3ForCausalLM(spec)
    ck = model.load_synthetic_moe(seed=13, keep_host_copy=True)
    ref = OracleQwen3(spec, ck, rope_cos_sin_cache(spec.head_dim, spec.max_position, spec.rope_theta, spec.act_dtype))
    shape = (spec.num_blocks, spec.block_size, spec.num_kv_heads, spec.head_dim)
    _fill_kv(torch, model, ref, spec.num_layers, [shape] * spec.num_layers, spec.kv_cache_dtype, seed=99)
    start, first = [0, 15, 16, 700], [1, 2, 3, 4]
    B, bs = 4, spec.block_size
    mb = (max(start) + 2 + bs - 1) // bs + 1
    bt = np.random.default_rng(5).permutation(spec.num_blocks)[:B * mb].reshape(B, mb).astype(np.int32)
    ids, pos = np.asarray(first, np.int32), np.asarray(start, np.int32)
    for step in range(2):
        slots = (bt[np.arange(B), pos // bs] * bs + pos % bs).astype(np.int32)
        sl = pos + 1
        logits = torch.empty((B, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
        out = model.forward_decode_batch(*(torch.from_numpy(a).cuda() for a in (ids, pos, slots, bt, sl)), max_seq_len=int(sl.max()), logits=logits)
        torch.cuda.synchronize()
        exp, exp_h, _ = ref.decode(ids, pos, slots, bt, sl)
        got = _act_f32(spec, _u16(torch, logits)); expf = _act_f32(spec, exp)
        rows_err = [float(np.sqrt(((got[b] - expf[b]) ** 2).sum() / (expf[b] ** 2).sum())) for b in range(B)]
        print(f"rows {rows} step {step} ids {ids.tolist()} per-row logits rel-L2 " + " ".join(f"{e:.4f}" for e in rows_err), flush=True)
        res[(rows, step)] = got
        ids = exp.view(np.uint16) if False else np.asarray([int(np.argmax(expf[b])) for b in range(B)], np.int32)   # teacher-forced: the oracle's tokens
        pos = pos + 1
    model.close()
for step in range(2):
    a, b = res[(0, step)], res[(16, step)]
    print(f"step {step}: per-row rel-L2 between the two forms " + " ".join(f"{float(np.sqrt(((a[r] - b[r]) ** 2).sum() / (b[r] ** 2).sum())):.4f}" for r in range(4)))

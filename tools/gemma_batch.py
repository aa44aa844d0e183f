"""Gemma-4-12B-shaped decode at several concurrent sequences; VMX_MMA_MIN_ROWS=17 keeps the per-row MMVQ kernels (A/B of the
matrix-core form)."""
import os, sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib
from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM
vmx.init(0)
if os.environ.get("VMX_MMA_MIN_ROWS"):
    lib.vmx_debug_iq_mma_min_rows(int(os.environ["VMX_MMA_MIN_ROWS"]))
batches = tuple(int(b) for b in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 2, 4, 8, 16)
prompt, steps, warmup = 64, 24, 4
total = prompt + warmup + steps + 8
per_seq = (total + 15) // 16 + 1
BM = max(batches)
spec = Gemma4Spec.gemma4_12b(sliding_window=1024, max_position=4096, kv_cache_dtype=1, num_blocks=per_seq * BM, max_batch=BM, max_seq_len=total + 1,
                             embed_ggml_type=11, final_logit_softcap=30.0, max_prefill_tokens=prompt)
model = QuantizedGemma4ForCausalLM(spec)
model.load_synthetic(seed=99)
dev = model.device
for B in batches:
    bt = torch.arange(per_seq * B, dtype=torch.int32, device=dev).reshape(B, per_seq).contiguous()
    ids = torch.empty(B, dtype=torch.int32, device=dev)
    g7 = torch.Generator(device=dev).manual_seed(7)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        for b in range(B):
            pids = torch.randint(0, spec.vocab_size, (prompt,), dtype=torch.int32, device=dev, generator=g7)
            model.prefill(pids, 0, bt[b], out_id=ids[b:b + 1])
        pos = torch.full((B,), prompt, dtype=torch.int32, device=dev)
        slots = (bt[:, prompt // 16] * 16 + prompt % 16).to(torch.int32).contiguous()
        sl = torch.full((B,), prompt + 1, dtype=torch.int32, device=dev)
        out = torch.empty(B, dtype=torch.int32, device=dev)
        def step():
            model.forward_decode_batch(ids, pos, slots, bt, sl, max_seq_len=spec.max_seq_len, out_ids=out)
            model.advance(ids, out, pos, slots, sl, bt)
        step()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            step()
        for _ in range(warmup):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        del g
    print(f"B={B}: {B * steps / dt:.1f} tok/s aggregate, {dt / steps * 1e3:.3f} ms/step", flush=True)

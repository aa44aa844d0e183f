"""Gemma-4-12B-shaped model, batch 1: a short prompt, then N graph-replayed decode steps and nothing else -- for a kernel trace
of the decode step alone (rocprofv3 --kernel-trace -- python3 tools/g4_decode_only.py 64)."""
import sys
import time
sys.path.insert(0, ".")
import torch
import vllm_rust_amd as vmx
from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM

vmx.init(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 256
total = prompt + steps + 16
n_blocks = (total + 15) // 16 + 1
spec = Gemma4Spec.gemma4_12b(sliding_window=1024, max_position=max(4096, total + 1), kv_cache_dtype=1, num_blocks=n_blocks, max_batch=1,
                             max_seq_len=total + 1, embed_ggml_type=11, final_logit_softcap=30.0)
model = QuantizedGemma4ForCausalLM(spec)
model.load_synthetic(seed=99)
dev = model.device
bt = torch.arange(n_blocks, dtype=torch.int32, device=dev).flip(0).reshape(1, n_blocks).contiguous()
ids = torch.full((1,), 17, dtype=torch.int32, device=dev)
pos = torch.full((1,), prompt, dtype=torch.int32, device=dev)
slots = (bt[:, prompt // 16] * 16 + prompt % 16).to(torch.int32).contiguous()
sl = torch.full((1,), prompt + 1, dtype=torch.int32, device=dev)
out = torch.empty(1, dtype=torch.int32, device=dev)


def step():
    model.forward_decode_batch(ids, pos, slots, bt, sl, max_seq_len=spec.max_seq_len, out_ids=out)
    model.advance(ids, out, pos, slots, sl, bt)


side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    step()   # the caches hold zeros below `prompt`: the step's cost does not depend on their contents
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        step()
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("decode tok/s %.1f  ms/step %.4f" % (steps / dt, dt / steps * 1e3))

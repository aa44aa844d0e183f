"""Per-wave timeline of the decode attention launch (split_decode_attn_kernel) of a Qwen3-8B-shaped layer: build variant
libvmx_stamps.so (make -C vllm-rust_amd/csrc STAMPS=1), run as
VMX_LIB_VARIANT=stamps python tools/attn_stamps.py [context]."""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
from vllm_rust_amd.qwen3 import Qwen3Spec, QuantizedQwen3ForCausalLM

vmx.init(0)
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 300
top = ctx + 64
n_blocks = (top + 15) // 16 + 1
spec = Qwen3Spec.qwen3_8b(max_position=top + 1, num_blocks=n_blocks, max_batch=1, max_seq_len=top + 1, max_prefill_tokens=2048)
spec.num_layers = 6
model = QuantizedQwen3ForCausalLM(spec)
model.load_synthetic(seed=4321)
dev = model.device
bt = torch.arange(n_blocks, dtype=torch.int32, device=dev).reshape(1, n_blocks).contiguous()
ids = torch.empty(1, dtype=torch.int32, device=dev)
g7 = torch.Generator(device=dev).manual_seed(ctx)
for c0 in range(0, ctx, 2048):
    n = min(2048, ctx - c0)
    model.prefill(torch.randint(0, spec.vocab_size, (n,), dtype=torch.int32, device=dev, generator=g7), c0, bt[0], out_id=ids)
pos = torch.full((1,), ctx, dtype=torch.int32, device=dev)
slots = (bt[:, ctx // 16] * 16 + ctx % 16).to(torch.int32).contiguous()
sl = torch.full((1,), ctx + 1, dtype=torch.int32, device=dev)
o = torch.empty(1, dtype=torch.int32, device=dev)

def step():
    model.forward_decode_batch(ids, pos, slots, bt, sl, max_seq_len=top + 1, out_ids=o)
    model.advance(ids, o, pos, slots, sl, bt)
step()
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    step()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
buf = np.zeros((4096, 8), dtype=np.uint64)
lib.vmx_debug_attn_stamps_read.restype = C.c_int
lib.vmx_debug_attn_stamps_read.argtypes = [C.c_void_p, C.c_size_t]
check(lib.vmx_debug_attn_stamps_read(buf.ctypes.data, buf.nbytes))
s = buf[buf[:, 0] > 0].astype(np.float64)
t0 = s[:, 0].min()
rel = (s - t0) / 100.0
names = ["entry", "split sums landed", "before A arithmetic (tile 0 requested)", "A done", "after barrier", "walk done", "merge barrier passed", "end"]
print(f"context {ctx}: {len(s)} waves; launch span first entry -> last end {rel[:, 7].max():.2f} us")
for i, n in enumerate(names):
    c = rel[:, i]
    print(f"  {n:40s} min {c.min():6.2f}  median {np.median(c):6.2f}  p90 {np.percentile(c, 90):6.2f}  max {c.max():6.2f} us")
# the slowest wave's own timeline
w = int(np.argmax(rel[:, 7]))
print("  slowest wave:", " ".join(f"{x:.2f}" for x in rel[w]))
model.close()

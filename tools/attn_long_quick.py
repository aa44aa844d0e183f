"""Time vmx.paged_attention (stand-alone decode attention) at long contexts, Qwen3-8B head geometry, one sequence.
usage: python tools/attn_long_quick.py TAG"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import vllm_rust_amd as vmx
vmx.init(0)
H, KVH, D, BS = 32, 8, 128, 16
for ctx in (2048, 4096, 8192, 16384):
    nb = ctx // BS + 1
    kc = torch.randn(nb, BS, KVH, D, device="cuda").to(torch.bfloat16).view(torch.int16)
    vc = torch.randn(nb, BS, KVH, D, device="cuda").to(torch.bfloat16).view(torch.int16)
    bt = torch.arange(nb, dtype=torch.int32, device="cuda").reshape(1, nb)
    sl = torch.tensor([ctx], dtype=torch.int32, device="cuda")
    q = torch.randn(1, H * D, device="cuda").to(torch.bfloat16)
    one = torch.ones(1, dtype=torch.float32, device="cuda")
    f = lambda: vmx.paged_attention(q, kc, vc, bt, sl, D ** -0.5, H, KVH, D, BS, ctx, 0, 0, one, one)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    print(sys.argv[1] if len(sys.argv) > 1 else "", ctx, "us/call", round(us, 1), "KV GB/s", round(2 * ctx * KVH * D * 2 / us / 1e3, 1))

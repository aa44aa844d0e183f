"""lm_head-shaped dense bf16 launch (151936 x 4096) at 1 / 4 / 8 / 16 activation rows, graph-replayed; every row count printed."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import vllm_rust_amd as vmx
vmx.init(0)
N, K = 151936, 4096
W = (torch.randn((N, K), device="cuda") * 0.02).to(torch.bfloat16)
s = torch.cuda.Stream()
for M in (1, 4, 8, 16):
    x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    with torch.cuda.stream(s):
        vmx.dense_gemv(x, W); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(4): vmx.dense_gemv(x, W)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 20 * 1e6
    print(f"M={M:2d}: {us:8.1f} us  {N * K * 2 / us / 1e6:6.2f} TB/s", flush=True)

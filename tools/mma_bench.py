"""Matrix-core MMVQ (vmx_iq_mmvq_mma_f32) against the per-row kernel on Gemma-4-12B projection shapes: microseconds per launch
from back-to-back graph replays over a pool of distinct weight copies (so that nothing is served from the caches)."""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib
vmx.init(0)
TS = {18: 98, 21: 110, 22: 82}
shapes = [("gate", 15360, 3840), ("down", 3840, 15360), ("q", 4096, 3840), ("o", 3840, 4096)]
types = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [18]
Ms = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16]
dev = torch.device("cuda")
for tid in types:
    for name, N, K in shapes:
        nbytes = N * (K // 256) * TS[tid]
        copies = max(2, min(16, (1 << 30) // nbytes))
        raws = []
        for c in range(copies):
            b = torch.randint(0, 256, (N * (K // 256), TS[tid]), dtype=torch.uint8, device=dev)
            b[:, 0] = 0; b[:, 1] = 0x2C
            raws.append(b.reshape(-1))
        packs = [vmx.iq_mma_pack(tid, r, N, K) for r in raws]
        for M in Ms:
            aq = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
            ad = torch.rand((M, K // 32), device=dev) * 0.01
            for label, fn, ws in (("mma", vmx.iq_mmvq_mma, packs), ("row", vmx.iq_mmvq, raws)):
                side = torch.cuda.Stream()
                with torch.cuda.stream(side):
                    for w in ws: fn(tid, w, aq, ad, N, K)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        for w in ws: fn(tid, w, aq, ad, N, K)
                    g.replay(); torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                    for _ in range(5): g.replay()
                    e1.record(side)
                    torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / (5 * len(ws))
                print(f"type {tid} {name:5s} [{N} x {K}] M={M:2d} {label}: {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s", flush=True)
        del raws, packs

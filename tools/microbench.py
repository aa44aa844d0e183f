#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X: achieved algorithmic GB/s per launch.

Weights are rotated over enough copies (>= 1 GiB in total) that no launch finds
its weights in the 256 MiB Infinity Cache, as in a real decode step where every
layer streams different weights.
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import vllm_rust_amd as vmx  # noqa: E402
from vllm_rust_amd._lib import lib, check  # noqa: E402

IQ = {"iq2_xs": (17, 74), "iq2_s": (22, 82), "iq3_xxs": (18, 98), "iq3_s": (21, 110), "iq4_xs": (23, 136)}


def timed(fn, n_iter, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n_iter):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n_iter  # us


def bench_mmvq(name, n_out, n_in, m, iters):
    tid, ts = IQ[name]
    row_bytes = n_in // 256 * ts
    wbytes = n_out * row_bytes
    copies = max(2, (1 << 30) // wbytes + 1)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = torch.randint(0, 256, (copies, wbytes), dtype=torch.uint8, device="cuda", generator=g)
    w.view(copies, -1, ts)[:, :, 1] = 0x30  # finite super-scales
    x = torch.randn((m, n_in), device="cuda")
    aq, ad = vmx.quantize_q8_1(x)
    y = torch.empty((m, n_out), device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run(i):
        check(lib.vmx_iq_mmvq_f32(tid, w[i % copies].data_ptr(), aq.data_ptr(), ad.data_ptr(), y.data_ptr(), n_out, n_in, m, st))

    us = timed(run, iters)
    alg = wbytes + m * n_in * (1 + 4 / 32) + m * n_out * 4
    return {"kernel": f"mmvq_{name}", "n_out": n_out, "n_in": n_in, "m": m, "us": round(us, 2), "GBps": round(alg / us / 1e3, 1),
            "frac_8TBs": round(alg / us / 1e3 / 8000, 3)}


def bench_int4(N, K, M, iters, group=128):
    nbytes = lib.vmx_int4_packed_bytes(K, N, group)
    copies = max(2, (1 << 30) // nbytes + 1)
    g = torch.Generator(device="cuda").manual_seed(2)
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device="cuda", generator=g)
    qz = torch.randint(-2**31, 2**31 - 1, (K // group, N // 8), dtype=torch.int32, device="cuda", generator=g)
    sc = (torch.rand((K // group, N), device="cuda") * 0.02).to(torch.bfloat16)
    one = vmx.int4_pack(qw, sc, qz, K, N, group)
    packed = one.unsqueeze(0).repeat(copies, 1).contiguous()
    x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ws = torch.empty(max(lib.vmx_int4_gemm_workspace_bytes(M, N, K), 8), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run(i):
        check(lib.vmx_int4_gemm_bf16(y.data_ptr(), x.data_ptr(), packed[i % copies].data_ptr(), None, M, N, K, group,
                                     ws.data_ptr(), ws.numel(), st))

    us = timed(run, iters)
    alg = K * N // 2 + (K // group) * N * 2 + (K // group) * N // 2 + M * K * 2 + M * N * 2
    return {"kernel": "int4_gemm", "N": N, "K": K, "M": M, "us": round(us, 2), "GBps": round(alg / us / 1e3, 1),
            "frac_8TBs": round(alg / us / 1e3 / 8000, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--out", default="gpurun_out/microbench.jsonl")
    ap.add_argument("--what", default="mmvq,int4")
    a = ap.parse_args()
    vmx.init(0)
    rows = []
    if "mmvq" in a.what:
        for name in IQ:
            for n_out, n_in in [(4096, 4096), (14336, 4096)]:
                rows.append(bench_mmvq(name, n_out, n_in, 1, a.iters))
                print(json.dumps(rows[-1]), flush=True)
        for m in (2, 4, 8, 16):
            rows.append(bench_mmvq("iq3_xxs", 4096, 4096, m, a.iters))
            print(json.dumps(rows[-1]), flush=True)
        rows.append(bench_mmvq("iq3_xxs", 15360, 3840, 1, a.iters))
        print(json.dumps(rows[-1]), flush=True)
    if "int4" in a.what:
        for N, K in [(4096, 4096), (6144, 4096), (24576, 4096), (4096, 12288), (1024, 4096)]:
            for M in (1, 16):
                rows.append(bench_int4(N, K, M, a.iters))
                print(json.dumps(rows[-1]), flush=True)
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    with open(a.out, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()

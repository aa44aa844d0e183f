"""int4 GEMM (mode 0, M=1) time per launch, graph of 40 launches over rotating packed weights."""
import sys, time, os
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
vmx.init(0)
s = torch.cuda.Stream()
M = int(os.environ.get("M", "1"))
for N, K in ((4096, 4096), (6144, 4096), (4096, 12288), (24576, 4096)):
    nbytes = lib.vmx_int4_packed_bytes(K, N, 128)
    stride = (nbytes + 255) // 256 * 256
    ncopies = max(2, (1200 << 20) // stride)
    g0 = torch.Generator(device="cuda").manual_seed(2)
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device="cuda", generator=g0)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device="cuda", generator=g0)
    sc = (torch.rand((K // 128, N), device="cuda") * 0.02).to(torch.bfloat16)
    one = vmx.int4_pack(qw, sc, qz, K, N, 128)
    pool = torch.empty(ncopies * stride, dtype=torch.uint8, device="cuda")
    for i in range(ncopies):
        pool[i * stride:i * stride + nbytes] = one
    x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ws = torch.empty(max(lib.vmx_int4_gemm_workspace_bytes(M, N, K), 8), dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(s):
        st = s.cuda_stream
        def chain(n, off=0):
            for i in range(n):
                check(lib.vmx_int4_gemm_bf16(y.data_ptr(), x.data_ptr(), pool.data_ptr() + ((i + off) % ncopies) * stride, None, M, N, K, 128,
                                             ws.data_ptr(), ws.numel(), st))
        chain(2); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); n = 40
        with torch.cuda.graph(g, stream=s):
            chain(n, 3)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 5 / n * 1e6
    print(f"N={N:5d} K={K:5d} M={M}: {us:6.2f} us/call (incl. reduce kernel when split)  {nbytes/us/1e3:7.0f} GB/s", flush=True)
    del pool

"""int4 GEMM time over row counts for the Qwen3-8B linear shapes (rotating weight copies, HIP events): usage gemm_m_sweep.py [M ...]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
vmx.init(0)
Ms = [int(a) for a in sys.argv[1:]] or [8, 16, 24, 32, 48, 64]
st = torch.cuda.current_stream().cuda_stream
for name, N, K in (("gate_up", 24576, 4096), ("down", 4096, 12288), ("qkv", 6144, 4096), ("o", 4096, 4096)):
    g0 = torch.Generator(device="cuda").manual_seed(2)
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device="cuda", generator=g0)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device="cuda", generator=g0)
    sc = (torch.rand((K // 128, N), device="cuda") * 0.02).to(torch.bfloat16)
    packs = [vmx.int4_pack(qw, sc, qz, K, N, 128) for _ in range(8)]
    row = []
    for M in Ms:
        x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
        y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        ws = torch.empty(max(lib.vmx_int4_gemm_workspace_bytes(M, N, K), 8), dtype=torch.uint8, device="cuda")
        def run(n):
            for i in range(n):
                check(lib.vmx_int4_gemm_bf16(y.data_ptr(), x.data_ptr(), packs[i % 8].data_ptr(), None, M, N, K, 128, ws.data_ptr(), ws.numel(), st))
        run(8)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(40); e1.record(); torch.cuda.synchronize()
        row.append(f"M={M}: {e0.elapsed_time(e1) / 40 * 1000:.1f}")
    print(name, " ".join(row), "us (whole call incl. split-K reduce)", flush=True)

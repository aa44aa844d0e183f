"""One int4 GEMM shape at M rows, a few eager launches (for counter passes): usage gemm_one.py N K M [glu]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
vmx.init(0)
N, K, M = (int(a) for a in sys.argv[1:4])
g0 = torch.Generator(device="cuda").manual_seed(2)
qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device="cuda", generator=g0)
qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device="cuda", generator=g0)
sc = (torch.rand((K // 128, N), device="cuda") * 0.02).to(torch.bfloat16)
packs = [vmx.int4_pack(qw, sc, qz, K, N, 128) for _ in range(12)]
x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
ws = torch.empty(max(lib.vmx_int4_gemm_workspace_bytes(M, N, K), 8), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for i in range(12):
    check(lib.vmx_int4_gemm_bf16(y.data_ptr(), x.data_ptr(), packs[i].data_ptr(), None, M, N, K, 128, ws.data_ptr(), ws.numel(), st))
torch.cuda.synchronize()
print("done")

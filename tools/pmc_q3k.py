"""Q3_K lm_head kernels on the Gemma-4-12B table shape (262144 x 3840), for rocprofv3 passes: the one-row f32 kernel and the
16-row matrix-core kernel, five launches each on random blocks."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check

vmx.init(0)
n_out, n_in = 262144, 3840
g = torch.Generator(device="cuda").manual_seed(1)
w = torch.randint(0, 256, (n_out * (n_in // 256) * 110,), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
blocks = w.view(-1, 110)
blocks[:, 108] = 0
blocks[:, 109] = 0x28          # small finite f16 d
x16 = torch.randn((16, n_in), device="cuda", generator=g).to(torch.bfloat16)
x1 = x16[:1].float().contiguous()
y16 = torch.empty((16, n_out), dtype=torch.float32, device="cuda")
y1 = torch.empty((1, n_out), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    check(lib.vmx_kquant_gemv_f32(11, w.data_ptr(), x1.data_ptr(), y1.data_ptr(), n_out, n_in, 1, st))
    check(lib.vmx_q3k_rows16_bf16(w.data_ptr(), x16.data_ptr(), y16.data_ptr(), n_out, n_in, 16, st))
torch.cuda.synchronize()
print("algorithmic bytes per launch", n_out * (n_in // 256) * 110)

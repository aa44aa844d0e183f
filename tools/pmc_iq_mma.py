"""The matrix-core MMVQ alone on Gemma-4-12B linear shapes at 16 rows (random IQ3_XXS / IQ3_S blocks), five launches each, for
rocprofv3 counter passes: 30720 x 3840 as two tensors is what gate|up launches, 3840 x 15360 the down projection."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check

vmx.init(0)
g = torch.Generator(device="cuda").manual_seed(1)
st = torch.cuda.current_stream().cuda_stream
TS = {18: 98, 21: 110}
for t, n_out, n_in in ((18, 30720, 3840), (21, 3840, 15360), (18, 8192, 3840)):
    nblk = n_out * (n_in // 256)
    raw = torch.randint(0, 256, (nblk, TS[t]), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    raw[:, 0] = 0
    raw[:, 1] = 0x28
    packed = torch.empty(lib.vmx_iq_mma_packed_bytes(t, n_out, n_in), dtype=torch.uint8, device="cuda")
    check(lib.vmx_iq_mma_pack(t, raw.data_ptr(), packed.data_ptr(), n_out, n_in, st))
    aq = torch.randint(-127, 128, (16, n_in), device="cuda", generator=g, dtype=torch.int32).to(torch.int8)
    ad = torch.rand((16, n_in // 32), device="cuda", generator=g) * 0.01
    y = torch.empty((16, n_out), dtype=torch.float32, device="cuda")
    for _ in range(5):
        check(lib.vmx_iq_mmvq_mma_f32(t, packed.data_ptr(), aq.data_ptr(), ad.data_ptr(), y.data_ptr(), n_out, n_in, 16, st))
    torch.cuda.synchronize()
    print(t, n_out, n_in, "bytes", nblk * TS[t])

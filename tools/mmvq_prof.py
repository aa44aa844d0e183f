"""A few plain launches of one I-quant MMVQ shape over a rotating pool of weight copies (for rocprofv3 passes).
env: TYPE (iq3_xxs), NOUT, NIN, N (launches)."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
vmx.init(0)
IQ = {"iq2_xs": (17, 74), "iq2_s": (22, 82), "iq3_xxs": (18, 98), "iq3_s": (21, 110), "iq4_xs": (23, 136)}
tid, ts = IQ[os.environ.get("TYPE", "iq3_xxs")]
n_out, n_in, n = int(os.environ.get("NOUT", 14336)), int(os.environ.get("NIN", 4096)), int(os.environ.get("N", 20))
wbytes = n_out * (n_in // 256) * ts
stride = (wbytes + 255) // 256 * 256
ncopies = max(2, (600 << 20) // stride)
pool = torch.randint(0, 256, (ncopies, stride), dtype=torch.uint8, device="cuda")
pool[:, :wbytes].view(ncopies, -1, ts)[:, :, 1] = 0x30
x = torch.randn((1, n_in), device="cuda")
aq, ad = vmx.quantize_q8_1(x)
y = torch.empty((1, n_out), device="cuda")
st = torch.cuda.current_stream().cuda_stream
for i in range(n):
    check(lib.vmx_iq_mmvq_f32(tid, pool[i % ncopies].data_ptr(), aq.data_ptr(), ad.data_ptr(), y.data_ptr(), n_out, n_in, 1, st))
torch.cuda.synchronize()
print("done", wbytes)

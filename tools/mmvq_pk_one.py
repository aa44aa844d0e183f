"""One packed-MMVQ shape replayed a few times (for rocprofv3 counter passes): gate|up of Gemma-4-12B, IQ3_XXS 30720 x 3840."""
import sys
import torch
sys.path.insert(0, ".")
import vllm_rust_amd as vmx
vmx.init(0)
t, n_out, n_in, ts = 18, 30720, 3840, 98
nbytes = n_out * (n_in // 256) * ts
copies = 12
g = torch.Generator(device="cuda").manual_seed(1)
raw = torch.randint(0, 256, (copies, nbytes), dtype=torch.uint8, device="cuda", generator=g)
raw.view(copies, -1, ts)[:, :, 1] = 0x2C
packed = [vmx.iq_pack(t, raw[c], n_out, n_in) for c in range(copies)]
aq = torch.randint(-127, 127, (1, n_in), dtype=torch.int8, device="cuda")
ad = torch.rand((1, n_in // 32), device="cuda") * 0.01
y = torch.empty((1, n_out), dtype=torch.float32, device="cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "packed"
for _ in range(3):
    for c in range(copies):
        if which == "raw":
            vmx.iq_mmvq(t, raw[c], aq, ad, n_out, n_in, out=y)
        else:
            vmx.iq_mmvq_packed(t, packed[c], aq, ad, n_out, n_in, out=y)
torch.cuda.synchronize()

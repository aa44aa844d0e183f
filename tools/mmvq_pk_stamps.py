"""Per-wave timeline of the packed MMVQ kernel on the Gemma-4-12B gate|up shape (IQ3_XXS 30720 x 3840): medians of the
in-kernel clock stamps, relative to the earliest wave entry of the launch."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
vmx.init(0)
t, n_out, n_in, ts = (18, 30720, 3840, 98) if len(sys.argv) < 2 else tuple(int(x) for x in sys.argv[1:5])
nbytes = n_out * (n_in // 256) * ts
copies = 8
g = torch.Generator(device="cuda").manual_seed(1)
raw = torch.randint(0, 256, (copies, nbytes), dtype=torch.uint8, device="cuda", generator=g)
raw.view(copies, -1, ts)[:, :, 1] = 0x2C
packed = [vmx.iq_pack(t, raw[c], n_out, n_in) for c in range(copies)]
aq = torch.randint(-127, 127, (1, n_in), dtype=torch.int8, device="cuda")
ad = torch.rand((1, n_in // 32), device="cuda") * 0.01
y = torch.empty((1, n_out), dtype=torch.float32, device="cuda")
stamps = torch.zeros((4096, 16, 10), dtype=torch.int64, device="cuda")
for c in range(copies):
    vmx.iq_mmvq_packed(t, packed[c], aq, ad, n_out, n_in, out=y)
torch.cuda.synchronize()
check(lib.vmx_debug_iq_packed_stamps(stamps.data_ptr()))
for c in range(copies):
    stamps.zero_()
    vmx.iq_mmvq_packed(t, packed[c], aq, ad, n_out, n_in, out=y)
    torch.cuda.synchronize()
check(lib.vmx_debug_iq_packed_stamps(None))
s = stamps.cpu().numpy().reshape(-1, 10)
s = s[s[:, 0] > 0]
rt0, rt1 = s[:, 8].copy(), s[:, 9].copy()
s = s[:, :8]
for k in (2, 3, 4):
    s[:, k] = s[:, 1]
life_us = (rt1 - rt0) / 100.0
print(f"wave lifetime (100 MHz wall clock): median {np.median(life_us):.2f} us, p10 {np.percentile(life_us, 10):.2f}, p90 {np.percentile(life_us, 90):.2f}; "
      f"launch span first entry -> last end {(rt1.max() - rt0.min()) / 100.0:.2f} us; entries spread p90-p10 {(np.percentile(rt0, 90) - np.percentile(rt0, 10)) / 100.0:.2f} us")
print(f"shader clock: {np.median((s[:, 7] - s[:, 0]) / np.maximum(life_us, 1e-9)) / 1000:.2f} GHz")

t0 = s[:, 0].min()
rel = (s - t0).astype(np.float64)
names = ["entry", "first requests issued", "-", "-", "-", "tables + pair 0 landed, barrier", "pair 0 done", "end"]
print("waves", len(s), "(stamps in shader clocks; 2.2 GHz ~ 2200 per us)")
for i, n in enumerate(names):
    col = rel[:, i]
    print(f"{n:14s} min {col.min():8.0f}  p10 {np.percentile(col, 10):8.0f}  median {np.median(col):8.0f}  p90 {np.percentile(col, 90):8.0f}  max {col.max():8.0f}")
d = np.diff(rel, axis=1)
print("per-wave durations (median): " + ", ".join(f"{names[i]}->{names[i+1]} {np.median(d[:, i]):.0f}" for i in range(7)))

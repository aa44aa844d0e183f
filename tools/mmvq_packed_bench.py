"""Raw-layout vs packed-layout I-quant MMVQ at the Gemma-4-12B shapes: graph-replayed launches over a pool of weight
copies larger than the caches, microseconds and TB/s per launch."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check

vmx.init(0)
TS = {18: 98, 21: 110, 22: 82}
shapes = [(18, 30720, 3840), (21, 3840, 15360), (18, 4096, 3840), (21, 3840, 4096), (22, 2048, 3840), (18, 14336, 4096), (18, 4096, 4096)]
POOL_BYTES = 1200e6
st = torch.cuda.Stream()
for t, n_out, n_in in shapes:
    nbytes = n_out * (n_in // 256) * TS[t]
    copies = max(2, int(POOL_BYTES // nbytes))
    g = torch.Generator(device="cuda").manual_seed(1)
    raw = torch.randint(0, 256, (copies, nbytes), dtype=torch.uint8, device="cuda", generator=g)
    raw.view(copies, -1, TS[t])[:, :, 1] = 0x2C
    packed = [vmx.iq_pack(t, raw[c], n_out, n_in) for c in range(copies)]
    aq = torch.randint(-127, 127, (1, n_in), dtype=torch.int8, device="cuda")
    ad = torch.rand((1, n_in // 32), device="cuda") * 0.01
    y = torch.empty((1, n_out), dtype=torch.float32, device="cuda")
    res = {}
    for name in ("raw", "packed"):
        def run():
            for c in range(copies):
                if name == "raw":
                    vmx.iq_mmvq(t, raw[c], aq, ad, n_out, n_in, out=y)
                else:
                    vmx.iq_mmvq_packed(t, packed[c], aq, ad, n_out, n_in, out=y)
        with torch.cuda.stream(st):
            run(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                run()
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record(st)
            for _ in range(reps):
                gr.replay()
            e1.record(st)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * copies)
        res[name] = us
    alg = nbytes + n_in * 1.125 + n_out * 4
    print(f"type {t} [{n_out} x {n_in}] {nbytes/1e6:.1f} MB: raw {res['raw']:.2f} us ({alg/res['raw']/1e6:.2f} TB/s)  packed {res['packed']:.2f} us ({alg/res['packed']/1e6:.2f} TB/s)", flush=True)
    del raw, packed

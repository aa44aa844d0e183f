"""I-quant MMVQ time per launch: graph of 40 launches over rotating weight copies (> 1 GiB)."""
import sys, time, os
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
vmx.init(0)
IQ = {"iq2_xs": (17, 74), "iq2_s": (22, 82), "iq3_xxs": (18, 98), "iq3_s": (21, 110), "iq4_xs": (23, 136)}
names = os.environ.get("TYPES", "iq3_xxs").split(",")
shapes = [(4096, 4096), (14336, 4096), (30720, 3840), (3840, 15360)]
M = int(os.environ.get("M", "1"))
s = torch.cuda.Stream()
for name in names:
    tid, ts = IQ[name]
    for n_out, n_in in shapes:
        wbytes = n_out * (n_in // 256) * ts
        stride = (wbytes + 255) // 256 * 256
        ncopies = max(2, (1200 << 20) // stride)
        g0 = torch.Generator(device="cuda").manual_seed(1)
        pool = torch.randint(0, 256, (ncopies, stride), dtype=torch.uint8, device="cuda", generator=g0)
        pool[:, :wbytes].view(ncopies, -1, ts)[:, :, 1] = 0x30
        x = torch.randn((M, n_in), device="cuda")
        aq, ad = vmx.quantize_q8_1(x)
        y = torch.empty((M, n_out), device="cuda")
        with torch.cuda.stream(s):
            st = s.cuda_stream
            def chain(n, off=0):
                for i in range(n):
                    check(lib.vmx_iq_mmvq_f32(tid, pool[(i + off) % ncopies].data_ptr(), aq.data_ptr(), ad.data_ptr(), y.data_ptr(), n_out, n_in, M, st))
            chain(2); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph(); n = 40
            with torch.cuda.graph(g, stream=s):
                chain(n, 3)
            g.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): g.replay()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 5 / n * 1e6
        alg = wbytes + M * n_in * (1 + 4 / 32) + M * n_out * 4
        print(f"{name:8s} {n_out:6d}x{n_in:5d} M={M}: {us:6.2f} us  {alg/us/1e3:7.0f} GB/s  {alg/us/1e3/8000:.3f} of 8 TB/s", flush=True)
        del pool

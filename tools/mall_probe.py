"""Per-launch streaming time against the working-set size the launches rotate over: 1 buffer (L2/infinity-cache hot),
~100 MB and ~200 MB (infinity-cache resident, beyond L2) and 1.5 GiB (HBM)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
vmx.init(0)
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
pool = torch.randint(0, 255, (1536 << 20,), dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
for mb in (6.3, 13.4, 26.7, 53.0):
    nbytes = int(mb * (1 << 20)) // 1024 * 1024
    for ws_mb in (0, 100, 200, 1536):
        ncopies = max(1, min(pool.numel(), ws_mb << 20) // nbytes)
        upw = 8
        with torch.cuda.stream(s):
            st = s.cuda_stream
            def chain(n, off=0):
                for i in range(n):
                    base = pool.data_ptr() + ((i + off) % ncopies) * nbytes
                    check(lib.vmx_debug_stream_read(base, nbytes, upw, sink.data_ptr(), st))
            chain(2); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            n = 40
            with torch.cuda.graph(g, stream=s):
                chain(n, 3)
            g.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): g.replay()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 5 / n * 1e6
        print(f"{mb:5.1f} MB  working set {ncopies * nbytes / 2**20:7.1f} MB: {us:6.2f} us/launch  {nbytes/us/1e3:7.0f} GB/s", flush=True)

"""Per-kernel cost of a dependent chain of trivial kernels replayed from one hipGraph."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
vmx.init(0)
z = torch.zeros(8, dtype=torch.int32, device="cuda"); bt = torch.zeros((1, 4096), dtype=torch.int32, device="cuda")
s = torch.cuda.Stream()
def chain(n):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(n):
        check(lib.vmx_decode_advance(z.data_ptr(), z[1:].data_ptr(), z[2:].data_ptr(), z[3:].data_ptr(), z[4:].data_ptr(), bt.data_ptr(), 1, 4096, 16, st))
for n in (100, 400):
    with torch.cuda.stream(s):
        chain(2); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain(n)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    print(f"graph of {n} trivial kernels: {dt*1e6:.1f} us per replay, {dt*1e6/n:.2f} us per kernel")

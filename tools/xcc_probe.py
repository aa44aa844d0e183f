"""Which XCD does workgroup i of a launch run on?  Prints the id sequence head for grids launched back to back."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
vmx.init(0)
st = torch.cuda.current_stream().cuda_stream
grids = [(769, 256), (48, 64), (512, 64), (777, 256), (1537, 256), (3, 256), (64, 256)]
outs = [torch.full((g,), -1, dtype=torch.int32, device="cuda") for g, _ in grids]
for rep in range(2):
    for (g, t), o in zip(grids, outs):
        check(lib.vmx_debug_xcc_ids(o.data_ptr(), g, t, st))
    torch.cuda.synchronize()
    for (g, t), o in zip(grids, outs):
        v = o.cpu().tolist()
        rr = sum(1 for i, x in enumerate(v) if x == (v[0] + i) % 8)
        print(f"rep {rep} grid {g:5d} x{t:3d}: first {v[:20]}  round-robin-from-first matches {rr}/{g}", flush=True)
# under graph replay
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for (gr, t), o in zip(grids, outs):
            check(lib.vmx_debug_xcc_ids(o.data_ptr(), gr, t, s.cuda_stream))
    for rep in range(2):
        g.replay(); torch.cuda.synchronize()
        for (gr, t), o in zip(grids, outs):
            v = o.cpu().tolist()
            rr = sum(1 for i, x in enumerate(v) if x == (v[0] + i) % 8)
            print(f"graph rep {rep} grid {gr:5d}: first {v[:12]}  rr {rr}/{gr}", flush=True)

"""Long-context decode sweep of bench.py on its own.  usage: python tools/ctx_sweep.py 1024,8192 [steps]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
import vllm_rust_amd as vmx
vmx.init(0)
ctxs = tuple(int(c) for c in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1024, 8192)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
print(bench.context_sweep_secondary(steps, 4, contexts=ctxs))

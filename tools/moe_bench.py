import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
import vllm_rust_amd as vmx
vmx.init(0)
t0 = time.time()
print(bench.mixtral_secondary(256, 32, 4), time.time() - t0)

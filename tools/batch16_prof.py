"""Decode steps at B = 16 only, for a kernel trace: which launches grow between B = 8 and B = 16."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
import vllm_rust_amd as vmx
vmx.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
print(bench.batch_sweep_secondary(256, 16, 2, batches=(B,)))

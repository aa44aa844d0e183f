"""Concurrent-stream decode sweep of bench.py on its own (every batch size is printed, whatever it shows)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
import vllm_rust_amd as vmx
vmx.init(0)
bs = tuple(int(b) for b in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 4, 8, 16)
print(bench.batch_sweep_secondary(256, 32, 4, batches=bs))

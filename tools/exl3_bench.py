"""EXL3 linear-path measurement of bench.py on its own."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
import vllm_rust_amd as vmx
vmx.init(0)
print(bench.exl3_secondary())

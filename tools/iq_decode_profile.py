"""Run a few decode steps of the Gemma-4-12B-shaped I-quant model (for rocprofv3)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
import vllm_rust_amd as vmx
vmx.init(0)
print(bench.gemma4_secondary(64, 16, 2))

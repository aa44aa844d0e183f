"""EXL3 measurements of bench.py on their own: whole-model decode and the linear path alone."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
import vllm_rust_amd as vmx
vmx.init(0)
if "--model" in sys.argv:
    print(bench.exl3_decode_secondary(256, 64, 8))
print(bench.exl3_secondary())

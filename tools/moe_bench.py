"""Mixtral-8x7B-shaped block of bench.py on its own; VMX_MOE_ROWS = 0 / 16 / 64 forces one form of the expert layer for an A/B."""
import os, sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, bench
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib
vmx.init(0)
if os.environ.get("VMX_MOE_ROWS"):
    lib.vmx_debug_moe_block_rows(int(os.environ["VMX_MOE_ROWS"]))
bs = tuple(int(b) for b in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 4, 8, 16)
t0 = time.time()
print(bench.mixtral_secondary(256, 32, 4, batches=bs), time.time() - t0)

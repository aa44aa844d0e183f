"""Mixtral-8x7B-shaped decode at a few batch sizes with the expert GEMM form forced: usage moe_rows_ab.py ROWS B[,B...]  (ROWS -1 = by size, 0 = per pair, 16 = 16-row blocks)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
vmx.init(0)
rows = int(sys.argv[1])
bs = tuple(int(b) for b in sys.argv[2].split(","))
check(lib.vmx_debug_moe_block_rows(rows))
r = bench.mixtral_secondary(64, 32, 4, batches=bs)
print("rows", rows, {k: v["tok_s"] for k, v in r["batch_sweep"].items()}, flush=True)

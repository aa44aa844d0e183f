"""The Gemma-4-12B-shaped secondary of bench.py alone (decode tok/s + MMVQ roofline), for A/B runs of the I-quant kernels."""
import json
import sys
sys.path.insert(0, ".")
import torch
import vllm_rust_amd as vmx
import bench
vmx.init(0)
print(json.dumps(bench.gemma4_secondary(256, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 4)))

"""Short program for PMC passes: 4 Qwen3-8B-shaped layers, the step's int4 linear launches alone, 3 times.  usage: pmc_linears.py [rows]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
vmx.init(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
spec = Qwen3Spec.qwen3_8b(num_blocks=8, max_seq_len=64, max_position=128, max_batch=rows)
spec.num_layers = 4
spec.vocab_size = 4096
m = QuantizedQwen3ForCausalLM(spec)
m.load_synthetic(seed=1)
for _ in range(3):
    m.linears_only(rows)
torch.cuda.synchronize()
print(m.linears_info(rows))

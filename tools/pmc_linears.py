"""Short program for PMC passes: 4 Qwen3-8B-shaped layers, the step's int4 linear launches alone, 3 times."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
vmx.init(0)
spec = Qwen3Spec.qwen3_8b(num_blocks=8, max_seq_len=64, max_position=128)
spec.num_layers = 4
spec.vocab_size = 4096
m = QuantizedQwen3ForCausalLM(spec)
m.load_synthetic(seed=1)
for _ in range(3):
    m.linears_only(1)
torch.cuda.synchronize()
print(m.linears_info(1))

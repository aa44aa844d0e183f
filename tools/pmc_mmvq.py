"""Short program for PMC passes: 4 Gemma-4-12B-shaped layers (3 sliding + 1 full), the step's MMVQ launches alone, 3 times."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM
vmx.init(0)
spec = Gemma4Spec(3840, 4, 16, 15360, 4096, full_every=4, num_blocks=8, max_seq_len=64, max_position=128)
m = QuantizedGemma4ForCausalLM(spec)
m.load_synthetic(seed=1)
for _ in range(3):
    m.mmvq_only(1)
torch.cuda.synchronize()
print(m.mmvq_info(1))

"""Prompt path at a longer prompt for PMC passes / traces: Qwen3-8B-shaped layers (4 of them), one P-token vmx_qwen3_prefill call, three times."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
vmx.init(0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
spec = Qwen3Spec.qwen3_8b(num_blocks=P // 16 + 2, max_seq_len=P + 16, max_position=P + 32, max_prefill_tokens=P)
spec.num_layers = 4
spec.vocab_size = 4096
m = QuantizedQwen3ForCausalLM(spec)
m.load_synthetic(seed=1)
bt = torch.arange(P // 16 + 2, dtype=torch.int32, device="cuda")
ids = torch.randint(0, spec.vocab_size, (P,), dtype=torch.int32, device="cuda")
for _ in range(3):
    m.prefill(ids, 0, bt)
torch.cuda.synchronize()

"""Prompt pass of a small dense Qwen3-0.6B-shaped model; prints logits statistics (run once per kernel variant and compare)."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
vmx.init(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
P = int(sys.argv[2]) if len(sys.argv) > 2 else 12
spec = Qwen3Spec(1024, L, 16, 8, 128, 3072, 8192, tie_word_embeddings=True, max_position=256, num_blocks=8, max_batch=1, max_seq_len=128,
                 max_prefill_tokens=64)
m = QuantizedQwen3ForCausalLM(spec)
m.load_synthetic_bf16(seed=4)
prompt = torch.from_numpy(np.random.default_rng(8).integers(0, spec.vocab_size, P).astype(np.int32)).cuda()
bt = torch.arange(8, dtype=torch.int32, device="cuda").flip(0).contiguous()
lg = torch.empty(spec.vocab_size, dtype=torch.bfloat16, device="cuda")
out = m.prefill(prompt, 0, bt, logits_last=lg)
torch.cuda.synchronize()
f = lg.float().cpu().numpy()
print("id", int(out[0]), "sum", float(f.sum()), "abs", float(np.abs(f).sum()), "first", f[:6].tolist())
np.save(sys.argv[3], f) if len(sys.argv) > 3 else None

"""Time vmx_gemma4_prefill on the Gemma-4-12B shape for several pass sizes (16 = q8_1 MMVQ passes; more = dequant + GEMM passes)."""
import sys
import time
import torch
sys.path.insert(0, ".")
import vllm_rust_amd as vmx
from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM
vmx.init(0)
prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 48
for mp in (0, 64, 256):
    n_blocks = (prompt + 32) // 16 + 1
    spec = Gemma4Spec.gemma4_12b(sliding_window=1024, max_position=4096, kv_cache_dtype=1, num_blocks=n_blocks, max_batch=1,
                                 max_seq_len=prompt + 32, embed_ggml_type=11, final_logit_softcap=30.0, max_prefill_tokens=mp)
    spec.num_layers = layers
    model = QuantizedGemma4ForCausalLM(spec)
    model.load_synthetic(seed=0)
    bt = torch.arange(n_blocks, dtype=torch.int32, device="cuda")
    ids = torch.randint(0, spec.vocab_size, (prompt,), dtype=torch.int32, device="cuda")
    model.prefill(ids[:max(mp, 16)], 0, bt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.prefill(ids, 0, bt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"pass {mp or 16}: {prompt} tokens in {dt * 1e3:.1f} ms = {prompt / dt:.0f} tok/s", flush=True)
    model.close()
    del model
    torch.cuda.empty_cache()

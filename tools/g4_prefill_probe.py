"""Probe: Gemma prefill vs token-by-token oracle for several (prompt, window) pairs."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import vllm_rust_amd as vmx
from oracle import oracle as orc
from tests.model_oracle import OracleGemma4
from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM
vmx.init(0)
for P, win, layers in ((5, 64, 3), (16, 64, 3), (17, 64, 3), (37, 64, 3), (37, 8, 3), (37, 8, 2)):
    spec = Gemma4Spec(256, layers, 4, 512, 1000, head_dim_sliding=64, head_dim_full=128, kv_heads_sliding=2, kv_heads_full=1,
                      sliding_window=win, full_every=3, max_position=128, kv_cache_dtype=0, num_blocks=8, max_batch=1, max_seq_len=96)
    model = QuantizedGemma4ForCausalLM(spec)
    ck = model.load_synthetic(seed=5, keep_host_copy=True)
    from vllm_rust_amd.gemma4 import T_Q_NORM, T_K_NORM
    gq = torch.Generator(device="cuda").manual_seed(9)
    for li in range(layers):   # logits of std ~1 instead of ~8: softmax no longer near one-hot
        hd = spec.head_dim_full if (li + 1) % 3 == 0 else spec.head_dim_sliding
        for which, name in ((T_Q_NORM, "q_norm"), (T_K_NORM, "k_norm")):
            w = ((1.0 + 0.1 * torch.randn(hd, device="cuda", generator=gq)) * (1.0 / (hd ** 0.25))).to(torch.bfloat16)
            model.load_tensor(which, li, w)
            ck[f"{li}.{name}"] = w.cpu()
    ref = OracleGemma4(spec, ck, model.cos_sin_sliding, model.cos_sin_full)
    bs = spec.block_size
    bt = np.asarray([[5, 2, 7, 0, 3, 1]], np.int32)
    prompt = np.random.default_rng(1).integers(0, spec.vocab_size, P).astype(np.int32)
    slot = lambda p: np.asarray([bt[0, p // bs] * bs + p % bs], np.int32)
    lg = None
    for p in range(P):
        lg, h_exp, _ = ref.decode(prompt[p:p + 1], np.asarray([p], np.int32), slot(p), bt, np.asarray([p + 1], np.int32))
    exp = orc.bf16_to_f32(lg)[0]
    out = torch.empty(spec.vocab_size, dtype=torch.bfloat16, device="cuda")
    model.prefill(torch.from_numpy(prompt).cuda(), 0, torch.from_numpy(bt).cuda()[0], logits_last=out)
    torch.cuda.synchronize()
    got = orc.bf16_to_f32(out.cpu().view(torch.int16).numpy().view(np.uint16))
    print(P, win, layers, "rel", float(np.sqrt(((got - exp) ** 2).sum() / (exp ** 2).sum())))
    model.close()

"""A real-format AWQ checkpoint directory (config.json with quantization_config, model.safetensors with the HF tensor names)
loaded through libvmx's detection + safetensors reader gives the same executor as loading the same tensors directly."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_awq_safetensors_directory_round_trip(tmp_path):
    import torch
    from safetensors.torch import save_file
    import vllm_rust_amd as vmx
    from vllm_rust_amd import checkpoint as ckp
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
    vmx.init(0)
    spec = Qwen3Spec(512, 2, 4, 2, 128, 1024, 2048, max_position=128, num_blocks=8, max_batch=2, max_seq_len=64)
    a = QuantizedQwen3ForCausalLM(spec)
    b = QuantizedQwen3ForCausalLM(spec)
    try:
        ck = a.load_synthetic(seed=3, keep_host_copy=True)
        proj = ckp._PROJ
        out = {"model.embed_tokens.weight": ck["embed"], "lm_head.weight": ck["lm_head"], "model.norm.weight": ck["final_norm"]}
        for li in range(spec.num_layers):
            pre = f"model.layers.{li}."
            out[pre + "input_layernorm.weight"] = ck[f"{li}.input_ln"]
            out[pre + "post_attention_layernorm.weight"] = ck[f"{li}.post_ln"]
            out[pre + "self_attn.q_norm.weight"] = ck[f"{li}.q_norm"]
            out[pre + "self_attn.k_norm.weight"] = ck[f"{li}.k_norm"]
            for slot, p in enumerate(proj):
                for n in ("qweight", "qzeros", "scales"):
                    out[f"{pre}{p}.{n}"] = ck[f"{li}.{slot}.{n}"]
        save_file({k: v.contiguous() for k, v in out.items()}, str(tmp_path / "model.safetensors"), metadata={"format": "pt"})
        (tmp_path / "config.json").write_text(json.dumps({"model_type": "qwen3", "quantization_config": {
            "quant_method": "awq", "bits": 4, "group_size": spec.group_size, "zero_point": True, "version": "gemm"}}))
        det = ckp.load_qwen3_awq_checkpoint(b, tmp_path)
        assert (det.method, det.bits, det.group_size) == ("awq", 4, spec.group_size)
        ids = torch.tensor([5, 77], dtype=torch.int32, device="cuda")
        pos = torch.tensor([0, 0], dtype=torch.int32, device="cuda")
        bt = torch.tensor([[0, 1], [2, 3]], dtype=torch.int32, device="cuda")
        slots = torch.tensor([0, 32], dtype=torch.int32, device="cuda")
        sl = torch.tensor([1, 1], dtype=torch.int32, device="cuda")
        la = torch.empty((2, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
        lb = torch.empty_like(la)
        a.forward_decode_batch(ids, pos, slots, bt, sl, logits=la)
        b.forward_decode_batch(ids, pos, slots, bt, sl, logits=lb)
        torch.cuda.synchronize()
        assert torch.equal(la.view(torch.int16), lb.view(torch.int16))
        # a GPTQ directory is refused by this loader with the detected method in the message
        (tmp_path / "config.json").write_text(json.dumps({"quantization_config": {"quant_method": "gptq", "bits": 4, "group_size": 128}}))
        with pytest.raises(ValueError, match="gptq"):
            ckp.load_qwen3_awq_checkpoint(b, tmp_path)
    finally:
        a.close()
        b.close()


def test_seeded_tp_checkpoint_at_world_1_is_the_single_gpu_checkpoint():
    """load_synthetic_tp draws the full seeded checkpoint and keeps this rank's slice; with one rank the slice is everything, so
    the model must be bit-identical to load_synthetic with the same seed (the property that makes a TP run checkable)."""
    import torch
    import vllm_rust_amd as vmx
    from vllm_rust_amd.parallel import shard_spec
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
    vmx.init(0)
    spec = Qwen3Spec(512, 2, 4, 2, 128, 1024, 2048, max_position=128, num_blocks=8, max_batch=1, max_seq_len=64)
    a = QuantizedQwen3ForCausalLM(spec)
    b = QuantizedQwen3ForCausalLM(shard_spec(spec, 0, 1))
    try:
        a.load_synthetic(seed=5)
        b.load_synthetic_tp(spec, seed=5)
        args = [torch.tensor(v, dtype=torch.int32, device="cuda") for v in ([9], [0], [0], [[0, 1]], [1])]
        la = torch.empty((1, spec.vocab_size), dtype=torch.bfloat16, device="cuda")
        lb = torch.empty_like(la)
        a.forward_decode_batch(*args, logits=la)
        b.forward_decode_batch(*args, logits=lb)
        torch.cuda.synchronize()
        assert torch.equal(la.view(torch.int16), lb.view(torch.int16))
    finally:
        a.close()
        b.close()

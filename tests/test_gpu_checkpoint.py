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


def _decode_logits(torch, model, spec):
    ids = torch.tensor([5, 77], dtype=torch.int32, device="cuda")
    pos = torch.tensor([0, 0], dtype=torch.int32, device="cuda")
    bt = torch.tensor([[0, 1], [2, 3]], dtype=torch.int32, device="cuda")
    slots = torch.tensor([0, 32], dtype=torch.int32, device="cuda")
    sl = torch.tensor([1, 1], dtype=torch.int32, device="cuda")
    out = torch.empty((2, spec.vocab_size), dtype=getattr(model, "act_torch_dtype", torch.bfloat16), device="cuda")
    model.forward_decode_batch(ids, pos, slots, bt, sl, logits=out)
    torch.cuda.synchronize()
    return out


def test_gptq_safetensors_directory_matches_the_same_weights_in_awq_form(tmp_path):
    """A GPTQ directory (qweight [in/8, out], qzeros [in/group, out/8], scales; sym false so the zeros are used as stored) built from
    the int4 values of a seeded AWQ checkpoint loads into an executor that is bit-identical to the AWQ one: both forms hold the
    same (nib - z) * s.  With "sym": true the zeros are ignored (zero point 8, the Marlin form): checked against AWQ zeros all 8."""
    import torch
    from safetensors.torch import save_file
    import vllm_rust_amd as vmx
    from vllm_rust_amd import checkpoint as ckp
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
    vmx.init(0)
    spec = Qwen3Spec(512, 2, 4, 2, 128, 1024, 2048, max_position=128, num_blocks=8, max_batch=2, max_seq_len=64)
    order = [0, 4, 1, 5, 2, 6, 3, 7]   # AWQ: output n0..n7 of a word sit at these nibble positions (awq.rs:20-38)

    def awq_to_nibbles(w):   # int32 [R, C/8] -> uint8 [R, C]
        u = w.numpy().view(np.uint32)
        out = np.zeros((u.shape[0], u.shape[1] * 8), np.uint8)
        for i in range(8):
            out[:, i::8] = (u >> np.uint32(4 * order[i])) & 0xF
        return out

    def pack_seq(nib, axis):   # eight consecutive values along `axis` per word, value j at bit 4j (GPTQ)
        nib = nib.astype(np.uint32)
        if axis == 0:
            w = np.zeros((nib.shape[0] // 8, nib.shape[1]), np.uint32)
            for j in range(8):
                w |= nib[j::8, :] << np.uint32(4 * j)
        else:
            w = np.zeros((nib.shape[0], nib.shape[1] // 8), np.uint32)
            for j in range(8):
                w |= nib[:, j::8] << np.uint32(4 * j)
        return torch.from_numpy(w.view(np.int32))

    for sym in (False, True):
        a = QuantizedQwen3ForCausalLM(spec)
        b = QuantizedQwen3ForCausalLM(spec)
        try:
            ck = a.load_synthetic(seed=4, keep_host_copy=True)
            out = {"model.embed_tokens.weight": ck["embed"], "lm_head.weight": ck["lm_head"], "model.norm.weight": ck["final_norm"]}
            for li in range(spec.num_layers):
                pre = f"model.layers.{li}."
                out[pre + "input_layernorm.weight"] = ck[f"{li}.input_ln"]
                out[pre + "post_attention_layernorm.weight"] = ck[f"{li}.post_ln"]
                out[pre + "self_attn.q_norm.weight"] = ck[f"{li}.q_norm"]
                out[pre + "self_attn.k_norm.weight"] = ck[f"{li}.k_norm"]
                for slot, p in enumerate(ckp._PROJ):
                    wn = awq_to_nibbles(ck[f"{li}.{slot}.qweight"])
                    zn = awq_to_nibbles(ck[f"{li}.{slot}.qzeros"])
                    if sym:   # reload the AWQ model with zero point 8 everywhere: what a symmetric GPTQ checkpoint means
                        zn[:] = 8
                        z8 = torch.full_like(ck[f"{li}.{slot}.qzeros"], int(np.uint32(0x88888888).view(np.int32)))
                        a.load_awq_linear(li, slot, ck[f"{li}.{slot}.qweight"].cuda(), z8.cuda(), ck[f"{li}.{slot}.scales"].cuda())
                    out[f"{pre}{p}.qweight"] = pack_seq(wn, 0)
                    out[f"{pre}{p}.qzeros"] = pack_seq(zn if not sym else np.full_like(zn, 7), 1)   # a v1 file stores zero - 1: must be ignored
                    out[f"{pre}{p}.scales"] = ck[f"{li}.{slot}.scales"]
                    out[f"{pre}{p}.g_idx"] = torch.arange(wn.shape[0], dtype=torch.int32) // spec.group_size
            d = tmp_path / ("sym" if sym else "asym")
            d.mkdir()
            save_file({k: v.contiguous() for k, v in out.items()}, str(d / "model.safetensors"), metadata={"format": "pt"})
            (d / "config.json").write_text(json.dumps({"model_type": "qwen3", "quantization_config": {
                "quant_method": "gptq", "bits": 4, "group_size": spec.group_size, "desc_act": False, "sym": sym}}))
            det = ckp.load_qwen3_gptq_checkpoint(b, d)
            assert det.method in ("gptq", "gptq_marlin") and det.bits == 4
            la, lb = _decode_logits(torch, a, spec), _decode_logits(torch, b, spec)
            assert torch.equal(la.view(torch.int16), lb.view(torch.int16))
            if not sym:
                (d / "config.json").write_text(json.dumps({"quantization_config": {"quant_method": "gptq", "bits": 4, "group_size": 128, "desc_act": True}}))
                with pytest.raises(ValueError, match="desc_act"):
                    ckp.load_qwen3_gptq_checkpoint(b, d)
        finally:
            a.close()
            b.close()


@pytest.mark.parametrize("act_dtype", ["bf16", "f16"])
def test_exl3_safetensors_directory_round_trip(tmp_path, act_dtype):
    """An ExLlamaV3-format directory (trellis / suh / svh per projection, the mcg flag tensor for the second codebook, fp16 backbone)
    loads into the same executor as the same tensors handed over directly -- a bf16 model converts the fp16 backbone, a model
    that runs in fp16 takes it as stored."""
    import torch
    from safetensors.torch import save_file
    import vllm_rust_amd as vmx
    from vllm_rust_amd import checkpoint as ckp
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
    vmx.init(0)
    spec = Qwen3Spec(512, 2, 4, 2, 128, 1024, 2048, max_position=128, num_blocks=8, max_batch=2, max_seq_len=64, act_dtype=act_dtype)
    a = QuantizedQwen3ForCausalLM(spec)
    b = QuantizedQwen3ForCausalLM(spec)
    adt = a.act_torch_dtype
    try:
        ck = a.load_synthetic_exl3(bits=3, codebook=1, seed=6, keep_host_copy=True)
        # fp16 backbone, as published EXL3 checkpoints have it: give model `a` the same fp16-rounded values
        f16 = lambda t: t.to(torch.float16)
        from vllm_rust_amd.qwen3 import T_EMBED, T_FINAL_NORM, T_INPUT_LN, T_K_NORM, T_LM_HEAD, T_POST_LN, T_Q_NORM
        out = {"model.embed_tokens.weight": f16(ck["embed"]), "lm_head.weight": f16(ck["lm_head"]), "model.norm.weight": f16(ck["final_norm"])}
        a.load_tensor(T_EMBED, 0, f16(ck["embed"]).to(adt).cuda())
        a.load_tensor(T_LM_HEAD, 0, f16(ck["lm_head"]).to(adt).cuda())
        a.load_tensor(T_FINAL_NORM, 0, f16(ck["final_norm"]).to(adt).cuda())
        for li in range(spec.num_layers):
            pre = f"model.layers.{li}."
            for name, key, tid in (("input_layernorm", "input_ln", T_INPUT_LN), ("post_attention_layernorm", "post_ln", T_POST_LN),
                                   ("self_attn.q_norm", "q_norm", T_Q_NORM), ("self_attn.k_norm", "k_norm", T_K_NORM)):
                out[f"{pre}{name}.weight"] = f16(ck[f"{li}.{key}"])
                a.load_tensor(tid, li, f16(ck[f"{li}.{key}"]).to(adt).cuda())
            for slot, p in enumerate(ckp._PROJ):
                bits, cb, tr, suh, svh = ck[f"{li}.{slot}.exl3"]
                out[f"{pre}{p}.trellis"], out[f"{pre}{p}.suh"], out[f"{pre}{p}.svh"] = tr, suh, svh
                out[f"{pre}{p}.mcg"] = torch.tensor([1], dtype=torch.int32)
        save_file({k: v.contiguous() for k, v in out.items()}, str(tmp_path / "model.safetensors"), metadata={"format": "pt"})
        (tmp_path / "config.json").write_text(json.dumps({"model_type": "qwen3", "quantization_config": {"quant_method": "exl3", "bits": 3.0}}))
        det = ckp.load_qwen3_exl3_checkpoint(b, tmp_path)
        assert det.method == "exl3"
        la, lb = _decode_logits(torch, a, spec), _decode_logits(torch, b, spec)
        assert torch.equal(la.view(torch.int16), lb.view(torch.int16))
    finally:
        a.close()
        b.close()

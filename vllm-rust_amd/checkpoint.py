"""Real checkpoints for the executors: quantization detection (the reference's quantization/detection.rs), the safetensors
container, and the HF-name loaders of the AWQ / GPTQ int4 format (weight_loader.rs:597-714) -- all parsing is done by
libvmx's host code; this file only moves bytes from the file to the executor's load calls."""
import ctypes as C
import json
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from ._lib import check, lib

QM_NAMES = {0: "none", 1: "awq", 2: "gptq", 3: "gptq_marlin", 4: "awq_marlin", 5: "exl3", 6: "fp8", 7: "gguf", 8: "other"}


class _QuantInfo(C.Structure):
    _fields_ = [("method", C.c_int), ("bits", C.c_int), ("group_size", C.c_int), ("desc_act", C.c_int), ("method_name", C.c_char * 32),
                ("activation_scheme", C.c_char * 32)]


@dataclass
class DetectedQuantConfig:
    """DetectedQuantConfig of the reference (detection.rs:15-23); absent fields are None."""
    method: str
    method_name: str
    bits: int | None
    group_size: int | None
    desc_act: bool | None
    activation_scheme: str | None


def _info(q: _QuantInfo) -> DetectedQuantConfig:
    return DetectedQuantConfig(QM_NAMES[q.method], q.method_name.decode(), None if q.bits < 0 else q.bits, None if q.group_size < 0 else q.group_size,
                               None if q.desc_act < 0 else bool(q.desc_act), q.activation_scheme.decode() or None)


def detect_from_json(config_json: str | None, quantize_config_json: str | None = None) -> DetectedQuantConfig:
    q = _QuantInfo()
    check(lib.vmx_quant_detect_json(None if config_json is None else config_json.encode(),
                                    None if quantize_config_json is None else quantize_config_json.encode(), C.byref(q)))
    return _info(q)


def detect_from_directory(model_dir) -> DetectedQuantConfig:
    q = _QuantInfo()
    check(lib.vmx_quant_detect_dir(str(model_dir).encode(), C.byref(q)))
    return _info(q)


class _StTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_char_p), ("element_bytes", C.c_int), ("n_dims", C.c_int), ("shape", C.c_int64 * 8),
                ("file_offset", C.c_uint64), ("n_bytes", C.c_uint64)]


_NP = {"BOOL": np.bool_, "U8": np.uint8, "I8": np.int8, "F8_E4M3": np.uint8, "F8_E5M2": np.uint8, "I16": np.int16, "U16": np.uint16,
       "F16": np.float16, "BF16": np.uint16, "I32": np.int32, "U32": np.uint32, "F32": np.float32, "F64": np.float64, "I64": np.int64,
       "U64": np.uint64}


class SafeTensorsFile:
    """One .safetensors file, parsed and bounds-checked by libvmx (vmx_safetensors_open)."""

    def __init__(self, path):
        self._h = C.c_void_p()
        check(lib.vmx_safetensors_open(str(path).encode(), C.byref(self._h)))
        self.names = []
        for i in range(lib.vmx_safetensors_tensor_count(self._h)):
            self.names.append(self.info(i)[0])

    def close(self):
        if self._h:
            lib.vmx_safetensors_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        self.close()

    def info(self, index: int):
        t = _StTensor()
        check(lib.vmx_safetensors_tensor_info(self._h, index, C.byref(t)))
        return t.name.decode(), t.dtype.decode(), tuple(t.shape[i] for i in range(t.n_dims)), int(t.file_offset), int(t.n_bytes)

    def __contains__(self, name: str) -> bool:
        return lib.vmx_safetensors_find(self._h, name.encode()) >= 0

    def get(self, name: str) -> np.ndarray:
        """The tensor as a numpy array (BF16 and FP8 come back as raw uint16 / uint8 bit patterns)."""
        i = lib.vmx_safetensors_find(self._h, name.encode())
        if i < 0:
            raise KeyError(name)
        _, dtype, shape, _, nb = self.info(i)
        a = np.empty(shape, dtype=_NP[dtype])
        assert a.nbytes == nb
        check(lib.vmx_safetensors_read_tensor(self._h, i, a.ctypes.data, a.nbytes))
        return a

    def metadata(self, key: str):
        v = lib.vmx_safetensors_metadata(self._h, key.encode())
        return None if v is None else v.decode()


class CheckpointDir:
    """All *.safetensors files of a model directory behind one name lookup (model.safetensors.index.json is honoured when
    present, otherwise every file is scanned)."""

    def __init__(self, model_dir):
        self.dir = Path(model_dir)
        self.files = {}
        self.where = {}
        idx = self.dir / "model.safetensors.index.json"
        if idx.exists():
            for name, fn in json.loads(idx.read_text())["weight_map"].items():
                self.where[name] = fn
        else:
            for p in sorted(self.dir.glob("*.safetensors")):
                f = self._file(p.name)
                for n in f.names:
                    self.where[n] = p.name

    def _file(self, fn):
        if fn not in self.files:
            self.files[fn] = SafeTensorsFile(self.dir / fn)
        return self.files[fn]

    def __contains__(self, name):
        return name in self.where

    def get(self, name) -> np.ndarray:
        return self._file(self.where[name]).get(name)

    def close(self):
        for f in self.files.values():
            f.close()


_PROJ = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


def load_qwen3_awq_checkpoint(model, model_dir):
    """Fill a QuantizedQwen3ForCausalLM from an AutoAWQ-format directory: detection must say awq (4 bits), the group size must
    match the model's spec; tensors are read under the HF names "model.layers.N.<proj>.{qweight,qzeros,scales}" (qweight i32
    [in, out/8], qzeros i32 [in/group, out/8], scales f16 [in/group, out]) plus the bf16 norms, embedding and lm_head."""
    import torch
    from .qwen3 import T_EMBED, T_FINAL_NORM, T_INPUT_LN, T_K_NORM, T_LM_HEAD, T_POST_LN, T_Q_NORM
    det = detect_from_directory(model_dir)
    if det.method not in ("awq", "awq_marlin") or det.bits not in (None, 4):
        raise ValueError(f"{model_dir}: detected {det.method} ({det.bits} bits); this loader takes 4-bit AWQ checkpoints")
    s = model.spec
    if det.group_size is not None and det.group_size != s.group_size:
        raise ValueError(f"{model_dir}: group size {det.group_size}, model built for {s.group_size}")
    ck = CheckpointDir(model_dir)

    def bf16(name):
        a = ck.get(name)
        return torch.from_numpy(a.view(np.int16)).view(torch.bfloat16)

    try:
        model.load_tensor(T_EMBED, 0, bf16("model.embed_tokens.weight").cuda())
        if not s.tie_word_embeddings:
            model.load_tensor(T_LM_HEAD, 0, bf16("lm_head.weight").cuda())
        model.load_tensor(T_FINAL_NORM, 0, bf16("model.norm.weight").cuda())
        for li in range(s.num_layers):
            pre = f"model.layers.{li}."
            model.load_tensor(T_INPUT_LN, li, bf16(pre + "input_layernorm.weight").cuda())
            model.load_tensor(T_POST_LN, li, bf16(pre + "post_attention_layernorm.weight").cuda())
            if s.qk_norm:
                model.load_tensor(T_Q_NORM, li, bf16(pre + "self_attn.q_norm.weight").cuda())
                model.load_tensor(T_K_NORM, li, bf16(pre + "self_attn.k_norm.weight").cuda())
            for slot, proj in enumerate(_PROJ):
                qw, qz, sc = (ck.get(f"{pre}{proj}.{n}") for n in ("qweight", "qzeros", "scales"))
                model.load_awq_linear(li, slot, torch.from_numpy(qw).cuda(), torch.from_numpy(qz).cuda(),
                                      torch.from_numpy(sc.view(np.int16)).view(torch.float16).cuda())
    finally:
        ck.close()
    return det


def _backbone_bf16(ck, name, want=None):
    """A norm / embedding / lm_head tensor in the model's 16-bit dtype `want` (torch.bfloat16 by default): tensors stored in
    that dtype go in as they are (the fp16 backbone of an EXL3 checkpoint into a model that runs in fp16), others are converted."""
    import torch
    want = want or torch.bfloat16
    a = ck.get(name)
    f = ck._file(ck.where[name])
    dtype = f.info(lib.vmx_safetensors_find(f._h, name.encode()))[1]
    if dtype == "BF16":
        return torch.from_numpy(a.view(np.int16)).view(torch.bfloat16).to(want)
    if dtype == "F16":
        return torch.from_numpy(a.view(np.int16)).view(torch.float16).to(want)
    if dtype == "F32":
        return torch.from_numpy(a).to(want)
    raise ValueError(f"{name}: backbone tensor of dtype {dtype}")


def _load_backbone(model, ck):
    from .qwen3 import T_EMBED, T_FINAL_NORM, T_INPUT_LN, T_K_NORM, T_LM_HEAD, T_POST_LN, T_Q_NORM
    s = model.spec
    want = getattr(model, "act_torch_dtype", None)
    model.load_tensor(T_EMBED, 0, _backbone_bf16(ck, "model.embed_tokens.weight", want).cuda())
    if not s.tie_word_embeddings:
        model.load_tensor(T_LM_HEAD, 0, _backbone_bf16(ck, "lm_head.weight", want).cuda())
    model.load_tensor(T_FINAL_NORM, 0, _backbone_bf16(ck, "model.norm.weight", want).cuda())
    for li in range(s.num_layers):
        pre = f"model.layers.{li}."
        model.load_tensor(T_INPUT_LN, li, _backbone_bf16(ck, pre + "input_layernorm.weight", want).cuda())
        model.load_tensor(T_POST_LN, li, _backbone_bf16(ck, pre + "post_attention_layernorm.weight", want).cuda())
        if s.qk_norm:
            model.load_tensor(T_Q_NORM, li, _backbone_bf16(ck, pre + "self_attn.q_norm.weight", want).cuda())
            model.load_tensor(T_K_NORM, li, _backbone_bf16(ck, pre + "self_attn.k_norm.weight", want).cuda())


def load_qwen3_gptq_checkpoint(model, model_dir, use_zeros=None):
    """Fill a QuantizedQwen3ForCausalLM from an AutoGPTQ-format directory (GptqWeightLoader, quantization/weight_loader.rs:
    tensors "model.layers.N.<proj>.{qweight,qzeros,scales,g_idx}": qweight i32 [in/8, out], qzeros i32 [in/group, out/8],
    scales f16 [in/group, out]).  Detection must say gptq / gptq_marlin at 4 bits with the model's group size; desc_act
    checkpoints (activation-ordered groups) are refused: like the reference's kernels this path takes group = k / group_size.
    use_zeros None: symmetric checkpoints ("sym": true, the Marlin form (nib - 8) * s) ignore qzeros, others use them as stored."""
    import torch
    det = detect_from_directory(model_dir)
    if det.method not in ("gptq", "gptq_marlin") or det.bits not in (None, 4):
        raise ValueError(f"{model_dir}: detected {det.method} ({det.bits} bits); this loader takes 4-bit GPTQ checkpoints")
    if det.desc_act:
        raise ValueError(f"{model_dir}: desc_act (activation-ordered) GPTQ checkpoints are not supported")
    s = model.spec
    if det.group_size is not None and det.group_size != s.group_size:
        raise ValueError(f"{model_dir}: group size {det.group_size}, model built for {s.group_size}")
    if use_zeros is None:
        sym = True
        for fn in ("quantize_config.json", "config.json"):
            p = Path(model_dir) / fn
            if p.exists():
                j = json.loads(p.read_text())
                j = j.get("quantization_config", j)
                if "sym" in j:
                    sym = bool(j["sym"])
                    break
        use_zeros = not sym
    ck = CheckpointDir(model_dir)
    try:
        _load_backbone(model, ck)
        for li in range(s.num_layers):
            pre = f"model.layers.{li}."
            for slot, proj in enumerate(_PROJ):
                qw, sc = ck.get(f"{pre}{proj}.qweight"), ck.get(f"{pre}{proj}.scales")
                qz = torch.from_numpy(ck.get(f"{pre}{proj}.qzeros")).cuda() if use_zeros else None
                model.load_gptq_linear(li, slot, torch.from_numpy(qw).cuda(), qz, torch.from_numpy(sc.view(np.int16)).view(torch.float16).cuda())
    finally:
        ck.close()
    return det


def load_qwen3_exl3_checkpoint(model, model_dir):
    """Fill a QuantizedQwen3ForCausalLM from an ExLlamaV3 directory (Exl3WeightLoader::load_linear, quantization/
    weight_loader.rs:470-559): per projection "<prefix>.trellis" i16 [in/16, out/16, 16 * bpw] (bits per weight from the last
    dimension, 2..8, may differ between tensors), ".suh" f16 [in], ".svh" f16 [out], and the codebook from the presence of the
    flag tensors ".mcg" / ".mul1" (exl3.rs:90-152).  Backbone tensors of such checkpoints are fp16: a model built with
    act_dtype "f16" takes them as stored and runs in fp16 end to end, as the reference does (weight_loader.rs:453-461); a
    bf16 model converts them on the way in."""
    import torch
    det = detect_from_directory(model_dir)
    if det.method != "exl3":
        raise ValueError(f"{model_dir}: detected {det.method}; this loader takes EXL3 checkpoints")
    s = model.spec
    shapes = model.linear_shapes()
    ck = CheckpointDir(model_dir)
    try:
        _load_backbone(model, ck)
        for li in range(s.num_layers):
            pre = f"model.layers.{li}."
            for slot, proj in enumerate(_PROJ):
                K, N = shapes[slot]
                tr = ck.get(f"{pre}{proj}.trellis")
                if tr.ndim != 3:
                    raise ValueError(f"exl3 loader: `{pre}{proj}.trellis` rank must be 3, got dims {tr.shape}")
                if tr.shape[0] != K // 16 or tr.shape[1] != N // 16:
                    raise ValueError(f"exl3 loader: `{pre}{proj}.trellis` shape {tr.shape} doesn't match (in_features/16, out_features/16, *) "
                                     f"= ({K // 16}, {N // 16}, *)")
                if tr.shape[2] == 0 or tr.shape[2] % 16 or not 2 <= tr.shape[2] // 16 <= 8:
                    raise ValueError(f"exl3 loader: `{pre}{proj}.trellis` last dim {tr.shape[2]} is not 16 * bpw with bpw in 2..8")
                codebook = 1 if f"{pre}{proj}.mcg" in ck else (2 if f"{pre}{proj}.mul1" in ck else 0)
                suh, svh = ck.get(f"{pre}{proj}.suh"), ck.get(f"{pre}{proj}.svh")
                model.load_exl3_linear(li, slot, torch.from_numpy(tr).cuda(), torch.from_numpy(suh.view(np.int16)).view(torch.float16).cuda(),
                                       torch.from_numpy(svh.view(np.int16)).view(torch.float16).cuda(), codebook)
    finally:
        ck.close()
    return det

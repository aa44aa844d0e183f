"""Host-side checkpoint code of libvmx (no GPU): quantization detection against the cases the reference's own detection tests
assert (crates/core/src/quantization/detection.rs:316-560: fp8 / gptq / awq in config.json, quantize_config.json in GPTQ and
AWQ form, gptq_marlin routing, unknown methods, GGUF directories), and the safetensors reader against files written by the
`safetensors` package plus crafted bad headers."""
import json
import struct

import numpy as np
import pytest

from vllm_rust_amd import checkpoint as ckp


def test_detection_from_config_json():
    d = ckp.detect_from_json(json.dumps({"model_type": "llama", "quantization_config": {"quant_method": "fp8", "activation_scheme": "dynamic"}}))
    assert (d.method, d.activation_scheme, d.bits) == ("fp8", "dynamic", None)
    d = ckp.detect_from_json(json.dumps({"quantization_config": {"quant_method": "gptq", "bits": 4, "group_size": 128, "desc_act": True}}))
    assert (d.method, d.bits, d.group_size, d.desc_act) == ("gptq", 4, 128, True)
    d = ckp.detect_from_json(json.dumps({"quantization_config": {"quant_method": "awq", "bits": 4, "group_size": 128}}))
    assert (d.method, d.bits, d.group_size, d.desc_act) == ("awq", 4, 128, None)
    assert ckp.detect_from_json(json.dumps({"quantization_config": {"quant_method": "awq_triton", "bits": 4, "group_size": 64}})).method == "awq"
    assert ckp.detect_from_json(json.dumps({"quantization_config": {"quant_method": "gptq_marlin", "bits": 4, "group_size": 128}})).method == "gptq_marlin"
    assert ckp.detect_from_json(json.dumps({"quantization_config": {"quant_method": "awq_marlin", "bits": 4, "group_size": 128}})).method == "awq_marlin"
    assert ckp.detect_from_json(json.dumps({"quantization_config": {"quant_method": "exl3", "bits": 4}})).method == "exl3"
    d = ckp.detect_from_json(json.dumps({"quantization_config": {"quant_method": "moe_wna16", "bits": 4, "group_size": 128}}))
    assert (d.method, d.method_name) == ("other", "moe_wna16")
    assert ckp.detect_from_json(json.dumps({"model_type": "llama"})).method == "none"
    assert ckp.detect_from_json(json.dumps({"quantization_config": {"quant_method": "made_up"}})).method == "none"
    assert ckp.detect_from_json("{ not json").method == "none"
    assert ckp.detect_from_json(None, None).method == "none"


def test_detection_modelopt_follows_quant_algo():
    """quant_method "modelopt" (detection.rs:143-160 and its tests :386-500): recognised for MXFP8 / FP8 / FP8_PER_CHANNEL_PER_TOKEN /
    FP8_PB_WO / NVFP4, nested under "quantization" or flat, case-insensitive; any other or a missing quant_algo is "not detected" and
    the search falls through to quantize_config.json."""
    mo = lambda inner: json.dumps({"quantization_config": dict({"quant_method": "modelopt"}, **inner)})
    for algo in ("MXFP8", "FP8", "FP8_PER_CHANNEL_PER_TOKEN", "FP8_PB_WO", "fp8"):
        d = ckp.detect_from_json(mo({"quantization": {"quant_algo": algo, "kv_cache_quant_algo": "FP8"}}))
        assert (d.method, d.method_name) == ("other", "modelopt"), algo
    assert ckp.detect_from_json(mo({"quant_algo": "NVFP4"})).method == "other"                      # flat layout
    assert ckp.detect_from_json(mo({"quantization": {"quant_algo": "INT8_SQ"}})).method == "none"   # unknown algorithm
    assert ckp.detect_from_json(mo({})).method == "none"                                            # no quantization block
    # not detected in config.json: quantize_config.json is consulted next
    assert ckp.detect_from_json(mo({"quantization": {"quant_algo": "INT8_SQ"}}), json.dumps({"w_bit": 4, "q_group_size": 128})).method == "awq"


def test_detection_from_quantize_config_and_directory(tmp_path):
    d = ckp.detect_from_json(None, json.dumps({"bits": 4, "group_size": 128, "desc_act": False, "sym": True, "damp_percent": 0.01}))
    assert (d.method, d.bits, d.group_size, d.desc_act) == ("gptq", 4, 128, False)
    d = ckp.detect_from_json(None, json.dumps({"bits": 4, "group_size": 128, "quant_method": "gptq_marlin"}))
    assert d.method == "gptq_marlin"
    d = ckp.detect_from_json(None, json.dumps({"zero_point": True, "q_group_size": 128, "w_bit": 4, "version": "GEMM"}))
    assert (d.method, d.bits, d.group_size) == ("awq", 4, 128)
    # config.json wins over quantize_config.json; a config.json without quantization_config falls through to it
    cfg = json.dumps({"quantization_config": {"quant_method": "fp8"}})
    assert ckp.detect_from_json(cfg, json.dumps({"w_bit": 4, "q_group_size": 128})).method == "fp8"
    assert ckp.detect_from_json(json.dumps({"hidden_size": 8}), json.dumps({"w_bit": 4, "q_group_size": 64})).group_size == 64
    (tmp_path / "a").mkdir()
    (tmp_path / "a" / "config.json").write_text(json.dumps({"hidden_size": 8}))
    (tmp_path / "a" / "quantize_config.json").write_text(json.dumps({"w_bit": 4, "q_group_size": 128}))
    assert ckp.detect_from_directory(tmp_path / "a").method == "awq"
    (tmp_path / "g").mkdir()
    (tmp_path / "g" / "model-q4.gguf").write_bytes(b"GGUF")
    assert ckp.detect_from_directory(tmp_path / "g").method == "gguf"
    (tmp_path / "e").mkdir()
    assert ckp.detect_from_directory(tmp_path / "e").method == "none"
    assert ckp.detect_from_directory(tmp_path / "missing").method == "none"


def test_safetensors_reader_matches_the_safetensors_package(tmp_path):
    from safetensors.numpy import save_file
    rng = np.random.default_rng(0)
    tensors = {"model.layers.0.self_attn.q_proj.qweight": rng.integers(-2 ** 31, 2 ** 31, (64, 16), dtype=np.int64).astype(np.int32),
               "model.layers.0.self_attn.q_proj.scales": rng.standard_normal((2, 128)).astype(np.float16),
               "model.norm.weight": rng.standard_normal(64).astype(np.float32), "scalar": np.asarray(3.5, np.float64),
               "empty": np.zeros((0, 4), np.uint8), "bytes": rng.integers(0, 256, (5, 3, 2), dtype=np.uint8)}
    p = tmp_path / "m.safetensors"
    save_file(tensors, str(p), metadata={"format": "pt", "note": "café \"quoted\""})
    f = ckp.SafeTensorsFile(p)
    assert sorted(f.names) == sorted(tensors)
    for name, t in tensors.items():
        assert name in f
        got = f.get(name)
        assert got.dtype == t.dtype and got.shape == t.shape and np.array_equal(got, t), name
    assert f.metadata("format") == "pt" and f.metadata("note") == "café \"quoted\"" and f.metadata("absent") is None
    assert "nope" not in f
    raw = p.read_bytes()
    hlen = struct.unpack("<Q", raw[:8])[0]
    for i in range(len(f.names)):   # file offsets point at the bytes the package wrote
        name, dtype, shape, off, nb = f.info(i)
        assert raw[off:off + nb] == tensors[name].tobytes() and off >= 8 + hlen
    f.close()


def _write(path, header: dict, data: bytes = b"", hlen=None):
    h = json.dumps(header).encode()
    path.write_bytes(struct.pack("<Q", len(h) if hlen is None else hlen) + h + data)


def test_safetensors_reader_rejects_bad_headers(tmp_path):
    from vllm_rust_amd._lib import lib
    p = tmp_path / "bad.safetensors"

    def fails(fragment):
        with pytest.raises(RuntimeError) as e:
            ckp.SafeTensorsFile(p)
        assert fragment in str(e.value), str(e.value)

    p.write_bytes(b"\x01\x02\x03")
    fails("shorter than")
    _write(p, {"t": {"dtype": "F32", "shape": [2], "data_offsets": [0, 8]}}, b"\0" * 8, hlen=1 << 40)
    fails("header length")
    _write(p, {"t": {"dtype": "F32", "shape": [4], "data_offsets": [0, 16]}}, b"\0" * 8)
    fails("outside the file")
    _write(p, {"t": {"dtype": "F32", "shape": [3], "data_offsets": [0, 8]}}, b"\0" * 8)
    fails("shape x dtype")
    _write(p, {"t": {"dtype": "Q4", "shape": [2], "data_offsets": [0, 8]}}, b"\0" * 8)
    fails("unknown dtype")
    _write(p, {"t": {"dtype": "F32", "shape": [1 << 62, 8], "data_offsets": [0, 8]}}, b"\0" * 8)
    fails("overflows")
    _write(p, {"t": {"dtype": "F32", "data_offsets": [0, 8]}}, b"\0" * 8)
    fails("lacks dtype")
    p.write_bytes(struct.pack("<Q", 9) + b'{"a": [1,' + b"\0" * 8)
    fails("not a JSON object")
    assert lib.vmx_safetensors_tensor_count(None) == -1

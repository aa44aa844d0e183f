"""GGUF container parser (host code of libvmx, no GPU): against the Python restatement of the reference's
GgufHeader::read on synthetic files of every version, plus the facts the reference's own header tests assert."""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import gguf_oracle as go


def _s(s, v):
    b = s.encode()
    return struct.pack("<Q" if v >= 2 else "<I", len(b)) + b


def _val(t, x, v):
    fmt = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<B", 10: "<Q", 11: "<q", 12: "<d"}
    if t == 8:
        return _s(x, v)
    if t == 9:
        et, items = x
        return struct.pack("<I", et) + struct.pack("<Q" if v >= 2 else "<I", len(items)) + b"".join(_val(et, i, v) for i in items)
    return struct.pack(fmt[t], x)


def build_gguf(version, metadata, tensors, data_align=32, payload=True):
    """metadata: [(key, type, value)], tensors: [(name, on_disk_dims, ggml_type)] -> (bytes, {name: payload bytes})"""
    cnt = "<Q" if version >= 2 else "<I"
    out = b"GGUF" + struct.pack("<I", version) + struct.pack(cnt, len(tensors)) + struct.pack(cnt, len(metadata))
    for k, t, x in metadata:
        out += _s(k, version) + struct.pack("<I", t) + _val(t, x, version)
    off, blobs, infos = 0, {}, b""
    rng = np.random.default_rng(len(tensors))
    for name, dims, tid in tensors:
        ts, bs = go.TYPE_GEOMETRY.get(tid, (1, 1))
        n = int(np.prod(dims)) // bs * ts
        infos += _s(name, version) + struct.pack("<I", len(dims)) + b"".join(struct.pack(cnt, d) for d in dims)
        infos += struct.pack("<I", tid) + struct.pack("<Q", off)
        blobs[name] = (off, rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        off += (n + data_align - 1) // data_align * data_align
    out += infos
    start = (len(out) + data_align - 1) // data_align * data_align
    if payload:
        out += b"\0" * (start - len(out))
        body = bytearray(off)
        for name, (o, b) in blobs.items():
            body[o:o + len(b)] = b
        out += bytes(body)
    return out, {k: b for k, (_, b) in blobs.items()}


GEMMA_TENSORS = [("token_embd.weight", [3840, 1024], 11), ("blk.0.attn_v.weight", [3840, 2048], 18), ("blk.0.attn_k.weight", [3840, 2048], 22),
                 ("blk.0.attn_output.weight", [4096, 3840], 21), ("blk.16.attn_k.weight", [3840, 2048], 17), ("blk.1.ffn_down.weight", [15360, 256], 23),
                 ("output_norm.weight", [3840], 0), ("mtp.token_ordering.weight", [100], 26), ("blk.0.attn_norm.weight", [3840], 30)]
GEMMA_META = [("general.architecture", 8, "gemma4"), ("gemma4.block_count", 4, 48), ("gemma4.attention.layer_norm_rms_epsilon", 6, 1e-6),
              ("gemma4.rope.freq_base", 6, 1000000.0), ("tokenizer.ggml.tokens", 9, (8, ["<pad>", "<eos>", "hello"])),
              ("gemma4.attention.sliding_window_pattern", 9, (7, [1, 1, 0])), ("some.i64", 11, -5), ("some.f64", 12, 2.5), ("some.u64", 10, 2 ** 40)]


def _open(lib, data):
    h = C.c_void_p()
    buf = C.create_string_buffer(data, len(data))
    rc = lib.vmx_gguf_parse(buf, len(data), C.byref(h))
    return rc, h


@pytest.mark.parametrize("version", [1, 2, 3])
@pytest.mark.parametrize("align", [None, 64])
def test_header_matches_oracle(version, align):
    from vllm_rust_amd._lib import GgufTensor, lib
    meta = list(GEMMA_META) + ([("general.alignment", 4, align)] if align else [])
    data, _ = build_gguf(version, meta, GEMMA_TENSORS, data_align=align or 32, payload=False)
    ref = go.parse(data)
    rc, h = _open(lib, data)
    assert rc == 0, lib.vmx_last_error()
    assert lib.vmx_gguf_version(h) == version == ref["version"]
    assert lib.vmx_gguf_tensor_data_offset(h) == ref["tensor_data_offset"]
    assert lib.vmx_gguf_tensor_count(h) == len(ref["tensors"])
    for i, (name, shape, tid, off, nbytes) in enumerate(ref["tensors"]):
        t = GgufTensor()
        assert lib.vmx_gguf_tensor_info(h, i, C.byref(t)) == 0
        assert t.name.decode() == name and t.n_dims == len(shape) and list(t.shape[:t.n_dims]) == shape
        assert (t.ggml_type, t.offset, t.n_bytes) == (tid, off, nbytes)
        assert lib.vmx_gguf_find_tensor(h, name.encode()) == i
    assert lib.vmx_gguf_find_tensor(h, b"nope") == -1
    # the facts the reference's real-file test asserts (header.rs:371-415), on the synthetic table
    v = GgufTensor(); lib.vmx_gguf_tensor_info(h, lib.vmx_gguf_find_tensor(h, b"blk.0.attn_v.weight"), C.byref(v))
    assert list(v.shape[:2]) == [2048, 3840] and v.ggml_type == 18 and v.n_bytes == 2048 * 3840 // 256 * 98
    # metadata
    assert lib.vmx_gguf_metadata_count(h) == len(meta)
    assert [lib.vmx_gguf_metadata_key(h, i).decode() for i in range(len(meta))] == [k for k, _, _ in meta]
    assert lib.vmx_gguf_metadata_string(h, b"general.architecture", -1) == b"gemma4"
    i64, f64, n, et = C.c_int64(), C.c_double(), C.c_uint64(), C.c_int()
    assert lib.vmx_gguf_metadata_int(h, b"gemma4.block_count", -1, C.byref(i64)) == 0 and i64.value == 48
    assert lib.vmx_gguf_metadata_int(h, b"some.i64", -1, C.byref(i64)) == 0 and i64.value == -5
    assert lib.vmx_gguf_metadata_int(h, b"some.u64", -1, C.byref(i64)) == 0 and i64.value == 2 ** 40
    assert lib.vmx_gguf_metadata_float(h, b"some.f64", -1, C.byref(f64)) == 0 and f64.value == 2.5
    assert lib.vmx_gguf_metadata_float(h, b"gemma4.rope.freq_base", -1, C.byref(f64)) == 0 and f64.value == 1000000.0
    assert lib.vmx_gguf_metadata_type(h, b"tokenizer.ggml.tokens", C.byref(n), C.byref(et)) == 9 and n.value == 3 and et.value == 8
    assert lib.vmx_gguf_metadata_string(h, b"tokenizer.ggml.tokens", 2) == b"hello"
    assert lib.vmx_gguf_metadata_int(h, b"gemma4.attention.sliding_window_pattern", 2, C.byref(i64)) == 0 and i64.value == 0
    assert lib.vmx_gguf_metadata_type(h, b"absent", None, None) == -1
    assert lib.vmx_gguf_metadata_int(h, b"general.architecture", -1, C.byref(i64)) == -1
    lib.vmx_gguf_close(h)


def test_open_and_read_tensor_bytes(tmp_path):
    from vllm_rust_amd._lib import GgufTensor, lib
    data, blobs = build_gguf(3, GEMMA_META, GEMMA_TENSORS)
    path = tmp_path / "m.gguf"
    path.write_bytes(data)
    h = C.c_void_p()
    assert lib.vmx_gguf_open(str(path).encode(), C.byref(h)) == 0, lib.vmx_last_error()
    for name, blob in blobs.items():
        i = lib.vmx_gguf_find_tensor(h, name.encode())
        t = GgufTensor(); lib.vmx_gguf_tensor_info(h, i, C.byref(t))
        buf = C.create_string_buffer(int(t.n_bytes))
        assert lib.vmx_gguf_read_tensor(h, i, buf, t.n_bytes) == 0
        assert buf.raw == blob
        assert lib.vmx_gguf_read_tensor(h, i, buf, max(t.n_bytes - 1, 0)) == -1
    lib.vmx_gguf_close(h)
    assert lib.vmx_gguf_open(str(tmp_path / "missing.gguf").encode(), C.byref(h)) == -1


def test_rejections():
    from vllm_rust_amd._lib import lib
    data, _ = build_gguf(3, GEMMA_META, GEMMA_TENSORS, payload=False)
    rc, _ = _open(lib, b"GGML" + data[4:])
    assert rc == -1 and b"not a GGUF file" in lib.vmx_last_error()
    bad, _ = build_gguf(3, [], [("w", [256, 4], 9999)], payload=False)
    rc, _ = _open(lib, bad)
    assert rc == -4 and b"'w': unsupported GGML type id 9999" in lib.vmx_last_error()
    with pytest.raises(go.GgufError):
        go.parse(bad)
    for cut in (6, 30, len(data) - 3):
        rc, _ = _open(lib, data[:cut])
        assert rc == -1
    badbool, _ = build_gguf(3, [("b", 7, 2)], [], payload=False)
    rc, _ = _open(lib, badbool)
    assert rc == -1 and b"bool" in lib.vmx_last_error()
    badtype = b"GGUF" + struct.pack("<IQQ", 3, 0, 1) + _s("k", 3) + struct.pack("<I", 13)
    rc, _ = _open(lib, badtype)
    assert rc == -1 and b"value-type 13" in lib.vmx_last_error()


def test_every_classified_type_id_and_plain_integers():
    """classify_dtype (header.rs:325-362): candle-native ids, I8..I64 (block 1), the five I-quants; 9999 is unknown."""
    from vllm_rust_amd._lib import GgufTensor, lib
    tens = [(f"t{tid}", [bs * 3, 2], tid) for tid, (ts, bs) in sorted(go.TYPE_GEOMETRY.items())]
    data, _ = build_gguf(3, [], tens, payload=False)
    rc, h = _open(lib, data)
    assert rc == 0
    for i, (name, dims, tid) in enumerate(tens):
        t = GgufTensor(); lib.vmx_gguf_tensor_info(h, i, C.byref(t))
        ts, bs = go.TYPE_GEOMETRY[tid]
        assert t.n_bytes == dims[0] * dims[1] // bs * ts
    lib.vmx_gguf_close(h)
    assert go.TYPE_GEOMETRY[26] == (4, 1)


def test_llama_cpp_name_map():
    from vllm_rust_amd._lib import lib
    cases = ["model.embed_tokens", "model.norm", "lm_head", "model.rope_freqs", "model.embed_tokens_per_layer",
             "model.layers.0.self_attn.q_proj", "model.layers.17.self_attn.k_norm", "model.layers.3.mlp.down_proj",
             "model.layers.5.post_attention_layernorm", "model.layers.5.pre_feedforward_layernorm",
             "model.layers.47.post_feedforward_layernorm", "model.layers.2.layer_scalar", "model.layers.2.input_layernorm",
             "model.layers.x.mlp.up_proj", "model.layers.3.unknown", "model.layers.3", "other.thing"]
    for c in cases:
        exp = go.llama_cpp_names(c)
        got = []
        for which in range(3):
            buf = C.create_string_buffer(64)
            n = lib.vmx_gguf_llama_cpp_name(c.encode(), which, buf, 64)
            if n == 0:
                break
            got.append(buf.value.decode())
        assert got == exp, c
    assert go.llama_cpp_names("model.layers.5.post_attention_layernorm") == ["blk.5.post_attention_norm", "blk.5.ffn_norm"]
    assert lib.vmx_gguf_llama_cpp_name(b"model.norm", 0, C.create_string_buffer(4), 4) == 0   # buffer too small


def test_crafted_lengths_do_not_allocate_or_overflow():
    """Length fields larger than the data that follows are refused before anything is resized (a 2^27-element array or a
    1 GiB string announced by a 40-byte header), and a tensor whose element count wraps around 2^64 or is not a whole
    number of blocks is an error instead of an undersized tensor."""
    from vllm_rust_amd._lib import lib
    hdr = b"GGUF" + struct.pack("<IQQ", 3, 0, 1) + _s("k", 3)
    big_array = hdr + struct.pack("<I", 9) + struct.pack("<I", 4) + struct.pack("<Q", 1 << 27)
    rc, _ = _open(lib, big_array)
    assert rc == -1 and b"unexpected end of data" in lib.vmx_last_error()
    big_string = hdr + struct.pack("<I", 8) + struct.pack("<Q", 1 << 30)
    rc, _ = _open(lib, big_string)
    assert rc == -1
    wrap = b"GGUF" + struct.pack("<IQQ", 3, 1, 0) + _s("w", 3) + struct.pack("<I", 2) + struct.pack("<QQ", 1 << 33, 1 << 33)
    wrap += struct.pack("<I", 0) + struct.pack("<Q", 0)
    rc, _ = _open(lib, wrap)
    assert rc == -1 and b"'w'" in lib.vmx_last_error() and b"overflows" in lib.vmx_last_error()
    ragged, _ = build_gguf(3, [], [("r", [100, 3], 18)], payload=False)   # 300 elements, 256-element blocks
    rc, _ = _open(lib, ragged)
    assert rc == -1 and b"not a multiple" in lib.vmx_last_error()

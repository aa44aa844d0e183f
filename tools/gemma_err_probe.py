"""Development probe: relative-L2 error of the Gemma-4-12B-geometry executor against the CPU oracle for several cache dtypes,
context lengths and cache magnitudes (how much of the error is q8_1 / FP8 rounding flips, how much depends on the context)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import vllm_rust_amd as vmx
from tests.model_oracle import OracleGemma4
from tests.test_gpu_true_geometry import _decode_and_compare, _fill_kv
from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM

vmx.init(0)
for kv, start, layers_full_every in [(1, 1023, 2), (0, 1023, 2), (1, 100, 2), (0, 100, 2), (0, 1023, 100), (0, 1023, 1)]:
    nl = 2 if layers_full_every == 2 else 1
    spec = Gemma4Spec(3840, nl, 16, 15360, 8192, full_every=layers_full_every, max_position=2048, num_blocks=320, max_batch=1,
                      max_seq_len=2048, embed_ggml_type=11, kv_cache_dtype=kv)
    model = QuantizedGemma4ForCausalLM(spec)
    ck = model.load_synthetic(seed=21, keep_host_copy=True)
    ref = OracleGemma4(spec, ck, model.cos_sin_sliding, model.cos_sin_full)
    shapes = [(spec.num_blocks, spec.block_size, spec.layer_geom(li)[1], spec.layer_geom(li)[0]) for li in range(nl)]
    _fill_kv(torch, model, ref, nl, shapes, kv, seed=3)
    print("kv", kv, "start", start, "full_every", layers_full_every, flush=True)
    try:
        _decode_and_compare(torch, model, ref, spec, [start], 2, [17], bar=1.0)
    except AssertionError as e:
        print("assert", e)
    model.close()

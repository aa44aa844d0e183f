"""Per-wave timeline of the gate|up int4 GEMM of a Qwen3-8B-shaped layer (build variant libvmx_stamps.so, made with make STAMPS=1:
-DVMX_WIDE_STAMPS): run as  VMX_LIB_VARIANT=stamps python tools/wide_stamps.py [mask]   (mask 4 = gate|up, 1 = q|k|v)."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
from vllm_rust_amd.qwen3 import Qwen3Spec, QuantizedQwen3ForCausalLM

vmx.init(0)
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 4
spec = Qwen3Spec(hidden_size=4096, num_layers=8, num_heads=32, num_kv_heads=8, head_dim=128, intermediate_size=12288, vocab_size=32000,
                 max_position=2048, num_blocks=64, max_batch=1, max_seq_len=1024)
m = QuantizedQwen3ForCausalLM(spec)
m.load_synthetic(seed=1)
m.linears_select(mask)
for _ in range(3):
    m.linears_only(1)
torch.cuda.synchronize()
nw = 1536 * 4 if mask == 4 else 768 * 4
buf = np.zeros((8192, 8), dtype=np.uint64)
lib.vmx_debug_wide_stamps_read.restype = C.c_int
lib.vmx_debug_wide_stamps_read.argtypes = [C.c_void_p, C.c_size_t]
check(lib.vmx_debug_wide_stamps_read(buf.ctypes.data, buf.nbytes))
s = buf[:nw].astype(np.float64)
rt0, rt1 = s[:, 0], s[:, 7]
print(f"waves {nw}; launch span (100 MHz clock) first entry -> last end: {(rt1.max() - rt0.min()) / 100:.2f} us")
print(f"entry spread: p10 {(np.percentile(rt0, 10) - rt0.min()) / 100:.2f}  median {(np.median(rt0) - rt0.min()) / 100:.2f}  p90 {(np.percentile(rt0, 90) - rt0.min()) / 100:.2f}  max {(rt0.max() - rt0.min()) / 100:.2f} us")
print(f"end   spread: p10 {(np.percentile(rt1, 10) - rt0.min()) / 100:.2f}  median {(np.median(rt1) - rt0.min()) / 100:.2f}  p90 {(np.percentile(rt1, 90) - rt0.min()) / 100:.2f} us")
life = (rt1 - rt0) / 100
print(f"wave lifetime: p10 {np.percentile(life, 10):.2f} median {np.median(life):.2f} p90 {np.percentile(life, 90):.2f} us")
d = s[:, 1:5] - s[:, 1:2]
names = ["entry", "weights requested", "activations staged + barrier", "last unit done"]
for i, n in enumerate(names):
    print(f"  {n:32s} median {np.median(d[:, i]):8.0f} p10 {np.percentile(d[:, i], 10):8.0f} p90 {np.percentile(d[:, i], 90):8.0f}  shader clocks after entry")
# by entry order: which waves start late?
order = np.argsort(rt0)
for q in (0, 0.25, 0.5, 0.75, 0.99):
    w = order[int(q * (nw - 1))]
    print(f"  wave at entry quantile {q:4.2f}: workgroup {w // 4:5d} entry {(rt0[w] - rt0.min()) / 100:6.2f} us, lifetime {life[w]:.2f} us")
m.close()

"""vllm_rust_amd: MI355X-native quantized-inference hot path behind a C ABI.

Python here is plumbing for tests and the benchmark (device memory through
torch, streams, torch.distributed); the operators live in lib/libvmx.so.
"""
from . import _lib  # noqa: F401  (raises if the HIP library is missing)
from .ops import *  # noqa: F401,F403

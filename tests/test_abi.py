"""CPU-side checks of the boundary: the library loads, exports every symbol that
include/vmx.h declares, and rejects bad arguments before touching a device."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "vmx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vmx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vllm_rust_amd import _lib
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), f"libvmx.so lacks {n}"
    assert _lib.lib.vmx_abi_version() == 1


def test_python_binding_covers_header():
    from vllm_rust_amd import _lib
    assert set(_lib.declared_symbols()) == set(_declared())


def test_argument_errors_need_no_gpu():
    from vllm_rust_amd._lib import lib
    assert lib.vmx_iq_type_size(18) == 98 and lib.vmx_iq_type_size(12) == -1
    assert lib.vmx_iq_mmvq_f32(18, 1, 1, 1, 1, 8, 256, 17, None) == -1
    assert b"IQ_GEMV_MAX_M" in lib.vmx_last_error()
    assert lib.vmx_int4_packed_bytes(4096, 4096, 128) == 4096 * 4096 // 2 + 32 * 4096 * 4
    assert lib.vmx_int4_gemm_bf16(1, 1, 1, None, 1, 4096, 4000, 128, None, 0, None) == -1

#!/usr/bin/env python3
"""Extract the GGML I-quant codebook constants into a binary data blob.

The codebooks (iq2xs_grid, iq2s_grid, iq3xxs_grid, iq3s_grid, ksigns, kmask,
kvalues_iq4nl) are constants of the GGUF file format (ggml-common.h).  The
reference keeps a generated copy in
crates/core/src/quantization/gguf/iq/tables.rs:9-327.  This tool parses the
NUMBERS out of that file (run once, in the build container where
/root/reference is mounted) and writes them as little-endian binary:

    offset  size   content
    0       8      kmask_iq2xs   u8[8]
    8       128    ksigns_iq2xs  u8[128]
    136     4096   iq2xs_grid    u64[512]
    4232    8192   iq2s_grid     u64[1024]
    12424   1024   iq3xxs_grid   u32[256]
    13448   2048   iq3s_grid     u32[512]
    15496   16     kvalues_iq4nl i8[16]
    total 15512 bytes

The blob is committed (vllm-rust_amd/csrc/data/iq_tables.bin); the C header
used by the kernels and by the oracle is generated from the blob at build time
by tools/gen_iq_tables_header.py, so neither build needs the reference tree.
"""
import re
import struct
import sys
from pathlib import Path

SRC = Path("/root/reference/crates/core/src/quantization/gguf/iq/tables.rs")
OUT = Path(__file__).resolve().parent.parent / "vllm-rust_amd/csrc/data/iq_tables.bin"

SPEC = [
    ("KMASK_IQ2XS", "B", 8),
    ("KSIGNS_IQ2XS", "B", 128),
    ("IQ2XS_GRID", "Q", 512),
    ("IQ2S_GRID", "Q", 1024),
    ("IQ3XXS_GRID", "I", 256),
    ("IQ3S_GRID", "I", 512),
    ("KVALUES_IQ4NL", "b", 16),
]


def main() -> int:
    text = SRC.read_text()
    blob = b""
    for name, fmt, count in SPEC:
        m = re.search(r"pub static %s: \[[^\]]+\] = \[(.*?)\];" % name, text, re.S)
        if not m:
            print(f"table {name} not found", file=sys.stderr)
            return 1
        vals = [int(tok, 0) for tok in re.findall(r"-?0x[0-9a-fA-F]+|-?\d+", m.group(1))]
        if len(vals) != count:
            print(f"{name}: {len(vals)} values, expected {count}", file=sys.stderr)
            return 1
        blob += struct.pack("<%d%s" % (count, fmt), *vals)
    assert len(blob) == 15512, len(blob)
    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_bytes(blob)
    print(f"wrote {OUT} ({len(blob)} bytes)")
    return 0


if __name__ == "__main__":
    sys.exit(main())

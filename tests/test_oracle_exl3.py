"""EXL3 oracle: the reference's zero-trellis constant, Hadamard properties, layout self-consistency, and the
reference's own self-consistency test (exl3_cuda.rs:1397-1432,1542-1633,1796-1868) restated on the CPU."""
import numpy as np

from oracle import oracle as orc


def lcg_trellis(k, n, bpw, seed=0x12345678):
    out = np.empty((k // 16) * (n // 16) * 16 * bpw, np.uint16)
    for i in range(out.size):
        seed = (seed * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = (seed >> 16) & 0xFFFF
    return out


def test_zero_trellis_is_constant():
    for bpw in (2, 3, 4, 6, 8):
        w = orc.exl3_reconstruct(np.zeros((32 // 16) * (128 // 16) * 16 * bpw, np.uint16), 32, 128, bpw, 0)
        assert (w == w.flat[0]).all()
        assert w.flat[0] == orc.exl3_decode(0, 0)
    # decode_3inst<0>(0): x = 64248484 = 0x03D45AA4 -> (x & 0x8fff8fff) ^ 0x3b603b60; hi 0x38B4 + lo 0x31C4 in fp16
    x = (64248484 & 0x8fff8fff) ^ 0x3b603b60
    lo, hi = np.array([x & 0xffff], np.uint16).view(np.float16)[0], np.array([x >> 16], np.uint16).view(np.float16)[0]
    assert np.array([orc.exl3_decode(0, 0)], np.uint16).view(np.float16)[0] == np.float16(np.float32(lo) + np.float32(hi))


def test_codebooks_are_bounded_and_distinct():
    for cb in (0, 1, 2):
        vals = orc.f16_to_f32(np.array([orc.exl3_decode(w, cb) for w in range(0, 65536, 7)], np.uint16))
        assert np.isfinite(vals).all() and np.abs(vals).max() < 8.0 and vals.std() > 0.3


def test_window_layout_bit_positions():
    """Setting one 16-bit window pattern for weight t must show up at the tile position of reconstruct.cu:36-72."""
    bpw = 4
    for t in (0, 1, 7, 8, 37, 255):
        words = np.zeros(8 * bpw, np.uint32)
        b0 = ((t + 257) * bpw - 16) % (256 * bpw)
        pat = 0xBEEF
        for j in range(16):
            i = (b0 + j) % (256 * bpw)
            if (pat >> (15 - j)) & 1:
                words[i // 32] |= np.uint32(1 << (31 - i % 32))
        w = orc.exl3_reconstruct(words.view(np.uint16), 16, 16, bpw, 0)
        lane, i = t >> 3, t & 7
        c, h, q = lane >> 3, (lane >> 2) & 1, lane & 3
        row, col = 2 * q + (i & 1) + 8 * ((i >> 1) & 1), 2 * c + h + 8 * (i >> 2)
        assert w[row, col] == orc.exl3_decode(pat, 0)


def test_hadamard_norm_preserving_and_involutive():
    i = np.arange(4 * 256)
    x = orc.f32_to_f16(np.clip(np.sin(i * 0.1234) * 0.5, -1, 1).astype(np.float32).reshape(4, 256))
    y = orc.exl3_had_fp16(x)
    xn, yn = (orc.f16_to_f32(x) ** 2).sum(), (orc.f16_to_f32(y) ** 2).sum()
    assert abs(yn / xn - 1.0) < 0.015                    # exl3_cuda.rs:1860-1866
    z = orc.f16_to_f32(orc.exl3_had_fp16(y))
    assert np.abs(z - orc.f16_to_f32(x)).max() < 4e-3    # H/sqrt(128) is its own inverse
    ones = orc.f32_to_f16(np.ones(256, np.float32))
    assert np.array_equal(orc.exl3_had_fp16(x, ones, 1), y) and np.array_equal(orc.exl3_had_fp16(x, ones, 2), y)


def test_gemm_matches_hadamard_then_matmul():
    """exl3_cuda.rs:1542-1633: M=2, K=512, N=128, bpw=4, LCG trellis, x = sin(0.013 i) * 0.05, suh = svh = 1."""
    m, k, n, bpw = 2, 512, 128, 4
    tr = lcg_trellis(k, n, bpw)
    x = orc.f32_to_f16((np.sin(np.arange(m * k) * np.float32(0.013)) * 0.05).astype(np.float32).reshape(m, k))
    ones_k, ones_n = orc.f32_to_f16(np.ones(k, np.float32)), orc.f32_to_f16(np.ones(n, np.float32))
    y = orc.f16_to_f32(orc.exl3_gemm(x, tr, ones_k, ones_n, k, n, bpw))
    xh = orc.f16_to_f32(orc.exl3_had_fp16(x, ones_k, 1)).astype(np.float64)
    w = orc.f16_to_f32(orc.exl3_reconstruct(tr, k, n, bpw)).astype(np.float64)
    y_pre = orc.f32_to_f16((xh @ w).astype(np.float32))                # the reference's fp16 matmul output
    y_ref = orc.f16_to_f32(orc.exl3_had_fp16(y_pre, ones_n, 2))
    assert np.abs(y - y_ref).max() / np.abs(y_ref).max() < 0.10       # the reference's own gate
    assert np.abs(y - y_ref).max() / np.abs(y_ref).max() < 5e-3       # and in practice far tighter

"""Pin the int4 oracle to the reference's known-answer tests
(awq_marlin.rs:721-809, marlin_cuda.rs:2032-2184)."""
import numpy as np

from oracle import oracle as orc
from tests.helpers import awq_lcg_case


def test_awq_to_gptq_u32_known_values():
    assert orc.awq_to_gptq_u32(0) == 0
    awq = 1 | (3 << 4) | (5 << 8) | (7 << 12) | (2 << 16) | (4 << 20) | (6 << 24) | (8 << 28)
    exp = 1 | (2 << 4) | (3 << 8) | (4 << 12) | (5 << 16) | (6 << 20) | (7 << 24) | (8 << 28)
    assert orc.awq_to_gptq_u32(awq) == exp          # awq_marlin.rs:727-738


def test_awq_to_gptq_qweight_known_value():
    k = n = 8                                        # awq_marlin.rs:761-799
    words = np.array([[sum(((ki + 1) & 0xF) << s for s in range(0, 32, 4))] for ki in range(k)], np.uint32)
    out = orc.awq_to_gptq_qweight(words, k, n)
    assert out.shape == (1, 8)
    exp = sum(((ki + 1) & 0xF) << (4 * ki) for ki in range(k))
    assert (out == exp).all()


def test_awq_to_gptq_qweight_general():
    rng = np.random.default_rng(0)
    K, N = 64, 32
    nib = rng.integers(0, 16, size=(K, N), dtype=np.uint32)     # logical int4 [K, N]
    order = [0, 4, 1, 5, 2, 6, 3, 7]                             # lane i -> nibble position
    awq = np.zeros((K, N // 8), np.uint32)
    for i in range(8):
        awq |= nib[:, i::8] << np.uint32(4 * order[i])
    gptq = orc.awq_to_gptq_qweight(awq, K, N)
    assert gptq.shape == (K // 8, N)
    for i in range(8):
        assert ((gptq >> np.uint32(4 * i)) & 0xF == nib[i::8, :]).all()
    z = orc.repack_awq_nibbles(awq)
    for i in range(8):
        assert ((z >> np.uint32(4 * i)) & 0xF == nib[:, i::8]).all()


def test_gemv_kernel_contract_vs_scalar_reference():
    """bf16-in/bf16-out contract vs the test's f32 scalar reference, with the
    reference test's tolerance max(1% max|y|, 5e-2)."""
    for m in (1, 4, 8, 12, 16):
        qw, qz, sc, x = awq_lcg_case(m, 256, 128, 128)
        sc_b, x_b = orc.f32_to_bf16(sc), orc.f32_to_bf16(x)
        exp = orc.int4_gemv_f32(orc.bf16_to_f32(x_b), qw, orc.bf16_to_f32(sc_b), qz, 256, 128, 128)
        got = orc.bf16_to_f32(orc.int4_gemv_bf16(x_b, qw, sc_b, qz, None, 256, 128, 128))
        tol = max(np.abs(exp).max() * 0.01, 5e-2)
        assert np.abs(got - exp).max() <= tol
        dense = orc.int4_dequant(qw, orc.bf16_to_f32(sc_b), qz, 256, 128, 128)
        ref = orc.bf16_to_f32(x_b).astype(np.float64) @ dense.astype(np.float64)
        assert np.abs(exp - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max())


def test_gptq_symmetric_zero_point():
    qw, _, sc, x = awq_lcg_case(1, 128, 64, 128)
    dense = orc.int4_dequant(qw, sc, None, 128, 64, 128)
    k, n = 5, 9
    nib = (int(qw[k // 8, n]) >> (4 * (k % 8))) & 0xF
    assert dense[k, n] == np.float32(nib - 8) * sc[0, n]        # marlin_gemm.cu:43-48


def test_gptq_int8_dequant_formula():
    """gptq_dequant.cu:128-170: (q - z) * s with byte k % 4 of word [k / 4][n] and byte n % 4 of the zero word; against numpy."""
    rng = np.random.default_rng(3)
    K, N, G = 64, 24, 32
    q = rng.integers(0, 256, (K, N)).astype(np.uint32)
    z = rng.integers(0, 256, (K // G, N)).astype(np.uint32)
    s = orc.f32_to_bf16((rng.random((K // G, N)) * 0.02 + 0.001).astype(np.float32))
    qw = np.zeros((K // 4, N), np.uint32)
    for j in range(4):
        qw |= q[j::4, :] << np.uint32(8 * j)
    qz = np.zeros((K // G, N // 4), np.uint32)
    for j in range(4):
        qz |= z[:, j::4] << np.uint32(8 * j)
    got = orc.bf16_to_f32(orc.gptq_dequant_int8(qw, s, qz, K, N, G))
    want = (q.astype(np.int64) - np.repeat(z, G, axis=0).astype(np.int64)).astype(np.float32) * np.repeat(orc.bf16_to_f32(s), G, axis=0)
    assert np.array_equal(got, orc.bf16_to_f32(orc.f32_to_bf16(want)))

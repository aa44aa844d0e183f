"""Pin the I-quant oracle to the reference's gguf-py golden vectors and restate
the reference's CPU tests (gguf/iq/mod.rs:905-1100)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import IQ_NAMES, load_iq_fixture, ramp, tile_blocks


@pytest.mark.parametrize("name", IQ_NAMES)
def test_dequant_matches_gguf_py_golden(name):
    f, raw, gold = load_iq_fixture(name)
    tid = f["ggml_type_id"]
    assert orc.IQ_TYPES[name] == tid
    assert orc.IQ_TYPE_SIZE[tid] == f["type_size"] and f["block_size"] == 256
    assert raw.size == f["n_blocks"] * f["type_size"]
    out = orc.iq_dequant(tid, raw)
    assert out.shape == gold.shape
    assert np.abs(out - gold).max() <= 1e-5          # mod.rs:925
    assert np.any(out != 0.0)


def test_type_ids_round_trip():
    assert sorted(orc.IQ_TYPE_SIZE) == [17, 18, 21, 22, 23]
    assert orc._lib.orc_iq_type_size(12) == -1       # Q4_K is not an I-quant


def test_q8_1_quantiser_rules():
    x = np.zeros((1, 64), np.float32)
    x[0, 32:] = np.linspace(-3, 3, 32, dtype=np.float32)
    aq, ad = orc.quantize_q8_1(x)
    assert ad[0, 0] == 0.0 and not aq[0, :32].any()  # all-zero block
    assert ad[0, 1] == np.float32(3.0) / np.float32(127.0)
    assert aq[0, 32] == -127 and aq[0, 63] == 127
    # half-away-from-zero rounding
    x2 = np.zeros((1, 32), np.float32); x2[0, 0] = 127.0; x2[0, 1] = 0.5; x2[0, 2] = -0.5; x2[0, 3] = 1.5
    aq2, ad2 = orc.quantize_q8_1(x2)
    assert ad2[0, 0] == 1.0 and list(aq2[0, :4]) == [127, 1, -1, 2]


@pytest.mark.parametrize("name", IQ_NAMES)
def test_subdots_equal_dense_integer_dot(name):
    """The integer sub-dot must equal the exact integer form of dequant * q8:
    cross-check of the two code paths of the restatement on the fixture blocks."""
    f, raw, _ = load_iq_fixture(name)
    tid, ts, nb = f["ggml_type_id"], f["type_size"], f["n_blocks"]
    x = ramp(4, 256, 23, 11, 0.03)
    aq, ad = orc.quantize_q8_1(x)
    sub = orc.iq_subdots(tid, raw, aq, nb, 256)
    # rebuild: dequant with d forced to 1.0 gives scale*q per weight
    raw1 = raw.copy().reshape(nb, ts); raw1[:, 0] = 0x00; raw1[:, 1] = 0x3C
    w1 = orc.iq_dequant(tid, raw1.reshape(-1)).reshape(nb, 8, 32).astype(np.float64)
    dots = np.einsum("ogk,mgk->omg", w1, aq.reshape(4, 8, 32).astype(np.float64))
    # float scales: iq2 (0.5+ls)/4, iq3_xxs (0.5+ls)/2, iq3_s (1+2ls), iq4_xs (ls-32);
    # the integer form truncates, so it may differ by < 1 from the real-valued product
    assert np.abs(sub - dots).max() < 1.0 + 1e-9


def test_iq_matmul_matches_dequant_then_matmul():
    """mod.rs:981-1046: q8_1 MMVQ vs dense f32 matmul, rel-L2 <= 5 %."""
    f, raw, _ = load_iq_fixture("iq3_xxs")
    tid, ts, nb = f["ggml_type_id"], f["type_size"], f["n_blocks"]
    for n_out, n_in in [(6, 256), (3, 512), (2, 768), (1, 2048), (1, 4096)]:
        w_raw = tile_blocks(raw, ts, nb, n_out * n_in // 256)
        w = orc.iq_dequant(tid, w_raw).reshape(n_out, n_in)
        for m in (1, 4, 16):
            x = ramp(m, n_in, 17, 8, 0.05)
            fused = orc.iq_matmul(tid, w_raw, x, n_out, n_in)
            ref = x.astype(np.float64) @ w.T.astype(np.float64)
            rel = np.sqrt(((fused - ref) ** 2).sum() / max((ref ** 2).sum(), 1e-12))
            assert (ref ** 2).sum() > 1e-6
            assert rel <= 0.05, (n_in, m, rel)


def test_q8_1_outlier_activation():
    """mod.rs:1054-1100."""
    f, raw, _ = load_iq_fixture("iq3_xxs")
    tid = f["ggml_type_id"]
    n_in = 1536
    w = orc.iq_dequant(tid, raw).reshape(1, n_in)
    i = np.arange(n_in)
    x = np.where(i % 32 == 0, 120.0, ((i % 7) - 3.0) * 0.2).astype(np.float32).reshape(1, n_in)
    fused = orc.iq_matmul(tid, raw, x, 1, n_in)
    ref = x.astype(np.float64) @ w.T.astype(np.float64)
    rel = np.sqrt(((fused - ref) ** 2).sum() / (ref ** 2).sum())
    assert rel <= 0.05


@pytest.mark.parametrize("name", IQ_NAMES)
def test_mt_variant_bit_identical(name):
    f, raw, _ = load_iq_fixture(name)
    tid, nb = f["ggml_type_id"], f["n_blocks"]
    aq, ad = orc.quantize_q8_1(ramp(4, 256, 23, 11, 0.03))
    a = orc.iq_mmvq_q8(tid, raw, aq, ad, nb, 256)
    b = orc.iq_mmvq_q8(tid, raw, aq, ad, nb, 256, threads=True)
    assert np.array_equal(a, b)

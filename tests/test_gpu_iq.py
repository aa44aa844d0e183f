"""GPU parity of the I-quant kernels against the CPU oracle and the reference's
golden vectors; restates gguf/iq/cuda_tests.rs and adds bit-exact integer checks."""
import numpy as np
import pytest

from tests.helpers import IQ_NAMES, load_iq_fixture, ramp, random_iq_blocks, tile_blocks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import vllm_rust_amd as vmx
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    vmx.init(0)
    return vmx, torch


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel_l2(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-12)))


@pytest.mark.parametrize("name", IQ_NAMES)
def test_dequant_matches_golden_and_oracle(gpu, name):
    from oracle import oracle as orc
    vmx, torch = gpu
    f, raw, gold = load_iq_fixture(name)
    tid = f["ggml_type_id"]
    out = vmx.iq_dequantize(tid, _dev(torch, raw), f["n_blocks"]).cpu().numpy()
    assert np.abs(out - gold).max() <= 1e-4            # cuda_tests.rs:51
    assert np.array_equal(out, orc.iq_dequant(tid, raw))   # same f32 expression order
    # random valid blocks, many of them (grid-stride path)
    rng = np.random.default_rng(tid)
    rb = random_iq_blocks(rng, f["type_size"], 5000)
    out = vmx.iq_dequantize(tid, _dev(torch, rb), 5000).cpu().numpy()
    ref = orc.iq_dequant(tid, rb)
    assert np.array_equal(out, ref)


def test_quantize_q8_1_bit_exact(gpu):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, 4096)).astype(np.float32)
    x[1, 64:96] = 0.0                                   # all-zero block
    x[2, :32] = np.float32(120.0) * (np.arange(32) % 2)  # outliers
    x[3, 5] = 1e-30
    aq, ad = vmx.quantize_q8_1(_dev(torch, x))
    raq, rad = orc.quantize_q8_1(x)
    assert np.array_equal(ad.cpu().numpy(), rad)
    assert np.array_equal(aq.cpu().numpy(), raq)


@pytest.mark.parametrize("name", IQ_NAMES)
def test_integer_subdots_bit_exact(gpu, name):
    from oracle import oracle as orc
    vmx, torch = gpu
    f, raw, _ = load_iq_fixture(name)
    tid, ts, nb = f["ggml_type_id"], f["type_size"], f["n_blocks"]
    rng = np.random.default_rng(7)
    for w_raw, n_out, n_in in [(raw, nb, 256), (random_iq_blocks(rng, ts, 37 * 3), 37, 768)]:
        for m in (1, 3):
            x = (rng.standard_normal((m, n_in)) * 2).astype(np.float32)
            aq, ad = orc.quantize_q8_1(x)
            got = vmx.iq_subdots(tid, _dev(torch, w_raw), _dev(torch, aq), n_out, n_in).cpu().numpy()
            assert np.array_equal(got, orc.iq_subdots(tid, w_raw, aq, n_out, n_in))


@pytest.mark.parametrize("name", IQ_NAMES)
def test_mmvq_matches_cpu_mmvq_fixture(gpu, name):
    """cuda_tests.rs:61-107: [n_blocks, 256] weight, m in {1,4,16}, rel-L2 <= 1e-3."""
    from oracle import oracle as orc
    vmx, torch = gpu
    f, raw, _ = load_iq_fixture(name)
    tid, nb = f["ggml_type_id"], f["n_blocks"]
    w = _dev(torch, raw)
    for m in (1, 4, 16):
        x = ramp(m, 256, 23, 11, 0.03)
        got = vmx.iq_matmul(tid, w, _dev(torch, x), nb, 256).cpu().numpy()
        ref = orc.iq_matmul(tid, raw, x, nb, 256)
        assert (ref.astype(np.float64) ** 2).sum() > 1e-6
        assert rel_l2(got, ref) <= 1e-3, (name, m)


def test_mmvq_multi_superblock(gpu):
    """cuda_tests.rs:115-176: tiled IQ3_XXS fixture at nb256 = 1,2,3,6,8,16."""
    from oracle import oracle as orc
    vmx, torch = gpu
    f, raw, _ = load_iq_fixture("iq3_xxs")
    tid, ts, nb = f["ggml_type_id"], f["type_size"], f["n_blocks"]
    for n_out, n_in in [(6, 256), (3, 512), (2, 768), (1, 1536), (1, 2048), (1, 4096)]:
        w_raw = tile_blocks(raw, ts, nb, n_out * n_in // 256)
        for m in (1, 4, 16):
            x = ramp(m, n_in, 23, 11, 0.03)
            got = vmx.iq_matmul(tid, _dev(torch, w_raw), _dev(torch, x), n_out, n_in).cpu().numpy()
            ref = orc.iq_matmul(tid, w_raw, x, n_out, n_in)
            assert rel_l2(got, ref) <= 1e-3, (n_in, m)


@pytest.mark.parametrize("name", IQ_NAMES)
@pytest.mark.parametrize("m", [1, 2, 5, 8, 13, 16])
def test_mmvq_random_blocks_all_m(gpu, name, m):
    """Ragged n_out (not a multiple of the rows per workgroup), Gemma width 3840."""
    from oracle import oracle as orc
    vmx, torch = gpu
    tid = orc.IQ_TYPES[name]; ts = orc.IQ_TYPE_SIZE[tid]
    rng = np.random.default_rng(100 + m)
    n_out, n_in = 203, 3840
    w_raw = random_iq_blocks(rng, ts, n_out * n_in // 256)
    x = rng.standard_normal((m, n_in)).astype(np.float32)
    got = vmx.iq_matmul(tid, _dev(torch, w_raw), _dev(torch, x), n_out, n_in).cpu().numpy()
    ref = orc.iq_matmul(tid, w_raw, x, n_out, n_in)
    assert rel_l2(got, ref) <= 1e-3


def test_mmvq_full_size_rows_subset(gpu):
    """Bench shape 14336 x 4096 (benches/iq_dequant_bench.rs:27): rows are independent,
    so a random subset of output rows is checked against the oracle run on those rows."""
    from oracle import oracle as orc
    vmx, torch = gpu
    tid, ts = 18, 98
    rng = np.random.default_rng(5)
    n_out, n_in = 14336, 4096
    w_raw = random_iq_blocks(rng, ts, n_out * n_in // 256)
    x = rng.standard_normal((1, n_in)).astype(np.float32)
    got = vmx.iq_matmul(tid, _dev(torch, w_raw), _dev(torch, x), n_out, n_in).cpu().numpy()
    rows = np.sort(rng.choice(n_out, 64, replace=False)); rows[0] = 0; rows[-1] = n_out - 1
    row_bytes = n_in // 256 * ts
    sub = np.concatenate([w_raw[r * row_bytes:(r + 1) * row_bytes] for r in rows])
    ref = orc.iq_matmul(tid, sub, x, len(rows), n_in)
    assert rel_l2(got[:, rows], ref) <= 1e-3
    # linearity in the (pre-quantised) activations' scales: y(2*ad) == 2*y(ad) exactly
    aq, ad = vmx.quantize_q8_1(_dev(torch, x))
    y1 = vmx.iq_mmvq(tid, _dev(torch, w_raw), aq, ad, n_out, n_in)
    y2 = vmx.iq_mmvq(tid, _dev(torch, w_raw), aq, ad * 2, n_out, n_in)
    assert torch.equal(y2, y1 * 2)


def test_preconditions_surface_as_errors(gpu):
    vmx, torch = gpu
    from vllm_rust_amd._lib import VmxError
    w = torch.zeros(98 * 16, dtype=torch.uint8, device="cuda")
    with pytest.raises(VmxError, match="exceeds IQ_GEMV_MAX_M"):
        vmx.iq_matmul(18, w, torch.zeros((17, 256), device="cuda"), 1, 256)
    with pytest.raises(VmxError, match="not a multiple of 256"):
        vmx.iq_matmul(18, w, torch.zeros((1, 320), device="cuda"), 1, 320)
    with pytest.raises(VmxError, match="unsupported ggml type"):
        vmx.iq_dequantize(12, w, 1)


def test_gather_dequant_q6k(gpu):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(9)
    rows, dim = 50, 768
    blk = rng.integers(0, 256, size=(rows * dim // 256, 210), dtype=np.uint8)
    blk[:, 209] = rng.integers(0x2C, 0x34, size=blk.shape[0]); blk[:, 208] = 0
    table = orc.q6k_dequant(blk.reshape(-1)).reshape(rows, dim)
    ids = np.array([3, 49, 0, 77, -1, 3], np.int32)
    got = vmx.gather_dequant_q6k(_dev(torch, blk.reshape(-1)), _dev(torch, ids), dim, rows).cpu().numpy()
    for i, t in enumerate(ids):
        exp = table[t] if 0 <= t < rows else np.zeros(dim, np.float32)
        assert np.array_equal(got[i], exp)      # gguf/mod.rs:2112-2154: gather == full dequant, bit-exact


PACKED_TYPES = [("iq3_xxs", 18, 98), ("iq3_s", 21, 110), ("iq2_s", 22, 82)]


@pytest.mark.parametrize("name,tid,ts", PACKED_TYPES)
@pytest.mark.parametrize("n_out,n_in", [(6, 256), (13, 768), (64, 3840), (40, 15360), (24, 11008), (8, 4096), (33, 14336), (2050, 512)])
def test_packed_mmvq_subdots_bit_exact_and_sums(gpu, name, tid, ts, n_out, n_in):
    """The load-time repacked stream (8-row x K-slice units, wave-ordered fields): integer sub-dots bit for bit those of the
    CPU oracle, row sums within the reference's MMVQ gate (rel-L2 1e-3, cuda_tests.rs:100-104).  Shapes: one super-block,
    ragged row counts, Gemma's 3840 (60 of 64 lanes live) and 15360 (4 equal slices), 11008 (3 uneven slices), 14336, and
    more row groups than one workgroup holds."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(n_out + n_in + tid)
    if n_in == 256 and n_out == 6:
        _, w_raw, _ = load_iq_fixture(name)      # the reference's real blocks
    else:
        w_raw = random_iq_blocks(rng, ts, n_out * (n_in // 256))
    x = (rng.standard_normal((1, n_in)) * 2).astype(np.float32)
    aq, ad = orc.quantize_q8_1(x)
    wd = _dev(torch, w_raw)
    packed = vmx.iq_pack(tid, wd, n_out, n_in)
    assert packed.numel() == (n_out + 7) // 8 * 8 * (n_in // 256) * ts
    # a permutation of the same bytes (plus zero rows up to a multiple of 8)
    assert np.array_equal(np.sort(packed.cpu().numpy()[: (n_out // 8) * 8 * (n_in // 256) * ts]),
                          np.sort(w_raw[: (n_out // 8) * 8 * (n_in // 256) * ts]))
    aqd, add = _dev(torch, aq), _dev(torch, ad)
    got_sub = vmx.iq_subdots_packed(tid, packed, aqd, n_out, n_in).cpu().numpy()
    assert np.array_equal(got_sub, orc.iq_subdots(tid, w_raw, aq, n_out, n_in))
    got = vmx.iq_mmvq_packed(tid, packed, aqd, add, n_out, n_in).cpu().numpy()
    ref = orc.iq_matmul(tid, w_raw, x, n_out, n_in)
    assert rel_l2(got, ref) <= 1e-3
    # and against the raw-layout kernel on the same q8_1 input
    raw_y = vmx.iq_mmvq(tid, wd, aqd, add, n_out, n_in).cpu().numpy()
    assert np.abs(got - raw_y).max() <= 1e-5 * np.abs(raw_y).max() + 1e-6


@pytest.mark.parametrize("name,tid,ts", PACKED_TYPES)
@pytest.mark.parametrize("m,n_out,n_in", [(16, 6, 256), (2, 13, 768), (16, 64, 3840), (5, 40, 15360), (16, 33, 11008), (9, 2050, 512), (1, 70, 1024),
                                           (16, 600, 4096)])
def test_mma_mmvq_subdots_bit_exact_and_sums(gpu, name, tid, ts, m, n_out, n_in):
    """2 .. 16 activation rows on the int8 matrix cores over the 16-row-tile form: integer sub-dots of every (row, activation
    row, sub-block) bit for bit those of the CPU oracle, sums within the reference's MMVQ gate (rel-L2 1e-3,
    cuda_tests.rs:100-104, M in {1, 4, 16}) and equal to the per-row kernel's up to f32 summation order.  Shapes: one
    super-block (the reference's real blocks), ragged row counts (not a multiple of 16 or 32), Gemma's 3840 and 15360, uneven
    K slices, more rows than one workgroup holds."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(m + n_out + n_in + tid)
    if n_in == 256 and n_out == 6:
        _, w_raw, _ = load_iq_fixture(name)
    else:
        w_raw = random_iq_blocks(rng, ts, n_out * (n_in // 256))
    x = (rng.standard_normal((m, n_in)) * 2).astype(np.float32)
    x[0, :64] = 0.0                                             # all-zero q8_1 blocks
    if m > 2:
        x[2, 5] = 300.0                                         # an outlier: the rest of its block quantises to 0 / +-1
    aq, ad = orc.quantize_q8_1(x)
    wd = _dev(torch, w_raw)
    packed = vmx.iq_mma_pack(tid, wd, n_out, n_in)
    assert packed.numel() == (n_out + 15) // 16 * 16 * (n_in // 256) * ts
    full = (n_out // 16) * 16 * (n_in // 256) * ts              # a permutation of the same bytes (plus zero rows up to a multiple of 16)
    assert np.array_equal(np.sort(packed.cpu().numpy()[:full]), np.sort(w_raw[:full]))
    aqd, add = _dev(torch, aq), _dev(torch, ad)
    got_sub = vmx.iq_subdots_mma(tid, packed, aqd, add, n_out, n_in).cpu().numpy()
    assert np.array_equal(got_sub, orc.iq_subdots(tid, w_raw, aq, n_out, n_in))
    got = vmx.iq_mmvq_mma(tid, packed, aqd, add, n_out, n_in).cpu().numpy()
    ref = orc.iq_mmvq_q8(tid, w_raw, aq, ad, n_out, n_in)
    assert np.isfinite(got).all() and rel_l2(got, ref) <= 1e-3
    raw_y = vmx.iq_mmvq(tid, wd, aqd, add, n_out, n_in).cpu().numpy()
    assert np.abs(got - raw_y).max() <= 1e-5 * np.abs(raw_y).max() + 1e-6


def test_packed_rejects_types_without_packed_form(gpu):
    from vllm_rust_amd._lib import lib
    assert lib.vmx_iq_packed_bytes(17, 8, 256) == 0 and lib.vmx_iq_packed_bytes(18, 9, 512) == 16 * 2 * 98
    assert lib.vmx_iq_pack(23, 1, 1, 8, 256, None) == -1
    assert b"packed form" in lib.vmx_last_error()


@pytest.mark.parametrize("name", IQ_NAMES)
@pytest.mark.parametrize("m,n_out,n_in", [(17, 203, 3840), (64, 512, 256), (130, 300, 1024), (300, 64, 512), (256, 1024, 2048)])
def test_iq_gemm_prompt_sized_rows(gpu, name, m, n_out, n_in):
    """More than 16 rows: the reference dequantises the weight to dense f32 and multiplies the unquantised activations
    (gguf/mod.rs:358-365).  Checked against the oracle's dequantised weight times the bf16 activations in float64.  IQ2_XS,
    IQ2_S, IQ3_XXS and IQ3_S run the exact one-pass form (fp16 integer x f32 super-block scale: only f32 summation order
    differs); IQ4_XS, and any call whose workspace cannot also hold the fp16 copy of the rows (m > n_out here), the two-term
    bf16 split (2^-17 per weight).  1e-3 is the stated bar; measured ~1e-6."""
    from oracle import oracle as orc
    vmx, torch = gpu
    tid = orc.IQ_TYPES[name]; ts = orc.IQ_TYPE_SIZE[tid]
    rng = np.random.default_rng(7 * m + n_out)
    w_raw = random_iq_blocks(rng, ts, n_out * n_in // 256)
    x = orc.f32_to_bf16(rng.standard_normal((m, n_in)).astype(np.float32))
    xt = torch.from_numpy(x.view(np.int16)).cuda().view(torch.bfloat16)
    got = vmx.iq_gemm(tid, _dev(torch, w_raw), xt, n_out).cpu().numpy()
    w = orc.iq_dequant(tid, w_raw).reshape(n_out, n_in).astype(np.float64)
    ref = orc.bf16_to_f32(x).astype(np.float64) @ w.T
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) <= 2e-5
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()


@pytest.mark.parametrize("name,tid,ts", PACKED_TYPES)
@pytest.mark.parametrize("inter,n_in", [(64, 256), (192, 3840), (15360, 3840), (1024, 4096)])
def test_packed_gate_up_geglu_fused_matches_separate_launches(gpu, name, tid, ts, inter, n_in):
    """gate|up MMVQ + GeGLU + q8_1 in one launch: the activation bits are those of the two separate packed launches followed by
    the GeGLU of Gemma's MLP (bf16 roundings as in gemma4_quantized.rs:1390-1400, restated in tests/model_oracle.py), and its
    q8_1 form is bit for bit the oracle quantiser's on that activation (iq/mod.rs:663-682).  15360 x 3840 is the 12B layer."""
    from oracle import oracle as orc
    from scipy.special import erf
    vmx, torch = gpu
    rng = np.random.default_rng(inter + n_in + tid)
    wg_raw = random_iq_blocks(rng, ts, inter * (n_in // 256))
    wu_raw = random_iq_blocks(rng, ts, inter * (n_in // 256))
    x = (rng.standard_normal((1, n_in)) * 0.02).astype(np.float32)   # gate values of order one, where gelu bends
    aq, ad = orc.quantize_q8_1(x)
    aqd, add = _dev(torch, aq), _dev(torch, ad)
    pg = vmx.iq_pack(tid, _dev(torch, wg_raw), inter, n_in)
    pu = vmx.iq_pack(tid, _dev(torch, wu_raw), inter, n_in)
    gate = vmx.iq_mmvq_packed(tid, pg, aqd, add, inter, n_in).cpu().numpy().reshape(-1)
    up = vmx.iq_mmvq_packed(tid, pu, aqd, add, inter, n_in).cpu().numpy().reshape(-1)
    act, act_q, act_d = vmx.iq_mmvq_packed_geglu(tid, pg, pu, aqd, add, inter, n_in)
    torch.cuda.synchronize()
    act = act.cpu().numpy().view(np.uint16)
    g = orc.bf16_to_f32(orc.f32_to_bf16(gate))
    u = orc.bf16_to_f32(orc.f32_to_bf16(up))
    # gelu_erf with f32 steps (1 + erf underflows to 0 below about -5.5, as in the kernel and in the reference's f32 erf)
    e32 = erf((g * np.float32(0.70710678118654752)).astype(np.float64)).astype(np.float32)
    ge = orc.bf16_to_f32(orc.f32_to_bf16((np.float32(0.5) * g) * (np.float32(1.0) + e32)))
    want = orc.f32_to_bf16((ge * u).astype(np.float32))
    # erff on the GPU against double erf: an absolute error of ~2e-7 in erf is 0.5 |g| 2e-7 in gelu (1 + erf cancels for
    # negative g), on top of one bf16 ulp each for the gelu rounding and the product rounding
    gd, ud = g.astype(np.float64), u.astype(np.float64)
    got_f, want_f = orc.bf16_to_f32(act).astype(np.float64), orc.bf16_to_f32(want).astype(np.float64)
    tol = np.abs(ud) * (2.0 ** -8 * np.abs(ge.astype(np.float64)) + np.abs(gd) * 2e-7) * 1.01 + 2.0 ** -8 * np.abs(want_f) + 1e-30
    bad = np.abs(got_f - want_f) > tol
    assert not bad.any(), (np.flatnonzero(bad)[:4], gd[bad][:4], ud[bad][:4], got_f[bad][:4], want_f[bad][:4])
    assert (act != want).mean() <= 0.02
    # q8_1 of exactly the activation the kernel wrote
    wq, wd = orc.quantize_q8_1(orc.bf16_to_f32(act).reshape(1, inter))
    assert np.array_equal(act_q.cpu().numpy().reshape(-1), wq.reshape(-1))
    assert np.array_equal(act_d.cpu().numpy().reshape(-1), wd.reshape(-1))

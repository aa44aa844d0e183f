"""K-quant / legacy GGML blocks: GPU dequant == CPU oracle (ggml loop restatement) bit for bit, GEMV vs float64
product of the dequantised weight.  No golden vectors exist for these formats in the reference (third-party
candle arithmetic): parity unpinned."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TYPES = {2: "Q4_0", 3: "Q4_1", 6: "Q5_0", 7: "Q5_1", 8: "Q8_0", 10: "Q2_K", 11: "Q3_K", 12: "Q4_K", 13: "Q5_K", 14: "Q6_K"}
D_OFF = {2: [0], 3: [0, 2], 6: [0], 7: [0, 2], 8: [0], 10: [80, 82], 11: [108], 12: [0, 2], 13: [0, 2], 14: [208]}


@pytest.fixture(scope="module")
def gpu():
    import torch
    import vllm_rust_amd as vmx
    assert torch.cuda.is_available()
    vmx.init(0)
    return vmx, torch


def _blocks(rng, t, n):
    from oracle import oracle as orc
    ts, _ = orc.GGML_BLOCK[t]
    b = rng.integers(0, 256, size=(n, ts), dtype=np.uint8)
    for off in D_OFF[t]:       # finite, small f16 scales
        b[:, off] = 0
        b[:, off + 1] = rng.integers(0x24, 0x2C, size=n)
    return b.reshape(-1)


@pytest.mark.parametrize("t", sorted(TYPES))
def test_dequant_bit_exact_and_gemv(gpu, t):
    from oracle import oracle as orc
    from vllm_rust_amd._lib import lib, check
    vmx, torch = gpu
    rng = np.random.default_rng(t)
    ts, qk = orc.GGML_BLOCK[t]
    assert lib.vmx_ggml_type_size(t) == ts and lib.vmx_ggml_block_elems(t) == qk
    n_out, n_in = 77, 1024
    raw = _blocks(rng, t, n_out * n_in // qk)
    ref = orc.ggml_dequant(t, raw)
    w = torch.from_numpy(raw).cuda()
    out = torch.empty(ref.size, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    check(lib.vmx_kquant_dequantize_f32(t, w.data_ptr(), out.data_ptr(), raw.size // ts, st))
    assert np.array_equal(out.cpu().numpy(), ref)
    for m in (1, 3, 8):
        x = rng.standard_normal((m, n_in)).astype(np.float32)
        y = torch.empty((m, n_out), dtype=torch.float32, device="cuda")
        xd = torch.from_numpy(x).cuda()
        check(lib.vmx_kquant_gemv_f32(t, w.data_ptr(), xd.data_ptr(), y.data_ptr(), n_out, n_in, m, st))
        exp = x.astype(np.float64) @ ref.reshape(n_out, n_in).astype(np.float64).T
        got = y.cpu().numpy()
        assert np.sqrt(((got - exp) ** 2).sum() / (exp ** 2).sum()) <= 1e-5
    # past 8 rows: dense product of the dequantised weight with the f32 activations (ragged row counts, both tile forms)
    for m in (9, 70, 200):
        x = rng.standard_normal((m, n_in)).astype(np.float32)
        y = torch.full((m, n_out), 7.0, dtype=torch.float32, device="cuda")
        xd = torch.from_numpy(x).cuda()
        need = lib.vmx_kquant_gemm_workspace_bytes(m, n_out, n_in)
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        check(lib.vmx_kquant_gemm_f32(t, w.data_ptr(), xd.data_ptr(), y.data_ptr(), m, n_out, n_in, ws.data_ptr(), need, st))
        exp = x.astype(np.float64) @ ref.reshape(n_out, n_in).astype(np.float64).T
        got = y.cpu().numpy()
        assert np.sqrt(((got - exp) ** 2).sum() / (exp ** 2).sum()) <= 1e-5
        assert np.abs(got - exp).max() <= 2e-5 * np.abs(exp).max()
        assert lib.vmx_kquant_gemm_f32(t, w.data_ptr(), xd.data_ptr(), y.data_ptr(), m, n_out, n_in, ws.data_ptr(), need - 1, st) != 0
    assert lib.vmx_kquant_gemm_f32(18, w.data_ptr(), xd.data_ptr(), y.data_ptr(), 9, n_out, n_in, ws.data_ptr(), need, st) != 0


def test_q6k_matches_reference_gather_kernel_path(gpu):
    """The Q6_K formula is the one the reference restates in-tree (gather_dequant.cu): the generic dequant and the
    gather entry point must agree exactly."""
    from vllm_rust_amd._lib import lib, check
    vmx, torch = gpu
    rng = np.random.default_rng(3)
    rows, dim = 10, 512
    raw = _blocks(rng, 14, rows * dim // 256)
    w = torch.from_numpy(raw).cuda()
    full = torch.empty(rows * dim, dtype=torch.float32, device="cuda")
    check(lib.vmx_kquant_dequantize_f32(14, w.data_ptr(), full.data_ptr(), rows * dim // 256, torch.cuda.current_stream().cuda_stream))
    ids = torch.arange(rows, dtype=torch.int32, device="cuda")
    g = vmx.gather_dequant_q6k(w, ids, dim, rows)
    assert torch.equal(g.reshape(-1), full)


@pytest.mark.parametrize("n_out,n_in", [(1003, 3840), (37, 256), (4100, 2048), (9, 4096)])
def test_q3k_lm_head_gemv(gpu, n_out, n_in):
    """The one-row Q3_K kernel (tied lm_head shape: rows of 15 super-blocks, ragged row counts) against the float64
    product of the oracle's dequantised table."""
    from oracle import oracle as orc
    from vllm_rust_amd._lib import lib, check
    vmx, torch = gpu
    rng = np.random.default_rng(n_out + n_in)
    raw = _blocks(rng, 11, n_out * n_in // 256)
    ref = orc.ggml_dequant(11, raw).reshape(n_out, n_in)
    x = rng.standard_normal((1, n_in)).astype(np.float32)
    w, xd = torch.from_numpy(raw).cuda(), torch.from_numpy(x).cuda()
    y = torch.full((1, n_out + 8), 7.0, dtype=torch.float32, device="cuda")
    check(lib.vmx_kquant_gemv_f32(11, w.data_ptr(), xd.data_ptr(), y.data_ptr(), n_out, n_in, 1, torch.cuda.current_stream().cuda_stream))
    got = y.cpu().numpy()[0]
    exp = (x.astype(np.float64) @ ref.astype(np.float64).T)[0]
    assert np.all(got[n_out:] == 7.0)                      # nothing written past the last row
    assert np.sqrt(((got[:n_out] - exp) ** 2).sum() / (exp ** 2).sum()) <= 1e-5


@pytest.mark.parametrize("n_out,n_in,m", [(1003, 3840, 16), (37, 256, 2), (4100, 2048, 5), (9, 4096, 13), (70000, 256, 16), (20000, 3840, 7), (16, 512, 1)])
def test_q3k_rows_on_the_matrix_cores(gpu, n_out, n_in, m):
    """Up to 16 bf16 activation rows over a Q3_K table in one pass (integer part exact on the matrix cores, d * scale on
    the f32 group sums) against the float64 product of the oracle's dequantised table; ragged row counts, rows past m and
    columns past n_out untouched; the one-row kernel's answer on the same rows agrees to f32 summation order."""
    from oracle import oracle as orc
    from vllm_rust_amd._lib import lib, check
    vmx, torch = gpu
    rng = np.random.default_rng(n_out + n_in + m)
    raw = _blocks(rng, 11, n_out * n_in // 256)
    ref = orc.ggml_dequant(11, raw).reshape(n_out, n_in)
    xb = torch.from_numpy(rng.standard_normal((m, n_in)).astype(np.float32)).to(torch.bfloat16)
    x = xb.float().numpy()
    w, xd = torch.from_numpy(raw).cuda(), xb.cuda()
    y = torch.full((m * n_out + 8,), 7.0, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    check(lib.vmx_q3k_rows16_bf16(w.data_ptr(), xd.data_ptr(), y.data_ptr(), n_out, n_in, m, st))
    got = y.cpu().numpy()
    assert np.all(got[m * n_out:] == 7.0)
    got = got[:m * n_out].reshape(m, n_out)
    exp = x.astype(np.float64) @ ref.astype(np.float64).T
    assert np.sqrt(((got - exp) ** 2).sum() / (exp ** 2).sum()) <= 1e-5
    assert np.abs(got - exp).max() <= 2e-5 * np.abs(exp).max()
    y1 = torch.empty((1, n_out), dtype=torch.float32, device="cuda")
    x1 = xb[m - 1:m].float().cuda()
    check(lib.vmx_kquant_gemv_f32(11, w.data_ptr(), x1.data_ptr(), y1.data_ptr(), n_out, n_in, 1, st))
    assert np.abs(y1.cpu().numpy()[0] - got[m - 1]).max() <= 2e-5 * np.abs(exp).max()
    assert lib.vmx_q3k_rows16_bf16(w.data_ptr(), xd.data_ptr(), y.data_ptr(), n_out, n_in, 17, st) != 0
    assert lib.vmx_q3k_rows16_bf16(w.data_ptr(), xd.data_ptr(), y.data_ptr(), n_out, 8192, m, st) != 0
    assert lib.vmx_q3k_rows16_bf16(w.data_ptr(), xd.data_ptr(), y.data_ptr(), n_out, 384, m, st) != 0

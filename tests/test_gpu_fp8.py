"""GPU parity of the FP8 (e4m3) weight path against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import vllm_rust_amd as vmx
    assert torch.cuda.is_available()
    vmx.init(0)
    return vmx, torch


def _bf(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).cuda().view(torch.bfloat16)


def _u16(torch, t):
    return t.cpu().view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize("M,N,K,per_channel,bias", [(1, 4096, 4096, False, False), (3, 1000, 2048, True, True), (16, 515, 1024, True, False)])
def test_fp8_gemm(gpu, M, N, K, per_channel, bias):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(M + N)
    W = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    W[(W & 0x7F) == 0x7F] = 0x38                      # no NaN encodings in weights
    x = orc.f32_to_bf16(rng.standard_normal((M, K)).astype(np.float32))
    sc = (rng.random(N if per_channel else 1).astype(np.float32) + 0.5) * 1e-3
    b = orc.f32_to_bf16(rng.standard_normal(N).astype(np.float32)) if bias else None
    got = orc.bf16_to_f32(_u16(torch, vmx.fp8_gemm(_bf(torch, x), torch.from_numpy(W).cuda(), torch.from_numpy(sc).cuda(),
                                                  _bf(torch, b) if bias else None)))
    ref = orc.bf16_to_f32(orc.fp8_gemm_bf16(x, W, sc, b, per_channel))
    assert np.abs(got - ref).max() <= (2 ** -7 + 1e-3) * np.abs(ref).max()
    rel = np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum())
    assert rel <= 1e-3 + 2 ** -8


def test_fp8_quant_dequant_byte_exact(gpu):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(3)
    x = orc.f32_to_bf16((rng.standard_normal((5, 4096)) * np.array([[0.01], [1.0], [30.0], [1000.0], [0.0]])).astype(np.float32))
    q, s = vmx.fp8_quant(_bf(torch, x))
    rq, rs = orc.fp8_quant_bf16(x, dynamic=True)
    assert np.array_equal(s.cpu().numpy(), rs)
    assert np.array_equal(q.cpu().numpy(), rq)
    st = np.array([0.05], np.float32)                  # static scale: values beyond 448*scale saturate
    q2, _ = vmx.fp8_quant(_bf(torch, x), torch.from_numpy(st).cuda())
    rq2, _ = orc.fp8_quant_bf16(x, st, dynamic=False)
    assert np.array_equal(q2.cpu().numpy(), rq2)
    assert (np.array([orc.fp8_e4m3_decode(int(v)) for v in rq2[3][:64]]).__abs__() <= 448).all()
    d = _u16(torch, vmx.fp8_dequant(q, s))
    assert np.array_equal(d, orc.fp8_dequant_bf16(rq, rs, True))


def _fp8_bytes(rng, shape):
    q = rng.integers(0, 256, size=shape, dtype=np.uint8)
    q[(q & 0x7F) == 0x7F] = 0x38                      # no NaN encodings
    return q


def _close(got, ref):
    assert np.abs(got - ref).max() <= (2 ** -7 + 1e-3) * np.abs(ref).max()
    assert np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum()) <= 1e-3 + 2 ** -8


@pytest.mark.parametrize("M,N,K", [(17, 515, 1024), (64, 4096, 4096), (130, 1000, 2064)])
def test_fp8_gemm_prompt_sized_m(gpu, M, N, K):
    """W8A16 beyond 16 rows: bf16 matrix cores over fp8 -> bf16 weights; ragged M, N and a K that is not a multiple of 64."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(M + N)
    W = _fp8_bytes(rng, (N, K))
    x = orc.f32_to_bf16(rng.standard_normal((M, K)).astype(np.float32))
    sc = (rng.random(N).astype(np.float32) + 0.5) * 1e-3
    b = orc.f32_to_bf16(rng.standard_normal(N).astype(np.float32))
    got = orc.bf16_to_f32(_u16(torch, vmx.fp8_gemm(_bf(torch, x), torch.from_numpy(W).cuda(), torch.from_numpy(sc).cuda(), _bf(torch, b))))
    _close(got, orc.bf16_to_f32(orc.fp8_gemm_bf16(x, W, sc, b, True)))


@pytest.mark.parametrize("M,N,K,a_rows,b_cols,bias,f32", [(1, 4096, 4096, False, False, False, False), (5, 1000, 2064, True, True, True, False),
                                                         (16, 515, 1024, True, False, False, True), (77, 2048, 4096, True, True, True, False),
                                                         (256, 1024, 1024, False, True, False, True)])
def test_fp8_scaled_mm_w8a8(gpu, M, N, K, a_rows, b_cols, bias, f32):
    """fp8_scaled_mm_{bf16,f32}: every scale mode, bias, ragged shapes, one row to prompt-sized M."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(M * 7 + N)
    A, B = _fp8_bytes(rng, (M, K)), _fp8_bytes(rng, (N, K))
    # keep magnitudes moderate: exponent field at most 9 (|v| < 8) so sums stay far from overflow
    A = (A & 0x87) | (np.minimum((A >> 3) & 0xF, 9) << 3).astype(np.uint8)
    B = (B & 0x87) | (np.minimum((B >> 3) & 0xF, 9) << 3).astype(np.uint8)
    sa = (rng.random(M if a_rows else 1).astype(np.float32) + 0.5) * 1e-2
    sb = (rng.random(N if b_cols else 1).astype(np.float32) + 0.5) * 1e-2
    b = orc.f32_to_bf16(rng.standard_normal(N).astype(np.float32) * 0.1) if bias else None
    out = vmx.fp8_scaled_mm(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), torch.from_numpy(sa).cuda(), torch.from_numpy(sb).cuda(),
                            _bf(torch, b) if bias else None, out_dtype=torch.float32 if f32 else torch.bfloat16)
    ref = orc.fp8_scaled_mm(A, B, sa, sb, b, a_rows, b_cols, f32)
    if f32:
        got, ref = out.cpu().numpy(), ref
        # f32 summation order only: both sums carry about sqrt(K) * 2^-24 of the sum of |terms|, which is ~30x the largest result here
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
    else:
        _close(orc.bf16_to_f32(_u16(torch, out)), orc.bf16_to_f32(ref))


def test_fp8_scaled_mm_rejects_bad_arguments(gpu):
    vmx, torch = gpu
    from vllm_rust_amd._lib import VmxError
    a = torch.zeros((2, 24), dtype=torch.uint8, device="cuda")
    s = torch.ones(1, device="cuda")
    with pytest.raises(VmxError, match="multiple of 16"):
        vmx.fp8_scaled_mm(a, a, s, s)


def test_fp8_block_scaled_weights(gpu):
    """Block-scaled weights: dequant with a [ceil(N/128), ceil(K/128)] scale grid (ragged last blocks) byte-exact; the linear on top of
    it against the oracle's bf16 matmul of the dequantised weights, bias added after the bf16 rounding."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(21)
    N, K, bn, bk = 300, 1040, 128, 128
    W = _fp8_bytes(rng, (N, K))
    sc = ((rng.random((-(-N // bn), -(-K // bk))).astype(np.float32) + 0.5) * 1e-3)
    ref_w = orc.fp8_dequant_block_bf16(W, sc, bn, bk)
    got_w = _u16(torch, vmx.fp8_dequant_block(torch.from_numpy(W).cuda(), torch.from_numpy(sc).cuda(), bn, bk))
    assert np.array_equal(got_w, ref_w)
    for M in (3, 40):
        x = orc.f32_to_bf16(rng.standard_normal((M, K)).astype(np.float32))
        b = orc.f32_to_bf16(rng.standard_normal(N).astype(np.float32))
        y = vmx.fp8_block_gemm(_bf(torch, x), torch.from_numpy(W).cuda(), torch.from_numpy(sc).cuda(), bn, bk, _bf(torch, b))
        mm = orc.bf16_to_f32(orc.dense_gemv_bf16(x, ref_w))                       # bf16 matmul, f32 accumulate
        ref = orc.bf16_to_f32(orc.f32_to_bf16(mm + orc.bf16_to_f32(b)[None, :]))    # bf16 broadcast add
        _close(orc.bf16_to_f32(_u16(torch, y)), ref)


def test_fused_rmsnorm_fp8(gpu):
    """fused_rmsnorm_fp8: per-row scales within f32 reduction-order noise of the oracle's, bytes equal except where that noise moves a
    value across a rounding boundary (then by one code), zero rows hit the 1e-12 floor."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(4)
    H = 4096
    x = orc.f32_to_bf16((rng.standard_normal((6, H)) * np.array([[0.01], [1.0], [30.0], [1000.0], [0.0], [3.0]])).astype(np.float32))
    w = orc.f32_to_bf16((1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32))
    q, s = vmx.fused_rmsnorm_fp8(_bf(torch, x), _bf(torch, w), 1e-6)
    rq, rs = orc.fused_rmsnorm_fp8(x, w, 1e-6)
    got_s, got_q = s.cpu().numpy(), q.cpu().numpy()
    assert np.allclose(got_s, rs, rtol=5e-6, atol=0) and got_s[4] == np.float32(1e-12)
    diff = got_q != rq
    assert diff.mean() <= 2e-3
    assert (np.abs(got_q[diff].astype(np.int16) - rq[diff].astype(np.int16)) <= 1).all()
    # the pair feeds the W8A8 GEMM as its activation operand
    B = _fp8_bytes(rng, (64, H))
    B = (B & 0x87) | (np.minimum((B >> 3) & 0xF, 9) << 3).astype(np.uint8)
    sb = np.array([0.01], np.float32)
    out = vmx.fp8_scaled_mm(q, torch.from_numpy(B).cuda(), s, torch.from_numpy(sb).cuda())
    ref = orc.fp8_scaled_mm(got_q, B, got_s, sb, None, True, False, False)
    _close(orc.bf16_to_f32(_u16(torch, out)), orc.bf16_to_f32(ref))

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

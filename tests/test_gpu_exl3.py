"""GPU parity of the EXL3 path against the CPU oracle: reconstruct and Hadamard bit-exact (integer bit extraction,
fp16 codebook arithmetic with single roundings), GEMM within fp16 rounding of the f32 sums."""
import numpy as np
import pytest

from tests.test_oracle_exl3 import lcg_trellis

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import vllm_rust_amd as vmx
    assert torch.cuda.is_available()
    vmx.init(0)
    return vmx, torch


def _d(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).cuda()


def _u16(torch, t):
    return t.cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("bits", [1, 2, 3, 4, 5, 6, 8])
@pytest.mark.parametrize("cb", [0, 1, 2])
def test_reconstruct_bit_exact(gpu, bits, cb):
    from oracle import oracle as orc
    from vllm_rust_amd._lib import lib, check
    vmx, torch = gpu
    K, N = 64, 128
    tr = lcg_trellis(K, N, bits, seed=0x12345678 + bits * 17 + cb)
    dense = torch.empty((K, N), dtype=torch.int16, device="cuda")
    t = _d(torch, tr)
    check(lib.vmx_exl3_reconstruct_f16(t.data_ptr(), dense.data_ptr(), K, N, bits, cb, torch.cuda.current_stream().cuda_stream))
    assert np.array_equal(_u16(torch, dense), orc.exl3_reconstruct(tr, K, N, bits, cb))


def test_zero_trellis_constant_and_hadamard(gpu):
    from oracle import oracle as orc
    from vllm_rust_amd._lib import lib, check
    vmx, torch = gpu
    st = torch.cuda.current_stream().cuda_stream
    z = torch.zeros((32 // 16) * (128 // 16) * 64, dtype=torch.int16, device="cuda")
    dense = torch.empty((32, 128), dtype=torch.int16, device="cuda")
    check(lib.vmx_exl3_reconstruct_f16(z.data_ptr(), dense.data_ptr(), 32, 128, 4, 0, st))
    w = _u16(torch, dense)
    assert (w == orc.exl3_decode(0, 0)).all()                         # exl3_cuda.rs:1397-1432
    rng = np.random.default_rng(0)
    x = orc.f32_to_f16(rng.standard_normal((5, 512)).astype(np.float32))
    sc = orc.f32_to_f16((rng.random(512) + 0.5).astype(np.float32) * rng.choice([-1, 1], 512))
    y = torch.empty((5, 512), dtype=torch.int16, device="cuda")
    xd, sd = _d(torch, x), _d(torch, sc)
    for mode in (0, 1, 2):
        check(lib.vmx_exl3_had_r_128_f16(y.data_ptr(), xd.data_ptr(), sd.data_ptr() if mode else None, 5, 512, mode, st))
        assert np.array_equal(_u16(torch, y), orc.exl3_had_fp16(x, sc if mode else None, mode)), mode


@pytest.mark.parametrize("M,K,N,bits,cb", [(2, 512, 128, 4, 0), (1, 4096, 1024, 4, 0), (16, 1024, 256, 3, 1), (5, 512, 384, 2, 2),
                                           (1, 2048, 512, 6, 0), (20, 256, 128, 8, 1)])
def test_gemm_matches_oracle(gpu, M, K, N, bits, cb):
    from oracle import oracle as orc
    from vllm_rust_amd._lib import lib, check
    vmx, torch = gpu
    rng = np.random.default_rng(M + K + bits)
    tr = lcg_trellis(K, N, bits, seed=0x12345678 + K + N)
    if (M, K, N, bits) == (2, 512, 128, 4):                          # the reference's test inputs (exl3_cuda.rs:1555-1577)
        x = orc.f32_to_f16((np.sin(np.arange(M * K) * np.float32(0.013)) * 0.05).astype(np.float32).reshape(M, K))
        suh = orc.f32_to_f16(np.ones(K, np.float32)); svh = orc.f32_to_f16(np.ones(N, np.float32))
    else:
        x = orc.f32_to_f16((rng.standard_normal((M, K)) * 0.05).astype(np.float32))
        suh = orc.f32_to_f16(rng.choice([-1.0, 1.0], K).astype(np.float32)); svh = orc.f32_to_f16(rng.choice([-1.0, 1.0], N).astype(np.float32))
    ws = torch.empty(lib.vmx_exl3_gemm_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")
    y = torch.empty((M, N), dtype=torch.int16, device="cuda")
    xd, td, sud, svd = _d(torch, x), _d(torch, tr), _d(torch, suh), _d(torch, svh)
    check(lib.vmx_exl3_gemm_f16(y.data_ptr(), xd.data_ptr(), td.data_ptr(), sud.data_ptr(), svd.data_ptr(), M, K, N, bits, cb,
                                ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
    got = orc.f16_to_f32(_u16(torch, y)); ref = orc.f16_to_f32(orc.exl3_gemm(x, tr, suh, svh, K, N, bits, cb))
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2 ** -9 * scale + 1e-6           # one fp16 ulp of the largest output + f32 re-association
    assert np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum()) <= 1e-3

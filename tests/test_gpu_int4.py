"""GPU parity of the AWQ/GPTQ int4 path: load-time repacks bit-exact, dequant
bit-exact, small-M GEMM within the reference test's tolerance
(marlin_cuda.rs:2032-2470) and a tighter one derived from bf16 output rounding."""
import numpy as np
import pytest

from tests.helpers import awq_lcg_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import vllm_rust_amd as vmx
    assert torch.cuda.is_available()
    vmx.init(0)
    return vmx, torch


def _dev(torch, a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    if a.dtype == np.uint16:
        a = a.view(np.int16)
    return torch.from_numpy(a).cuda()


def _u16(t):
    return t.cpu().view(__import__("torch").int16).numpy().view(np.uint16)


def test_awq_repacks_bit_exact(gpu):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(0)
    K, N = 512, 384
    awq = rng.integers(0, 2 ** 32, size=(K, N // 8), dtype=np.uint32)
    got = vmx.awq_to_gptq_qweight(_dev(torch, awq), K, N).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, orc.awq_to_gptq_qweight(awq, K, N))     # awq_marlin.rs:1005-1044
    z = rng.integers(0, 2 ** 32, size=(K // 128, N // 8), dtype=np.uint32)
    gz = vmx.awq_repack_nibbles(_dev(torch, z)).cpu().numpy().view(np.uint32)
    assert np.array_equal(gz, orc.repack_awq_nibbles(z))


def _random_case(rng, M, K, N, group, with_zeros=True):
    qw = rng.integers(0, 2 ** 32, size=(K // 8, N), dtype=np.uint32)
    qz = rng.integers(0, 2 ** 32, size=(K // group, N // 8), dtype=np.uint32) if with_zeros else None
    sc = ((rng.integers(0, 2000, size=(K // group, N)) - 1000) * 1e-4).astype(np.float32)
    x = rng.standard_normal((M, K)).astype(np.float32)
    return qw, qz, sc, x


@pytest.mark.parametrize("K,N,group", [(256, 128, 128), (4096, 1024, 128), (1024, 4096, 1024)])
def test_pack_then_dequant_bit_exact(gpu, K, N, group):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(K + N)
    for with_zeros in (True, False):
        qw, qz, sc, _ = _random_case(rng, 1, K, N, group, with_zeros)
        sc_b = orc.f32_to_bf16(sc)
        packed = vmx.int4_pack(_dev(torch, qw), _dev(torch, sc_b), _dev(torch, qz) if qz is not None else None, K, N, group)
        dense = _u16(vmx.int4_dequant(packed, K, N, group))
        ref = orc.f32_to_bf16(orc.int4_dequant(qw, orc.bf16_to_f32(sc_b), qz, K, N, group))
        assert np.array_equal(dense, ref)


def _check_gemm(vmx, torch, orc, qw, qz, sc, x, K, N, group, bias=None):
    sc_b, x_b = orc.f32_to_bf16(sc), orc.f32_to_bf16(x)
    b_b = orc.f32_to_bf16(bias) if bias is not None else None
    packed = vmx.int4_pack(_dev(torch, qw), _dev(torch, sc_b), _dev(torch, qz) if qz is not None else None, K, N, group)
    y = vmx.int4_gemm(_dev(torch, x_b).view(torch.bfloat16), packed, _dev(torch, b_b).view(torch.bfloat16) if b_b is not None else None,
                      N, K, group)
    got = orc.bf16_to_f32(_u16(y))
    exp_bf16 = orc.bf16_to_f32(orc.int4_gemv_bf16(x_b, qw, sc_b, qz, b_b, K, N, group))
    exp_f32 = orc.int4_gemv_f32(orc.bf16_to_f32(x_b), qw, orc.bf16_to_f32(sc_b), qz, K, N, group)
    if bias is not None:
        exp_f32 = exp_f32 + orc.bf16_to_f32(b_b)[None, :]
    # reference test tolerance
    tol = max(np.abs(exp_f32).max() * 0.01, 5e-2)
    assert np.abs(got - exp_f32).max() <= tol
    # tighter: one bf16 ulp of the result plus f32 re-association noise relative to sum|terms|
    scale = np.abs(exp_f32).max()
    assert np.abs(got - exp_bf16).max() <= 2 ** -7 * scale + 1e-6
    # north-star bound on the f32 accumulate (before the bf16 store): rel-L2 <= 1e-3 + bf16 rounding
    rel = np.sqrt(((got - exp_f32) ** 2).sum() / (exp_f32 ** 2).sum())
    assert rel <= 1e-3 + 2 ** -8


@pytest.mark.parametrize("m", [1, 4, 8, 12, 16])
def test_gemm_reference_lcg_cases(gpu, m):
    from oracle import oracle as orc
    vmx, torch = gpu
    qw, qz, sc, x = awq_lcg_case(m, 256, 128, 128)
    _check_gemm(vmx, torch, orc, qw, qz, sc, x, 256, 128, 128)


@pytest.mark.parametrize("M,K,N,group,zeros,bias", [
    (1, 4096, 4096, 128, True, False),     # Qwen3-8B q/o
    (1, 4096, 1024, 128, True, True),      # k/v
    (3, 12288, 4096, 128, True, False),    # down
    (16, 4096, 12288, 128, True, False),   # gate/up, full M tile
    (1, 1024, 3072, 128, True, False),     # Qwen3-0.6B
    (2, 4096, 14336, 128, False, False),   # Mixtral expert, GPTQ symmetric
    (5, 11008 // 128 * 128, 4096, 128, True, True),  # K/128 = 86, not a multiple of 4*UPW
    (1, 2048, 2048, 2048, True, False),    # one group spanning K
    (37, 1024, 2048, 128, True, True),     # prompt-sized M, ragged last row tile: 64 x 128 output tiles
    (100, 2048, 3072, 128, True, False),   # two row blocks, split-K slices summed
    (64, 12288, 1024, 128, False, True),   # long k, three slices, bias
    (40, 1024, 208, 128, True, False),     # N not a multiple of 128: passes of 16 rows
    # 5..16 rows with K a whole number of 512-deep passes: the ring form (tiles staged ahead through registers and LDS)
    (5, 4096, 6144, 128, True, True),      # q|k|v: two k slices, three column tiles per workgroup, rows 5..15 of the tile idle
    (9, 12288, 4096, 128, True, False),    # down: three k slices, two column tiles per workgroup
    (13, 4096, 4112, 128, False, True),    # 257 column tiles: one per workgroup, symmetric, bias
    (16, 2048, 1024, 256, True, False),    # group of two k-blocks, short k (four passes), small grid
    (24, 4096, 2048, 128, True, True),     # 16 + 8 rows: two ring launches below the prompt kernel's threshold
    (48, 4096, 1024, 128, True, False),    # three ring launches, k slices summed per launch
])
def test_gemm_model_shapes(gpu, M, K, N, group, zeros, bias):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(M * 7 + K + N)
    qw, qz, sc, x = _random_case(rng, M, K, N, group, zeros)
    b = rng.standard_normal(N).astype(np.float32) if bias else None
    _check_gemm(vmx, torch, orc, qw, qz, sc, x, K, N, group, b)


def test_gemm_linearity_and_determinism(gpu):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(11)
    K, N = 4096, 4096
    qw, qz, sc, x = _random_case(rng, 1, K, N, 128)
    sc_b = orc.f32_to_bf16(sc)
    packed = vmx.int4_pack(_dev(torch, qw), _dev(torch, sc_b), _dev(torch, qz), K, N, 128)
    xb = _dev(torch, orc.f32_to_bf16(x)).view(torch.bfloat16)
    y1 = vmx.int4_gemm(xb, packed, None, N, K, 128)
    y2 = vmx.int4_gemm(xb, packed, None, N, K, 128)
    assert torch.equal(y1, y2)                            # no atomics: bitwise reproducible
    y4 = vmx.int4_gemm(xb * 4, packed, None, N, K, 128)   # power-of-two scaling is exact
    assert torch.equal(y4, y1 * 4)


def test_preconditions(gpu):
    vmx, torch = gpu
    from vllm_rust_amd._lib import VmxError
    z = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    x = torch.zeros((17, 256), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(VmxError, match="multiple of 128"):
        vmx.int4_gemm(x[:1, :200].contiguous(), z, None, 128, 200, 128)
    with pytest.raises(VmxError, match="multiple of 16"):
        vmx.int4_gemm(x[:1], z, None, 120, 256, 128)


@pytest.mark.parametrize("K,N,G,M", [(256, 128, 128, 1), (512, 200, 64, 5), (4096, 1024, 128, 16), (1024, 512, -1, 40)])
def test_gptq_int8_dequant_and_gemm(gpu, K, N, G, M):
    """GPTQ 8-bit: dequantised weights bit for bit the oracle's in both layouts (gptq_dequant.cu:128-170); y = x @ W + bias within a
    bf16 ulp of the float64 product of the same bf16 operands (the reference multiplies the dequantised matrix with cuBLAS: parity
    unpinned beyond that)."""
    from oracle import oracle as orc
    from vllm_rust_amd._lib import lib, check
    vmx, torch = gpu
    rng = np.random.default_rng(K + N + M)
    groups = K // G if G > 0 else 1
    qw = rng.integers(0, 2 ** 32, (K // 4, N), dtype=np.uint64).astype(np.uint32)
    qz = rng.integers(0, 2 ** 32, (groups, (N + 3) // 4), dtype=np.uint64).astype(np.uint32)
    sc = orc.f32_to_bf16((rng.random((groups, N)) * 0.01 + 0.0005).astype(np.float32))
    ref_w = orc.gptq_dequant_int8(qw, sc, qz, K, N, G)
    dq, dz, ds = (torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a.view(np.int16)).cuda() for a in (qw, qz, sc))
    out = torch.empty((K, N), dtype=torch.int16, device="cuda")
    check(lib.vmx_gptq_dequant_int8_bf16(out.data_ptr(), dq.data_ptr(), ds.data_ptr(), dz.data_ptr(), K, N, G, 0, None))
    assert np.array_equal(out.cpu().numpy().view(np.uint16), ref_w)
    out_t = torch.empty((N, K), dtype=torch.int16, device="cuda")
    check(lib.vmx_gptq_dequant_int8_bf16(out_t.data_ptr(), dq.data_ptr(), ds.data_ptr(), dz.data_ptr(), K, N, G, 1, None))
    assert np.array_equal(out_t.cpu().numpy().view(np.uint16), ref_w.T)
    x = orc.f32_to_bf16((rng.standard_normal((M, K)) * 0.5).astype(np.float32))
    bias = orc.f32_to_bf16(rng.standard_normal(N).astype(np.float32))
    y = torch.empty((M, N), dtype=torch.int16, device="cuda")
    ws = torch.empty((N, K), dtype=torch.int16, device="cuda")
    xd, bd = torch.from_numpy(x.view(np.int16)).cuda(), torch.from_numpy(bias.view(np.int16)).cuda()
    check(lib.vmx_gptq_int8_gemm_bf16(y.data_ptr(), xd.data_ptr(), dq.data_ptr(), ds.data_ptr(), dz.data_ptr(), bd.data_ptr(), M, N, K, G,
                                      ws.data_ptr(), None))
    torch.cuda.synchronize()
    got = orc.bf16_to_f32(y.cpu().numpy().view(np.uint16)).astype(np.float64)
    prod = orc.bf16_to_f32(x).astype(np.float64) @ orc.bf16_to_f32(ref_w).astype(np.float64)
    want = orc.bf16_to_f32(orc.f32_to_bf16(prod.astype(np.float32))).astype(np.float64) + orc.bf16_to_f32(bias).astype(np.float64)
    tol = 2.0 ** -7 * np.abs(want) + 2.0 ** -8 * np.abs(prod) + 1e-3 * np.abs(prod).mean() + 1e-6
    assert (np.abs(got - want) <= tol).all()
    assert lib.vmx_gptq_int8_gemm_bf16(y.data_ptr(), y.data_ptr(), dq.data_ptr(), ds.data_ptr(), dz.data_ptr(), None, 1, N, 6, G, ws.data_ptr(), None) != 0

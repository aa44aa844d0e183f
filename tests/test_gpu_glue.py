"""GPU parity of the glue operators and paged-attention decode against the CPU oracle."""
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


def _bf(torch, a_u16):
    return torch.from_numpy(np.ascontiguousarray(a_u16).view(np.int16)).cuda().view(torch.bfloat16)


def _u16(torch, t):
    return t.cpu().view(torch.int16).numpy().view(np.uint16)


def _close_bf16(got_u16, ref_u16, orc, ulps=1):
    g, r = orc.bf16_to_f32(got_u16), orc.bf16_to_f32(ref_u16)
    tol = np.maximum(np.abs(r), 1e-30) * (2.0 ** -7) * ulps + 1e-7
    bad = np.abs(g - r) > tol
    assert not bad.any(), (g[bad][:5], r[bad][:5])


@pytest.mark.parametrize("hidden", [1024, 3840, 4096])
def test_rms_norm_and_fused_add(gpu, hidden):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(hidden)
    x = orc.f32_to_bf16(rng.standard_normal((3, hidden)).astype(np.float32) * 3)
    w = orc.f32_to_bf16(rng.standard_normal(hidden).astype(np.float32))
    res = orc.f32_to_bf16(rng.standard_normal((3, hidden)).astype(np.float32))
    for plus_one in (False, True):
        got = _u16(torch, vmx.rms_norm(_bf(torch, x), _bf(torch, w), 1e-6, weight_plus_one=plus_one))
        _close_bf16(got, orc.rms_norm_bf16(x, w, 1e-6, plus_one), orc)
    r_dev = _bf(torch, res)
    got = _u16(torch, vmx.rms_norm(_bf(torch, x), _bf(torch, w), 1e-6, residual=r_dev))
    ref_out, ref_res = orc.fused_add_rms_norm_bf16(x, res, w, 1e-6)
    assert np.array_equal(_u16(torch, r_dev), ref_res)       # the add itself is bit-exact
    _close_bf16(got, ref_out, orc)


def test_rope_neox(gpu):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(2)
    T, H, KVH, D = 5, 32, 8, 128
    q = orc.f32_to_bf16(rng.standard_normal((T, H * D)).astype(np.float32))
    k = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32))
    pos = np.array([0, 3, 17, 255, 1000], np.int32)
    inv = 1.0 / (1e6 ** (np.arange(0, D, 2) / D))
    ang = np.arange(1024)[:, None] * inv[None, :]
    cs = np.concatenate([np.cos(ang), np.sin(ang)], -1).astype(np.float32)
    qd, kd = _bf(torch, q), _bf(torch, k)
    vmx.rope_neox(torch.from_numpy(pos).cuda(), qd, kd, torch.from_numpy(cs).cuda(), D, D, H, KVH)
    rq, rk = orc.rope_neox_bf16(pos, q, k, cs, D, D, H, KVH)
    _close_bf16(_u16(torch, qd), rq, orc)     # fma contraction may move the last bit
    _close_bf16(_u16(torch, kd), rk, orc)


def test_silu_and_mul(gpu):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(3)
    g = orc.f32_to_bf16(rng.standard_normal((2, 12288)).astype(np.float32) * 4)
    u = orc.f32_to_bf16(rng.standard_normal((2, 12288)).astype(np.float32))
    got = _u16(torch, vmx.silu_and_mul(_bf(torch, g), _bf(torch, u)))
    _close_bf16(got, orc.silu_and_mul_bf16(g, u), orc)


def test_argmax(gpu):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(4)
    for vocab in (151936, 262144, 1000, 7):
        lg = orc.f32_to_bf16(rng.standard_normal((3, vocab)).astype(np.float32))
        got = vmx.argmax(_bf(torch, lg)).cpu().numpy()
        ref = orc.argmax_bf16(lg)
        vals = orc.bf16_to_f32(lg)
        # bf16 logits tie often: compare VALUES (sampling.cu:20-30 leaves the index on ties open),
        # and our documented rule (smallest index) equals the oracle's first-max.
        assert np.array_equal(vals[np.arange(3), got], vals[np.arange(3), ref])
        assert np.array_equal(got, ref)
    neg = orc.f32_to_bf16(np.full((1, 100), -5.0, np.float32)); neg[0, 63] = orc.f32_to_bf16(np.float32(-1.0))
    assert vmx.argmax(_bf(torch, neg)).item() == 63


@pytest.mark.parametrize("kv_dtype", [0, 1, 2])
def test_reshape_and_cache(gpu, kv_dtype):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(5)
    KVH, D, BS, NB = 8, 128, 16, 6
    k = orc.f32_to_bf16(rng.standard_normal((4, KVH * D)).astype(np.float32) * 2)
    v = orc.f32_to_bf16(rng.standard_normal((4, KVH * D)).astype(np.float32) * 2)
    slots = np.array([5, -1, 40, 95], np.int32)
    dt = np.uint16 if kv_dtype == 0 else np.uint8
    kc_ref = np.zeros((NB, BS, KVH, D), dt); vc_ref = np.zeros_like(kc_ref)
    ks = float(np.abs(orc.bf16_to_f32(k)).max() * 1.5 / 448.0); vs = float(np.abs(orc.bf16_to_f32(v)).max() * 1.5 / 448.0)
    orc.reshape_and_cache(k, v, kc_ref, vc_ref, slots, KVH, D, BS, kv_dtype, ks, vs)
    tdt = torch.int16 if kv_dtype == 0 else torch.uint8
    kc = torch.zeros((NB, BS, KVH, D), dtype=tdt, device="cuda"); vc = torch.zeros_like(kc)
    sk = torch.tensor([ks], dtype=torch.float32, device="cuda"); sv = torch.tensor([vs], dtype=torch.float32, device="cuda")
    vmx.reshape_and_cache(_bf(torch, k), _bf(torch, v), kc, vc, torch.from_numpy(slots).cuda(), KVH, D, BS, kv_dtype, sk, sv)
    assert np.array_equal(kc.cpu().numpy().view(dt), kc_ref)       # byte-exact, FP8 and INT8 (value + 128) included
    assert np.array_equal(vc.cpu().numpy().view(dt), vc_ref)


def test_kv_scales_first_write_calibration(gpu):
    """KVScales::calibrate_if_needed on the device: first call pins max(absmax * 1.5 / 448, 1e-12) bit-exactly, later calls and
    larger inputs do not move it, reset() re-arms it, explicit scales count as calibrated; the FP8 cache written through the
    calibrated state equals the oracle's bytes.  Edge cases: all-zero tensors (the 1e-12 floor), bad headroom."""
    from oracle import oracle as orc
    vmx, torch = gpu
    from vllm_rust_amd._lib import VmxError
    from vllm_rust_amd.ops import KVScales
    rng = np.random.default_rng(11)
    KVH, D, BS, NB, T = 4, 64, 16, 4, 37
    k = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32) * 0.3)
    v = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32) * 3.0)
    ref = orc.KVScales()
    st = KVScales()
    assert not st.is_calibrated and st.k_scale.item() == 1.0
    ref.calibrate_if_needed(k, v)
    st.calibrate_if_needed(_bf(torch, k), _bf(torch, v))
    got = st.state[:2].cpu().numpy()
    assert st.is_calibrated
    assert got[0].view(np.uint32) == ref.k_scale.view(np.uint32) and got[1].view(np.uint32) == ref.v_scale.view(np.uint32), (got, ref.k_scale)
    # the cache bytes written with the device-resident scales
    slots = rng.permutation(NB * BS)[:T].astype(np.int32)
    kc_ref = np.zeros((NB, BS, KVH, D), np.uint8); vc_ref = np.zeros_like(kc_ref)
    orc.reshape_and_cache(k, v, kc_ref, vc_ref, slots, KVH, D, BS, 1, float(ref.k_scale), float(ref.v_scale))
    kc = torch.zeros((NB, BS, KVH, D), dtype=torch.uint8, device="cuda"); vc = torch.zeros_like(kc)
    vmx.reshape_and_cache(_bf(torch, k), _bf(torch, v), kc, vc, torch.from_numpy(slots).cuda(), KVH, D, BS, 1, st.k_scale, st.v_scale)
    assert np.array_equal(kc.cpu().numpy(), kc_ref) and np.array_equal(vc.cpu().numpy(), vc_ref)
    # pinned: a louder second batch does not move the scales
    k2 = orc.f32_to_bf16(orc.bf16_to_f32(k) * 8)
    st.calibrate_if_needed(_bf(torch, k2), _bf(torch, v))
    assert np.array_equal(st.state[:2].cpu().numpy(), got)
    # reset re-arms; headroom 1.0; zero tensors hit the floor
    st.reset(); ref.reset()
    st.headroom = 1.0; ref.set_headroom_factor(1.0)
    z = np.zeros((3, 8), np.uint16)
    ref.calibrate_if_needed(k2, z)
    st.calibrate_if_needed(_bf(torch, k2), _bf(torch, z))
    got = st.state[:2].cpu().numpy()
    assert got[0].view(np.uint32) == ref.k_scale.view(np.uint32) and got[1] == np.float32(1e-12) == ref.v_scale
    # explicit scales are calibrated scales
    st.set(0.25, 0.5)
    st.calibrate_if_needed(_bf(torch, k), _bf(torch, v))
    assert st.state[:2].cpu().numpy().tolist() == [0.25, 0.5]
    st.reset()
    st.headroom = float("nan")
    with pytest.raises(VmxError, match="headroom"):
        st.calibrate_if_needed(_bf(torch, k), _bf(torch, v))
    with pytest.raises(ValueError):
        ref.set_headroom_factor(0.0)


# (40, 8) = 5, (28, 4) = 7, (24, 8) = 3, (12, 1) = 12 query heads per KV head: ratios between the compiled capacities
@pytest.mark.parametrize("H,KVH,D", [(32, 8, 128), (16, 8, 256), (16, 1, 512), (4, 4, 64), (40, 8, 128), (28, 4, 128), (24, 8, 64),
                                     (12, 1, 256)])
@pytest.mark.parametrize("kv_dtype", [0, 1])
def test_paged_attention(gpu, H, KVH, D, kv_dtype):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(H + D + kv_dtype)
    BS, NB, MB = 16, 80, 40
    seq_lens = np.array([1, 37, 256, 257, 600], np.int32)
    S = len(seq_lens)
    kf = rng.standard_normal((NB, BS, KVH, D)).astype(np.float32)
    vf = rng.standard_normal((NB, BS, KVH, D)).astype(np.float32)
    ks = vs = 1.0
    if kv_dtype == 0:
        kc, vc = orc.f32_to_bf16(kf), orc.f32_to_bf16(vf)
        kcd = torch.from_numpy(kc.view(np.int16)).cuda(); vcd = torch.from_numpy(vc.view(np.int16)).cuda()
    else:
        ks = float(np.abs(kf).max() * 1.5 / 448.0); vs = float(np.abs(vf).max() * 1.5 / 448.0)
        enc = np.vectorize(orc.fp8_e4m3_encode_rne, otypes=[np.uint8])
        kc, vc = enc(kf / ks), enc(vf / vs)
        kcd = torch.from_numpy(kc).cuda(); vcd = torch.from_numpy(vc).cuda()
    bt = np.stack([rng.permutation(NB)[:MB] for _ in range(S)]).astype(np.int32)
    q = orc.f32_to_bf16(rng.standard_normal((S, H * D)).astype(np.float32))
    scale = float(1.0 / np.sqrt(D))
    sk = torch.tensor([ks], dtype=torch.float32, device="cuda"); sv = torch.tensor([vs], dtype=torch.float32, device="cuda")
    for window in (0, 100):
        got = _u16(torch, vmx.paged_attention(_bf(torch, q), kcd, vcd, torch.from_numpy(bt).cuda(), torch.from_numpy(seq_lens).cuda(),
                                              scale, H, KVH, D, BS, int(seq_lens.max()), window, kv_dtype, sk, sv))
        for psize in (256, 64):   # 256 = this library's partitioning; 64 = the reference's for ctx <= 4096
            ref = orc.paged_attention_v2(q, kc, vc, bt, seq_lens, scale, H, KVH, D, BS, psize, window, kv_dtype, ks, vs)
            g, r = orc.bf16_to_f32(got), orc.bf16_to_f32(ref)
            # f32 accumulate within 1e-3 relative (north star) + one bf16 ulp of the stored output
            assert np.abs(g - r).max() <= (1e-3 + 2 ** -7) * np.abs(r).max(), (window, psize)
            rel = np.sqrt(((g - r) ** 2).sum() / (r ** 2).sum())
            assert rel <= 1e-3 + 2 ** -8


@pytest.mark.parametrize("H,KVH,D,kv_dtype", [(32, 8, 128, 0), (8, 2, 128, 1), (4, 4, 64, 0), (40, 8, 128, 0)])
def test_paged_attention_long_context(gpu, H, KVH, D, kv_dtype):
    """Contexts of several thousand tokens: partitions of four and more tiles are walked by four waves each (a partition that
    ends inside its first quarter, a sequence far shorter than the bound and a window in the same batch)."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(H * D + kv_dtype)
    BS, NB, MB = 16, 560, 520
    seq_lens = np.array([8192, 4097, 5000, 130, 1], np.int32)
    S = len(seq_lens)
    kf = rng.standard_normal((NB, BS, KVH, D)).astype(np.float32)
    vf = rng.standard_normal((NB, BS, KVH, D)).astype(np.float32)
    ks = vs = 1.0
    if kv_dtype == 0:
        kc, vc = orc.f32_to_bf16(kf), orc.f32_to_bf16(vf)
        kcd = torch.from_numpy(kc.view(np.int16)).cuda(); vcd = torch.from_numpy(vc.view(np.int16)).cuda()
    else:
        ks = float(np.abs(kf).max() * 1.5 / 448.0); vs = float(np.abs(vf).max() * 1.5 / 448.0)
        enc = np.vectorize(orc.fp8_e4m3_encode_rne, otypes=[np.uint8])
        kc, vc = enc(kf / ks), enc(vf / vs)
        kcd = torch.from_numpy(kc).cuda(); vcd = torch.from_numpy(vc).cuda()
    bt = np.stack([rng.permutation(NB)[:MB] for _ in range(S)]).astype(np.int32)
    q = orc.f32_to_bf16(rng.standard_normal((S, H * D)).astype(np.float32))
    scale = float(1.0 / np.sqrt(D))
    sk = torch.tensor([ks], dtype=torch.float32, device="cuda"); sv = torch.tensor([vs], dtype=torch.float32, device="cuda")
    for window in (0, 4500):
        got = _u16(torch, vmx.paged_attention(_bf(torch, q), kcd, vcd, torch.from_numpy(bt).cuda(), torch.from_numpy(seq_lens).cuda(),
                                              scale, H, KVH, D, BS, int(seq_lens.max()), window, kv_dtype, sk, sv))
        ref = orc.paged_attention_v2(q, kc, vc, bt, seq_lens, scale, H, KVH, D, BS, 512, window, kv_dtype, ks, vs)
        g, r = orc.bf16_to_f32(got), orc.bf16_to_f32(ref)
        assert np.abs(g - r).max() <= (1e-3 + 2 ** -7) * np.abs(r).max(), window
        rel = np.sqrt(((g - r) ** 2).sum() / (r ** 2).sum())
        assert rel <= 1e-3 + 2 ** -8


def test_fused_qknorm_rope_add_gelu(gpu):
    from oracle import oracle as orc
    from vllm_rust_amd._lib import lib, check
    vmx, torch = gpu
    rng = np.random.default_rng(11)
    Bt, Hq, Hkv, S, D = 2, 4, 2, 3, 128
    q = orc.f32_to_bf16(rng.standard_normal((Bt, Hq, S, D)).astype(np.float32))
    k = orc.f32_to_bf16(rng.standard_normal((Bt, Hkv, S, D)).astype(np.float32))
    qw = orc.f32_to_bf16(1 + 0.1 * rng.standard_normal(D).astype(np.float32)); kw = orc.f32_to_bf16(1 + 0.1 * rng.standard_normal(D).astype(np.float32))
    pos = rng.integers(0, 100, size=(Bt, S)).astype(np.int32)
    inv = 1.0 / (1e6 ** (np.arange(0, D, 2) / D)); ang = np.arange(128)[:, None] * inv[None, :]
    cos, sin = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    oq = torch.empty((Bt, Hq, S, D), dtype=torch.bfloat16, device="cuda"); ok = torch.empty((Bt, Hkv, S, D), dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    d = lambda a: torch.from_numpy(a).cuda()
    tq, tk, tqw, tkw, tc, ts, tp = _bf(torch, q), _bf(torch, k), _bf(torch, qw), _bf(torch, kw), d(cos), d(sin), d(pos)
    check(lib.vmx_fused_qknorm_rope_bf16(oq.data_ptr(), ok.data_ptr(), tq.data_ptr(), tk.data_ptr(), tqw.data_ptr(), tkw.data_ptr(),
                                         tc.data_ptr(), ts.data_ptr(), tp.data_ptr(), Bt, Hq, Hkv, S, D, st))
    def ref(x, w):  # qknorm_rope.cu:47-136 in float64, one rounding
        xf = orc.bf16_to_f32(x).astype(np.float64); wf = orc.bf16_to_f32(w).astype(np.float64)
        n = xf / np.sqrt((xf ** 2).mean(-1, keepdims=True) + 1e-6) * wf
        c = cos[pos][:, None, :, :]; s_ = sin[pos][:, None, :, :]
        lo, hi = n[..., :D // 2], n[..., D // 2:]
        return np.concatenate([lo * c - hi * s_, hi * c + lo * s_], -1)
    for got, exp in ((oq, ref(q, qw)), (ok, ref(k, kw))):
        g = orc.bf16_to_f32(_u16(torch, got))
        assert np.abs(g - exp).max() <= 2 ** -7 * np.abs(exp).max()
    a = orc.f32_to_bf16(rng.standard_normal(5000).astype(np.float32) * 3); b = orc.f32_to_bf16(rng.standard_normal(5000).astype(np.float32))
    out = torch.empty(5000, dtype=torch.bfloat16, device="cuda"); ta, tb = _bf(torch, a), _bf(torch, b)
    check(lib.vmx_add_bf16(out.data_ptr(), ta.data_ptr(), tb.data_ptr(), 5000, st))
    assert np.array_equal(_u16(torch, out), orc.f32_to_bf16(orc.bf16_to_f32(a) + orc.bf16_to_f32(b)))
    from math import erf
    af, bf = orc.bf16_to_f32(a).astype(np.float64), orc.bf16_to_f32(b).astype(np.float64)
    for tanh_mode, fn in ((0, lambda x: x * 0.5 * (1 + np.vectorize(erf)(x / np.sqrt(2)))),
                          (1, lambda x: 0.5 * x * (1 + np.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3))))):
        check(lib.vmx_gelu_and_mul_bf16(out.data_ptr(), ta.data_ptr(), tb.data_ptr(), 5000, tanh_mode, st))
        g = orc.bf16_to_f32(_u16(torch, out)); exp = fn(af) * bf
        assert np.abs(g - exp).max() <= 2 ** -7 * np.abs(exp).max() + 1e-6


def test_rope_gptj(gpu):
    """rotary_embedding_gptj_bf16 (rope.cu:119-141): bit for bit the oracle, partial rotation, k optional."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(11)
    T, H, KVH, D, R = 5, 8, 2, 128, 64
    q = orc.f32_to_bf16(rng.standard_normal((T, H * D)).astype(np.float32))
    k = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32))
    pos = rng.integers(0, 64, T).astype(np.int32)
    ang = np.arange(64)[:, None] * (1.0 / (10000.0 ** (np.arange(R // 2) * 2.0 / R)))[None, :]
    cs = np.concatenate([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)
    qd, kd = _bf(torch, q), _bf(torch, k)
    vmx.rope_gptj(torch.from_numpy(pos).cuda(), qd, kd, torch.from_numpy(cs).cuda(), R, D, H, KVH)
    rq, rk = orc.rope_gptj_bf16(pos, q, k, cs, R, D, H, KVH)
    assert np.array_equal(_u16(torch, qd).reshape(rq.shape), rq)
    assert np.array_equal(_u16(torch, kd).reshape(rk.shape), rk)
    qd2 = _bf(torch, q)
    vmx.rope_gptj(torch.from_numpy(pos).cuda(), qd2, None, torch.from_numpy(cs).cuda(), R, D, H, KVH)
    assert np.array_equal(_u16(torch, qd2).reshape(rq.shape), rq)


def test_reshape_and_cache_hnd_and_copy_blocks(gpu):
    """reshape_and_cache_hnd_* and copy_blocks_* (cache_ops.cu:106-202): byte for byte the oracle, skipped slots, 256-wide heads."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(12)
    T, KVH, D, BS, NB = 9, 8, 256, 16, 12
    k = rng.integers(0, 65536, (T, KVH * D)).astype(np.uint16)
    v = rng.integers(0, 65536, (T, KVH * D)).astype(np.uint16)
    slots = rng.permutation(NB * BS)[:T].astype(np.int32)
    slots[3] = -1
    kc_ref = rng.integers(0, 65536, (NB, KVH, BS, D)).astype(np.uint16)
    vc_ref = rng.integers(0, 65536, (NB, KVH, BS, D)).astype(np.uint16)
    kc, vc = torch.from_numpy(kc_ref.view(np.int16).copy()).cuda(), torch.from_numpy(vc_ref.view(np.int16).copy()).cuda()
    orc.reshape_and_cache_hnd(k, v, kc_ref, vc_ref, slots, KVH, D, BS)
    vmx.reshape_and_cache_hnd(torch.from_numpy(k.view(np.int16)).cuda(), torch.from_numpy(v.view(np.int16)).cuda(), kc, vc,
                              torch.from_numpy(slots).cuda(), KVH, D, BS)
    assert np.array_equal(kc.cpu().numpy().view(np.uint16), kc_ref) and np.array_equal(vc.cpu().numpy().view(np.uint16), vc_ref)
    mapping = np.array([[0, 5], [7, 2], [11, 9]], np.int32)
    bpb = KVH * BS * D * 2
    orc.copy_blocks(kc_ref.view(np.uint8), vc_ref.view(np.uint8), mapping, bpb)
    vmx.copy_blocks(kc, vc, torch.from_numpy(mapping).cuda(), bpb)
    assert np.array_equal(kc.cpu().numpy().view(np.uint16), kc_ref) and np.array_equal(vc.cpu().numpy().view(np.uint16), vc_ref)
    vmx.copy_blocks(kc, vc, torch.zeros((0, 2), dtype=torch.int32).cuda(), bpb)     # no pairs: nothing to do


@pytest.mark.parametrize("H,KVH,D", [(32, 8, 128), (16, 8, 256), (16, 1, 512), (4, 4, 64)])
def test_paged_attention_int8_kv(gpu, H, KVH, D):
    """INT8 KV cache (KVCacheDtype::Int8: bytes = round(clamp(x / scale, +-127)) + 128, read back as (byte - 128) * scale,
    kv_cache/quantization.rs:330-345, paged_attention.cu:83-95): the cache written by vmx_reshape_and_cache is byte for byte the
    oracle's, and decode attention over it matches the oracle within the floating-point bar."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(H + D)
    BS, NB, MB = 16, 48, 24
    seq_lens = np.array([1, 37, 300], np.int32)
    S = len(seq_lens)
    T = NB * BS
    k = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32))
    v = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32))
    ks = float(np.abs(orc.bf16_to_f32(k)).max() / 127.0)
    vs = float(np.abs(orc.bf16_to_f32(v)).max() * 0.5 / 127.0)      # half the range: exercises the clamp
    slots = rng.permutation(T).astype(np.int32)
    kc = np.zeros((NB, BS, KVH, D), np.uint8); vc = np.zeros_like(kc)
    orc.reshape_and_cache(k, v, kc, vc, slots, KVH, D, BS, 2, ks, vs)
    assert vc.min() == 1 and vc.max() == 255 and kc.min() >= 1
    kcd = torch.zeros((NB, BS, KVH, D), dtype=torch.uint8, device="cuda"); vcd = torch.zeros_like(kcd)
    sk = torch.tensor([ks], dtype=torch.float32, device="cuda"); sv = torch.tensor([vs], dtype=torch.float32, device="cuda")
    vmx.reshape_and_cache(_bf(torch, k), _bf(torch, v), kcd, vcd, torch.from_numpy(slots).cuda(), KVH, D, BS, 2, sk, sv)
    assert np.array_equal(kcd.cpu().numpy(), kc) and np.array_equal(vcd.cpu().numpy(), vc)
    bt = np.stack([rng.permutation(NB)[:MB] for _ in range(S)]).astype(np.int32)
    q = orc.f32_to_bf16(rng.standard_normal((S, H * D)).astype(np.float32))
    scale = float(1.0 / np.sqrt(D))
    for window in (0, 100):
        got = _u16(torch, vmx.paged_attention(_bf(torch, q), kcd, vcd, torch.from_numpy(bt).cuda(), torch.from_numpy(seq_lens).cuda(),
                                              scale, H, KVH, D, BS, int(seq_lens.max()), window, 2, sk, sv))
        ref = orc.paged_attention_v2(q, kc, vc, bt, seq_lens, scale, H, KVH, D, BS, 256, window, 2, ks, vs)
        g, r = orc.bf16_to_f32(got), orc.bf16_to_f32(ref)
        assert np.abs(g - r).max() <= (1e-3 + 2 ** -7) * np.abs(r).max(), window
        assert np.sqrt(((g - r) ** 2).sum() / (r ** 2).sum()) <= 1e-3 + 2 ** -8


@pytest.mark.parametrize("H,KVH,D", [(32, 8, 128), (16, 8, 256), (4, 4, 64)])
def test_paged_attention_fp8_e5m2_kv(gpu, H, KVH, D):
    """FP8 e5m2 KV cache (KVCacheDtype::Fp8E5m2; the read side is load_kv_to_f32<fp8_e5m2_byte>, kernels/paged_attention.cu:77-82):
    the bytes vmx_reshape_and_cache writes are the oracle's round-to-nearest-even e5m2 of clamp(x / scale, +-57344), and decode
    attention over them matches the oracle within the floating-point bar.  (The reference's own cache engine still writes
    e4m3 bytes in this mode, cache_engine.rs:217-223; a caller that wants those passes VMX_KV_FP8_E4M3 to the write.)"""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(H + D + 11)
    BS, NB, MB = 16, 48, 24
    seq_lens = np.array([1, 37, 300], np.int32)
    S = len(seq_lens)
    T = NB * BS
    k = orc.f32_to_bf16((rng.standard_normal((T, KVH * D)) * np.exp(rng.uniform(-12, 3, (T, 1)))).astype(np.float32))   # down into subnormals
    v = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32))
    ks = 1.0
    vs = float(np.abs(orc.bf16_to_f32(v)).max() * 0.5 / 57344.0)    # half the range: exercises the clamp
    slots = rng.permutation(T).astype(np.int32)
    kc = np.zeros((NB, BS, KVH, D), np.uint8); vc = np.zeros_like(kc)
    orc.reshape_and_cache(k, v, kc, vc, slots, KVH, D, BS, 4, ks, vs)
    assert (vc & 0x7F).max() == 0x7B and (kc & 0x7F).min() == 0 and 0 < (kc & 0x7F)[(kc & 0x7F) > 0].min() < 4
    kcd = torch.zeros((NB, BS, KVH, D), dtype=torch.uint8, device="cuda"); vcd = torch.zeros_like(kcd)
    sk = torch.tensor([ks], dtype=torch.float32, device="cuda"); sv = torch.tensor([vs], dtype=torch.float32, device="cuda")
    vmx.reshape_and_cache(_bf(torch, k), _bf(torch, v), kcd, vcd, torch.from_numpy(slots).cuda(), KVH, D, BS, 4, sk, sv)
    assert np.array_equal(kcd.cpu().numpy(), kc) and np.array_equal(vcd.cpu().numpy(), vc)
    # attention over a cache of ordinary magnitudes
    k2 = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32))
    ks2 = float(np.abs(orc.bf16_to_f32(k2)).max() / 57344.0)
    orc.reshape_and_cache(k2, v, kc, vc, slots, KVH, D, BS, 4, ks2, vs)
    kcd, vcd = torch.from_numpy(kc).cuda(), torch.from_numpy(vc).cuda()
    sk = torch.tensor([ks2], dtype=torch.float32, device="cuda")
    bt = np.stack([rng.permutation(NB)[:MB] for _ in range(S)]).astype(np.int32)
    q = orc.f32_to_bf16(rng.standard_normal((S, H * D)).astype(np.float32))
    scale = float(1.0 / np.sqrt(D))
    for window in (0, 100):
        got = _u16(torch, vmx.paged_attention(_bf(torch, q), kcd, vcd, torch.from_numpy(bt).cuda(), torch.from_numpy(seq_lens).cuda(),
                                              scale, H, KVH, D, BS, int(seq_lens.max()), window, 4, sk, sv))
        ref = orc.paged_attention_v2(q, kc, vc, bt, seq_lens, scale, H, KVH, D, BS, 256, window, 4, ks2, vs)
        g, r = orc.bf16_to_f32(got), orc.bf16_to_f32(ref)
        assert np.abs(g - r).max() <= (1e-3 + 2 ** -7) * np.abs(r).max(), window
        assert np.sqrt(((g - r) ** 2).sum() / (r ** 2).sum()) <= 1e-3 + 2 ** -8


@pytest.mark.parametrize("H,KVH,D,kvt", [(32, 8, 128, 0), (16, 8, 256, 0), (4, 4, 64, 1), (32, 8, 128, 1), (16, 1, 512, 0)])
def test_paged_attention_with_lse_and_context_parallel_merge(gpu, H, KVH, D, kvt):
    """batched_decode_attention_with_lse (layers/attention/backend.rs:113, naive.rs:151-205): the output is the plain entry point's
    bit for bit and lse = ln sum exp(scale * q . k) within 1e-4 of the oracle's float64 value (absolute; the statistic is O(1..10)),
    -inf for an empty context.  Then the use it exists for (lse_correct_and_reduce, distributed/context_parallel.rs:281-315): the
    context dealt to 3 ranks in interleaved blocks, each rank's attention over its own slice corrected with the gathered
    statistics, and the shares summed -- equal to attention over the whole context within the bf16 bar."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(H + D + kvt)
    BS, NB, MB = 16, 160, 130
    seq_lens = np.array([0, 1, 37, 300, 2050], np.int32)
    S = len(seq_lens)
    T = NB * BS
    k = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32))
    v = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32))
    ks, vs = (float(np.abs(orc.bf16_to_f32(k)).max() / 448.0), float(np.abs(orc.bf16_to_f32(v)).max() / 448.0)) if kvt else (1.0, 1.0)
    kc = np.zeros((NB, BS, KVH, D), np.uint8 if kvt else np.uint16); vc = np.zeros_like(kc)
    orc.reshape_and_cache(k, v, kc, vc, np.arange(T, dtype=np.int32), KVH, D, BS, kvt, ks, vs)
    dev = lambda a: torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).cuda()
    kcd, vcd = dev(kc), dev(vc)
    if not kvt:
        kcd, vcd = kcd.view(torch.bfloat16), vcd.view(torch.bfloat16)
    sk = torch.tensor([ks], dtype=torch.float32, device="cuda"); sv = torch.tensor([vs], dtype=torch.float32, device="cuda")
    bt = np.stack([rng.permutation(NB)[:MB] for _ in range(S)]).astype(np.int32)
    q = orc.f32_to_bf16(rng.standard_normal((S, H * D)).astype(np.float32))
    scale = float(1.0 / np.sqrt(D))
    btd, sld = torch.from_numpy(bt).cuda(), torch.from_numpy(seq_lens).cuda()
    for window in (0, 100):
        plain = vmx.paged_attention(_bf(torch, q), kcd, vcd, btd, sld, scale, H, KVH, D, BS, int(seq_lens.max()), window, kvt, sk, sv)
        out, lse = vmx.paged_attention_with_lse(_bf(torch, q), kcd, vcd, btd, sld, scale, H, KVH, D, BS, int(seq_lens.max()), window, kvt, sk, sv)
        assert torch.equal(out.view(torch.int16), plain.view(torch.int16))
        want = orc.paged_attention_lse(q, kc, bt, seq_lens, scale, H, KVH, D, BS, window, kvt, ks)
        got = lse.cpu().numpy().astype(np.float64)
        assert np.all(np.isneginf(got[0])) and np.all(np.isneginf(want[0]))
        assert np.abs(got[1:] - want[1:]).max() <= 1e-4, (window, np.abs(got[1:] - want[1:]).max())
    # ---- context-parallel merge: rank r holds the context's blocks r, r + R, ... (interleave = one cache block)
    R = 3
    full = orc.bf16_to_f32(_u16(torch, vmx.paged_attention(_bf(torch, q), kcd, vcd, btd, sld, scale, H, KVH, D, BS, int(seq_lens.max()), 0, kvt, sk, sv)))
    outs, lses = [], []
    for r in range(R):
        nblk = (seq_lens + BS - 1) // BS
        bt_r = np.zeros_like(bt); sl_r = np.zeros_like(seq_lens)
        for s in range(S):
            mine = [b for b in range(nblk[s]) if b % R == r]
            bt_r[s, :len(mine)] = bt[s, mine]
            sl_r[s] = sum(min(BS, seq_lens[s] - b * BS) for b in mine)
        o, l = vmx.paged_attention_with_lse(_bf(torch, q), kcd, vcd, torch.from_numpy(bt_r).cuda(), torch.from_numpy(sl_r).cuda(), scale,
                                            H, KVH, D, BS, int(max(sl_r.max(), 1)), 0, kvt, sk, sv)
        outs.append(o); lses.append(l)
    lse_all = torch.stack(lses).contiguous()                      # [R, S, H]: what the all-gather hands every rank
    merged = sum(vmx.lse_correct(outs[r], lse_all, r, D) for r in range(R)).cpu().numpy().reshape(S, H * D)
    assert np.all(merged[0] == 0)
    assert np.abs(merged[1:] - full[1:]).max() <= (1e-3 + 2 ** -7) * np.abs(full[1:]).max()
    assert np.sqrt(((merged[1:] - full[1:]) ** 2).sum() / (full[1:] ** 2).sum()) <= 1e-3 + 2 ** -8
    from vllm_rust_amd._lib import lib
    assert lib.vmx_lse_correct_f32(None, None, None, 1, 0, 1, 64, None) != 0
    assert lib.vmx_lse_correct_f32(merged.ctypes.data, merged.ctypes.data, merged.ctypes.data, 2, 2, 1, 64, None) != 0


@pytest.mark.parametrize("H,KVH,D", [(32, 8, 128), (16, 8, 256), (4, 4, 64), (16, 1, 512)])
def test_paged_attention_f16(gpu, H, KVH, D):
    """The fp16 kernel variants (paged_attention_v2_f16 / _reduce_f16): fp16 q, cache and output against the oracle's fp16 form."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(H + D + 3)
    BS, NB, MB = 16, 48, 24
    seq_lens = np.array([1, 37, 256, 300], np.int32)
    S = len(seq_lens)
    kc = rng.standard_normal((NB, BS, KVH, D)).astype(np.float16)
    vc = rng.standard_normal((NB, BS, KVH, D)).astype(np.float16)
    q = rng.standard_normal((S, H * D)).astype(np.float16)
    bt = np.stack([rng.permutation(NB)[:MB] for _ in range(S)]).astype(np.int32)
    scale = float(1.0 / np.sqrt(D))
    for window in (0, 100):
        got = vmx.paged_attention_f16(torch.from_numpy(q).cuda(), torch.from_numpy(kc).cuda(), torch.from_numpy(vc).cuda(),
                                      torch.from_numpy(bt).cuda(), torch.from_numpy(seq_lens).cuda(), scale, H, KVH, D, BS,
                                      int(seq_lens.max()), window).cpu().numpy().astype(np.float32)
        ref = orc.paged_attention_v2(q.view(np.uint16), kc.view(np.uint16), vc.view(np.uint16), bt, seq_lens, scale, H, KVH, D, BS, 256,
                                     window, 3, 1.0, 1.0).view(np.float16).astype(np.float32)
        assert np.abs(got - ref).max() <= (1e-3 + 2 ** -10) * np.abs(ref).max(), window
        assert np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum()) <= 1e-3 + 2 ** -11
    from vllm_rust_amd._lib import lib
    assert lib.vmx_paged_attention_f16(None, None, None, None, None, None, 1.0, 1, 4, 4, 1, 16, 64, 16, 0, None, 0, None) != 0


def test_register_only_lane_exchanges_equal_the_lds_crossbar_forms():
    """The wave / row butterflies and lane exchanges every kernel reduces with (DPP row operations, v_permlane swaps) give the
    bits of the __shfl_xor forms they replaced, on values whose sums depend on the order of the additions."""
    import torch
    import vllm_rust_amd as vmx
    from vllm_rust_amd._lib import check, lib
    vmx.init(0)
    out = torch.full((64,), -1, dtype=torch.int32, device="cuda")
    for seed in (1, 12345, 987654321):
        check(lib.vmx_debug_lane_ops_check(out.data_ptr(), seed, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert out.cpu().tolist() == [0] * 64

"""GPU parity of the non-greedy sampling kernels against a line-by-line numpy restatement of
crates/core/kernels/sampling.cu:197-293 (single-thread reference algorithm)."""
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


def ref_sample(p, rnd, top_k, top_p):
    p = p.astype(np.float32); V = p.size
    thr = np.float32(0)
    if 0 < top_k < V:
        vals = np.unique(p)[::-1]
        thr = vals[min(top_k, vals.size) - 1]
    f = np.float32(0)
    for v in p[p >= thr]:
        f = np.float32(f + v)
    if 0.0 < top_p < 1.0:
        target = np.float32(np.float32(top_p) * f); cum = np.float32(0); tpt = np.float32(0)
        for best in np.unique(p[p >= thr])[::-1]:
            if not cum < target:
                break
            tpt = best
            for _ in range(int((p == best).sum())):
                cum = np.float32(cum + best)
                if cum >= target:
                    break
        if tpt > thr:
            thr = tpt
        f = np.float32(0)
        for v in p[p >= thr]:
            f = np.float32(f + v)
    target = np.float32(np.float32(rnd) * f); cum = np.float32(0)
    lo_hi = []
    for i in range(V):
        if p[i] >= thr:
            prev = cum; cum = np.float32(cum + p[i])
            if cum > target:
                return i, float(target - prev), float(cum - target)
    return V - 1, 1.0, 1.0


def test_softmax_temperature_and_sampling(gpu):
    from oracle import oracle as orc
    from vllm_rust_amd._lib import lib, check
    vmx, torch = gpu
    rng = np.random.default_rng(0)
    rows, V = 6, 5000
    lg = orc.f32_to_bf16((rng.standard_normal((rows, V)) * 3).astype(np.float32))
    t = torch.from_numpy(lg.view(np.int16)).cuda().view(torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    check(lib.vmx_temperature_scale_bf16(t.data_ptr(), 0.7, rows, V, st))
    scaled = orc.f32_to_bf16(orc.bf16_to_f32(lg) * np.float32(0.7))
    assert np.array_equal(t.cpu().view(torch.int16).numpy().view(np.uint16), scaled)
    probs = torch.empty((rows, V), dtype=torch.float32, device="cuda")
    check(lib.vmx_softmax_to_probs(probs.data_ptr(), t.data_ptr(), 1, rows, V, st))
    x = orc.bf16_to_f32(scaled).astype(np.float64)
    e = np.exp(x - x.max(-1, keepdims=True)); ref = e / (e.sum(-1, keepdims=True) + 1e-6)
    pn = probs.cpu().numpy()
    assert np.abs(pn - ref).max() <= 2e-6 * ref.max() + 1e-9
    out = torch.empty(rows, dtype=torch.int32, device="cuda")
    for top_k, top_p in ((0, 1.0), (50, 1.0), (0, 0.9), (40, 0.8), (1, 1.0)):
        for trial in range(3):
            rv = rng.random(rows).astype(np.float32)
            check(lib.vmx_top_k_top_p_sample(out.data_ptr(), probs.data_ptr(), torch.from_numpy(rv).cuda().data_ptr(), None, None,
                                             top_k, top_p, rows, V, st))
            got = out.cpu().numpy()
            for r in range(rows):
                sel, below, above = ref_sample(pn[r], rv[r], top_k, top_p)
                # identical unless the draw sits within f32 summation noise of an interval boundary
                if min(below, above) > 1e-5:
                    assert got[r] == sel, (top_k, top_p, r, got[r], sel)
                if top_k == 1:
                    assert got[r] == int(np.argmax(pn[r]))
    # per-sequence parameters
    tk = torch.tensor([1, 0, 5, 0, 1, 3], dtype=torch.int32, device="cuda"); tp = torch.tensor([1.0, 0.5, 1.0, 1.0, 0.3, 0.9], device="cuda")
    rv = rng.random(rows).astype(np.float32)
    check(lib.vmx_top_k_top_p_sample(out.data_ptr(), probs.data_ptr(), torch.from_numpy(rv).cuda().data_ptr(), tk.data_ptr(), tp.data_ptr(),
                                     0, 1.0, rows, V, st))
    got = out.cpu().numpy()
    for r in range(rows):
        sel, below, above = ref_sample(pn[r], rv[r], int(tk[r]), float(tp[r]))
        if min(below, above) > 1e-5:
            assert got[r] == sel


def ref_multinomial_no_filter(p, rnd):
    """multinomial_sample_no_filter_parallel (sampling.cu:566-631) with its order of f32 additions: np.add.accumulate is a
    strict left-to-right sum."""
    V = p.size
    chunk = (V + 255) // 256
    ser = lambda a, start=np.float32(0): np.add.accumulate(np.concatenate([[start], a]).astype(np.float32), dtype=np.float32)
    sums = np.array([ser(p[min(c * chunk, V):min((c + 1) * chunk, V)])[-1] for c in range(256)], np.float32)
    target = np.float32(np.float32(rnd) * ser(sums)[-1])
    pre = ser(sums)                       # pre[c] = cdf before chunk c, pre[c + 1] after it
    hit = np.nonzero(pre[1:] > target)[0]
    tc = int(hit[0]) if hit.size else 255
    cdf = pre[tc] if hit.size else pre[256]    # without a hit the walk has added every chunk and the last chunk is scanned on top
    lo, hi = min(tc * chunk, V), min((tc + 1) * chunk, V)
    run = ser(p[lo:hi], cdf)[1:]
    inner = np.nonzero(run > target)[0]
    return lo + int(inner[0]) if inner.size else V - 1


@pytest.mark.parametrize("V", [100, 5000, 151936])
def test_per_seq_temperature_and_parallel_multinomial_bit_exact(gpu, V):
    """temperature_scale_f32_per_seq and multinomial_sample_no_filter_parallel (sampling.cu:529-631): the scaled logits are the
    f32 products bit for bit, and the drawn token ids are the reference's for the same probabilities and uniforms (integer
    result, same addition order, so exact) -- over ordinary draws, rand = 0, rand just under 1 and a peaked distribution."""
    from vllm_rust_amd._lib import lib, check
    vmx, torch = gpu
    rng = np.random.default_rng(V)
    rows = 8
    lg = (rng.standard_normal((rows, V)) * 2).astype(np.float32)
    lg[3, V // 3] = 40.0                                                    # one token holds nearly all the mass
    inv_t = (1.0 / rng.uniform(0.3, 1.5, rows)).astype(np.float32)
    t = torch.from_numpy(lg.copy()).cuda()
    st = torch.cuda.current_stream().cuda_stream
    check(lib.vmx_temperature_scale_f32_per_seq(t.data_ptr(), torch.from_numpy(inv_t).cuda().data_ptr(), rows, V, st))
    assert np.array_equal(t.cpu().numpy(), lg * inv_t[:, None])
    probs = torch.empty((rows, V), dtype=torch.float32, device="cuda")
    check(lib.vmx_softmax_to_probs(probs.data_ptr(), t.data_ptr(), 0, rows, V, st))
    pn = probs.cpu().numpy()
    out = torch.empty(rows, dtype=torch.int32, device="cuda")
    for trial in range(4):
        rv = rng.random(rows).astype(np.float32)
        if trial == 1:
            rv[:] = 0.0
        if trial == 2:
            rv[:] = np.nextafter(np.float32(1.0), np.float32(0.0))
        check(lib.vmx_multinomial_sample_no_filter(out.data_ptr(), probs.data_ptr(), torch.from_numpy(rv).cuda().data_ptr(), rows, V, st))
        got = out.cpu().numpy()
        want = np.array([ref_multinomial_no_filter(pn[r], rv[r]) for r in range(rows)])
        assert np.array_equal(got, want), (trial, got, want)
        if trial == 1:
            assert np.array_equal(got, [int(np.nonzero(pn[r] > 0)[0][0]) for r in range(rows)])
    assert lib.vmx_multinomial_sample_no_filter(None, None, None, 1, 8, None) != 0 and lib.vmx_temperature_scale_f32_per_seq(None, None, 1, 8, None) != 0

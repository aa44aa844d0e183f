"""GPU parity of the MoE operators: router top-k softmax, block alignment and the device-side int4
expert layer, against numpy / oracle restatements of the reference semantics."""
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


def _ref_topk(logits, k, renorm, round_bf16, orc):
    lg = logits.astype(np.float32)
    e = np.exp(lg - lg.max(-1, keepdims=True)).astype(np.float32)
    p = (e / e.sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)
    if round_bf16:
        p = orc.bf16_to_f32(orc.f32_to_bf16(p))
    ids = np.argsort(-p, axis=-1, kind="stable")[:, :k].astype(np.int32)    # ties -> smaller index
    w = np.take_along_axis(p, ids, -1)
    if renorm:
        w = w / w.sum(-1, keepdims=True)
    return w.astype(np.float32), ids


@pytest.mark.parametrize("E,k", [(8, 2), (64, 6), (200, 8)])
@pytest.mark.parametrize("round_bf16", [False, True])
def test_topk_softmax(gpu, E, k, round_bf16):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(E + k)
    lg = rng.standard_normal((33, E)).astype(np.float32) * 2
    lg[0, :] = 0.0                                     # all-equal row: ties resolve to the smallest indices
    for dt in ("f32", "bf16"):
        if dt == "bf16":
            lgq = orc.bf16_to_f32(orc.f32_to_bf16(lg))
            dev = torch.from_numpy(orc.f32_to_bf16(lg).view(np.int16)).cuda().view(torch.bfloat16)
        else:
            lgq, dev = lg, torch.from_numpy(lg).cuda()
        w, ids = vmx.topk_softmax(dev, k, True, round_bf16)
        rw, rids = _ref_topk(lgq, k, True, round_bf16, orc)
        w, ids = w.cpu().numpy(), ids.cpu().numpy()
        assert list(ids[0]) == list(range(k))
        same = (ids == rids).all(-1)
        assert same.mean() > 0.9                       # differing rows can only be near-ties of expf rounding
        assert np.abs(w[same] - rw[same]).max() <= 2e-6 + (4e-3 if round_bf16 else 0)
        assert np.allclose(w.sum(-1), 1.0, atol=1e-5)


def test_moe_align_block_size(gpu):
    vmx, torch = gpu
    rng = np.random.default_rng(1)
    E, bs = 8, 16
    ids = rng.integers(0, E, size=(37, 2)).astype(np.int32)
    s, e, tot = vmx.moe_align_block_size(torch.from_numpy(ids).cuda(), E, bs)
    s, e, tot = s.cpu().numpy(), e.cpu().numpy(), int(tot.item())
    flat = ids.reshape(-1)
    counts = np.bincount(flat, minlength=E)
    assert tot == int(((counts + bs - 1) // bs * bs).sum())
    pos = 0
    for ex in range(E):
        n = counts[ex]; padded = (n + bs - 1) // bs * bs
        assert list(s[pos:pos + n]) == list(np.nonzero(flat == ex)[0])
        assert (s[pos + n:pos + padded] == flat.size).all()
        assert (e[pos // bs:(pos + padded) // bs] == ex).all()
        pos += padded


@pytest.mark.parametrize("numel,E,bs", [(2, 8, 16), (1, 8, 64), (5000, 8, 64), (16384, 8, 64), (1500, 64, 16), (3001, 200, 16), (4096, 256, 64),
                                         (700, 300, 16)])
def test_moe_align_block_size_large_and_sparse(gpu, numel, E, bs):
    """The O(numel) form (up to 256 experts) and the per-expert walk (more): long id lists, experts without rows, several
    rounds of 1024 ids; an expert's indices ascend, padding is `numel`, every block names its expert."""
    vmx, torch = gpu
    rng = np.random.default_rng(numel + E)
    live = rng.permutation(E)[:max(1, E * 2 // 3)]             # a third of the experts gets nothing
    flat = live[rng.integers(0, len(live), size=numel)].astype(np.int32)
    s, e, tot = vmx.moe_align_block_size(torch.from_numpy(flat.reshape(-1, 1)).cuda(), E, bs)
    s, e, tot = s.cpu().numpy(), e.cpu().numpy(), int(tot.item())
    counts = np.bincount(flat, minlength=E)
    assert tot == int(((counts + bs - 1) // bs * bs).sum())
    pos = 0
    for ex in range(E):
        n = counts[ex]; padded = (n + bs - 1) // bs * bs
        assert np.array_equal(s[pos:pos + n], np.nonzero(flat == ex)[0])
        assert (s[pos + n:pos + padded] == numel).all()
        assert (e[pos // bs:(pos + padded) // bs] == ex).all()
        pos += padded


@pytest.fixture(scope="module")
def mixtral_experts(gpu):
    """Eight GPTQ-symmetric int4 experts at the Mixtral-8x7B shape (4096 -> 14336 -> 4096, g128), packed for the device and
    kept on the host for the oracle."""
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(77)
    E, H, I, g = 8, 4096, 14336, 128
    packs13, packs2, experts = [], [], []
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if a.dtype == np.uint32 else np.int16)).cuda()
    for ex in range(E):
        def mk(K, N):
            qw = rng.integers(0, 2 ** 32, size=(K // 8, N), dtype=np.uint32)
            sc = orc.f32_to_bf16(((rng.random((K // g, N)) * 0.5 + 0.75) / (6.5 * np.sqrt(K))).astype(np.float32))
            return qw, sc
        w1, w3, w2 = mk(H, I), mk(H, I), mk(I, H)
        experts.append((w1, w3, w2))
        packs13.append(vmx.int4_pack(d(np.concatenate([w1[0], w3[0]], 1)), d(np.concatenate([w1[1], w3[1]], 1)), None, H, 2 * I, g))
        packs2.append(vmx.int4_pack(d(w2[0]), d(w2[1]), None, I, H, g))
    t13 = torch.tensor([p.data_ptr() for p in packs13], dtype=torch.int64, device="cuda")
    t2 = torch.tensor([p.data_ptr() for p in packs2], dtype=torch.int64, device="cuda")
    return dict(E=E, H=H, I=I, g=g, experts=experts, t13=t13, t2=t2, keep=(packs13, packs2))


@pytest.mark.parametrize("T", [1, 7, 64, 256])
def test_moe_token_grouped_true_shape_matches_expert_loop(gpu, mixtral_experts, T):
    """The three forms of the expert layer -- one pass per (token, slot) pair, 16-row blocks, 64-row blocks on the matrix
    cores -- at the true Mixtral-8x7B expert shape, E = 8 top-2, against the oracle's per-token expert loop
    (mixtral_quantized.rs:138-195; rows of one expert go through the oracle together, each row's arithmetic is its own)."""
    from oracle import oracle as orc
    from vllm_rust_amd._lib import lib
    vmx, torch = gpu
    m = mixtral_experts
    E, H, I, g, k = m["E"], m["H"], m["I"], m["g"], 2
    rng = np.random.default_rng(1000 + T)
    x = orc.f32_to_bf16(rng.standard_normal((T, H)).astype(np.float32))
    logits = rng.standard_normal((T, E)).astype(np.float32)
    if T >= 64:
        logits[:, 5] -= 30.0          # an expert nobody picks: its blocks must not exist
    xd = torch.from_numpy(x.view(np.int16)).cuda().view(torch.bfloat16)
    w, ids = vmx.topk_softmax(torch.from_numpy(logits).cuda(), k, True, True)
    wn, idn = w.cpu().numpy(), ids.cpu().numpy()
    eo = np.zeros((T, k, H), np.float32)
    for ex in range(E):
        tt, jj = np.nonzero(idn == ex)
        if not len(tt):
            continue
        w1, w3, w2 = m["experts"][ex]
        xe = np.ascontiguousarray(x[tt])
        gate = orc.int4_gemv_bf16(xe, w1[0], w1[1], None, None, H, I, g)
        up = orc.int4_gemv_bf16(xe, w3[0], w3[1], None, None, H, I, g)
        hid = orc.silu_and_mul_bf16(gate, up)
        eo[tt, jj] = orc.bf16_to_f32(orc.int4_gemv_bf16(hid, w2[0], w2[1], None, None, I, H, g))
    ref = np.zeros((T, H), np.float32)
    for j in range(k):
        ws = orc.bf16_to_f32(orc.f32_to_bf16(wn[:, j].astype(np.float32)))[:, None]
        ref = orc.bf16_to_f32(orc.f32_to_bf16(ref + orc.bf16_to_f32(orc.f32_to_bf16(eo[:, j] * ws))))
    try:
        for rows in (-1, 0, 16, 64):
            if rows == 0 and T > 64:
                continue                                  # 512 passes over the experts: covered at the smaller sizes
            lib.vmx_debug_moe_block_rows(rows)
            out = vmx.moe_int4_forward(xd, w, ids, m["t13"], m["t2"], E, H, I, g)
            torch.cuda.synchronize()
            got = orc.bf16_to_f32(out.cpu().view(torch.int16).numpy().view(np.uint16))
            rel = np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum())
            assert np.isfinite(got).all() and rel <= 1e-2, (rows, rel)
            assert np.abs(got - ref).max() <= 2 ** -5 * np.abs(ref).max(), (rows, np.abs(got - ref).max(), np.abs(ref).max())
    finally:
        lib.vmx_debug_moe_block_rows(-1)


@pytest.mark.parametrize("T", [1, 3])
def test_moe_int4_forward_matches_reference_loop(gpu, T):
    from oracle import oracle as orc
    vmx, torch = gpu
    rng = np.random.default_rng(T)
    E, k, H, I, g = 4, 2, 256, 512, 128
    packs13, packs2, experts = [], [], []
    for ex in range(E):
        def mk(K, N):
            qw = rng.integers(0, 2 ** 32, size=(K // 8, N), dtype=np.uint32)
            sc = orc.f32_to_bf16(((rng.random((K // g, N)) * 0.5 + 0.75) / (6.5 * np.sqrt(K))).astype(np.float32))
            return qw, sc
        w1, w3, w2 = mk(H, I), mk(H, I), mk(I, H)          # GPTQ symmetric experts (no qzeros)
        experts.append((w1, w3, w2))
        qw13 = np.concatenate([w1[0], w3[0]], 1); sc13 = np.concatenate([w1[1], w3[1]], 1)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if a.dtype == np.uint32 else np.int16)).cuda()
        packs13.append(vmx.int4_pack(d(qw13), d(sc13), None, H, 2 * I, g))
        packs2.append(vmx.int4_pack(d(w2[0]), d(w2[1]), None, I, H, g))
    t13 = torch.tensor([p.data_ptr() for p in packs13], dtype=torch.int64, device="cuda")
    t2 = torch.tensor([p.data_ptr() for p in packs2], dtype=torch.int64, device="cuda")
    x = orc.f32_to_bf16(rng.standard_normal((T, H)).astype(np.float32))
    logits = rng.standard_normal((T, E)).astype(np.float32)
    xd = torch.from_numpy(x.view(np.int16)).cuda().view(torch.bfloat16)
    w, ids = vmx.topk_softmax(torch.from_numpy(logits).cuda(), k, True, True)
    out = vmx.moe_int4_forward(xd, w, ids, t13, t2, E, H, I, g)
    got = orc.bf16_to_f32(out.cpu().view(torch.int16).numpy().view(np.uint16))
    wn, idn = w.cpu().numpy(), ids.cpu().numpy()
    ref = np.zeros((T, H), np.float32)
    for t in range(T):                                      # mixtral_quantized.rs:152-190
        acc = np.zeros(H, np.float32)
        for j in range(k):
            w1, w3, w2 = experts[idn[t, j]]
            gate = orc.int4_gemv_bf16(x[t:t + 1], w1[0], w1[1], None, None, H, I, g)
            up = orc.int4_gemv_bf16(x[t:t + 1], w3[0], w3[1], None, None, H, I, g)
            hid = orc.silu_and_mul_bf16(gate, up)
            eo = orc.bf16_to_f32(orc.int4_gemv_bf16(hid, w2[0], w2[1], None, None, I, H, g))[0]
            ws = orc.bf16_to_f32(orc.f32_to_bf16(np.float32(wn[t, j])))
            scaled = orc.bf16_to_f32(orc.f32_to_bf16(eo * ws))
            acc = orc.bf16_to_f32(orc.f32_to_bf16(acc + scaled))
        ref[t] = acc
    assert np.abs(got - ref).max() <= 2 ** -6 * np.abs(ref).max()
    rel = np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum())
    assert rel <= 1e-2

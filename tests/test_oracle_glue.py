"""FP8 known answers (kv_cache/quantization.rs:654-692) and self-checks of the
glue-operator oracle against float64 numpy."""
import numpy as np

from oracle import oracle as orc


def test_fp8_known_answers():
    assert orc.fp8_e4m3_encode_trunc(0.0) == 0 and orc.fp8_e4m3_decode(0) == 0.0
    for enc in (orc.fp8_e4m3_encode_trunc, orc.fp8_e4m3_encode_rne):
        assert orc.fp8_e4m3_decode(enc(1.0)) == 1.0
        assert orc.fp8_e4m3_decode(enc(-2.0)) == -2.0
        assert orc.fp8_e4m3_decode(enc(1000.0)) == 448.0        # clamp
        assert orc.fp8_e4m3_decode(enc(-1000.0)) == -448.0


def test_fp8_rne_exhaustive_round_trip_and_midpoints():
    vals = [orc.fp8_e4m3_decode(b) for b in range(0x7F)]        # 0 .. 448, increasing
    assert all(vals[i] < vals[i + 1] for i in range(len(vals) - 1))
    for b, v in enumerate(vals):
        assert orc.fp8_e4m3_encode_rne(v) == b
        assert orc.fp8_e4m3_encode_rne(-v) == (b | 0x80)
    for b in range(0x7E):
        mid = (vals[b] + vals[b + 1]) / 2.0
        assert orc.fp8_e4m3_encode_rne(mid) == (b if b % 2 == 0 else b + 1)   # ties to even
        assert orc.fp8_e4m3_encode_rne(np.nextafter(np.float32(mid), np.float32(1e9))) == b + 1
        assert orc.fp8_e4m3_encode_rne(np.nextafter(np.float32(mid), np.float32(-1e9))) == b


def test_rms_norm_and_fused_add():
    rng = np.random.default_rng(1)
    x = orc.f32_to_bf16(rng.standard_normal((3, 256)).astype(np.float32))
    w = orc.f32_to_bf16(rng.standard_normal(256).astype(np.float32))
    out = orc.bf16_to_f32(orc.rms_norm_bf16(x, w, 1e-6))
    xf, wf = orc.bf16_to_f32(x).astype(np.float64), orc.bf16_to_f32(w).astype(np.float64)
    ref = xf / np.sqrt((xf ** 2).mean(-1, keepdims=True) + 1e-6) * wf
    assert np.abs(out - ref).max() <= 2 ** -7 * np.abs(ref).max()
    res = orc.f32_to_bf16(rng.standard_normal((3, 256)).astype(np.float32))
    out2, res2 = orc.fused_add_rms_norm_bf16(x, res, w, 1e-6)
    assert np.array_equal(res2, orc.f32_to_bf16(orc.bf16_to_f32(x) + orc.bf16_to_f32(res)))


def test_rope_rotation_preserves_norm():
    rng = np.random.default_rng(2)
    T, H, KVH, D = 2, 4, 2, 64
    q = orc.f32_to_bf16(rng.standard_normal((T, H * D)).astype(np.float32))
    k = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32))
    pos = np.array([3, 17], np.int32)
    inv = 1.0 / (10000.0 ** (np.arange(0, D, 2) / D))
    ang = np.arange(32)[:, None] * inv[None, :]
    cs = np.concatenate([np.cos(ang), np.sin(ang)], -1).astype(np.float32)
    q2, k2 = orc.rope_neox_bf16(pos, q, k, cs, D, D, H, KVH)
    n0 = np.linalg.norm(orc.bf16_to_f32(q).reshape(T, H, D), axis=-1)
    n1 = np.linalg.norm(orc.bf16_to_f32(q2).reshape(T, H, D), axis=-1)
    assert np.allclose(n0, n1, rtol=1e-2)


def test_paged_attention_matches_dense_softmax():
    rng = np.random.default_rng(3)
    S, H, KVH, D, BS = 2, 4, 2, 64, 16
    seq_lens = np.array([37, 150], np.int32)
    nblk = 32
    kc = orc.f32_to_bf16(rng.standard_normal((nblk, BS, KVH, D)).astype(np.float32))
    vc = orc.f32_to_bf16(rng.standard_normal((nblk, BS, KVH, D)).astype(np.float32))
    bt = np.stack([rng.permutation(nblk)[:10] for _ in range(S)]).astype(np.int32)
    q = orc.f32_to_bf16(rng.standard_normal((S, H * D)).astype(np.float32))
    scale = 1.0 / np.sqrt(D)
    for window in (0, 64):
        out = orc.bf16_to_f32(orc.paged_attention_v2(q, kc, vc, bt, seq_lens, scale, H, KVH, D, BS, 64, window))
        kf, vf, qf = orc.bf16_to_f32(kc).astype(np.float64), orc.bf16_to_f32(vc).astype(np.float64), orc.bf16_to_f32(q).astype(np.float64)
        for s in range(S):
            L = int(seq_lens[s]); start = L - window if (window and L > window) else 0
            toks = np.arange(start, L)
            kk = kf[bt[s, toks // BS], toks % BS]; vv = vf[bt[s, toks // BS], toks % BS]
            for h in range(H):
                lg = kk[:, h // 2] @ qf[s, h * D:(h + 1) * D] * scale
                p = np.exp(lg - lg.max()); p /= p.sum()
                ref = p @ vv[:, h // 2]
                assert np.abs(out[s, h * D:(h + 1) * D] - ref).max() <= 2e-2 * max(1.0, np.abs(ref).max())


def test_kv_fp8_cache_round_trip():
    rng = np.random.default_rng(4)
    KVH, D, BS = 2, 32, 16
    k = orc.f32_to_bf16(rng.standard_normal((3, KVH * D)).astype(np.float32))
    v = orc.f32_to_bf16(rng.standard_normal((3, KVH * D)).astype(np.float32))
    kc = np.zeros((4, BS, KVH, D), np.uint8); vc = np.zeros_like(kc)
    ks = float(np.abs(orc.bf16_to_f32(k)).max() * 1.5 / 448.0)
    orc.reshape_and_cache(k, v, kc, vc, np.array([5, -1, 40], np.int32), KVH, D, BS, 1, ks, ks)
    row = kc.reshape(-1, KVH * D)
    dec = np.array([orc.fp8_e4m3_decode(int(b)) for b in row[5]], np.float32) * np.float32(ks)
    assert np.abs(dec - orc.bf16_to_f32(k[0])).max() <= 0.07 * np.abs(orc.bf16_to_f32(k[0])).max()
    assert not row[6].any() and row[40].any()


def test_q6k_dequant_structure():
    blk = np.zeros(210, np.uint8)
    blk[192:208] = 1                       # scales = 1
    blk[208:210] = np.frombuffer(np.float16(1.0).tobytes(), np.uint8)
    out = orc.q6k_dequant(blk)
    assert (out == -32.0).all()            # q = 0 - 32
    blk[0] = 0x21; blk[128] = 0b00_01_00_11
    out = orc.q6k_dequant(blk)
    assert out[0] == (1 | (3 << 4)) - 32 and out[64] == (2 | (1 << 4)) - 32


def test_legacy_ggml_blocks_against_numpy_formulas():
    """Q4_1 / Q5_0 / Q5_1 (public ggml block specs; parity unpinned): the C loops against vectorised numpy formulas."""
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    for t, (ts, _) in ((3, orc.GGML_BLOCK[3]), (6, orc.GGML_BLOCK[6]), (7, orc.GGML_BLOCK[7])):
        nb = 50
        b = rng.integers(0, 256, size=(nb, ts), dtype=np.uint8)
        for off in ([0, 2] if t in (3, 7) else [0]):
            b[:, off] = 0; b[:, off + 1] = rng.integers(0x24, 0x2C, size=nb)
        d = b[:, 0:2].copy().view(np.float16).astype(np.float32)[:, 0]
        m = b[:, 2:4].copy().view(np.float16).astype(np.float32)[:, 0] if t in (3, 7) else None
        qoff = {3: 4, 6: 6, 7: 8}[t]
        qs = b[:, qoff:qoff + 16].astype(np.int32)
        lo, hi = qs & 0xF, qs >> 4
        if t != 3:
            hoff = 2 if t == 6 else 4
            qh = b[:, hoff:hoff + 4].copy().view(np.uint32)[:, 0].astype(np.int64)
            j = np.arange(16)
            lo = lo | (((qh[:, None] >> j) & 1) << 4)
            hi = hi | (((qh[:, None] >> (j + 16)) & 1) << 4)
        if t == 6:
            exp = np.concatenate([(lo - 16) * d[:, None], (hi - 16) * d[:, None]], axis=1)
        else:
            exp = np.concatenate([lo.astype(np.float32) * d[:, None] + m[:, None], hi.astype(np.float32) * d[:, None] + m[:, None]], axis=1)
        got = orc.ggml_dequant(t, b.reshape(-1)).reshape(nb, 32)
        assert np.array_equal(got, exp.astype(np.float32)), t


def test_fp8_w8a8_block_and_fused_norm_restatements_against_numpy():
    """The FP8 restatements checked against independent float64 numpy forms on small cases: scaled_mm (all scale modes, bias),
    block-scaled dequant (ragged blocks), fused rmsnorm + dynamic quant (scale and decoded values), KVScales first-write rule."""
    rng = np.random.default_rng(0)
    dec = np.array([orc.fp8_e4m3_decode(i) for i in range(256)], np.float64)
    M, N, K = 3, 5, 48
    A = rng.integers(0, 0x60, (M, K)).astype(np.uint8); B = rng.integers(0, 0x60, (N, K)).astype(np.uint8)
    sa = rng.random(M).astype(np.float32) + 0.5; sb = rng.random(N).astype(np.float32) + 0.5
    bias = orc.f32_to_bf16(rng.standard_normal(N).astype(np.float32))
    exact = (dec[A] * sa[:, None]) @ (dec[B] * sb[:, None]).T + orc.bf16_to_f32(bias)[None, :]
    got = orc.fp8_scaled_mm(A, B, sa, sb, bias, True, True, True)
    assert np.allclose(got, exact, rtol=1e-5)
    got1 = orc.fp8_scaled_mm(A, B, sa[:1], sb[:1], None, False, False, False)
    exact1 = (dec[A] * sa[0]) @ (dec[B] * sb[0]).T
    assert np.allclose(orc.bf16_to_f32(got1), exact1, rtol=2 ** -7)
    W = rng.integers(0, 0x78, (5, 12)).astype(np.uint8)
    sc = (rng.random((3, 2)).astype(np.float32) + 0.5)
    exp = dec[W] * np.repeat(np.repeat(sc, 2, axis=0), 8, axis=1)[:5, :12]
    assert np.array_equal(orc.fp8_dequant_block_bf16(W, sc, 2, 8), orc.f32_to_bf16(exp.astype(np.float32)))
    x = orc.f32_to_bf16(rng.standard_normal((2, 64)).astype(np.float32)); w = orc.f32_to_bf16(np.ones(64, np.float32) * 1.5)
    q, s = orc.fused_rmsnorm_fp8(x, w, 1e-6)
    xf = orc.bf16_to_f32(x).astype(np.float64)
    normed = xf / np.sqrt((xf ** 2).mean(axis=1, keepdims=True) + 1e-6) * 1.5
    assert np.allclose(s, np.abs(normed).max(axis=1) / 448.0, rtol=1e-6)
    assert np.abs(dec[q] * s[:, None] - normed).max() <= np.abs(normed).max() * 2 ** -4     # e4m3: 3 mantissa bits
    assert (np.abs(dec[q]).max(axis=1) == 448.0).all()                                       # the row maximum lands on +-448
    kv = orc.KVScales()
    k = orc.f32_to_bf16(np.array([0.5, -2.0, 1.0], np.float32)); v = orc.f32_to_bf16(np.zeros(3, np.float32))
    kv.calibrate_if_needed(k, v)
    assert kv.calibrated and kv.k_scale == np.float32(np.float32(2.0 * 1.5) * np.float32(1 / 448)) and kv.v_scale == np.float32(1e-12)
    kv.calibrate_if_needed(orc.f32_to_bf16(np.array([100.0], np.float32)), v)
    assert kv.k_scale == np.float32(np.float32(3.0) * np.float32(1 / 448))


def test_rope_gptj_pairs_adjacent_elements_and_matches_float64():
    """rope.cu:34-45: pair (2i, 2i + 1) rotates by angle i; against a float64 rotation, one bf16 rounding."""
    rng = np.random.default_rng(5)
    T, H, KVH, D, R = 3, 2, 1, 16, 8
    q = orc.f32_to_bf16(rng.standard_normal((T, H * D)).astype(np.float32))
    k = orc.f32_to_bf16(rng.standard_normal((T, KVH * D)).astype(np.float32))
    pos = np.array([0, 3, 7], np.int32)
    ang = np.arange(8)[:, None] * (1.0 / (10000.0 ** (np.arange(R // 2) * 2.0 / R)))[None, :]
    cs = np.concatenate([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)
    q2, k2 = orc.rope_gptj_bf16(pos, q, k, cs, R, D, H, KVH)
    qf = orc.bf16_to_f32(q).reshape(T, H, D).astype(np.float64)
    want = qf.copy()
    for t in range(T):
        c, s = cs[pos[t], : R // 2].astype(np.float64), cs[pos[t], R // 2:].astype(np.float64)
        x, y = qf[t, :, 0:R:2], qf[t, :, 1:R:2]
        want[t, :, 0:R:2] = x * c - y * s
        want[t, :, 1:R:2] = y * c + x * s
    got = orc.bf16_to_f32(q2).reshape(T, H, D)
    assert np.abs(got - want).max() <= 2.0 ** -8 * np.abs(want).max() + 1e-6
    assert np.array_equal(got[:, :, R:], qf[:, :, R:].astype(np.float32))     # dims past rot_dim untouched
    assert np.array_equal(orc.bf16_to_f32(q2[0]), orc.bf16_to_f32(q[0]))       # position 0: identity
    assert k2.shape == k.shape


def test_reshape_and_cache_hnd_and_copy_blocks():
    """cache_ops.cu:106-202: HND placement [blocks, kv_heads, block_size, head_dim]; copy_blocks copies whole blocks in both caches."""
    rng = np.random.default_rng(6)
    T, KVH, D, BS, NB = 3, 2, 8, 4, 5
    k = rng.integers(0, 65536, (T, KVH * D)).astype(np.uint16)
    v = rng.integers(0, 65536, (T, KVH * D)).astype(np.uint16)
    kc = np.zeros((NB, KVH, BS, D), np.uint16)
    vc = np.zeros_like(kc)
    slots = np.array([6, -1, 17], np.int32)
    orc.reshape_and_cache_hnd(k, v, kc, vc, slots, KVH, D, BS)
    assert np.array_equal(kc[1, :, 2, :], k[0].reshape(KVH, D)) and np.array_equal(vc[4, :, 1, :], v[2].reshape(KVH, D))
    assert kc.astype(np.int64).sum() == k[0].astype(np.int64).sum() + k[2].astype(np.int64).sum()     # the skipped row wrote nothing
    mapping = np.array([[1, 0], [4, 3]], np.int32)
    k_before = kc.copy()
    orc.copy_blocks(kc.view(np.uint8), vc.view(np.uint8), mapping, KVH * BS * D * 2)
    assert np.array_equal(kc[0], k_before[1]) and np.array_equal(kc[3], k_before[4]) and np.array_equal(kc[1], k_before[1])
    assert np.array_equal(vc[3], vc[4])


def test_int8_kv_quantiser_known_values():
    """quantize_int8 (kv_cache/quantization.rs:330-345): clamp(x / scale, +-127), round half away from zero, + 128."""
    vals = np.array([[0.0, 1.0, -1.0, 0.5, -0.5, 1.5, 200.0, -200.0]], np.float32)
    k = orc.f32_to_bf16(vals)
    kc = np.zeros((1, 1, 1, 8), np.uint8); vc = np.zeros_like(kc)
    orc.reshape_and_cache(k, k, kc, vc, np.array([0], np.int32), 1, 8, 1, 2, 1.0, 0.5)
    assert kc.reshape(-1).tolist() == [128, 129, 127, 129, 127, 130, 255, 1]
    assert vc.reshape(-1).tolist() == [128, 130, 126, 129, 127, 131, 255, 1]


def test_fp8_e5m2_is_the_top_byte_of_a_half_and_rounds_to_nearest_even():
    """load_kv_to_f32<fp8_e5m2_byte> (kernels/paged_attention.cu:77-82): the byte is the high half of an IEEE fp16."""
    vals = [orc.fp8_e5m2_decode(b) for b in range(0x7C)]         # 0 .. 57344, increasing; 0x7C is inf
    assert vals[0x3C] == 1.0 and vals[0x7B] == 57344.0 and vals[1] == 2.0 ** -16
    for b, v in enumerate(vals):
        assert v == float(np.array([b << 8], np.uint16).view(np.float16)[0])
        assert orc.fp8_e5m2_encode_rne(v) == b and orc.fp8_e5m2_encode_rne(-v) == (b | 0x80)
    for b in range(0x7B):
        mid = (vals[b] + vals[b + 1]) / 2.0
        assert orc.fp8_e5m2_encode_rne(mid) == (b if b % 2 == 0 else b + 1)
        assert orc.fp8_e5m2_encode_rne(np.nextafter(np.float32(mid), np.float32(1e9))) == b + 1
        assert orc.fp8_e5m2_encode_rne(np.nextafter(np.float32(mid), np.float32(-1e9))) == b
    assert orc.fp8_e5m2_encode_rne(1e9) == 0x7B and orc.fp8_e5m2_encode_rne(-1e9) == 0xFB     # clamp, never inf


def test_paged_attention_lse_matches_dense_logsumexp():
    """per_seq_decode_with_lse (layers/attention/naive.rs:151-205) on a paged cache against float64 numpy."""
    rng = np.random.default_rng(5)
    H, KVH, D, BS, NB, MB = 4, 2, 64, 16, 8, 4
    seq_lens = np.array([0, 1, 37], np.int32)
    kc = orc.f32_to_bf16(rng.standard_normal((NB, BS, KVH, D)).astype(np.float32))
    q = orc.f32_to_bf16(rng.standard_normal((3, H * D)).astype(np.float32))
    bt = np.stack([rng.permutation(NB)[:MB] for _ in range(3)]).astype(np.int32)
    lse = orc.paged_attention_lse(q, kc, bt, seq_lens, 0.125, H, KVH, D, BS, window=20)
    assert np.all(np.isneginf(lse[0]))
    kf, qf = orc.bf16_to_f32(kc).astype(np.float64), orc.bf16_to_f32(q).astype(np.float64).reshape(3, H, D)
    for s in (1, 2):
        n = int(seq_lens[s]); start = max(n - 20, 0)
        for h in range(H):
            keys = np.stack([kf[bt[s, t // BS], t % BS, h // (H // KVH)] for t in range(start, n)])
            sc = keys @ qf[s, h] * 0.125
            assert abs(lse[s, h] - (sc.max() + np.log(np.exp(sc - sc.max()).sum()))) < 1e-12

"""MFMA prefill attention over the paged cache against a float64 restatement of the reference's naive backend
(write K/V, read the history back, softmax(QK^T * scale + causal/window mask) V; layers/attention/naive.rs:38-83).
Parity unpinned (the reference holds no golden vectors; its GPU build uses a third-party kernel): tolerance only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dequant_cache(orc, cache, kv_dtype, scale):
    if kv_dtype == 0:
        return orc.bf16_to_f32(cache).astype(np.float64)
    lut = np.array([orc.fp8_e4m3_decode(b) for b in range(256)], np.float64)
    return lut[cache] * scale


@pytest.mark.parametrize("hd,H,KVH,q_len,offset,window,kv_dtype", [
    (128, 8, 2, 37, 0, 0, 0),       # ragged last tile, QPK 4
    (128, 10, 2, 41, 3, 0, 0),      # 5 query heads per KV head (Qwen3-14B's ratio): capacity 8, three idle head slots
    (64, 6, 2, 33, 0, 12, 1),       # 3 per KV head: capacity 4, one idle wave per workgroup
    (64, 4, 4, 16, 5, 0, 0),        # QPK 1: four token tiles per workgroup, chunk with history
    (256, 4, 2, 50, 20, 8, 1),      # sliding window smaller than a tile, FP8 cache, QPK 2
    (512, 16, 1, 19, 3, 0, 1),      # Gemma full-attention geometry: 16 query heads on one 512-wide KV head
    (128, 8, 1, 70, 0, 33, 0),      # QPK 8 (two head groups), window across tiles
    (128, 32, 8, 1, 64, 0, 0),      # a single new token (decode-shaped call)
    # 64-token KV tiles x 32-row query blocks (64- / 128-wide heads): several tiles and blocks, ragged ends, history, windows
    (128, 32, 8, 300, 0, 0, 0),     # Qwen3-8B heads: 10 query blocks, 5 KV tiles, four heads of a KV head per workgroup
    (128, 8, 4, 200, 77, 0, 1),     # two heads x two query blocks per workgroup, FP8 cache, history not aligned to a tile
    (64, 4, 4, 130, 50, 0, 0),      # one head x four query blocks per workgroup, 64-wide heads
    (128, 8, 2, 150, 100, 70, 0),   # window across tiles with history: the first tiles of a workgroup are skipped
    (64, 12, 4, 97, 0, 40, 1),      # three heads per KV head (one staging-only wave), window, FP8, 64-wide heads
    (128, 8, 2, 64, 64, 0, 0),      # exactly one tile of history + one tile of new tokens
    (128, 16, 8, 12, 0, 0, 0),      # Qwen3-0.6B heads, a prompt shorter than one query block (second block of the workgroup idle)
    (128, 8, 4, 100, 3, 0, 0),      # two heads x two query blocks, bf16 cache
    (128, 8, 2, 90, 0, 0, 3),       # fp16 queries, cache and output (models that run in fp16)
    (64, 4, 2, 70, 9, 30, 3),       # ... 64-wide heads, window, history
    (256, 4, 2, 40, 5, 0, 3),       # ... wide heads (16-row kernel)
])
def test_prefill_attention_matches_float64(hd, H, KVH, q_len, offset, window, kv_dtype):
    import torch
    import vllm_rust_amd as vmx
    from oracle import oracle as orc
    from vllm_rust_amd._lib import check, lib
    vmx.init(0)
    rng = np.random.default_rng(hd + H + q_len + offset)
    bs, total = 16, offset + q_len
    nblk = (total + bs - 1) // bs + 1
    block_table = rng.permutation(nblk + 3)[:nblk].astype(np.int32)
    f16 = kv_dtype == 3
    kv_np = np.uint8 if kv_dtype == 1 else np.uint16
    kc = np.zeros((nblk + 3, bs, KVH, hd), kv_np); vc = np.zeros_like(kc)
    to16 = (lambda a: a.astype(np.float16).view(np.uint16)) if f16 else orc.f32_to_bf16
    from16 = (lambda u: u.view(np.float16).astype(np.float32)) if f16 else orc.bf16_to_f32
    k = to16(rng.standard_normal((total, KVH * hd)).astype(np.float32))
    v = to16(rng.standard_normal((total, KVH * hd)).astype(np.float32))
    pos = np.arange(total)
    slots = (block_table[pos // bs] * bs + pos % bs).astype(np.int32)
    ks, vs = (0.02, 0.015) if kv_dtype == 1 else (1.0, 1.0)
    if f16:   # 16-bit copy
        kc.reshape(-1, KVH * hd)[slots] = k; vc.reshape(-1, KVH * hd)[slots] = v
    else:
        orc.reshape_and_cache(k, v, kc, vc, slots, KVH, hd, bs, kv_dtype, ks, vs)
    q = to16((rng.standard_normal((q_len, H, hd)) * 0.7).astype(np.float32))
    scale = 1.0 / np.sqrt(hd)
    # oracle
    if f16:
        kf = from16(kc).astype(np.float64).reshape(-1, KVH, hd)[slots]
        vf = from16(vc).astype(np.float64).reshape(-1, KVH, hd)[slots]
    else:
        kf = _dequant_cache(orc, kc, kv_dtype, ks).reshape(-1, KVH, hd)[slots]      # [total, KVH, hd] in position order
        vf = _dequant_cache(orc, vc, kv_dtype, vs).reshape(-1, KVH, hd)[slots]
    qf = from16(q).astype(np.float64)
    exp = np.zeros((q_len, H, hd))
    for h in range(H):
        kvh = h // (H // KVH)
        s = qf[:, h, :] @ kf[:, kvh, :].T * scale                                # [q_len, total]
        qp = offset + np.arange(q_len)[:, None]
        tp = np.arange(total)[None, :]
        mask = tp <= qp
        if window > 0:
            mask &= tp > qp - window
        s = np.where(mask, s, -np.inf)
        p = np.exp(s - s.max(axis=1, keepdims=True))
        p /= p.sum(axis=1, keepdims=True)
        exp[:, h, :] = p @ vf[:, kvh, :]
    # device
    dev = "cuda"
    out = torch.full((q_len + 1, H * hd), 7.0, dtype=torch.float16 if f16 else torch.bfloat16, device=dev)
    qd = torch.from_numpy(q.view(np.int16)).to(dev)
    kcd, vcd = torch.from_numpy(kc.view(np.uint8 if kv_dtype == 1 else np.int16)).to(dev), torch.from_numpy(vc.view(np.uint8 if kv_dtype == 1 else np.int16)).to(dev)
    btd = torch.from_numpy(block_table).to(dev)
    sc = torch.tensor([ks, vs], dtype=torch.float32, device=dev)
    if f16:
        check(lib.vmx_prefill_attention_f16(out.data_ptr(), qd.data_ptr(), kcd.data_ptr(), vcd.data_ptr(), btd.data_ptr(), q_len, offset,
                                            float(scale), H, KVH, hd, bs, window, torch.cuda.current_stream().cuda_stream))
    else:
        check(lib.vmx_prefill_attention_bf16(out.data_ptr(), qd.data_ptr(), kcd.data_ptr(), vcd.data_ptr(), btd.data_ptr(), q_len, offset,
                                             float(scale), H, KVH, hd, bs, window, kv_dtype, sc.data_ptr(), sc.data_ptr() + 4,
                                             torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.all(got[q_len] == 7.0)                                             # nothing written past the last query row
    got = got[:q_len].reshape(q_len, H, hd)
    assert np.isfinite(got).all()
    rel = np.sqrt(((got - exp) ** 2).sum() / (exp ** 2).sum())
    print(f"rel-L2 {rel:.3e} max-abs {np.abs(got - exp).max():.3e}")
    assert rel <= (1.5e-3 if f16 else 6e-3), rel                                 # probabilities and output rounded to the 16-bit type
    assert np.abs(got - exp).max() <= 0.03 * np.abs(exp).max() + 1e-3


def test_prefill_attention_equals_decode_kernel_per_token():
    """Each query row must agree with the decode kernel run on the same token as a pseudo-sequence (the way prompt chunks
    were served before this kernel existed)."""
    import torch
    import vllm_rust_amd as vmx
    from oracle import oracle as orc
    from vllm_rust_amd._lib import check, lib
    vmx.init(0)
    rng = np.random.default_rng(9)
    hd, H, KVH, q_len, offset, bs = 128, 8, 2, 40, 7, 16
    total = offset + q_len
    nblk = (total + bs - 1) // bs
    kc = torch.from_numpy(orc.f32_to_bf16(rng.standard_normal((nblk, bs, KVH, hd)).astype(np.float32)).view(np.int16)).cuda()
    vc = torch.from_numpy(orc.f32_to_bf16(rng.standard_normal((nblk, bs, KVH, hd)).astype(np.float32)).view(np.int16)).cuda()
    q = torch.from_numpy(orc.f32_to_bf16(rng.standard_normal((q_len, H, hd)).astype(np.float32)).view(np.int16)).cuda()
    bt1 = torch.arange(nblk, dtype=torch.int32, device="cuda")
    out = torch.empty((q_len, H * hd), dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    scale = float(1.0 / np.sqrt(hd))
    check(lib.vmx_prefill_attention_bf16(out.data_ptr(), q.data_ptr(), kc.data_ptr(), vc.data_ptr(), bt1.data_ptr(), q_len, offset, scale,
                                         H, KVH, hd, bs, 0, 0, None, None, st))
    sl = torch.arange(offset + 1, total + 1, dtype=torch.int32, device="cuda")
    ref = vmx.paged_attention(q.view(torch.bfloat16), kc.view(torch.bfloat16), vc.view(torch.bfloat16), bt1.reshape(1, -1).expand(q_len, -1).contiguous(),
                              sl, scale, H, KVH, hd, bs, total)
    torch.cuda.synchronize()
    a, b = out.float(), ref.float().reshape(q_len, H * hd)
    assert float(((a - b) ** 2).sum().sqrt() / (b ** 2).sum().sqrt()) <= 6e-3

"""vmx_paged_attention_bf16_static (split-KV partials in the library's static per-GPU array, no allocation) against
vmx_paged_attention_bf16 (caller workspace): bit-identical outputs on two calls in a row, the second with a smaller input,
so that stale partials of the first call would show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(torch, vmx, rng, S, H, KVH, D, max_len, kv_dtype, window):
    BS, NB = 16, 160
    MB = (max_len + BS - 1) // BS
    seq_lens = rng.integers(1, max_len + 1, size=S).astype(np.int32)
    seq_lens[0] = max_len
    if kv_dtype == 0:
        kc = torch.randn((NB, BS, KVH, D), device="cuda").to(torch.bfloat16)
        vc = torch.randn((NB, BS, KVH, D), device="cuda").to(torch.bfloat16)
    else:
        kc = torch.randint(0, 120, (NB, BS, KVH, D), dtype=torch.uint8, device="cuda")   # finite e4m3 bytes
        vc = torch.randint(0, 120, (NB, BS, KVH, D), dtype=torch.uint8, device="cuda")
    bt = torch.from_numpy(np.stack([rng.permutation(NB)[:MB] for _ in range(S)]).astype(np.int32)).cuda()
    q = torch.randn((S, H * D), device="cuda").to(torch.bfloat16)
    sl = torch.from_numpy(seq_lens).cuda()
    sk = torch.tensor([0.05], dtype=torch.float32, device="cuda"); sv = torch.tensor([0.07], dtype=torch.float32, device="cuda")
    args = (q, kc, vc, bt, sl, float(1.0 / np.sqrt(D)), H, KVH, D, BS, int(max_len), window, kv_dtype, sk, sv)
    ref = vmx.paged_attention(*args)
    out = torch.full_like(ref, 7.0)
    free0 = torch.cuda.mem_get_info()[0]
    vmx.paged_attention_static(*args, out=out)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0          # the call took no device memory
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


def test_static_scratch_attention_equals_workspace_attention():
    import torch
    import vllm_rust_amd as vmx
    from vllm_rust_amd._lib import VmxError, lib
    vmx.init(0)
    rng = np.random.default_rng(11)
    _case(torch, vmx, rng, S=6, H=32, KVH=8, D=128, max_len=1500, kv_dtype=0, window=0)     # many partitions
    _case(torch, vmx, rng, S=2, H=32, KVH=8, D=128, max_len=70, kv_dtype=0, window=0)       # smaller: 5 tiles, old partials beside it
    _case(torch, vmx, rng, S=3, H=16, KVH=8, D=256, max_len=300, kv_dtype=1, window=100)    # FP8 cache, sliding window
    _case(torch, vmx, rng, S=1, H=10, KVH=2, D=64, max_len=17, kv_dtype=0, window=0)        # 5 query heads per KV head
    # beyond the static array: rejected before any launch, with the limit in the message
    assert lib.vmx_paged_attention_static_max_bytes() == 32 << 20
    z = torch.zeros(16, dtype=torch.int32, device="cuda")
    with pytest.raises(VmxError, match="static"):
        vmx.paged_attention_static(z, z, z, z.reshape(16, 1), z, 1.0, 64, 8, 512, 16, 65536)

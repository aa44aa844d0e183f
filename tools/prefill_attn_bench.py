"""Time vmx_prefill_attention_bf16 on Qwen3-8B / Gemma geometries: causal FLOPs = 4 * H * hd * S * (S + 1) / 2."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vllm_rust_amd as vmx
from vllm_rust_amd._lib import lib, check
vmx.init(0)
bs = 16
for (H, KVH, hd, window, fp8) in [(32, 8, 128, 0, 0), (16, 8, 256, 1024, 1), (16, 1, 512, 0, 1)]:
    for S in (512, 2048, 8192):
        nblk = (S + bs - 1) // bs
        dt = torch.uint8 if fp8 else torch.bfloat16
        if fp8:
            kc = torch.randint(0, 120, (nblk, bs, KVH, hd), dtype=torch.uint8, device="cuda"); vc = kc.clone()
        else:
            kc = torch.randn((nblk, bs, KVH, hd), device="cuda").to(dt); vc = torch.randn_like(kc)
        q = torch.randn((S, H, hd), device="cuda").to(torch.bfloat16)
        out = torch.empty((S, H * hd), dtype=torch.bfloat16, device="cuda")
        bt = torch.arange(nblk, dtype=torch.int32, device="cuda")
        sc = torch.tensor([0.02, 0.02], dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        def run():
            check(lib.vmx_prefill_attention_bf16(out.data_ptr(), q.data_ptr(), kc.data_ptr(), vc.data_ptr(), bt.data_ptr(), S, 0, hd ** -0.5,
                                                 H, KVH, hd, bs, window, fp8, sc.data_ptr(), sc.data_ptr() + 4, st))
        run(); torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n): run()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        if window:
            pairs = sum(min(p + 1, window) for p in range(S))
        else:
            pairs = S * (S + 1) // 2
        fl = 4.0 * H * hd * pairs
        print(f"H={H} KVH={KVH} hd={hd} window={window} fp8={fp8} S={S}: {us:9.1f} us  {fl/us/1e6:7.1f} TFLOP/s", flush=True)

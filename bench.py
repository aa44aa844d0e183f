#!/usr/bin/env python3
"""Headline benchmark: greedy decode tok/s of a Qwen3-8B-shaped AWQ int4 model on MI355X.

A "step" is one decode token for the whole batch (batch = 1 per GPU: BASELINE.json configs[1],
"Qwen3-8B AWQ int4 TP=1 on one MI355X, packed-int GEMM decode, batch=1 seq=1").  Weights are a
seeded random-init AWQ checkpoint of the Qwen3-8B architecture (no network, no model files); a
256-token prompt first goes through the native prefill entry point (untimed for `value`, reported
under `prefill`), then W warm-up and K timed steps are replayed from one hipGraph (decode step +
device-side step bookkeeping), with all inputs resident in HBM.

With --gpus N > 1 each rank (one process per GPU, launched by torch.distributed.run) runs an
independent replica of the same job (data-parallel serving, no exchange step on the data path);
value is the sum over ranks, time is the max over ranks.  --tp instead shards ONE model over the
ranks (column/row-parallel int4 linears, RCCL all-reduce over xGMI, vocab-parallel lm_head).

Prints one JSON line on rank 0 with `roofline` (the int4 linear kernel, timed with HIP events on
its launch stream over back-to-back replays of the step's linear launches) and `cpu_baseline` (the CPU oracle's
int4 GEMV, OpenMP over output columns, on a bounded sample of the same per-token work).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def _host_cores():
    cores = len(os.sched_getaffinity(0))
    return max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores))))


def _timed(fn, budget_s, max_reps=10 ** 6):
    """Average seconds of fn() over as many repetitions as fit the budget (at least one)."""
    t_sum, reps, t_start = 0.0, 0, time.perf_counter()
    while True:
        t0 = time.perf_counter()
        fn()
        t_sum += time.perf_counter() - t0
        reps += 1
        if time.perf_counter() - t_start > budget_s or reps >= max_reps:
            return t_sum / reps, reps


def cpu_baseline(spec, budget_s=10.0):
    """The CPU oracle timed on the GPU box's host cores, on a bounded sample of the same workload: the int4 GEMV (bf16 in/out,
    f32 accumulate) of ONE decoder layer's seven projections and the dense bf16 lm_head dot on 1/16 of the vocabulary, with all
    the cores this process may use (OpenMP over output columns) and again with one thread; tokens/s = 1 / (layers x per-layer
    time + lm_head time).  Norms, RoPE and attention are left out (a few per cent of the CPU step)."""
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    H, I = spec.hidden_size, spec.intermediate_size
    nq, nkv = spec.num_heads * spec.head_dim, spec.num_kv_heads * spec.head_dim
    shapes = [(H, nq), (H, nkv), (H, nkv), (nq, H), (H, I), (H, I), (I, H)]
    mats = []
    for K, N in shapes:
        qw = rng.integers(0, 2 ** 32, size=(K // 8, N), dtype=np.uint32)
        qz = rng.integers(0, 2 ** 32, size=(K // spec.group_size, N // 8), dtype=np.uint32)
        sc = orc.f32_to_bf16((rng.random((K // spec.group_size, N), dtype=np.float32) * 0.01))
        x = orc.f32_to_bf16(rng.standard_normal((1, K)).astype(np.float32))
        mats.append((x, qw, sc, qz, K, N))
    Vs = spec.vocab_size // 16
    Wh = orc.f32_to_bf16(rng.standard_normal((Vs, H)).astype(np.float32) * 0.02)
    xh = orc.f32_to_bf16(rng.standard_normal((1, H)).astype(np.float32))

    def layer():
        for x, qw, sc, qz, K, N in mats:
            orc.int4_gemv_bf16(x, qw, sc, qz, None, K, N, spec.group_size)

    def head():
        orc.dense_gemv_bf16(xh, Wh)

    cores = _host_cores()
    prev = orc.set_threads(cores)
    t_layer, reps = _timed(layer, budget_s * 0.5)
    t_head, _ = _timed(head, budget_s * 0.1)
    t_head *= 16
    orc.set_threads(1)
    t_layer1, reps1 = _timed(layer, budget_s * 0.3, max_reps=2)
    t_head1, _ = _timed(head, budget_s * 0.1, max_reps=2)
    t_head1 *= 16
    orc.set_threads(prev)
    tok_s = 1.0 / (spec.num_layers * t_layer + t_head)
    tok_s1 = 1.0 / (spec.num_layers * t_layer1 + t_head1)
    return {"value": round(tok_s, 3), "unit": "tok/s", "cores": cores, "kind": "port",
            "sample": f"oracle int4 GEMV of one decoder layer's 7 projections x{reps} ({t_layer * 1e3:.1f} ms/layer), scaled to "
                      f"{spec.num_layers} layers + oracle bf16 lm_head dot on 1/16 of the vocabulary ({t_head * 1e3:.1f} ms for all of it)",
            "single_thread": {"value": round(tok_s1, 4), "unit": "tok/s", "cores": 1,
                              "sample": f"the same sample with one OpenMP thread x{reps1} ({t_layer1 * 1e3:.0f} ms/layer, lm_head {t_head1 * 1e3:.0f} ms)"}}


def cpu_baseline_mmvq(spec, slot_types, budget_s=8.0):
    """CPU leg of the GGUF I-quant path: the oracle's q8_1 MMVQ (integer sub-dots, f32 scales) of one Gemma-4-12B sliding layer's
    seven projections in the model's tensor-type mix, all cores and one thread; tokens/s over the layer count (token embedding
    and lm_head left out)."""
    from oracle import oracle as orc
    from vllm_rust_amd.gemma4 import IQ_TYPE_SIZE
    rng = np.random.default_rng(1)
    mats = []
    for slot, (K, N) in spec.linear_shapes(0).items():
        t = slot_types[slot]
        w = rng.integers(0, 256, size=(N * (K // 256), IQ_TYPE_SIZE[t]), dtype=np.uint8)
        w[:, 0:2] = np.frombuffer(np.float16(0.01).tobytes(), dtype=np.uint8)
        aq = rng.integers(-127, 128, size=(1, K), dtype=np.int8)
        ad = (rng.random((1, K // 32), dtype=np.float32) * 0.01).astype(np.float32)
        mats.append((t, w, aq, ad, N, K))

    def layer(threads):
        for t, w, aq, ad, N, K in mats:
            orc.iq_mmvq_q8(t, w, aq, ad, N, K, threads=threads)

    cores = _host_cores()
    prev = orc.set_threads(cores)
    t_mt, reps = _timed(lambda: layer(True), budget_s * 0.6)
    orc.set_threads(prev)
    t_1, reps1 = _timed(lambda: layer(False), budget_s * 0.4, max_reps=2)
    return {"value": round(1.0 / (spec.num_layers * t_mt), 3), "unit": "tok/s", "cores": cores, "kind": "port",
            "sample": f"oracle q8_1 MMVQ of one sliding layer's 7 I-quant projections x{reps} ({t_mt * 1e3:.1f} ms/layer), scaled to {spec.num_layers} layers",
            "single_thread": {"value": round(1.0 / (spec.num_layers * t_1), 4), "unit": "tok/s", "cores": 1,
                              "sample": f"the same layer with one thread x{reps1} ({t_1 * 1e3:.0f} ms/layer)"}}


def cpu_baseline_exl3(budget_s=8.0):
    """CPU leg of the EXL3 path: the oracle's Exl3Linear forward (Hadamard, trellis decode, fp16 GEMM, Hadamard) on a 1/8-wide
    slice of a Qwen3-8B q_proj at 4.0 bpw, all cores; scaled by weight count to the 36 layers' seven projections."""
    from oracle import oracle as orc
    rng = np.random.default_rng(2)
    K, N, bits = 4096, 512, 4
    tr = rng.integers(0, 65536, size=(K // 16, N // 16, 16 * bits), dtype=np.uint16)
    suh = orc.f32_to_f16(np.sign(rng.standard_normal(K)).astype(np.float32))
    svh = orc.f32_to_f16(np.sign(rng.standard_normal(N)).astype(np.float32))
    x = orc.f32_to_f16(rng.standard_normal((1, K)).astype(np.float32))
    cores = _host_cores()
    prev = orc.set_threads(cores)
    t, reps = _timed(lambda: orc.exl3_gemm(x, tr, suh, svh, K, N, bits, 0), budget_s)
    orc.set_threads(prev)
    per_weight = t / (K * N)
    layer_weights = 4096 * (4096 + 1024 + 1024) + 4096 * 4096 + 2 * 4096 * 12288 + 12288 * 4096
    return {"value": round(1.0 / (36 * layer_weights * per_weight), 4), "unit": "tok/s", "cores": cores, "kind": "port",
            "sample": f"oracle EXL3 linear [4096 -> 512] at 4.0 bpw x{reps} ({t * 1e3:.1f} ms), scaled by weight count to 36 layers x 7 projections"}


def prefill_block(tokens, seconds, spec):
    """The prompt pass of the primary: tok/s, the arithmetic rate of its linears (2 flop per weight per token, lm_head on the last
    row only) against the dense bf16 MFMA peak, and the MFMA-busy share measured by the committed counter passes of the same
    kernels (rocprofv3 cannot run inside this process)."""
    H, I = spec.hidden_size, spec.intermediate_size
    nq, nkv = spec.num_heads * spec.head_dim, spec.num_kv_heads * spec.head_dim
    flop = 2.0 * tokens * spec.num_layers * (H * (nq + 2 * nkv) + nq * H + 3 * H * I) + 2.0 * spec.vocab_size * H
    blk = {"tokens": tokens, "ms": round(seconds * 1e3, 3), "tok_s": round(tokens / seconds, 1),
           "linear_TFLOP_s": round(flop / seconds / 1e12, 1), "mfma_peak_TFLOP_s": 2500.0,
           "frac_of_mfma_peak": round(flop / seconds / 1e12 / 2500.0, 4),
           "note": "one vmx_qwen3_prefill call (64x256-tile MFMA int4 GEMM, 32x32-tile MFMA prompt attention); untimed part of the job"}
    try:
        j = json.loads((ROOT / "profiles" / "r03_pmc_prefill_mfma.json").read_text())
        blk["mfma_busy_frac"] = {f"{tag}: {name.split(' (')[0]}": v["mfma_busy_frac"] for tag in ("prompt_256", "prompt_2048")
                                 for name, v in j[tag].items()}
        blk["mfma_busy_source"] = ("profiles/r03_pmc_prefill_mfma.json (committed rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES pass over the same "
                                   "kernels at 256- and 2048-token prompts; not measured in this run)")
    except Exception:
        pass
    return blk


def time_linear_kinds(model, side, B, reps):
    """Each kind of linear launch of a decode step alone (and all four), captured into a graph on the model's own buffers and
    replayed back to back between two HIP events on the stream the kernels run on: duration per launch = elapsed / launches."""
    kinds = {}
    for name, mask in (("qkv", 1), ("o_proj", 2), ("gate_up", 4), ("down_proj", 8), ("all", 15)):
        model.linears_select(mask)
        launches, alg = model.linears_info(B)
        with torch.cuda.stream(side):
            model.linears_only(B)
            lg = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(lg, stream=side):
                model.linears_only(B)
            lg.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(reps):
                lg.replay()
            e1.record(side)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        del lg
        kinds[name] = {"launches_per_step": launches, "avg_launch_us": round(us / launches, 3), "alg_bytes_per_launch": round(alg / launches),
                       "GB_s": round(alg / us / 1e3, 1), "frac": round(alg / us / 1e3 / HBM_PEAK_GBS, 4), "us_per_step": round(us, 1)}
    model.linears_select(15)
    return kinds


def gemma4_secondary(prompt, steps, warmup, cpu_leg=True, batches=(4, 16)):
    """BASELINE.json configs[2]: decode tok/s of a Gemma-4-12B-shaped GGUF model (UD-IQ3_XXS tensor-type mix, Q3_K tied token
    embedding, FP8 e4m3 KV cache with per-layer sliding/full geometry) through the Gemma 4 executor, plus the roofline of
    its dominant kernel (the q8_1 MMVQ): the step's MMVQ launches alone replayed back to back between two HIP events."""
    from vllm_rust_amd.gemma4 import Gemma4Spec, QuantizedGemma4ForCausalLM
    total = prompt + warmup + steps + 8
    n_blocks = (total + 15) // 16 + 1
    BM = max((1,) + tuple(batches))
    spec = Gemma4Spec.gemma4_12b(sliding_window=1024, max_position=max(4096, total + 1), kv_cache_dtype=1, num_blocks=n_blocks * BM,
                                 max_batch=BM, max_seq_len=total + 1, embed_ggml_type=11, final_logit_softcap=30.0,
                                 max_prefill_tokens=prompt)   # prompt passes of more than 16 rows: dequantise once + matrix-core GEMM
    model = QuantizedGemma4ForCausalLM(spec)
    model.load_synthetic(seed=99)
    dev = model.device
    bt = torch.arange(n_blocks, dtype=torch.int32, device=dev).flip(0).reshape(1, n_blocks).contiguous()
    prompt_ids = torch.randint(0, spec.vocab_size, (prompt,), dtype=torch.int32, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    ids = torch.empty(1, dtype=torch.int32, device=dev)
    pos = torch.full((1,), prompt, dtype=torch.int32, device=dev)
    slots = (bt[:, prompt // 16] * 16 + prompt % 16).to(torch.int32).contiguous()
    sl = torch.full((1,), prompt + 1, dtype=torch.int32, device=dev)
    out = torch.empty(1, dtype=torch.int32, device=dev)

    def step():
        model.forward_decode_batch(ids, pos, slots, bt, sl, max_seq_len=spec.max_seq_len, out_ids=out)
        model.advance(ids, out, pos, slots, sl, bt)

    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        model.prefill(prompt_ids[:16], 0, bt[0], out_id=ids)   # untimed: first-call set-up
        torch.cuda.synchronize()
        t_pre = time.perf_counter()
        model.prefill(prompt_ids, 0, bt[0], out_id=ids)        # the prompt through the native prompt path (16-row passes)
        torch.cuda.synchronize()
        t_pre = time.perf_counter() - t_pre
        step()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            step()
        for _ in range(warmup):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # roofline of the MMVQ kernels: each kind of projection alone (and all four), replayed back to back from a graph between
        # two HIP events on the stream they run on
        def time_mmvq(mask):
            model.mmvq_select(mask)
            launches, alg = model.mmvq_info(1)
            model.mmvq_only(1)
            lg = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(lg, stream=side):
                model.mmvq_only(1)
            reps = max(8, steps // 4)
            lg.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(reps):
                lg.replay()
            e1.record(side)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            del lg
            return launches, alg, us

        kinds = {}
        for name, mask in (("qkv", 1), ("o_proj", 2), ("gate_up", 4), ("down_proj", 8), ("all", 15)):
            launches, alg, us = time_mmvq(mask)
            kinds[name] = {"launches_per_step": launches, "avg_launch_us": round(us / launches, 3), "alg_bytes_per_launch": round(alg / launches),
                           "GB_s": round(alg / us / 1e3, 1), "frac": round(alg / us / 1e3 / HBM_PEAK_GBS, 4), "us_per_step": round(us, 1)}
        model.mmvq_select(15)
    # concurrent sequences: from five rows on the I-quant linears run on the int8 matrix cores (one decode of a sub-block for all rows)
    by_batch = {"1": {"tok_s": round(steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4)}}
    bsteps = max(8, steps // 2)
    for B in batches:
        btb = torch.arange(n_blocks * B, dtype=torch.int32, device=dev).reshape(B, n_blocks).contiguous()
        g7 = torch.Generator(device=dev).manual_seed(11)
        idb = torch.empty(B, dtype=torch.int32, device=dev)
        with torch.cuda.stream(side):
            for b in range(B):
                pids = torch.randint(0, spec.vocab_size, (prompt,), dtype=torch.int32, device=dev, generator=g7)
                model.prefill(pids, 0, btb[b], out_id=idb[b:b + 1])
            posb = torch.full((B,), prompt, dtype=torch.int32, device=dev)
            slotb = (btb[:, prompt // 16] * 16 + prompt % 16).to(torch.int32).contiguous()
            slb = torch.full((B,), prompt + 1, dtype=torch.int32, device=dev)
            outb = torch.empty(B, dtype=torch.int32, device=dev)

            def stepb():
                model.forward_decode_batch(idb, posb, slotb, btb, slb, max_seq_len=spec.max_seq_len, out_ids=outb)
                model.advance(idb, outb, posb, slotb, slb, btb)
            stepb()
            gb = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(gb, stream=side):
                stepb()
            for _ in range(warmup):
                gb.replay()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(bsteps):
                gb.replay()
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb
            del gb
        by_batch[str(B)] = {"tok_s": round(B * bsteps / tb, 1), "ms_per_step": round(tb / bsteps * 1e3, 4)}
    dom = max(("qkv", "o_proj", "gate_up", "down_proj"), key=lambda k: kinds[k]["us_per_step"])
    # the launches are the variants the step makes (vmx_gemma4_mmvq_only mirrors run_layers): norms in the prologues included
    mmvq_kernel = {"qkv": "iq_pk_mmvq_kernel<NORM> (q|k|v, three packed I-quant tensors in one launch, layer-boundary norms + q8_1 in the prologue; "
                          "the first layer: iq_mmvq_multi_kernel)", "o_proj": "iq_mmvq_kernel (o_proj)",
                   "gate_up": "iq_pk_mmvq_kernel<GLU, NORM> (gate|up, packed 8-row units, codebook + sign tables in LDS, post-attention / "
                              "pre-feedforward norms + q8_1 in the prologue, GeGLU + q8_1 epilogue)",
                   "down_proj": "iq_mmvq_pipe_kernel (down_proj, rows split over 4 waves)"}
    traffic, traffic_source = None, None
    for name in ("r03_pmc_traffic_iq_mmvq.json", "r02_pmc_traffic_iq_mmvq.json"):
        try:   # HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/pmc_mmvq.py), when one exists for it
            traffic = json.loads((ROOT / "profiles" / name).read_text())[dom]["hbm_bytes_per_launch"]
            traffic_source = f"profiles/{name} (committed rocprofv3 --pmc pass; not measured in this run)"
            break
        except Exception:
            pass
    wbytes = model.iq_weight_bytes()
    embed_bytes = spec.vocab_size * (spec.hidden_size // 256) * 110
    model.close()
    from vllm_rust_amd.gemma4 import GEMMA4_12B_IQ3_MIX
    cpu = None
    try:
        if cpu_leg:
            cpu = cpu_baseline_mmvq(spec, GEMMA4_12B_IQ3_MIX)
    except Exception as e:   # the CPU leg must not cost the GPU numbers
        cpu = {"error": repr(e)}
    return {"metric": "decode tok/s (greedy, batch=1), Gemma-4-12B GGUF UD-IQ3_XXS mix + FP8 KV", "cpu_baseline": cpu, "value": round(steps / dt, 2), "unit": "tok/s",
            "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
            "prefill": {"tokens": prompt, "ms": round(t_pre * 1e3, 3), "tok_s": round(prompt / t_pre, 1),
                        "note": "one vmx_gemma4_prefill call: each I-quant weight dequantised once into bf16 hi/lo halves, two matrix-core passes per "
                                "linear (the reference's large-M IqLinear form), MFMA prefill attention; untimed part of the job"},
            "config": {"workload": "Gemma-4-12B-shaped GGUF (187 IQ3_XXS / IQ2_S / IQ3_S projections per the UD-IQ3_XXS mix, Q3_K tied "
                                   "token embedding), 48 layers (40 sliding 8x256 window 1024, 8 full 1x512), FP8 e4m3 KV, soft-cap 30",
                       "weights": "seeded random-init blocks of the architecture", "execution": "one hipGraph per step"},
            "weight_bytes_per_token": wbytes + embed_bytes,
            "batch_sweep": by_batch,
            "roofline": {"bound": "hbm", "kernel": mmvq_kernel[dom], "achieved": kinds[dom]["GB_s"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": kinds[dom]["frac"], "traffic": traffic, "traffic_source": traffic_source,
                         "launches_per_step": kinds[dom]["launches_per_step"], "avg_launch_us": kinds[dom]["avg_launch_us"],
                         "alg_bytes_per_launch": kinds[dom]["alg_bytes_per_launch"], "mmvq": kinds}}


def exl3_secondary(reps=8):
    """BASELINE.json configs[3] at TP=1, kernel level: the seven EXL3 4.0 bpw projections of every Qwen3-8B layer (random trellis,
    +-1 sign vectors), M = 1, replayed from one hipGraph: Hadamard -> trellis-decode MFMA GEMM -> Hadamard per projection.
    Reports the weight-stream rate of the EXL3 linear path; the f16 model around it is not built yet."""
    import vllm_rust_amd as vmx
    from vllm_rust_amd._lib import check, lib
    dev = torch.device("cuda", torch.cuda.current_device())
    H, I, nq, nkv, L, bits = 4096, 12288, 4096, 1024, 36, 4
    shapes = [(H, nq), (H, nkv), (H, nkv), (nq, H), (H, I), (H, I), (I, H)]
    g = torch.Generator(device=dev).manual_seed(3)
    layers = []
    for _ in range(L):
        lay = []
        for K, N in shapes:
            tr = torch.randint(-32768, 32767, ((K // 16) * (N // 16) * 16 * bits,), dtype=torch.int16, device=dev, generator=g)
            suh = (torch.randint(0, 2, (K,), device=dev, generator=g) * 2 - 1).to(torch.float16)
            svh = (torch.randint(0, 2, (N,), device=dev, generator=g) * 2 - 1).to(torch.float16)
            lay.append((K, N, tr, suh, svh))
        layers.append(lay)
    x = {K: (torch.randn((1, K), device=dev, generator=g) * 0.05).to(torch.float16) for K in (H, I)}
    y = {N: torch.empty((1, N), dtype=torch.float16, device=dev) for N in (H, nkv, I)}
    ws = torch.empty(max(lib.vmx_exl3_gemm_workspace_bytes(1, K, N) for K, N in shapes), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(device=dev)

    def chain():
        st = side.cuda_stream
        for lay in layers:
            for K, N, tr, suh, svh in lay:
                check(lib.vmx_exl3_gemm_f16(y[N].data_ptr(), x[K].data_ptr(), tr.data_ptr(), suh.data_ptr(), svh.data_ptr(), 1, K, N, bits, 0,
                                            ws.data_ptr(), ws.numel(), st))
    with torch.cuda.stream(side):
        chain()
        gph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(gph, stream=side):
            chain()
        gph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            gph.replay()
        e1.record(side)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = L * sum(K * N * bits / 8 + 2 * K + 2 * N + 2 * K + 2 * N for K, N in shapes)
    return {"metric": "EXL3 4.0 bpw linear path, Qwen3-8B shapes, M=1, TP=1 (252 projections per token)", "ms_per_token_linears": round(ms, 4),
            "tok_s_bound_from_linears": round(1e3 / ms, 1), "alg_bytes_per_token": int(alg), "GBps": round(alg / ms / 1e6, 1),
            "frac_of_8TBs": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4)}


def exl3_decode_secondary(prompt, steps, warmup):
    """BASELINE.json configs[3] at TP=1: decode tok/s of a Qwen3-8B-shaped model whose 252 projections are EXL3 4.0 bpw
    trellis-coded linears, run in fp16 end to end as the reference runs an EXL3 checkpoint (fp16 embedding, norms, lm_head,
    activations and KV cache; weight_loader.rs:453-461)."""
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
    total = prompt + warmup + steps + 8
    n_blocks = (total + 15) // 16 + 1
    spec = Qwen3Spec.qwen3_8b(max_position=max(4096, total + 1), num_blocks=n_blocks, max_batch=1, max_seq_len=total + 1,
                              max_prefill_tokens=prompt, act_dtype="f16")
    model = QuantizedQwen3ForCausalLM(spec)
    model.load_synthetic_exl3(bits=4, codebook=0, seed=9)
    dev = model.device
    bt = torch.arange(n_blocks, dtype=torch.int32, device=dev).flip(0).reshape(1, n_blocks).contiguous()
    prompt_ids = torch.randint(0, spec.vocab_size, (prompt,), dtype=torch.int32, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    ids = torch.empty(1, dtype=torch.int32, device=dev)
    pos = torch.full((1,), prompt, dtype=torch.int32, device=dev)
    slots = (bt[:, prompt // 16] * 16 + prompt % 16).to(torch.int32).contiguous()
    sl = torch.full((1,), prompt + 1, dtype=torch.int32, device=dev)
    out = torch.empty(1, dtype=torch.int32, device=dev)

    def step():
        model.forward_decode_batch(ids, pos, slots, bt, sl, max_seq_len=spec.max_seq_len, out_ids=out)
        model.advance(ids, out, pos, slots, sl, bt)

    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        model.prefill(prompt_ids, 0, bt[0], out_id=ids)
        step()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            step()
        for _ in range(warmup):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    # the step's EXL3 linear launches alone, per kind (q|k|v and gate|up are one multi-tensor GEMM each, input Hadamard inside
    # the GEMM workgroups, output Hadamard in the consumer except for q|k|v)
    kinds = time_linear_kinds(model, side, 1, max(8, steps // 4))
    H, I = spec.hidden_size, spec.intermediate_size
    nq, nkv = spec.num_heads * spec.head_dim, spec.num_kv_heads * spec.head_dim
    ex = lambda K, N: K * N / 2 + 2 * K + 2 * N
    per_tok = spec.num_layers * (ex(H, nq) + 2 * ex(H, nkv) + ex(nq, H) + 2 * ex(H, I) + ex(I, H)) + spec.vocab_size * H * 2
    model.close()
    return {"metric": "decode tok/s (greedy, batch=1), Qwen3-8B EXL3 4.0 bpw, TP=1", "value": round(steps / dt, 2), "unit": "tok/s",
            "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
            "config": {"workload": "Qwen3-8B-shaped: 36 layers, 252 EXL3 4.0 bpw projections (default codebook), fp16 activations, lm_head and KV",
                       "weights": "seeded random trellis words and sign/scale vectors", "execution": "one hipGraph per step"},
            "weight_bytes_per_token": int(per_tok), "GBps_weights_only": round(per_tok / (dt / steps) / 1e9, 1),
            "roofline": {"bound": "valu (trellis decode: 7 vector-ALU slots per weight); shown against HBM", "kernel": "exl3_gemm_kernel<4 bpw>",
                         "achieved": kinds["all"]["GB_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kinds["all"]["frac"], "traffic": None,
                         "linears": kinds}}


def batch_sweep_secondary(prompt, steps, warmup, batches=(1, 4, 8, 16)):
    """Concurrent decode streams on one GPU (the reference's c = 1/4/8/16 table, docs/perf/bench-history/
    2026-05-09-python-vllm-comparison.md): Qwen3-8B AWQ, B sequences of `prompt` tokens each, one hipGraph per step.
    The small-M GEMM streams the weights once for up to 16 rows, but a step still grows with B (activation staging, the
    16-row lm_head); every batch size of the sweep is reported as measured, whether or not it scales."""
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
    total = prompt + warmup + steps + 8
    per_seq = (total + 15) // 16 + 1
    BM = max(batches)
    spec = Qwen3Spec.qwen3_8b(max_position=max(4096, total + 1), num_blocks=per_seq * BM, max_batch=BM, max_seq_len=total + 1,
                              max_prefill_tokens=prompt)
    model = QuantizedQwen3ForCausalLM(spec)
    model.load_synthetic(seed=1234)
    dev = model.device
    out = {}
    for B in batches:
        bt = torch.arange(per_seq * B, dtype=torch.int32, device=dev).reshape(B, per_seq).contiguous()
        g7 = torch.Generator(device=dev).manual_seed(7)
        ids = torch.empty(B, dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            for b in range(B):   # every sequence gets its own prompt through the native prefill
                pids = torch.randint(0, spec.vocab_size, (prompt,), dtype=torch.int32, device=dev, generator=g7)
                model.prefill(pids, 0, bt[b], out_id=ids[b:b + 1])
            pos = torch.full((B,), prompt, dtype=torch.int32, device=dev)
            slots = (bt[:, prompt // 16] * 16 + prompt % 16).to(torch.int32).contiguous()
            sl = torch.full((B,), prompt + 1, dtype=torch.int32, device=dev)
            o = torch.empty(B, dtype=torch.int32, device=dev)

            def step():
                model.forward_decode_batch(ids, pos, slots, bt, sl, max_seq_len=spec.max_seq_len, out_ids=o)
                model.advance(ids, o, pos, slots, sl, bt)
            step()
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                step()
            for _ in range(warmup):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[str(B)] = {"tok_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4)}
    model.close()
    return {"metric": "aggregate decode tok/s at B concurrent sequences, Qwen3-8B AWQ, TP=1", "prompt": prompt, "steps": steps, "by_batch": out}


def context_sweep_secondary(steps, warmup, contexts=(1024, 8192)):
    """Decode tok/s of the headline model (Qwen3-8B AWQ, batch 1) at longer contexts: the prompt goes through the native
    prompt pass in chunks of 2048 tokens, then `steps` timed tokens from one hipGraph.  Beyond 512 tokens the attention of a
    layer is the split kernel with one workgroup of 8 waves per compute unit and the separate merge launch."""
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
    chunk = 2048
    top = max(contexts) + warmup + steps + 8
    n_blocks = (top + 15) // 16 + 1
    spec = Qwen3Spec.qwen3_8b(max_position=top + 1, num_blocks=n_blocks, max_batch=1, max_seq_len=top + 1, max_prefill_tokens=chunk)
    model = QuantizedQwen3ForCausalLM(spec)
    model.load_synthetic(seed=4321)
    dev = model.device
    out = {}
    bt = torch.arange(n_blocks, dtype=torch.int32, device=dev).reshape(1, n_blocks).contiguous()
    side = torch.cuda.Stream(device=dev)
    for ctx in contexts:
        total = ctx + warmup + steps + 8
        g7 = torch.Generator(device=dev).manual_seed(ctx)
        ids = torch.empty(1, dtype=torch.int32, device=dev)
        with torch.cuda.stream(side):
            t_pre = time.perf_counter()
            for c0 in range(0, ctx, chunk):
                n = min(chunk, ctx - c0)
                pids = torch.randint(0, spec.vocab_size, (n,), dtype=torch.int32, device=dev, generator=g7)
                model.prefill(pids, c0, bt[0], out_id=ids)
            torch.cuda.synchronize()
            t_pre = time.perf_counter() - t_pre
            pos = torch.full((1,), ctx, dtype=torch.int32, device=dev)
            slots = (bt[:, ctx // 16] * 16 + ctx % 16).to(torch.int32).contiguous()
            sl = torch.full((1,), ctx + 1, dtype=torch.int32, device=dev)
            o = torch.empty(1, dtype=torch.int32, device=dev)

            def step():
                model.forward_decode_batch(ids, pos, slots, bt, sl, max_seq_len=total + 1, out_ids=o)
                model.advance(ids, o, pos, slots, sl, bt)
            step()
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                step()
            for _ in range(warmup):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            del g
        kv_bytes = 2 * spec.num_layers * ctx * spec.num_kv_heads * spec.head_dim * 2
        out[str(ctx)] = {"tok_s": round(steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "kv_bytes_read_per_token": kv_bytes,
                         "prefill_tok_s": round(ctx / t_pre, 1)}
    model.close()
    return {"metric": "decode tok/s at longer contexts, Qwen3-8B AWQ, batch 1, bf16 KV, TP=1", "steps": steps, "by_context": out}


def mixtral_secondary(prompt, steps, warmup, batches=(1, 4, 16)):
    """BASELINE.json configs[4] on one GPU: decode tok/s of a Mixtral-8x7B-shaped model (AWQ attention projections, 8 GPTQ
    int4 experts top-2 per layer behind a dense router) through the executor's device-side fused MoE path; the prompt pass
    (token-grouped expert GEMM on the matrix cores) timed on its own; aggregate decode at several concurrent sequences (from
    12 (token, slot) pairs on an expert's weights are streamed once per block of its rows, not once per pair)."""
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM, Qwen3Spec
    total = prompt + warmup + steps + 8
    per_seq = (total + 15) // 16 + 1
    BM = max(batches)
    spec = Qwen3Spec.mixtral_8x7b(max_position=max(4096, total + 1), num_blocks=per_seq * BM, max_batch=BM, max_seq_len=total + 1,
                                  max_prefill_tokens=prompt)
    model = QuantizedQwen3ForCausalLM(spec)
    model.load_synthetic_moe(seed=5)
    dev = model.device
    by_batch = {}
    prefill_s = None
    for B in batches:
        bt = torch.arange(per_seq * B, dtype=torch.int32, device=dev).flip(0).reshape(B, per_seq).contiguous()
        g7 = torch.Generator(device=dev).manual_seed(7)
        ids = torch.empty(B, dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            for b in range(B):
                pids = torch.randint(0, spec.vocab_size, (prompt,), dtype=torch.int32, device=dev, generator=g7)
                model.prefill(pids, 0, bt[b], out_id=ids[b:b + 1])
                if prefill_s is None:   # the first call paid the lazy set-up: time a second pass over the same prompt
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    model.prefill(pids, 0, bt[b], out_id=ids[b:b + 1])
                    torch.cuda.synchronize()
                    prefill_s = time.perf_counter() - t0
            pos = torch.full((B,), prompt, dtype=torch.int32, device=dev)
            slots = (bt[:, prompt // 16] * 16 + prompt % 16).to(torch.int32).contiguous()
            sl = torch.full((B,), prompt + 1, dtype=torch.int32, device=dev)
            out = torch.empty(B, dtype=torch.int32, device=dev)

            def step():
                model.forward_decode_batch(ids, pos, slots, bt, sl, max_seq_len=spec.max_seq_len, out_ids=out)
                model.advance(ids, out, pos, slots, sl, bt)
            step()
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                step()
            for _ in range(warmup):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            del g
        by_batch[str(B)] = {"tok_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4)}
    H, I, E, k = spec.hidden_size, spec.intermediate_size, spec.num_experts, spec.experts_per_tok
    nq, nkv = spec.num_heads * spec.head_dim, spec.num_kv_heads * spec.head_dim
    int4 = lambda K, N: K * N / 2 + (K / 128) * N * 2.5
    per_tok = spec.num_layers * (int4(H, nq + 2 * nkv) + int4(nq, H) + k * (int4(H, 2 * I) + int4(I, H)) + E * H * 2) + spec.vocab_size * H * 2
    model.close()
    b1 = by_batch[str(batches[0])]
    flop = 2.0 * prompt * spec.num_layers * (H * (nq + 2 * nkv) + nq * H + k * 3 * H * I)
    return {"metric": "decode tok/s (greedy, batch=1), Mixtral-8x7B GPTQ int4 fused MoE, TP=1", "value": b1["tok_s"] / batches[0], "unit": "tok/s",
            "steps": steps, "ms_per_step": b1["ms_per_step"],
            "config": {"workload": "Mixtral-8x7B-shaped: 32 layers, hidden 4096, 8 experts x 14336 top-2 (GPTQ int4 g128), AWQ attention, bf16 KV",
                       "weights": "seeded random-init checkpoint of the architecture", "execution": "one hipGraph per step; routing stays on the device"},
            "weight_bytes_per_token": int(per_tok), "GBps_weights_only": round(per_tok / (b1["ms_per_step"] / 1e3) / 1e9, 1),
            "prefill": {"tokens": prompt, "ms": round(prefill_s * 1e3, 3), "tok_s": round(prompt / prefill_s, 1),
                        "linear_TFLOP_s": round(flop / prefill_s / 1e12, 1),
                        "note": "one vmx_qwen3_prefill call: (token, slot) pairs sorted by expert on the device, 64-row blocks through the 64 x 128-tile "
                                "int4 matrix-core GEMM (an expert's weights are read once per block of its rows)"},
            "batch_sweep": by_batch}


def plan_job(gpus, world, replicas_flag, weights, local_ranks):
    """What a launch of this script measures -- pure argument logic, no device.  N = 1: the single-GPU job.  N > 1 (one process
    per GPU under torch.distributed.run): ONE model tensor-parallel over the N ranks (BASELINE.json's metric is the TP curve at
    1/2/4/8 GPUs: strong scaling, all-reduce after every row-parallel linear), then N independent replicas as a secondary;
    --replicas makes the replicas the primary (weak scaling, no data-path collective) and skips the TP job."""
    from vllm_rust_amd.parallel import shard_spec
    from vllm_rust_amd.qwen3 import Qwen3Spec
    if world != max(gpus, 1):
        raise ValueError(f"--gpus {gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node {gpus}")
    if world == 1:
        return {"mode": "single", "scaling": "weak", "tp": 1, "secondary": None, "shards": None}
    if replicas_flag:
        return {"mode": "replicas", "scaling": "weak", "tp": 1, "secondary": None, "shards": None}
    full = Qwen3Spec.qwen3_8b()
    shards = [shard_spec(full, r, world) for r in range(world)]   # raises ValueError when the geometry does not divide
    return {"mode": "tp", "scaling": "strong", "tp": world, "secondary": "replicas",
            "shards": [(s.num_heads, s.num_kv_heads, s.intermediate_size, s.tp_rank) for s in shards],
            "transport": "peer-mapped (one GPU, development)" if local_ranks else "rccl+peer-mapped", "weights": weights}


class _Dist:
    """torch.distributed plumbing of the benchmark: one process per GPU over RCCL; VMX_BENCH_LOCAL_RANKS=1 (development, one-GPU
    boxes) puts every rank on GPU 0 and joins them over gloo + peer-mapped memory instead, to rehearse the multi-rank flow."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.on = self.world > 1
        self.local = os.environ.get("VMX_BENCH_LOCAL_RANKS", "0") not in ("", "0")
        self.device = 0 if self.local else self.local_rank
        if self.on:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.local:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            self.dist = dist
        torch.cuda.set_device(self.device)

    def barrier(self):
        if self.on:
            self.dist.barrier()

    def max_over_ranks(self, v):
        if not self.on:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if self.local else torch.device("cuda", self.device))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_agree(self, ok):
        if not self.on:
            return bool(ok)
        flags = [None] * self.world
        self.dist.all_gather_object(flags, bool(ok))
        return all(flags)

    def make_comm(self):
        """(communicator, description of the all-reduce path, note).  The transport is agreed between the ranks."""
        from vllm_rust_amd.parallel import Communicator, LocalCommunicator
        if self.local:
            def exchange(mine):
                got = [None] * self.world
                self.dist.all_gather_object(got, mine)
                return got
            # no RCCL between processes that share a GPU: prompts of at most 32 rows so that every all-reduce fits the staging slot
            return LocalCommunicator(self.rank, self.world, exchange, max_bytes=256 * 1024), "one-shot over peer-mapped memory (ranks share one GPU)", ""
        comm = Communicator(self.rank, self.world)
        if os.environ.get("VMX_NO_P2P_ALLREDUCE", "0") not in ("", "0"):
            return comm, "RCCL", "peer-mapped path switched off (VMX_NO_P2P_ALLREDUCE)"
        if comm.enable_p2p(64 * 1024):   # the 8 KiB row-parallel sums go over peer-mapped memory in one step; RCCL keeps the rest
            return comm, "one-shot over peer-mapped memory (xGMI) for the row-parallel sums and argmax keys, RCCL for larger messages", ""
        return comm, "RCCL", f"peer-mapped path unavailable on some rank: {comm.p2p_note}"

    def finish(self):
        if self.on:
            self.dist.barrier()
            self.dist.destroy_process_group()


def run_decode_job(d, args, spec, full_spec, comm, exl3, seed_per_rank):
    """Load the model of this rank, run the prompt through the native prefill, capture one decode step (+ device-side
    bookkeeping) into a hipGraph, replay `warmup` steps, then time exactly `steps` replays between barrier + synchronise on
    both sides; the time is the maximum over the ranks.  Returns (seconds, prefill seconds, model, graph, side stream)."""
    from vllm_rust_amd.qwen3 import QuantizedQwen3ForCausalLM
    model = QuantizedQwen3ForCausalLM(spec, device=d.device)
    if exl3:
        model.load_synthetic_exl3(bits=4, codebook=0, seed=1234 + (d.rank if seed_per_rank or comm is not None else 0))
    elif comm is not None:   # every rank slices the SAME seeded checkpoint: the TP job computes the model of the single-GPU run
        model.load_synthetic_tp(full_spec, seed=1234)
    else:
        model.load_synthetic(seed=1234 + (d.rank if seed_per_rank else 0))
    if comm is not None:
        model.set_comm(comm)
    dev = model.device
    B, block_size, P = 1, spec.block_size, args.prompt
    n_blocks = spec.num_blocks
    bt = torch.arange(n_blocks, dtype=torch.int32, device=dev).flip(0).reshape(B, n_blocks).contiguous()
    prompt_ids = torch.randint(0, spec.vocab_size, (P,), dtype=torch.int32, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    ids = torch.empty(B, dtype=torch.int32, device=dev)
    pos = torch.full((B,), P, dtype=torch.int32, device=dev)
    slots = (bt[:, P // block_size] * block_size + P % block_size).to(torch.int32).contiguous()
    seq_lens = torch.full((B,), P + 1, dtype=torch.int32, device=dev)
    out = torch.empty(B, dtype=torch.int32, device=dev)

    def step():
        model.forward_decode_batch(ids, pos, slots, bt, seq_lens, max_seq_len=spec.max_seq_len, out_ids=out)
        model.advance(ids, out, pos, slots, seq_lens, bt)

    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        # prompt: one native prefill pass (all rows through the linears, MFMA prefill attention); run twice, the second timed
        model.prefill(prompt_ids, 0, bt[0], out_id=ids)
        torch.cuda.synchronize()
        d.barrier()
        tp0 = time.perf_counter()
        model.prefill(prompt_ids, 0, bt[0], out_id=ids)
        torch.cuda.synchronize()
        prefill_s = time.perf_counter() - tp0
        step()  # first decode call outside capture (lazy module load); position P -> P + 1
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            step()
        for _ in range(args.warmup):
            graph.replay()
    torch.cuda.synchronize()
    d.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        for _ in range(args.steps):
            graph.replay()
    torch.cuda.synchronize()
    d.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pos_after = int(pos.item())
    assert pos_after == P + 1 + args.warmup + args.steps, (pos_after, P, args.warmup, args.steps)
    return d.max_over_ranks(elapsed), prefill_s, model, graph, side


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--prompt", type=int, default=256)
    ap.add_argument("--layers", type=int, default=0, help="debug only: override the layer count (invalidates the number)")
    ap.add_argument("--weights", choices=["awq", "exl3"], default="awq",
                    help="exl3: the same Qwen3-8B geometry with EXL3 4.0 bpw projections (BASELINE.json configs[3]: its TP curve for N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (other models at N = 1, replicas at N > 1)")
    ap.add_argument("--no-moe", action="store_true", help="skip the Mixtral-8x7B-shaped fused-MoE secondary measurement")
    ap.add_argument("--workload", choices=["all", "qwen3", "gemma4", "mixtral", "exl3"], default="all",
                    help="profiling aid: run ONE workload alone (qwen3 = the primary without secondaries; the others print only "
                         "their own block), so that a kernel trace of the run holds that workload's launches only")
    ap.add_argument("--tp", action="store_true", help="accepted for compatibility: with --gpus N > 1 the tensor-parallel job is the default")
    ap.add_argument("--replicas", action="store_true",
                    help="with --gpus N > 1: N independent replicas as the primary (weak scaling, no collective) instead of ONE model "
                         "tensor-parallel over the N ranks")
    args = ap.parse_args()

    d = _Dist()
    rank, world = d.rank, d.world
    import vllm_rust_amd as vmx
    from vllm_rust_amd.qwen3 import Qwen3Spec
    vmx.init(d.device)
    plan = plan_job(args.gpus, world, args.replicas, args.weights, d.local)

    K_steps, W_steps = args.steps, args.warmup
    if args.workload in ("gemma4", "mixtral", "exl3"):   # one secondary alone (single GPU)
        if args.workload == "gemma4":
            blk = gemma4_secondary(args.prompt, max(16, K_steps // 2), W_steps, cpu_leg=not args.no_cpu_baseline)
        elif args.workload == "mixtral":
            blk = mixtral_secondary(args.prompt, max(16, K_steps // 2), W_steps)
        else:
            blk = exl3_decode_secondary(args.prompt, min(args.steps, 64), args.warmup)
            blk["linears_only"] = exl3_secondary()
        print(json.dumps(blk), flush=True)
        return
    if args.workload == "qwen3":
        args.no_secondary = True
    total_tokens = args.prompt + W_steps + 2 * K_steps + 8
    block_size = 16
    n_blocks = (total_tokens + block_size - 1) // block_size + 1
    full_spec = Qwen3Spec.qwen3_8b(max_position=max(4096, total_tokens + 1), num_blocks=n_blocks, block_size=block_size, max_batch=1,
                                   max_seq_len=total_tokens + 1, max_prefill_tokens=args.prompt)
    if args.layers:
        full_spec.num_layers = args.layers
    exl3 = args.weights == "exl3"
    if exl3:   # an EXL3 checkpoint runs in fp16 end to end (weight_loader.rs:453-461)
        full_spec.act_dtype = "f16"
    B = 1

    # ---- the primary job
    comm, allreduce_path, comm_note = None, None, ""
    spec = full_spec
    mode = plan["mode"]
    if mode == "tp":
        from vllm_rust_amd.parallel import shard_spec
        ok = True
        try:
            comm, allreduce_path, comm_note = d.make_comm()
        except Exception as e:   # noqa: BLE001 - a rank that cannot join the group must take the others with it, not hang them
            comm_note = f"communicator set-up failed on rank {rank}: {e!r}"
            ok = False
        if not d.all_agree(ok):   # no tensor-parallel group: every rank falls back to replicas and the line says so
            if comm is not None:
                comm.close()
            comm, mode = None, "replicas"
            plan = dict(plan, mode="replicas", scaling="weak", tp=1, secondary=None)
            comm_note = comm_note or "communicator set-up failed on a peer rank"
        else:
            spec = shard_spec(full_spec, rank, world)   # this rank's heads / MLP columns / lm_head rows
    elapsed, prefill_s, model, graph, side = run_decode_job(d, args, spec, full_spec, comm, exl3, seed_per_rank=mode == "replicas")
    comm_state = None
    if comm is not None:
        comm_state = {"p2p_timeouts": comm.p2p_timeouts(), "error_code": comm.error_code()}

    # ---- N > 1, tensor-parallel primary: the N-replica job as a secondary (full model per rank, no collective)
    replicas = None
    if mode == "tp" and not args.no_secondary:
        model.close()
        del model, graph
        torch.cuda.empty_cache()
        r_elapsed, _, model, graph, side = run_decode_job(d, args, full_spec, full_spec, None, exl3, seed_per_rank=True)
        replicas = {"metric": "aggregate decode tok/s of N independent replicas (batch=1 each, no data-path collective)",
                    "value": round(world * B * K_steps / r_elapsed, 2), "unit": "tok/s", "scaling": "weak", "ms_per_step": round(r_elapsed / K_steps * 1e3, 4)}
        roofline_model = model          # full-size kernels: the same launches as the N = 1 run
    else:
        roofline_model = model if mode != "tp" else None

    # ---- roofline of the dominant kernel: the step's int4 linear launches alone (same weights, same buffers), one kind at
    # a time, captured into a graph and replayed back to back between two HIP events recorded on the stream the kernels run
    # on; duration per launch = elapsed / launches (boundaries included, as in rocprofv3's back-to-back kernel timestamps).
    # The four linears of a layer are four different kernels (norm prologue / residual epilogue variants); `roofline` is the
    # one the step spends most time in (gate|up), `linears` lists all four and their launch-weighted total.
    roofline = None
    if rank == 0 and not exl3 and roofline_model is not None:
        kinds = time_linear_kinds(roofline_model, side, B, max(8, K_steps // 4))
        dom = max(("qkv", "o_proj", "gate_up", "down_proj"), key=lambda k: kinds[k]["us_per_step"])
        # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE; rocprofv3 cannot run inside this process):
        # committed summary of `rocprofv3 --pmc ... -- python3 tools/pmc_linears.py`, same kernel, same shapes.
        traffic, traffic_source = None, None
        for name in ("r03_pmc_traffic_int4_gemm.json", "r02_pmc_traffic_int4_gemm.json"):
            try:
                traffic = json.loads((ROOT / "profiles" / name).read_text())[dom]["hbm_bytes_per_launch"]
                traffic_source = f"profiles/{name} (committed rocprofv3 --pmc pass over the same kernel and shape; not measured in this run)"
                break
            except Exception:
                pass
        kernel_names = {"qkv": "int4_gemm_wide_kernel (q|k|v: RMSNorm prologue)", "o_proj": "int4_gemm_resid_kernel (o_proj: residual epilogue)",
                        "gate_up": "int4_gemm_wide_kernel<GLU> (gate|up: RMSNorm prologue, SiLU*mul epilogue)",
                        "down_proj": "int4_gemm_resid_kernel (down_proj: residual epilogue)"}
        k = kinds[dom]
        roofline = {"bound": "hbm", "kernel": kernel_names[dom], "achieved": k["GB_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["frac"],
                    "traffic": traffic, "traffic_source": traffic_source, "launches_per_step": k["launches_per_step"],
                    "avg_launch_us": k["avg_launch_us"], "alg_bytes_per_launch": k["alg_bytes_per_launch"], "linears": kinds}

    # ---- secondaries (rank 0, N = 1 only): BASELINE.json's other models
    secondary = None
    secondary_moe = None
    secondary_exl3 = None
    batch_sweep = None
    context_sweep = None
    if rank == 0 and not d.on and not args.no_secondary and not args.layers and not exl3:
        model.close()
        del model, graph
        torch.cuda.empty_cache()
        secondary = gemma4_secondary(args.prompt, max(16, K_steps // 2), W_steps)
        torch.cuda.empty_cache()
        if not args.no_moe:
            secondary_moe = mixtral_secondary(args.prompt, max(16, K_steps // 2), W_steps)
        torch.cuda.empty_cache()
        secondary_exl3 = exl3_decode_secondary(args.prompt, min(args.steps, 64), args.warmup)
        secondary_exl3["linears_only"] = exl3_secondary()
        if not args.no_cpu_baseline:
            try:
                secondary_exl3["cpu_baseline"] = cpu_baseline_exl3()
            except Exception as e:
                secondary_exl3["cpu_baseline"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        batch_sweep = batch_sweep_secondary(args.prompt, 32, 4)
        torch.cuda.empty_cache()
        context_sweep = context_sweep_secondary(32, 4)

    if rank == 0:
        tok_s = (1 if mode == "tp" else world) * B * K_steps / elapsed
        geometry = f"{full_spec.num_layers} layers, hidden {full_spec.hidden_size}, vocabulary {full_spec.vocab_size}"
        if mode == "tp":
            parallelism = f"tp{world}: ONE model over {world} ranks, column/row-parallel linears, all-reduce: {allreduce_path}"
        elif mode == "replicas":
            parallelism = f"dp{world}: {world} independent replicas (tp1 each), no data-path collective"
        else:
            parallelism = "tp1"
        line = {
            "metric": "decode tok/s (greedy, batch=1 per GPU), Qwen3-8B EXL3 4.0 bpw" if exl3 else
                      "decode tok/s (greedy, batch=1) + %HBM-peak of the int4 linear kernel, Qwen3-8B AWQ",
            "value": round(tok_s, 2), "unit": "tok/s", "n_gpus": world, "steps": K_steps, "warmup": W_steps,
            "ms_per_step": round(elapsed / K_steps * 1e3, 4), "higher_is_better": True, "scaling": plan["scaling"], "vs_baseline": None,
            "dtype": "4-bit trellis weights x fp16 activations, f32 accumulate" if exl3 else "int4 weights x bf16 activations, f32 accumulate",
            "data": "synthetic",
            "config": {"workload": (f"Qwen3-8B EXL3 4.0 bpw ({geometry}) greedy decode, batch=1, {args.prompt}-token prompt then timed tokens, fp16 activations and KV" if exl3 else
                                    f"Qwen3-8B AWQ int4 g128 ({geometry}) greedy decode, batch=1, {args.prompt}-token prompt (native prefill) then timed tokens, bf16 KV"),
                       "parallelism": parallelism,
                       "weights": "seeded random-init checkpoint of the Qwen3-8B architecture" + (" (every rank slices the same one)" if mode == "tp" else ""),
                       "execution": "one hipGraph per step (decode + device-side bookkeeping)"},
            "prefill": prefill_block(args.prompt, prefill_s, full_spec if mode != "tp" else spec),
            "roofline": roofline,
            "cpu_baseline": None if (args.no_cpu_baseline or exl3 or d.on) else cpu_baseline(full_spec),
            "secondary": secondary,
            "secondary_moe": secondary_moe,
            "secondary_exl3": secondary_exl3,
            "batch_sweep": batch_sweep,
            "context_sweep": context_sweep,
        }
        if d.on:
            line["comm"] = {"state": comm_state, "note": comm_note}
            line["replicas"] = replicas
        print(json.dumps(line), flush=True)
    if d.on:
        d.barrier()
    if comm is not None:
        comm.close()
    d.finish()


if __name__ == "__main__":
    main()

"""World-size-2 gloo test (CPU) of the tensor-parallel sharding rules: column shards concatenate to the
full linear, row shards sum (all_reduce) to it.  The arithmetic is the CPU oracle's."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _awq_case(rng, K, N, g):
    qw = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(K, N // 8), dtype=np.int64).astype(np.int32)
    qz = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(K // g, N // 8), dtype=np.int64).astype(np.int32)
    sc = ((rng.random((K // g, N)) * 0.5 + 0.75) / (6.5 * np.sqrt(K))).astype(np.float16)
    return qw, qz, sc


def _oracle_linear(orc, x_u16, qw, qz, sc, g):
    K, N = qw.shape[0], qw.shape[1] * 8
    gq = orc.awq_to_gptq_qweight(qw.view(np.uint32), K, N)
    gz = orc.repack_awq_nibbles(qz.view(np.uint32))
    sb = orc.f32_to_bf16(sc.astype(np.float32))
    return orc.int4_gemv_bf16(x_u16, gq, sb, gz, None, K, N, g)


def _worker(rank, world, port, ret):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from vllm_rust_amd.parallel import shard_awq_column, shard_awq_row, shard_spec
    from vllm_rust_amd.qwen3 import Qwen3Spec
    rng = np.random.default_rng(0)                      # same data on every rank
    K, N, g = 512, 256, 128
    qw, qz, sc = _awq_case(rng, K, N, g)
    x = orc.f32_to_bf16(rng.standard_normal((2, K)).astype(np.float32))
    full = orc.bf16_to_f32(_oracle_linear(orc, x, qw, qz, sc, g))
    t = lambda a: torch.from_numpy(a)
    # column parallel: all_gather of the shards' outputs == full output, bit for bit
    cqw, cqz, csc = (a.numpy() for a in shard_awq_column(t(qw), t(qz), t(sc), rank, world))
    part = torch.from_numpy(orc.bf16_to_f32(_oracle_linear(orc, x, cqw, cqz, csc, g)))
    parts = [torch.empty_like(part) for _ in range(world)]
    dist.all_gather(parts, part)
    ok_col = np.array_equal(torch.cat(parts, 1).numpy(), full)
    # row parallel: all_reduce(sum) of the bf16-rounded shard outputs == full output up to bf16 rounding of the partials
    rqw, rqz, rsc = (a.numpy() for a in shard_awq_row(t(qw), t(qz), t(sc), rank, world, g))
    xs = np.ascontiguousarray(x[:, rank * (K // world):(rank + 1) * (K // world)])
    rpart = torch.from_numpy(orc.bf16_to_f32(_oracle_linear(orc, xs, rqw, rqz, rsc, g)))
    dist.all_reduce(rpart)
    err = float(np.abs(rpart.numpy() - full).max() / np.abs(full).max())
    # EXL3 linear: column shards (whole 128-column blocks) concatenate bit for bit; row shards sum to the full output up to
    # the fp16 rounding of each rank's output
    from vllm_rust_amd.parallel import shard_exl3_column, shard_exl3_row
    Ke, Ne, bits = 512, 256, 4
    tr = rng.integers(0, 65536, size=(Ke // 16, Ne // 16, 16 * bits)).astype(np.uint16)
    suh = orc.f32_to_f16(rng.choice([-1.0, 1.0], Ke).astype(np.float32))
    svh = orc.f32_to_f16((rng.choice([-1.0, 1.0], Ne) / np.sqrt(Ke)).astype(np.float32))
    xe = orc.f32_to_f16((rng.standard_normal((2, Ke)) * 0.5).astype(np.float32))
    efull = orc.f16_to_f32(orc.exl3_gemm(xe, tr.reshape(-1), suh, svh, Ke, Ne, bits, 0))
    ctr, csu, csv = (a.numpy() for a in shard_exl3_column(t(tr.view(np.int16)), t(suh.view(np.int16)), t(svh.view(np.int16)), rank, world))
    epart = torch.from_numpy(orc.f16_to_f32(orc.exl3_gemm(xe, ctr.view(np.uint16).reshape(-1), csu.view(np.uint16), csv.view(np.uint16),
                                                          Ke, Ne // world, bits, 0)))
    eparts = [torch.empty_like(epart) for _ in range(world)]
    dist.all_gather(eparts, epart)
    ok_ecol = np.array_equal(torch.cat(eparts, 1).numpy(), efull)
    rtr, rsu, rsv = (a.numpy() for a in shard_exl3_row(t(tr.view(np.int16)), t(suh.view(np.int16)), t(svh.view(np.int16)), rank, world))
    xes = np.ascontiguousarray(xe[:, rank * (Ke // world):(rank + 1) * (Ke // world)])
    erp = torch.from_numpy(orc.f16_to_f32(orc.exl3_gemm(xes, rtr.view(np.uint16).reshape(-1), rsu.view(np.uint16), rsv.view(np.uint16),
                                                        Ke // world, Ne, bits, 0)))
    dist.all_reduce(erp)
    eerr = float(np.abs(erp.numpy() - efull).max() / np.abs(efull).max())
    s = shard_spec(Qwen3Spec.qwen3_8b(), rank, world)
    ok_spec = (s.num_heads, s.num_kv_heads, s.intermediate_size, s.vocab_size, s.tp_rank) == (16, 4, 6144, 151936, rank)
    if rank == 0:
        ret["col"], ret["row_err"], ret["spec"] = ok_col, err, ok_spec
        ret["ecol"], ret["erow_err"] = ok_ecol, eerr
    dist.barrier()
    dist.destroy_process_group()


def test_tp_sharding_world2_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["col"] is True
    assert ret["row_err"] <= 2 ** -6                    # each rank rounds its partial to bf16 before the sum
    assert ret["spec"] is True
    assert ret["ecol"] is True
    assert ret["erow_err"] <= 2 ** -8                   # each rank rounds its output to fp16 before the sum


def test_shard_spec_rejects_bad_divisors():
    sys.path.insert(0, str(ROOT))
    from vllm_rust_amd.parallel import shard_spec
    from vllm_rust_amd.qwen3 import Qwen3Spec
    with pytest.raises(ValueError):
        shard_spec(Qwen3Spec.qwen3_8b(), 0, 3)
    with pytest.raises(ValueError):
        shard_spec(Qwen3Spec.qwen3_8b(), 0, 16)         # 8 KV heads


def test_one_seeded_checkpoint_shards_reassemble():
    """The slicing rules load_synthetic_tp applies to one seeded checkpoint (columns for q/k/v/gate/up, whole-group rows for
    o/down): the ranks' shards put back together are the full tensors, for 2 and 4 ranks."""
    import torch
    from vllm_rust_amd.parallel import shard_awq_column, shard_awq_row
    g = torch.Generator().manual_seed(0)
    K, N, group = 512, 256, 128
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, generator=g)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // group, N // 8), dtype=torch.int32, generator=g)
    sc = torch.rand((K // group, N), generator=g).to(torch.float16)
    for world in (2, 4):
        cols = [shard_awq_column(qw, qz, sc, r, world) for r in range(world)]
        assert torch.equal(torch.cat([c[0] for c in cols], 1), qw) and torch.equal(torch.cat([c[1] for c in cols], 1), qz)
        assert torch.equal(torch.cat([c[2] for c in cols], 1), sc)
        rows = [shard_awq_row(qw, qz, sc, r, world, group) for r in range(world)]
        assert torch.equal(torch.cat([c[0] for c in rows], 0), qw) and torch.equal(torch.cat([c[1] for c in rows], 0), qz)
        assert torch.equal(torch.cat([c[2] for c in rows], 0), sc)


def test_expert_parallel_spec_and_ownership():
    """Mixture-of-experts under tensor parallelism: attention dims divide, experts stay whole and full width, each rank owns a
    contiguous block of them (ExpertPlacement::Linear), together every expert exactly once."""
    sys.path.insert(0, str(ROOT))
    from vllm_rust_amd.parallel import local_experts, shard_spec
    from vllm_rust_amd.qwen3 import Qwen3Spec
    full = Qwen3Spec.mixtral_8x7b()
    for world in (2, 4, 8):
        owned = []
        for rank in range(world):
            s = shard_spec(full, rank, world)
            assert (s.num_heads, s.num_kv_heads, s.intermediate_size, s.num_experts) == (32 // world, 8 // world, 14336, 8)
            owned += list(local_experts(full.num_experts, rank, world))
        assert owned == list(range(8))
    assert list(local_experts(8, 1, 2)) == [4, 5, 6, 7]          # expert_map.rs: rank 1 of 2 holds [4, 8)
    with pytest.raises(ValueError):
        shard_spec(full, 0, 3)


def test_gemma4_shard_spec_and_iq_block_slices():
    """Gemma 4 under tensor parallelism: the 12B geometry divides for 2 and 4 ranks (8 would split down_proj inside a 256-block),
    the single full-attention KV head stays whole; I-quant block slices of a column- and a row-parallel weight reassemble."""
    sys.path.insert(0, str(ROOT))
    import torch
    from vllm_rust_amd.gemma4 import Gemma4Spec, L_DOWN, L_GATE, shard_gemma4_spec, shard_iq_blocks
    full = Gemma4Spec.gemma4_12b()
    for world in (2, 4):
        s = shard_gemma4_spec(full, 1, world)
        assert (s.num_heads, s.kv_heads_sliding, s.kv_heads_full, s.intermediate_size, s.vocab_size) == \
            (16 // world, 8 // world, 1, 15360 // world, 262144)
    with pytest.raises(ValueError, match="256-element"):
        shard_gemma4_spec(full, 0, 8)
    N, K, ts = 8, 1024, 98
    raw = torch.arange(N * (K // 256) * ts, dtype=torch.int32).to(torch.uint8)
    cols = [shard_iq_blocks(raw, L_GATE, N, K, ts, r, 2).reshape(N // 2, K // 256, ts) for r in range(2)]
    assert torch.equal(torch.cat(cols, 0).reshape(-1), raw)
    rows = [shard_iq_blocks(raw, L_DOWN, N, K, ts, r, 2).reshape(N, K // 512, ts) for r in range(2)]
    assert torch.equal(torch.cat(rows, 1).reshape(-1), raw)


def test_bench_job_plan_needs_no_gpu():
    """bench.py's argument logic: N = 1 is the single-GPU job; N > 1 measures ONE model tensor-parallel over the N ranks (the
    metric's TP curve) with the replicas as a secondary, --replicas restores the weak-scaling job; the Qwen3-8B geometry
    divides for 2, 4 and 8 ranks and every rank gets whole KV heads and whole quantisation groups."""
    sys.path.insert(0, str(ROOT))
    import bench
    assert bench.plan_job(1, 1, False, "awq", False)["mode"] == "single"
    for n in (2, 4, 8):
        p = bench.plan_job(n, n, False, "awq", False)
        assert (p["mode"], p["scaling"], p["tp"], p["secondary"]) == ("tp", "strong", n, "replicas")
        assert p["shards"] == [(32 // n, 8 // n, 12288 // n, r) for r in range(n)]
        assert (12288 // n) % 128 == 0 and (32 // n * 128) % 128 == 0
        q = bench.plan_job(n, n, True, "awq", False)
        assert (q["mode"], q["scaling"], q["tp"]) == ("replicas", "weak", 1)
    with pytest.raises(ValueError):
        bench.plan_job(3, 3, False, "awq", False)          # 8 KV heads do not divide
    with pytest.raises(ValueError):
        bench.plan_job(4, 2, False, "awq", False)          # --gpus must match the launcher's world size

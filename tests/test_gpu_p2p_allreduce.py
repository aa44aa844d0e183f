"""One-shot all-reduce over peer-mapped memory (vmx_comm_p2p_*): two processes on the one GPU of the test box map each other's
staging buffers through hipIpc handles and all-reduce bf16 rows of decode sizes; every rank must hold the rank-ordered f32 sum
rounded to bf16, bit for bit, also when the call is replayed from a hipGraph.  (Across GPUs the same code runs over xGMI; this
test covers the protocol -- slots, flags, call counters, parity -- not the link.)"""
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [4096, 8, 2048 * 3 + 8, 5120, 32768]


def _inputs(it, n, world):
    import torch
    return [(torch.randn(n, generator=torch.Generator().manual_seed(1000 * it + r)) * 2.0).to(torch.bfloat16) for r in range(world)]


def _expected(xs):
    import torch
    acc = torch.zeros_like(xs[0], dtype=torch.float32)
    for x in xs:   # rank order, f32
        acc = acc + x.float()
    return acc.to(torch.bfloat16)


def _worker(rank, world, conns, result_q):
    try:
        import torch
        import vllm_rust_amd as vmx
        from vllm_rust_amd.parallel import LocalCommunicator
        vmx.init(0)

        def exchange(mine):
            for c in conns[rank]:
                c.send((rank, mine))
            got = {rank: mine}
            for c in conns[rank]:
                r, b = c.recv()
                got[r] = b
            return [got[r] for r in range(world)]

        comm = LocalCommunicator(rank, world, exchange, max_bytes=64 * 1024)
        bad = []
        it = 0
        for rep in range(3):
            for n in SIZES:
                xs = _inputs(it, n, world)
                buf = xs[rank].cuda()
                comm.all_reduce_sum_bf16(buf)
                torch.cuda.synchronize()
                if not torch.equal(buf.cpu().view(torch.int16), _expected(xs).view(torch.int16)):
                    bad.append(("eager", rep, n))
                it += 1
        # the same call replayed from a graph with fresh inputs each time
        n = 4096
        buf = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
        src = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            buf.copy_(src)
            comm.all_reduce_sum_bf16(buf)   # warm-up call (counts as one all-reduce on every rank)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                buf.copy_(src)
                comm.all_reduce_sum_bf16(buf)
            for k in range(5):
                xs = _inputs(500 + k, n, world)
                src.copy_(xs[rank].cuda())
                g.replay()
                torch.cuda.synchronize()
                if not torch.equal(buf.cpu().view(torch.int16), _expected(xs).view(torch.int16)):
                    bad.append(("graph", k, n))
        result_q.put((rank, bad, comm.p2p_timeouts()))
        comm.close()
    except Exception as e:   # report instead of hanging the parent
        import traceback
        result_q.put((rank, [("exception", repr(e), traceback.format_exc())], -1))


def test_two_ranks_on_one_gpu_one_shot_all_reduce():
    world = 2
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    conns = {0: [a], 1: [b]}
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, conns, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            r, bad, timeouts = q.get(timeout=150)
            results[r] = (bad, timeouts)
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    assert set(results) == {0, 1}
    for r, (bad, timeouts) in results.items():
        assert timeouts == 0, f"rank {r}: {timeouts} flag waits timed out"
        assert not bad, f"rank {r}: {bad[:3]}"

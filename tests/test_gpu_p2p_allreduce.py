"""Collectives over peer-mapped memory (vmx_comm_p2p_*): 2, 4 and 8 processes on the one GPU of the test box map each other's
staging buffers through hipIpc handles.  All-reduce of bf16 rows of decode sizes: every rank must hold the rank-ordered f32 sum
rounded to bf16, bit for bit, also when the call is replayed from a hipGraph.  The exchange collectives (all_gather,
reduce_scatter, broadcast, barrier, all_to_all, all_to_all_v with host counts and with device-resident counts) against their
numpy statement.  A rank that misses a call must turn its peers' results into NaN and their next call into an error, never into
stale data.  (Across GPUs the same code runs over xGMI; these tests cover the protocol -- slots, flags, call counters, parity,
eight-rank indexing -- not the link.)"""
import numpy as np
import pytest

from helpers import make_exchange, spawn_ranks

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


def _allreduce_worker(rank, world, conns, result_q):
    try:
        import torch
        import vllm_rust_amd as vmx
        from vllm_rust_amd.parallel import LocalCommunicator
        vmx.init(0)
        comm = LocalCommunicator(rank, world, make_exchange(rank, world, conns), max_bytes=64 * 1024)
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
        # fp16 rows (a model that runs in fp16): the same protocol, sum rounded to fp16
        for n in SIZES[:3]:
            xs = [x.float().to(torch.float16) for x in _inputs(it, n, world)]
            buf = xs[rank].cuda()
            comm.all_reduce_sum_f16(buf)
            torch.cuda.synchronize()
            acc = torch.zeros(n, dtype=torch.float32)
            for x in xs:
                acc = acc + x.float()
            if not torch.equal(buf.cpu().view(torch.int16), acc.to(torch.float16).view(torch.int16)):
                bad.append(("eager fp16", n))
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
        result_q.put((rank, (bad, comm.p2p_timeouts(), comm.error_code())))
        comm.close()
    except Exception as e:   # report instead of hanging the parent
        import traceback
        result_q.put((rank, ([("exception", repr(e), traceback.format_exc())], -1, -1)))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_on_one_gpu_one_shot_all_reduce(world):
    results = spawn_ranks(world, _allreduce_worker)
    for r, (bad, timeouts, err) in results.items():
        assert not bad, f"rank {r}: {bad[:3]}"
        assert timeouts == 0 and err == 0, f"rank {r}: {timeouts} flag waits timed out, error code {err}"


def _collectives_worker(rank, world, conns, result_q):
    try:
        import torch
        import vllm_rust_amd as vmx
        from vllm_rust_amd.parallel import LocalCommunicator
        vmx.init(0)
        comm = LocalCommunicator(rank, world, make_exchange(rank, world, conns), max_bytes=64 * 1024, exchange_bytes=1 << 20)
        bad = []
        rng = lambda seed: np.random.default_rng(seed)
        for rep in range(3):
            # all_gather: bf16 rows and an odd number of bytes
            for n, dt in ((4096, np.uint16), (24, np.uint16), (7, np.uint8), (3 * 8192, np.float32)):
                parts = [rng(10 * rep + r).integers(0, 200, size=(2, n)).astype(dt) for r in range(world)]
                got = comm.all_gather(torch.from_numpy(parts[rank].view(np.uint8).reshape(2, -1)).cuda()).cpu().numpy()
                if not np.array_equal(got, np.concatenate([p.view(np.uint8).reshape(2, -1) for p in parts], 0)):
                    bad.append(("all_gather", rep, n))
            # reduce_scatter(sum): bf16 (f32 accumulation in rank order, one rounding) and f32
            n = 2048
            ins = [(torch.randn((world, n), generator=torch.Generator().manual_seed(77 * rep + r)) * 3).to(torch.bfloat16) for r in range(world)]
            got = comm.reduce_scatter_sum(ins[rank].cuda()).cpu()
            acc = torch.zeros(n)
            for r in range(world):
                acc = acc + ins[r][rank].float()
            if not torch.equal(got.view(torch.int16).reshape(-1), acc.to(torch.bfloat16).view(torch.int16)):
                bad.append(("reduce_scatter_bf16", rep))
            insf = [torch.randn((world, 512), generator=torch.Generator().manual_seed(5 * rep + r)) for r in range(world)]
            got = comm.reduce_scatter_sum(insf[rank].cuda()).cpu().reshape(-1)
            accf = torch.zeros(512)
            for r in range(world):
                accf = accf + insf[r][rank]
            if not torch.equal(got, accf):
                bad.append(("reduce_scatter_f32", rep))
            # broadcast from every root
            for root in range(world):
                src = torch.from_numpy(rng(1000 + 10 * rep + root).integers(0, 255, size=6000 + 4 * root).astype(np.uint8))
                t = (src.clone() if rank == root else torch.zeros_like(src)).cuda()
                comm.broadcast(t, root)
                if not torch.equal(t.cpu(), src):
                    bad.append(("broadcast", rep, root))
            comm.barrier()
            # all_to_all: the count exchange of the expert dispatch (world u32 values per rank)
            counts = [rng(300 + rep + r).integers(0, 50, size=world).astype(np.int32) for r in range(world)]
            got = comm.all_to_all(torch.from_numpy(counts[rank]).cuda()).cpu().numpy()
            if not np.array_equal(got, np.array([counts[r][rank] for r in range(world)], np.int32)):
                bad.append(("all_to_all", rep))
            # all_to_all_v: rows of 96 bf16; rank s sends (s + d + rep) % 4 rows to rank d
            H = 96
            send = [[(s + d + rep) % 4 for d in range(world)] for s in range(world)]
            rows = [rng(900 + rep + s).integers(0, 60000, size=(sum(send[s]), H)).astype(np.uint16) for s in range(world)]
            recv = [send[s][rank] for s in range(world)]
            got = comm.all_to_all_v(torch.from_numpy(rows[rank].view(np.int16)).cuda(), send[rank], recv).cpu().numpy().view(np.uint16)
            exp = []
            for s in range(world):
                o = sum(send[s][:rank])
                exp.append(rows[s][o:o + send[s][rank]])
            exp = np.concatenate(exp, 0) if exp else np.zeros((0, H), np.uint16)
            if not np.array_equal(got, exp):
                bad.append(("all_to_all_v", rep))
            # the same exchange with the counts resident on the device
            got, rr = comm.all_to_all_v_dev(torch.from_numpy(rows[rank].view(np.int16)).cuda() if rows[rank].size else torch.zeros((1, H), dtype=torch.int16, device="cuda"),
                                            torch.tensor(send[rank], dtype=torch.int32, device="cuda"), 4 * world)
            torch.cuda.synchronize()
            if rr.cpu().tolist() != recv or not np.array_equal(got.cpu().numpy().view(np.uint16)[:sum(recv)], exp):
                bad.append(("all_to_all_v_dev", rep, rr.cpu().tolist(), recv))
        # a graph holding three different collectives, replayed
        x = torch.zeros(1024, dtype=torch.bfloat16, device="cuda")
        src = torch.zeros(1024, dtype=torch.bfloat16, device="cuda")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            def body():
                x.copy_(src)
                comm.all_reduce_sum_bf16(x)
                g = comm.all_gather(x)
                return comm.reduce_scatter_sum(g)
            body()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                out = body()
            for k in range(4):
                xs = _inputs(40 + k, 1024, world)
                src.copy_(xs[rank].cuda())
                graph.replay()
                torch.cuda.synchronize()
                s = _expected(xs).float()
                acc = torch.zeros(1024)
                for _ in range(world):   # every rank gathered the same sums: reduce_scatter adds `world` copies of its chunk
                    acc = acc + s
                if not torch.equal(out.cpu().view(torch.int16), acc.to(torch.bfloat16).view(torch.int16)):
                    bad.append(("graph", k))
        result_q.put((rank, (bad, comm.p2p_timeouts(), comm.error_code())))
        comm.close()
    except Exception as e:
        import traceback
        result_q.put((rank, ([("exception", repr(e), traceback.format_exc())], -1, -1)))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_exchange_collectives_match_their_statement(world):
    results = spawn_ranks(world, _collectives_worker)
    for r, (bad, timeouts, err) in results.items():
        assert not bad, f"rank {r}: {bad[:3]}"
        assert timeouts == 0 and err == 0, f"rank {r}: {timeouts} flag waits timed out, error code {err}"


def _failure_worker(rank, world, conns, result_q):
    try:
        import torch
        import vllm_rust_amd as vmx
        from vllm_rust_amd._lib import VmxError
        from vllm_rust_amd.parallel import LocalCommunicator
        vmx.init(0)
        exchange = make_exchange(rank, world, conns)
        comm = LocalCommunicator(rank, world, exchange, max_bytes=64 * 1024, exchange_bytes=1 << 16)
        comm.set_timeout_ms(300)
        x = torch.ones(4096, dtype=torch.bfloat16, device="cuda")
        comm.all_reduce_sum_bf16(x)
        torch.cuda.synchronize()
        first_ok = bool((x.float() == world).all().item())
        exchange("healthy call done")   # every rank is past the first call
        report = {"first_ok": first_ok}
        if rank == world - 1:
            # this rank skips the second call: its peers must not get a sum
            exchange("peers have timed out")
            y = torch.ones(4096, dtype=torch.bfloat16, device="cuda")
            try:   # the peers poisoned this rank when they gave up: the next call fails on the host or yields NaN
                comm.all_reduce_sum_bf16(y)
                torch.cuda.synchronize()
                report["late_call"] = "nan" if bool(torch.isnan(y.float()).all().item()) else "data"
            except VmxError as e:
                report["late_call"] = "error"
                report["msg"] = e.msg
        else:
            y = torch.full((4096,), 2.0, dtype=torch.bfloat16, device="cuda")
            comm.all_reduce_sum_bf16(y)
            torch.cuda.synchronize()
            report["all_nan"] = bool(torch.isnan(y.float()).all().item())
            report["code"] = comm.error_code()
            report["timeouts"] = comm.p2p_timeouts()
            try:
                comm.all_reduce_sum_bf16(y)
                report["next_call"] = "accepted"
            except VmxError as e:
                report["next_call"] = "error"
                report["msg"] = e.msg
            exchange("peers have timed out")
        result_q.put((rank, report))
        comm.close()
    except Exception as e:
        import traceback
        result_q.put((rank, {"exception": repr(e), "trace": traceback.format_exc()}))


@pytest.mark.parametrize("world", [2, 4])
def test_a_missing_rank_is_an_error_not_stale_data(world):
    results = spawn_ranks(world, _failure_worker)
    for r, rep in results.items():
        assert "exception" not in rep, rep
        assert rep["first_ok"]
        if r == world - 1:
            assert rep["late_call"] in ("error", "nan"), rep
        else:
            assert rep["all_nan"], rep                 # no stale or partial sum reaches the caller
            assert rep["code"] in (1, 3), rep          # timed out itself, or told by a peer that did
            assert rep["next_call"] == "error" and "peer-mapped transport" in rep["msg"], rep

"""Expert parallelism with token dispatch (vmx_moe_ep_int4_forward): 2 and 4 ranks as processes on the one GPU of the test box,
every rank with its OWN tokens (data-parallel attention) and E / world of the experts; rows travel to the experts' ranks and
back through the device-count all_to_all_v over peer-mapped memory.  Every rank's output must equal, bit for bit, the
single-GPU expert layer (vmx_moe_int4_forward with all experts resident, same block form) on that rank's tokens -- and that
layer is checked against the CPU oracle's per-token expert loop in tests/test_gpu_moe.py.  Linear and round-robin placement
(moe/expert_map.rs:100-115); a captured graph replayed with new tokens; a rank that sends nothing."""
import numpy as np
import pytest

from helpers import make_exchange, spawn_ranks

pytestmark = pytest.mark.gpu

E, K, H, I, G = 8, 2, 256, 512, 128


def _experts(torch, vmx, orc, placement_world, rank, round_robin):
    rng = np.random.default_rng(31)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if a.dtype == np.uint32 else np.int16)).cuda()
    p13, p2 = [], []
    for ex in range(E):
        def mk(Kd, N):
            qw = rng.integers(0, 2 ** 32, size=(Kd // 8, N), dtype=np.uint32)
            sc = orc.f32_to_bf16(((rng.random((Kd // G, N)) * 0.5 + 0.75) / (6.5 * np.sqrt(Kd))).astype(np.float32))
            return qw, sc
        w13, w2 = mk(H, 2 * I), mk(I, H)
        p13.append(vmx.int4_pack(d(w13[0]), d(w13[1]), None, H, 2 * I, G))
        p2.append(vmx.int4_pack(d(w2[0]), d(w2[1]), None, I, H, G))
    per = E // placement_world
    mine = [e for e in range(E) if (e % placement_world if round_robin else e // per) == rank]   # in local order
    tab = lambda packs, ids: torch.tensor([packs[e].data_ptr() for e in ids], dtype=torch.int64, device="cuda")
    return (p13, p2), tab(p13, range(E)), tab(p2, range(E)), tab(p13, mine), tab(p2, mine)


def _worker(rank, world, conns, result_q, round_robin):
    try:
        import torch
        import vllm_rust_amd as vmx
        from oracle import oracle as orc
        from vllm_rust_amd._lib import lib
        from vllm_rust_amd.parallel import LocalCommunicator
        vmx.init(0)
        tmax = 48
        comm = LocalCommunicator(rank, world, make_exchange(rank, world, conns), max_bytes=64 * 1024, exchange_bytes=tmax * K * H * 2 + 4096)
        keep, t13, t2, l13, l2 = _experts(torch, vmx, orc, world, rank, round_robin)
        bad = []

        def case(tag, T, seed, cap_rows):
            g = torch.Generator().manual_seed(seed + 17 * rank)
            x = (torch.randn((max(T, 1), H), generator=g)).to(torch.bfloat16).cuda()
            logits = torch.randn((max(T, 1), E), generator=g).cuda()
            w, ids = vmx.topk_softmax(logits, K, True, True)
            if T == 0:   # this rank routes nothing in this call: every id out of range
                ids = torch.full_like(ids, -1)
            got = vmx.moe_ep_int4_forward(x, w, ids, l13, l2, E, H, I, G, comm, cap_rows, round_robin=round_robin)
            torch.cuda.synchronize()
            if T == 0:
                if not bool((got == 0).all().item()):
                    bad.append((tag, "rows of a rank that routed nothing must be zero"))
                return
            lib.vmx_debug_moe_block_rows(64 if cap_rows >= 96 else 16)   # the block form the owners used
            ref = vmx.moe_int4_forward(x, w, ids, t13, t2, E, H, I, G)
            lib.vmx_debug_moe_block_rows(-1)
            torch.cuda.synchronize()
            if not torch.equal(got.view(torch.int16), ref.view(torch.int16)):
                diff = (got.float() - ref.float()).abs().max().item()
                bad.append((tag, T, "max abs diff", diff))

        # a few rows per rank (16-row blocks at the owners), ranks with different token counts
        case("small", 3 + 2 * rank, 100, cap_rows=world * 16 * K)
        case("small-again", 5, 200, cap_rows=world * 16 * K)
        # one rank sends nothing
        case("idle-rank", 0 if rank == world - 1 else 7, 300, cap_rows=world * 16 * K)
        # enough rows for the 64-row matrix-core blocks
        case("blocks64", 40 + rank, 400, cap_rows=world * tmax * K)
        # captured once, replayed with new tokens: the counts change on the device only
        T = 6
        xg = torch.zeros((T, H), dtype=torch.bfloat16, device="cuda")
        lg = torch.zeros((T, E), device="cuda")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            def body():
                w, ids = vmx.topk_softmax(lg, K, True, True)
                return w, ids, vmx.moe_ep_int4_forward(xg, w, ids, l13, l2, E, H, I, G, comm, world * 16 * K, round_robin=round_robin)
            body()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                w, ids, out = body()
            for it in range(3):
                g = torch.Generator().manual_seed(900 + it * 10 + rank)
                xg.copy_(torch.randn((T, H), generator=g).to(torch.bfloat16))
                lg.copy_(torch.randn((T, E), generator=g))
                graph.replay()
                torch.cuda.synchronize()
                lib.vmx_debug_moe_block_rows(64 if world * 16 * K >= 96 else 16)   # the block form the owners used
                ref = vmx.moe_int4_forward(xg, w, ids, t13, t2, E, H, I, G)
                lib.vmx_debug_moe_block_rows(-1)
                torch.cuda.synchronize()
                if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
                    bad.append(("graph", it))
        result_q.put((rank, (bad, comm.p2p_timeouts(), comm.error_code())))
        comm.close()
    except Exception as e:
        import traceback
        result_q.put((rank, ([("exception", repr(e), traceback.format_exc())], -1, -1)))


@pytest.mark.parametrize("world,round_robin", [(2, False), (4, False), (4, True), (8, False)])
def test_expert_parallel_dispatch_equals_single_gpu_layer(world, round_robin):
    results = spawn_ranks(world, _worker, (round_robin,))
    for r, (bad, timeouts, err) in results.items():
        assert not bad, f"rank {r}: {bad[:3]}"
        assert timeouts == 0 and err == 0, f"rank {r}: {timeouts} flag waits timed out, error code {err}"

"""Tensor-parallel plumbing: communicator set-up over torch.distributed and Megatron-style shard
selection of AWQ checkpoint tensors.

Sharding follows the reference's dense TP layers (crates/core/src/distributed/parallel_layers.rs:
ColumnParallelLinear :36-175 shards the output dim, RowParallelLinear :181-325 shards the input dim
and all-reduces; models/tp_layers.rs:113-226): q/k/v/gate/up column-parallel, o/down row-parallel,
lm_head split by vocabulary rows.  The reference has no quantized TP path; the alignment rules are the
int4 tile stream's: column shards multiples of 16 (and of 8 for the packed zeros), row shards
multiples of the group size.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math

import torch

from ._lib import check, lib


VMX_COMM_BF16, VMX_COMM_F32 = 9, 7


class Communicator:
    """RCCL communicator of libvmx; the unique id is broadcast through torch.distributed."""

    def __init__(self, rank: int, world: int, group=None):
        import torch.distributed as dist
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            check(lib.vmx_comm_unique_id(buf))
            uid = torch.tensor(list(buf), dtype=torch.uint8)
        if world > 1:
            if dist.get_backend(group) == "nccl":
                dev = uid.cuda()
                dist.broadcast(dev, src=0, group=group)
                uid = dev.cpu()
            else:
                dist.broadcast(uid, src=0, group=group)
        raw = (C.c_uint8 * 128)(*uid.tolist())
        self.handle = C.c_void_p()
        check(lib.vmx_comm_create(raw, rank, world, C.byref(self.handle)))
        self.rank, self.world = rank, world

    def enable_p2p(self, max_bytes: int = 64 * 1024, exchange_bytes: int = 0, group=None, self_test: bool = True) -> bool:
        """Set up the peer-mapped transport for all-reduce messages of at most max_bytes (and exchange collectives of at most
        exchange_bytes): every rank exports its staging buffer and flag words as hipIpc handles, the handles are all-gathered
        through torch.distributed, every rank maps its peers and (self_test) all-reduces a known row over the new transport.
        The outcome is agreed between the ranks after every stage: if any rank failed, ALL ranks drop the transport and stay on
        RCCL, so the ranks of a group never take different transports.  Returns whether the transport is in use."""
        import torch.distributed as dist

        def all_agree(ok: bool) -> bool:
            if self.world == 1:
                return ok
            flags = [None] * self.world
            dist.all_gather_object(flags, bool(ok), group=group)
            return all(flags)

        self.p2p_note = ""
        mine = (C.c_uint8 * 128)()
        rc = lib.vmx_comm_p2p_export(self.handle, int(max_bytes), int(exchange_bytes), mine)
        if rc != 0:
            self.p2p_note = lib.vmx_last_error().decode("utf-8", "replace")
        if not all_agree(rc == 0):
            check(lib.vmx_comm_p2p_disable(self.handle))
            self.p2p_note = self.p2p_note or "a peer rank could not export its staging memory"
            return False
        gathered = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(gathered, bytes(mine), group=group)
        else:
            gathered = [bytes(mine)]
        allh = (C.c_uint8 * (128 * self.world))(*b"".join(gathered))
        rc = lib.vmx_comm_p2p_import(self.handle, allh)
        if rc != 0:
            self.p2p_note = lib.vmx_last_error().decode("utf-8", "replace")
        if not all_agree(rc == 0):
            check(lib.vmx_comm_p2p_disable(self.handle))
            self.p2p_note = self.p2p_note or "a peer rank could not map the staging memory"
            return False
        if self_test:
            ok = True
            try:   # rank r contributes r + 1 in every element: the sum is world (world + 1) / 2 on every rank
                t = torch.full((4096,), float(self.rank + 1), dtype=torch.bfloat16, device="cuda")
                for _ in range(3):
                    u = t.clone()
                    self.all_reduce_sum_bf16(u)
                    torch.cuda.synchronize()
                    ok = ok and bool((u.float() == self.world * (self.world + 1) / 2).all().item())
                ok = ok and self.error_code() == 0
            except Exception as e:   # noqa: BLE001 - any failure means "do not use the transport"
                self.p2p_note = repr(e)
                ok = False
            if not all_agree(ok):
                check(lib.vmx_comm_p2p_disable(self.handle))
                self.p2p_note = self.p2p_note or "the self-test all-reduce over peer-mapped memory gave a wrong sum on some rank"
                return False
        return True

    def p2p_ready(self) -> bool:
        return bool(lib.vmx_comm_p2p_ready(self.handle))

    def p2p_timeouts(self) -> int:
        n = C.c_int()
        check(lib.vmx_comm_p2p_timeouts(self.handle, C.byref(n)))
        return n.value

    def error_code(self) -> int:
        """Sticky error of the peer-mapped transport (0 = healthy); a host read, no synchronisation."""
        n = C.c_int()
        check(lib.vmx_comm_error(self.handle, C.byref(n)))
        return n.value

    def set_timeout_ms(self, ms: int) -> None:
        check(lib.vmx_comm_p2p_set_timeout_ms(self.handle, int(ms)))

    @staticmethod
    def _st():
        return torch.cuda.current_stream().cuda_stream

    def all_reduce_sum_bf16(self, t: torch.Tensor) -> None:
        assert t.dtype == torch.bfloat16 and t.is_contiguous()
        check(lib.vmx_comm_all_reduce_sum_bf16(self.handle, t.data_ptr(), t.numel(), self._st()))

    def all_reduce_sum_f16(self, t: torch.Tensor) -> None:
        """The same sum over fp16 rows (models that run in fp16): f32 accumulation in rank order, one rounding to fp16."""
        assert t.dtype == torch.float16 and t.is_contiguous()
        check(lib.vmx_comm_all_reduce_sum_f16(self.handle, t.data_ptr(), t.numel(), self._st()))

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """all_gather along dimension 0 (DeviceCommunicator::all_gather, communicator.rs:44)."""
        assert t.is_contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        check(lib.vmx_comm_all_gather(self.handle, t.data_ptr(), out.data_ptr(), t.numel() * t.element_size(), self._st()))
        return out

    def reduce_scatter_sum(self, t: torch.Tensor) -> torch.Tensor:
        """reduce_scatter(Sum) along dimension 0 (communicator.rs:51); bf16 or f32."""
        assert t.is_contiguous() and t.shape[0] % self.world == 0 and t.dtype in (torch.bfloat16, torch.float32)
        out = torch.empty((t.shape[0] // self.world,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        check(lib.vmx_comm_reduce_scatter_sum(self.handle, t.data_ptr(), out.data_ptr(), out.numel(),
                                              VMX_COMM_BF16 if t.dtype == torch.bfloat16 else VMX_COMM_F32, self._st()))
        return out

    def broadcast(self, t: torch.Tensor, root: int) -> None:
        assert t.is_contiguous()
        check(lib.vmx_comm_broadcast(self.handle, t.data_ptr(), t.numel() * t.element_size(), int(root), self._st()))

    def barrier(self) -> None:
        check(lib.vmx_comm_barrier(self.handle, self._st()))

    def all_to_all(self, t: torch.Tensor) -> torch.Tensor:
        """Chunk r of dimension 0 goes to rank r (communicator.rs:67-76)."""
        assert t.is_contiguous() and t.shape[0] % self.world == 0
        out = torch.empty_like(t)
        check(lib.vmx_comm_all_to_all(self.handle, t.data_ptr(), out.data_ptr(), t.numel() * t.element_size() // self.world, self._st()))
        return out

    def all_to_all_v(self, t: torch.Tensor, send_splits, recv_splits) -> torch.Tensor:
        """Rows of dimension 0 (communicator.rs:78-93): send_splits[r] rows to rank r, recv_splits[r] rows from rank r."""
        assert t.is_contiguous() and len(send_splits) == len(recv_splits) == self.world and sum(send_splits) == t.shape[0]
        row = math.prod(t.shape[1:])
        out = torch.empty((sum(recv_splits),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        sc = (C.c_size_t * self.world)(*[int(v) for v in send_splits])
        rc = (C.c_size_t * self.world)(*[int(v) for v in recv_splits])
        check(lib.vmx_comm_all_to_all_v(self.handle, t.data_ptr(), sc, out.data_ptr(), rc, row * t.element_size(), self._st()))
        return out

    def all_to_all_v_dev(self, t: torch.Tensor, send_rows: torch.Tensor, capacity_rows: int):
        """all_to_all_v of rows with the counts on the device: returns (rows [capacity_rows, ...], recv_rows int32 [world])."""
        assert t.is_contiguous() and send_rows.dtype == torch.int32 and send_rows.numel() == self.world
        out = torch.zeros((capacity_rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        recv = torch.zeros(self.world, dtype=torch.int32, device=t.device)
        row = t.numel() // t.shape[0] * t.element_size()
        check(lib.vmx_comm_all_to_all_v_dev(self.handle, t.data_ptr(), send_rows.data_ptr(), out.data_ptr(), recv.data_ptr(), row,
                                            int(capacity_rows), self._st()))
        return out, recv

    def close(self):
        if self.handle:
            lib.vmx_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class LocalCommunicator(Communicator):
    """A communicator without RCCL (vmx_comm_create_local): only the peer-mapped transport, with the handle exchange left to
    the caller (`exchange(my_128_bytes) -> list of every rank's bytes in rank order`)."""

    def __init__(self, rank: int, world: int, exchange, max_bytes: int = 64 * 1024, exchange_bytes: int = 0):
        self.handle = C.c_void_p()
        check(lib.vmx_comm_create_local(rank, world, C.byref(self.handle)))
        self.rank, self.world = rank, world
        mine = (C.c_uint8 * 128)()
        check(lib.vmx_comm_p2p_export(self.handle, int(max_bytes), int(exchange_bytes), mine))
        gathered = exchange(bytes(mine))
        allh = (C.c_uint8 * (128 * world))(*b"".join(gathered))
        check(lib.vmx_comm_p2p_import(self.handle, allh))


def shard_spec(spec, tp_rank: int, tp_size: int):
    """Per-rank model geometry: heads and MLP width divided, hidden and vocabulary global."""
    if tp_size == 1:
        return dataclasses.replace(spec, tp_size=1, tp_rank=0)
    moe = getattr(spec, "num_experts", 0) > 0
    for name in ("num_heads", "num_kv_heads", "num_experts" if moe else "intermediate_size", "vocab_size"):
        if getattr(spec, name) % tp_size:
            raise ValueError(f"{name}={getattr(spec, name)} is not divisible by tp_size={tp_size}")
    if not moe and (spec.intermediate_size // tp_size) % max(spec.group_size, 128):
        raise ValueError("row-parallel shard of the down projection must be a multiple of the group size (and of 128)")
    if (spec.num_heads // tp_size * spec.head_dim) % max(spec.group_size, 128):
        raise ValueError("row-parallel shard of the o projection must be a multiple of the group size (and of 128)")
    # mixture-of-experts: attention is tensor-parallel, the MLP expert-parallel (whole experts of full width per rank)
    inter = spec.intermediate_size if moe else spec.intermediate_size // tp_size
    return dataclasses.replace(spec, num_heads=spec.num_heads // tp_size, num_kv_heads=spec.num_kv_heads // tp_size,
                               intermediate_size=inter, tp_size=tp_size, tp_rank=tp_rank)


def local_experts(num_experts: int, tp_rank: int, tp_size: int) -> range:
    """Experts rank `tp_rank` holds: a contiguous block (ExpertPlacement::Linear, crates/core/src/moe/expert_map.rs:100-109)."""
    per = num_experts // max(tp_size, 1)
    return range(tp_rank * per, (tp_rank + 1) * per)


def shard_awq_column(qweight, qzeros, scales, tp_rank: int, tp_size: int):
    """Column-parallel shard (output features): AWQ qweight [K, N/8], qzeros [G, N/8], scales [G, N]."""
    n8 = qweight.shape[1]
    assert n8 % tp_size == 0 and (n8 // tp_size * 8) % 16 == 0, "column shard must be a multiple of 16 outputs"
    w = n8 // tp_size
    sl8 = slice(tp_rank * w, (tp_rank + 1) * w)
    sl = slice(tp_rank * w * 8, (tp_rank + 1) * w * 8)
    return qweight[:, sl8].contiguous(), qzeros[:, sl8].contiguous(), scales[:, sl].contiguous()


def shard_awq_row(qweight, qzeros, scales, tp_rank: int, tp_size: int, group_size: int):
    """Row-parallel shard (input features): rows of qweight, whole groups of qzeros/scales."""
    K, G = qweight.shape[0], qzeros.shape[0]
    assert K % tp_size == 0 and (K // tp_size) % group_size == 0, "row shard must be whole quantisation groups"
    k, g = K // tp_size, G // tp_size
    return (qweight[tp_rank * k:(tp_rank + 1) * k].contiguous(), qzeros[tp_rank * g:(tp_rank + 1) * g].contiguous(),
            scales[tp_rank * g:(tp_rank + 1) * g].contiguous())


def shard_exl3_column(trellis, suh, svh, tp_rank: int, tp_size: int):
    """Column-parallel shard of an EXL3 linear (output features): trellis [K/16, N/16, 16*bits] by column tiles, svh [N]
    by columns, suh [K] whole.  A shard must hold whole 128-column blocks (the output Hadamard works on them), so the
    shards' outputs concatenate to the unsharded output bit for bit."""
    n16 = trellis.shape[1]
    assert n16 % tp_size == 0 and (n16 // tp_size * 16) % 128 == 0, "column shard must be a multiple of 128 outputs"
    w = n16 // tp_size
    return (trellis[:, tp_rank * w:(tp_rank + 1) * w].contiguous(), suh.contiguous(),
            svh[tp_rank * w * 16:(tp_rank + 1) * w * 16].contiguous())


def shard_exl3_row(trellis, suh, svh, tp_rank: int, tp_size: int):
    """Row-parallel shard (input features): trellis by row tiles, suh [K] by rows (whole 128-row input Hadamard blocks),
    svh [N] whole; every rank runs the complete linear on its slice of the input and the fp16 outputs are summed."""
    k16 = trellis.shape[0]
    assert k16 % tp_size == 0 and (k16 // tp_size * 16) % 128 == 0, "row shard must be a multiple of 128 inputs"
    h = k16 // tp_size
    return (trellis[tp_rank * h:(tp_rank + 1) * h].contiguous(), suh[tp_rank * h * 16:(tp_rank + 1) * h * 16].contiguous(),
            svh.contiguous())

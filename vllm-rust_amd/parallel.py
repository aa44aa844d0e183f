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

import torch

from ._lib import check, lib


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

    def enable_p2p(self, max_bytes: int = 64 * 1024, group=None) -> None:
        """Set up the one-shot peer-mapped all-reduce for messages of at most max_bytes: every rank exports its staging buffer
        and flag words as hipIpc handles, the handles are all-gathered through torch.distributed, every rank maps its peers."""
        import torch.distributed as dist
        mine = (C.c_uint8 * 128)()
        check(lib.vmx_comm_p2p_export(self.handle, int(max_bytes), mine))
        gathered = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(gathered, bytes(mine), group=group)
        else:
            gathered = [bytes(mine)]
        allh = (C.c_uint8 * (128 * self.world))(*b"".join(gathered))
        check(lib.vmx_comm_p2p_import(self.handle, allh))

    def p2p_timeouts(self) -> int:
        n = C.c_int()
        check(lib.vmx_comm_p2p_timeouts(self.handle, C.byref(n)))
        return n.value

    def all_reduce_sum_bf16(self, t: torch.Tensor) -> None:
        assert t.dtype == torch.bfloat16 and t.is_contiguous()
        check(lib.vmx_comm_all_reduce_sum_bf16(self.handle, t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream))

    def close(self):
        if self.handle:
            lib.vmx_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class LocalCommunicator(Communicator):
    """A communicator without RCCL (vmx_comm_create_local): only the peer-mapped one-shot all-reduce, with the handle
    exchange left to the caller (`exchange(my_128_bytes) -> list of every rank's bytes in rank order`)."""

    def __init__(self, rank: int, world: int, exchange, max_bytes: int = 64 * 1024):
        self.handle = C.c_void_p()
        check(lib.vmx_comm_create_local(rank, world, C.byref(self.handle)))
        self.rank, self.world = rank, world
        mine = (C.c_uint8 * 128)()
        check(lib.vmx_comm_p2p_export(self.handle, int(max_bytes), mine))
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

"""Tensor-parallel process groups and the decode all-reduce.

One process per GPU.  ``torch.distributed`` with backend ``"nccl"`` IS RCCL on ROCm (xGMI inside
a node); ``"gloo"`` is used by the CPU unit tests.  The reference never writes a collective
itself: they are implicit in upstream ``RowParallelLinear`` for ``o_proj`` / ``down_proj``
(vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:54-60,118-124) on the group that
``init_worker_distributed_environment(..., current_platform.dist_backend)`` creates
(vllm_omni/worker/gpu_ar_worker.py:69-75).

World layout: ``world = dp * tp``; rank r belongs to TP group ``r // tp`` (consecutive ranks, so a
TP group sits on neighbouring GPUs of the node) and has ``tp_rank = r % tp``.  Data-parallel
replicas never communicate on the data path.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass
class ParallelState:
    world_size: int = 1
    rank: int = 0
    tp_size: int = 1
    tp_rank: int = 0
    dp_size: int = 1
    dp_rank: int = 0
    tp_group: Optional["dist.ProcessGroup"] = None
    simulated: bool = False   # one rank of a TP plan run alone, collectives stubbed (sizing estimates only)


_STATE = ParallelState()


def get_parallel_state() -> ParallelState:
    return _STATE


def get_tensor_model_parallel_world_size() -> int:
    return _STATE.tp_size


def get_tensor_model_parallel_rank() -> int:
    return _STATE.tp_rank


def get_tp_group():
    return _STATE.tp_group


def init_distributed_environment(world_size: int, rank: int, tp_size: int, backend: str = "nccl",
                                 init_method: Optional[str] = None, device: Optional[torch.device] = None) -> ParallelState:
    """Create the default group (if needed) and one TP subgroup per data-parallel replica."""
    global _STATE
    if world_size % tp_size:
        raise ValueError(f"world_size {world_size} is not a multiple of tp_size {tp_size}")
    if world_size > 1 and not dist.is_initialized():
        kw = {}
        if device is not None and backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend=backend, init_method=init_method or "env://", world_size=world_size,
                                rank=rank, **kw)
    dp_size = world_size // tp_size
    tp_group = None
    if tp_size > 1:
        # every rank must create every subgroup, in the same order
        for g in range(dp_size):
            ranks = list(range(g * tp_size, (g + 1) * tp_size))
            grp = dist.new_group(ranks=ranks, backend=backend)
            if rank in ranks:
                tp_group = grp
    _STATE = ParallelState(world_size=world_size, rank=rank, tp_size=tp_size, tp_rank=rank % tp_size,
                           dp_size=dp_size, dp_rank=rank // tp_size, tp_group=tp_group)
    return _STATE


def simulate_tensor_parallel_rank(tp_size: int, tp_rank: int = 0) -> ParallelState:
    """ONE rank of a ``tp_size`` plan without its peers: shapes, shards and kernels are those of the real rank,
    the all-reduce is skipped and the all-gather repeats the local shard.  Results are meaningless; the step TIME
    is the per-rank compute of that plan (bench.py --simulate-shard).  Never used by the product path."""
    global _STATE
    _STATE = ParallelState(world_size=tp_size, rank=tp_rank, tp_size=tp_size, tp_rank=tp_rank, simulated=True)
    return _STATE


def destroy_distributed_environment() -> None:
    global _STATE
    _STATE = ParallelState()
    if dist.is_initialized():
        dist.destroy_process_group()


def tensor_model_parallel_all_reduce(x: torch.Tensor) -> torch.Tensor:
    """In-place sum over the TP group (no-op at tp=1).  2 per decoder layer: after o_proj and
    after down_proj; message [T, H] in the model dtype (1.8 MB at the bench shape)."""
    if _STATE.tp_size > 1 and not _STATE.simulated:
        dist.all_reduce(x, group=_STATE.tp_group)
    return x


def tensor_model_parallel_all_gather(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """Concatenate the TP shards along ``dim`` (vocab-parallel logits)."""
    if _STATE.tp_size == 1:
        return x
    if _STATE.simulated:
        return torch.cat([x] * _STATE.tp_size, dim=dim)
    parts = [torch.empty_like(x) for _ in range(_STATE.tp_size)]
    dist.all_gather(parts, x.contiguous(), group=_STATE.tp_group)
    return torch.cat(parts, dim=dim)


# ---- head partition rules (qwen2_old.py:89-102) --------------------------------------
def partition_heads(total_q_heads: int, total_kv_heads: int, tp_size: int) -> tuple[int, int, int]:
    """(q heads per rank, kv heads per rank, kv replication factor).

    Q heads must divide by tp; KV heads are partitioned when ``total_kv_heads >= tp`` and
    otherwise replicated so that each rank holds exactly one (``tp % total_kv_heads == 0``).
    """
    if total_q_heads % tp_size:
        raise ValueError(f"num_attention_heads {total_q_heads} not divisible by tp_size {tp_size}")
    if total_kv_heads >= tp_size:
        if total_kv_heads % tp_size:
            raise ValueError(f"num_key_value_heads {total_kv_heads} not divisible by tp_size {tp_size}")
        return total_q_heads // tp_size, total_kv_heads // tp_size, 1
    if tp_size % total_kv_heads:
        raise ValueError(f"tp_size {tp_size} not a multiple of num_key_value_heads {total_kv_heads}")
    return total_q_heads // tp_size, 1, tp_size // total_kv_heads


def kv_head_for_rank(total_kv_heads: int, tp_size: int, tp_rank: int) -> range:
    """Global KV-head indices held by ``tp_rank``."""
    if total_kv_heads >= tp_size:
        per = total_kv_heads // tp_size
        return range(tp_rank * per, (tp_rank + 1) * per)
    rep = tp_size // total_kv_heads
    return range(tp_rank // rep, tp_rank // rep + 1)

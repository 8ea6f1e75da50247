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
    custom_ar: Optional["CustomAllReduce"] = None   # the in-tree two-shot all-reduce (VLLM_OMNI_AMD_ALLREDUCE=custom)


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
    if _STATE.custom_ar is not None:      # a previous plan's communicator (bench.py measures several plans)
        _STATE.custom_ar.close()
    _STATE = ParallelState(world_size=world_size, rank=rank, tp_size=tp_size, tp_rank=rank % tp_size,
                           dp_size=dp_size, dp_rank=rank // tp_size, tp_group=tp_group)
    if tp_size > 1 and os.environ.get("VLLM_OMNI_AMD_ALLREDUCE", "rccl") == "custom":
        enable_custom_all_reduce(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    return _STATE


class CustomAllReduce:
    """The in-tree decode all-reduce (``csrc/allreduce.hip``, C ABI ``mi_allreduce_*``): two-shot over IPC-mapped
    peer buffers, fp32 accumulation in rank order, in-stream, hipGraph-replayable.  The 64-byte IPC handles are
    exchanged over the TP group itself (``all_gather_object``); RCCL stays the transport for everything else and
    the DEFAULT for the all-reduce too until a multi-GPU run has validated this path."""

    def __init__(self, group, tp_rank: int, tp_size: int, device: torch.device, max_bytes: int = 8 << 20,
                 timeout_ms: int = 10000):
        import ctypes

        from . import _C

        self._C, self.lib = _C, _C.lib
        self.rank, self.world, self.device, self.max_bytes = tp_rank, tp_size, device, int(max_bytes)
        self.dead = False
        self._h = ctypes.c_void_p()
        mine = (ctypes.c_ubyte * _C.MI_ALLREDUCE_HANDLE_BYTES)()
        with torch.cuda.device(device):
            _C.check("mi_allreduce_init", self.lib.mi_allreduce_init(ctypes.byref(self._h), tp_rank, tp_size,
                                                                     self.max_bytes, int(timeout_ms), mine))
            handles: list = [None] * tp_size
            dist.all_gather_object(handles, bytes(mine), group=group)
            blob = (ctypes.c_ubyte * (_C.MI_ALLREDUCE_HANDLE_BYTES * tp_size)).from_buffer_copy(b"".join(handles))
            _C.check("mi_allreduce_register_buffer", self.lib.mi_allreduce_register_buffer(self._h, blob))
        dist.barrier(group=group)          # nobody runs before every rank has mapped every peer

    def supports(self, x: torch.Tensor) -> bool:
        return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous()
                and x.numel() % 8 == 0 and -(-x.numel() // (8 * self.world)) * self.world * 16 <= self.max_bytes)

    def all_reduce(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = x if out is None else out
        dt = self._C.MI_DTYPE_BF16 if x.dtype == torch.bfloat16 else self._C.MI_DTYPE_F16
        self._C.check("mi_allreduce_run", self.lib.mi_allreduce_run(
            self._h, out.data_ptr(), x.data_ptr(), x.numel(), 0, None, 0, dt, torch.cuda.current_stream().cuda_stream))
        return out

    def all_reduce_partials(self, partials: torch.Tensor, split: int, bias: Optional[torch.Tensor], rows: int,
                            cols: int, dtype: torch.dtype) -> torch.Tensor:
        """All-reduce of a row-parallel projection handed over as the streaming GEMM's fp32 split-K partials
        [split, rows, cols]: their reduction (+ bias, one rounding) happens in the kernel's copy-in phase."""
        out = torch.empty(rows, cols, dtype=dtype, device=partials.device)
        dt = self._C.MI_DTYPE_BF16 if dtype == torch.bfloat16 else self._C.MI_DTYPE_F16
        self._C.check("mi_allreduce_run", self.lib.mi_allreduce_run(
            self._h, out.data_ptr(), partials.data_ptr(), rows * cols, int(split),
            bias.data_ptr() if bias is not None else None, cols, dt, torch.cuda.current_stream().cuda_stream))
        return out

    def check(self) -> None:
        """Synchronising health check: raises once any spin has timed out (a peer never arrived)."""
        self._C.check("mi_allreduce_status", self.lib.mi_allreduce_status(self._h))

    def poll(self) -> None:
        """Health check without a device round trip (``mi_allreduce_poll``: the kernel mirrors its sticky status word
        into host-mapped pinned memory).  Covers every launch that had completed at the caller's last synchronisation
        point — the runner calls it right after the event wait of a step's device->host output copy, so a timed-out
        all-reduce (whose output was never written) raises BEFORE that step's tokens reach the scheduler.  After a
        timeout the communicator is dead: the peers' call counters no longer match, ``mi_allreduce_run`` refuses."""
        if self.lib.mi_allreduce_poll(self._h) != self._C.MI_OK:
            self.dead = True
            raise RuntimeError("in-tree all-reduce: a peer did not arrive within the timeout; this step's hidden states "
                               "are invalid and the communicator is dead (restart the TP group)")

    def close(self) -> None:
        if self._h:
            self.lib.mi_allreduce_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def enable_custom_all_reduce(device: torch.device, max_bytes: int = 8 << 20, timeout_ms: int = 10000) -> bool:
    """Build the in-tree all-reduce for the current TP group (collective: every rank of the group calls)."""
    if _STATE.tp_size <= 1 or _STATE.simulated:
        return False
    if _STATE.tp_size > 8:
        raise ValueError("the in-tree all-reduce serves one node: tp_size <= 8")
    _STATE.custom_ar = CustomAllReduce(_STATE.tp_group, _STATE.tp_rank, _STATE.tp_size, device, max_bytes, timeout_ms)
    return True


def simulate_tensor_parallel_rank(tp_size: int, tp_rank: int = 0) -> ParallelState:
    """ONE rank of a ``tp_size`` plan without its peers: shapes, shards and kernels are those of the real rank,
    the all-reduce is skipped and the all-gather repeats the local shard.  Results are meaningless; the step TIME
    is the per-rank compute of that plan (bench.py --simulate-shard).  Never used by the product path."""
    global _STATE
    _STATE = ParallelState(world_size=tp_size, rank=tp_rank, tp_size=tp_size, tp_rank=tp_rank, simulated=True)
    return _STATE


def destroy_distributed_environment() -> None:
    global _STATE
    if _STATE.custom_ar is not None:
        # no rank may free its shared region while a peer can still be reading it in phase 2 of a launch: drain this
        # rank's queue, then meet the others, THEN unmap / free
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist.is_initialized() and not getattr(_STATE.custom_ar, "dead", False):
            dist.barrier(group=_STATE.tp_group)
        _STATE.custom_ar.close()
    _STATE = ParallelState()
    if dist.is_initialized():
        dist.destroy_process_group()


def tensor_model_parallel_all_reduce(x: torch.Tensor) -> torch.Tensor:
    """In-place sum over the TP group (no-op at tp=1).  2 per decoder layer: after o_proj and
    after down_proj; message [T, H] in the model dtype (1.8 MB at the bench shape)."""
    if _STATE.tp_size > 1 and not _STATE.simulated:
        ar = _STATE.custom_ar
        if ar is not None and ar.supports(x):
            return ar.all_reduce(x)
        dist.all_reduce(x, group=_STATE.tp_group)
    return x


def tensor_model_parallel_all_reduce_partials(part) -> Optional[torch.Tensor]:
    """All-reduce of an ``ops.SplitKPartials`` (a row-parallel projection's split-K partials) without a separate
    reduce launch — only with the in-tree all-reduce; returns None when that path is not active (the caller
    materialises the partials and calls ``tensor_model_parallel_all_reduce``)."""
    ar = _STATE.custom_ar
    if ar is None or _STATE.tp_size <= 1 or _STATE.simulated:
        return None
    rows, cols = part.m, part.n
    if cols % 8 or -(-rows * cols // (8 * ar.world)) * ar.world * 16 > ar.max_bytes:
        return None
    part._live()
    out = ar.all_reduce_partials(part.part, part.split, part.bias, rows, cols, part.dtype)
    return out.view(*part.lead, cols)


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

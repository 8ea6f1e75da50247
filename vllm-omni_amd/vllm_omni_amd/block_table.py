"""Persistent block table + slot mapping of the decode step.

Mirrors the surface the reference touches on upstream's ``BlockTable`` /
``MultiGroupBlockTable``:

  * ``input_batch.block_table.append_row(new_block_ids, req_index)``
                                   vllm_omni/worker/gpu_model_runner.py:501-502
  * ``input_batch.block_table[0].block_table.cpu[req, :n]``   vllm_omni/core/prefix_cache.py:264
  * ``input_batch.block_table[0].slot_mapping.cpu``           vllm_omni/worker/gpu_ar_model_runner.py:231
  * ``input_batch.block_table.block_tables``                  vllm_omni/core/prefix_cache.py:233

``block_table[B, max_blocks]`` int32; ``slot_mapping[T]`` int64 =
``block_table[req, pos // b] * b + pos % b`` with a -1 padded tail for graph replay.  Host
arrays are pinned; ``commit_*`` issues one async H2D copy each into persistent device
buffers (whose addresses a captured hipGraph keeps reading).
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _C
from ._C import lib

PAD_SLOT_ID = -1


class CpuGpuBuffer:
    """A pinned host tensor (with a numpy view) and its persistent device twin."""

    def __init__(self, *shape: int, dtype: torch.dtype, device: Optional[torch.device], pin: bool = True):
        on_gpu = device is not None and torch.device(device).type == "cuda"
        self.cpu = torch.zeros(*shape, dtype=dtype, pin_memory=bool(pin and on_gpu and torch.cuda.is_available()))
        self.np = self.cpu.numpy()
        # without a GPU (host-logic unit tests) the "device" twin is the host tensor itself
        self.gpu = torch.zeros(*shape, dtype=dtype, device=device) if on_gpu else self.cpu

    def copy_to_gpu(self, n: Optional[int] = None) -> torch.Tensor:
        if self.gpu is self.cpu:
            return self.cpu if n is None else self.cpu[:n]
        if n is None:
            self.gpu.copy_(self.cpu, non_blocking=True)
            return self.gpu
        self.gpu[:n].copy_(self.cpu[:n], non_blocking=True)
        return self.gpu[:n]


class StepArena:
    """One pinned host block and its device twin holding the step's SMALL index arrays side by side (positions, sequence
    lengths, query offsets, slot mapping, reply / sampler / predictor rows): ``buffer()`` hands out ``CpuGpuBuffer``-shaped
    views, their ``copy_to_gpu(n)`` only marks bytes dirty, and ``flush()`` sends the dirty extent with ONE host->device
    copy where each array used to be its own (seven launches of ~5.5 us + ~6.5 us of gap each per decode step on the
    captured path: profiles/r03_shard_tp4dp2_timeline.md).  Bytes between two dirty arrays are copied along — host and
    device twins of an arena array are only ever written through the host side, so that rewrites equal data.  ``flush()``
    must run before the first kernel that reads any of the arrays is enqueued (the runner does, at the end of its input
    commit); arrays the DEVICE also writes (the last sampled token ids) do not belong in an arena."""

    ALIGN = 256

    def __init__(self, nbytes: int, device: Optional[torch.device], pin: bool = True):
        on_gpu = device is not None and torch.device(device).type == "cuda"
        self.cpu = torch.zeros(nbytes, dtype=torch.uint8, pin_memory=bool(pin and on_gpu and torch.cuda.is_available()))
        self.gpu = torch.zeros(nbytes, dtype=torch.uint8, device=device) if on_gpu else self.cpu
        self._used = 0
        self._lo: Optional[int] = None
        self._hi = 0

    def buffer(self, *shape: int, dtype: torch.dtype) -> "ArenaBuffer":
        n = 1
        for d in shape:
            n *= d
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        off = -(-self._used // self.ALIGN) * self.ALIGN
        if off + nbytes > self.cpu.numel():
            raise MemoryError("StepArena is too small for its buffers")
        self._used = off + nbytes
        return ArenaBuffer(self, off, shape, dtype, nbytes)

    def mark(self, lo: int, hi: int) -> None:
        self._lo = lo if self._lo is None else min(self._lo, lo)
        self._hi = max(self._hi, hi)

    def flush(self) -> None:
        if self._lo is None:
            return
        if self.gpu is not self.cpu:
            self.gpu[self._lo:self._hi].copy_(self.cpu[self._lo:self._hi], non_blocking=True)
        self._lo, self._hi = None, 0


class ArenaBuffer:
    """``CpuGpuBuffer`` interface over a slice of a ``StepArena``."""

    def __init__(self, arena: StepArena, off: int, shape, dtype: torch.dtype, nbytes: int):
        self._arena, self._off = arena, off
        self.cpu = arena.cpu[off:off + nbytes].view(dtype).view(*shape)
        self.np = self.cpu.numpy()
        self.gpu = arena.gpu[off:off + nbytes].view(dtype).view(*shape)
        self._row_bytes = nbytes // shape[0] if shape and shape[0] else 0
        self._nbytes = nbytes

    def copy_to_gpu(self, n: Optional[int] = None) -> torch.Tensor:
        nb = self._nbytes if n is None else n * self._row_bytes
        if nb:
            self._arena.mark(self._off, self._off + nb)
        return self.gpu if n is None else self.gpu[:n]


class BlockTable:
    def __init__(self, block_size: int, max_num_reqs: int, max_num_blocks_per_req: int, max_num_batched_tokens: int,
                 device: Optional[torch.device] = None, pin_memory: bool = True, arena: Optional[StepArena] = None):
        self.block_size = block_size
        self.max_num_reqs = max_num_reqs
        self.max_num_blocks_per_req = max_num_blocks_per_req
        self.max_num_batched_tokens = max_num_batched_tokens
        self.block_table = CpuGpuBuffer(max_num_reqs, max_num_blocks_per_req, dtype=torch.int32, device=device,
                                        pin=pin_memory)
        self.slot_mapping = (arena.buffer(max_num_batched_tokens, dtype=torch.int64) if arena is not None else
                             CpuGpuBuffer(max_num_batched_tokens, dtype=torch.int64, device=device, pin=pin_memory))
        self.num_blocks_per_row = np.zeros(max_num_reqs, dtype=np.int32)

    # ---- rows -----------------------------------------------------------------
    def append_row(self, block_ids: Sequence[int], row_idx: int) -> None:
        if not len(block_ids):
            return
        n = int(self.num_blocks_per_row[row_idx])
        m = len(block_ids)
        if n + m > self.max_num_blocks_per_req:
            raise ValueError(f"row {row_idx}: {n}+{m} blocks exceed max_num_blocks_per_req="
                             f"{self.max_num_blocks_per_req}")
        self.block_table.np[row_idx, n:n + m] = block_ids
        self.num_blocks_per_row[row_idx] = n + m

    def add_row(self, block_ids: Sequence[int], row_idx: int) -> None:
        self.num_blocks_per_row[row_idx] = 0
        self.append_row(block_ids, row_idx)

    def move_row(self, src: int, tgt: int) -> None:
        n = int(self.num_blocks_per_row[src])
        self.block_table.np[tgt, :n] = self.block_table.np[src, :n]
        self.num_blocks_per_row[tgt] = n

    def swap_row(self, a: int, b: int) -> None:
        self.block_table.np[[a, b]] = self.block_table.np[[b, a]]
        self.num_blocks_per_row[[a, b]] = self.num_blocks_per_row[[b, a]]

    def clear_row(self, row_idx: int) -> None:
        self.num_blocks_per_row[row_idx] = 0

    def clear(self) -> None:
        self.block_table.np.fill(0)
        self.num_blocks_per_row.fill(0)

    # ---- slot mapping -----------------------------------------------------------
    def compute_slot_mapping(self, req_indices: np.ndarray, positions: np.ndarray) -> None:
        """slot_mapping.np[:T] for the tokens scheduled this step (host, vectorised in C)."""
        t = int(positions.shape[0])
        if t > self.max_num_batched_tokens:
            raise ValueError("more tokens than max_num_batched_tokens")
        ri = np.ascontiguousarray(req_indices, dtype=np.int32)
        ps = np.ascontiguousarray(positions, dtype=np.int64)
        _C.check("mi_compute_slot_mapping", lib.mi_compute_slot_mapping(
            self.slot_mapping.np.ctypes.data, self.block_table.np.ctypes.data, self.max_num_blocks_per_req,
            ri.ctypes.data, ps.ctypes.data, t, self.block_size))

    def commit_block_table(self, num_reqs: int) -> torch.Tensor:
        return self.block_table.copy_to_gpu(num_reqs)

    def commit_slot_mapping(self, num_tokens: int, num_tokens_padded: Optional[int] = None) -> torch.Tensor:
        """H2D of the first ``num_tokens`` slots; the tail up to ``num_tokens_padded`` is set to
        PAD_SLOT_ID so padded graph-replay tokens never write the cache."""
        n_pad = num_tokens if num_tokens_padded is None else num_tokens_padded
        if n_pad > num_tokens:
            self.slot_mapping.np[num_tokens:n_pad] = PAD_SLOT_ID
        return self.slot_mapping.copy_to_gpu(n_pad)

    def get_device_tensor(self, num_reqs: int) -> torch.Tensor:
        return self.block_table.gpu[:num_reqs]

    def get_cpu_tensor(self) -> torch.Tensor:
        return self.block_table.cpu

    def get_numpy_array(self) -> np.ndarray:
        return self.block_table.np


class MultiGroupBlockTable:
    """One ``BlockTable`` per KV-cache group (the talker has exactly one)."""

    def __init__(self, max_num_reqs: int, max_model_len: int, max_num_batched_tokens: int, block_sizes: Sequence[int],
                 device: Optional[torch.device] = None, pin_memory: bool = True, arena: Optional[StepArena] = None):
        self.block_tables = [
            BlockTable(bs, max_num_reqs, -(-max_model_len // bs), max_num_batched_tokens, device, pin_memory, arena)
            for bs in block_sizes
        ]

    def append_row(self, block_ids: Sequence[Sequence[int]], row_idx: int) -> None:
        for bt, ids in zip(self.block_tables, block_ids):
            bt.append_row(ids, row_idx)

    def add_row(self, block_ids: Sequence[Sequence[int]], row_idx: int) -> None:
        for bt, ids in zip(self.block_tables, block_ids):
            bt.add_row(ids, row_idx)

    def move_row(self, src: int, tgt: int) -> None:
        for bt in self.block_tables:
            bt.move_row(src, tgt)

    def swap_row(self, a: int, b: int) -> None:
        for bt in self.block_tables:
            bt.swap_row(a, b)

    def clear_row(self, row_idx: int) -> None:
        for bt in self.block_tables:
            bt.clear_row(row_idx)

    def compute_slot_mapping(self, req_indices: np.ndarray, positions: np.ndarray) -> None:
        for bt in self.block_tables:
            bt.compute_slot_mapping(req_indices, positions)

    def commit_block_table(self, num_reqs: int) -> None:
        for bt in self.block_tables:
            bt.commit_block_table(num_reqs)

    def commit_slot_mapping(self, num_tokens: int, num_tokens_padded: Optional[int] = None) -> None:
        for bt in self.block_tables:
            bt.commit_slot_mapping(num_tokens, num_tokens_padded)

    def clear(self) -> None:
        for bt in self.block_tables:
            bt.clear()

    def __getitem__(self, idx: int) -> BlockTable:
        return self.block_tables[idx]

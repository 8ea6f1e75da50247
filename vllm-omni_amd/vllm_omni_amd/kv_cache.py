"""KV-cache block manager, cache tensors and budget.

``KVCacheManager`` exposes the method set the reference's schedulers call on upstream's
``vllm.v1.core.kv_cache_manager.KVCacheManager``:

  * ``new_step_starts``                  vllm_omni/core/sched/omni_generation_scheduler.py:49
  * ``allocate_slots(request, n, num_lookahead_tokens=)`` -> ``KVCacheBlocks | None``
                                          omni_generation_scheduler.py:100-110,167-177
  * ``get_block_ids(request_id)``         vllm_omni/core/sched/omni_ar_scheduler.py:628-630
  * ``free(request)``                     omni_ar_scheduler.py:528
  * ``take_events()``                     omni_ar_scheduler.py:480
  * ``get_num_common_prefix_blocks``      omni_generation_scheduler.py:210
  * ``take_new_block_ids``                omni_generation_scheduler.py:256

The free list / reference counts live in the C++ pool behind the C ABI
(``mi_blockpool_*``); block id 0 is the reserved null block.  One KV-cache group only (the
talker is a uniform full-attention stack), so every tuple below has length 1.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Any, Iterable, Optional, Sequence

import torch

from . import _C
from ._C import lib


def cdiv(a: int, b: int) -> int:
    return -(-a // b)


# --------------------------------------------------------------------------- pool
class BlockPool:
    """Thin owner of a ``mi_blockpool`` (FIFO free list, refcounts, null block 0)."""

    def __init__(self, num_blocks: int):
        if num_blocks < 2:
            raise ValueError("num_blocks must be >= 2 (block 0 is the null block)")
        self._h = lib.mi_blockpool_create(num_blocks)
        if not self._h:
            raise MemoryError("mi_blockpool_create failed")
        self.num_blocks = num_blocks

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:  # `lib` is already gone at interpreter shutdown
            lib.mi_blockpool_destroy(h)

    def get_num_free_blocks(self) -> int:
        return int(lib.mi_blockpool_num_free(self._h))

    def get_new_blocks(self, n: int) -> Optional[list[int]]:
        """``n`` fresh block ids, or None when the pool cannot serve all of them."""
        if n == 0:
            return []
        buf = (ctypes.c_int32 * n)()
        rc = lib.mi_blockpool_alloc(self._h, n, buf)
        if rc == _C.MI_ERR_EXHAUSTED:
            return None
        _C.check("mi_blockpool_alloc", rc)
        return list(buf)

    def touch(self, ids: Sequence[int]) -> None:
        if ids:
            arr = (ctypes.c_int32 * len(ids))(*ids)
            _C.check("mi_blockpool_ref", lib.mi_blockpool_ref(self._h, arr, len(ids)))

    def free_blocks(self, ids: Sequence[int]) -> None:
        if ids:
            arr = (ctypes.c_int32 * len(ids))(*ids)
            _C.check("mi_blockpool_free", lib.mi_blockpool_free(self._h, arr, len(ids)))

    def ref_cnt(self, block_id: int) -> int:
        return int(lib.mi_blockpool_refcount(self._h, block_id))

    @property
    def usage(self) -> float:
        total = self.num_blocks - 1
        return 1.0 - self.get_num_free_blocks() / total if total else 0.0


# ------------------------------------------------------------------------ blocks
@dataclass
class KVCacheBlocks:
    """What ``allocate_slots`` returns: the NEW blocks, one list per KV-cache group."""

    blocks: tuple[list[int], ...]

    def get_block_ids(self) -> tuple[list[int], ...]:
        return tuple(list(b) for b in self.blocks)

    def __add__(self, other: "KVCacheBlocks") -> "KVCacheBlocks":
        return KVCacheBlocks(tuple(a + b for a, b in zip(self.blocks, other.blocks)))


@dataclass
class BlockStored:
    block_ids: list[int]
    request_id: str


@dataclass
class BlockRemoved:
    block_ids: list[int]
    request_id: str


# ----------------------------------------------------------------------- manager
class KVCacheManager:
    """Request -> physical block ids.  Grows by ceil((computed + new + lookahead) / b)."""

    def __init__(self, num_blocks: int, block_size: int, enable_kv_cache_events: bool = False,
                 track_new_block_ids: bool = False):
        self.block_size = block_size
        self.num_blocks = num_blocks
        self.block_pool = BlockPool(num_blocks)
        self.req_to_blocks: dict[str, list[int]] = {}
        self._events: list[Any] = []
        self._events_on = enable_kv_cache_events
        self._new_block_ids: Optional[list[int]] = [] if track_new_block_ids else None
        # requests whose KV is being handed to the next stage: their blocks outlive `free` until the runner reports
        # the extraction (OmniARScheduler.active_kv_transfers / waiting_for_transfer_free, omni_ar_scheduler.py:559-605)
        self.active_kv_transfers: set[str] = set()
        self.waiting_for_transfer_free: dict[str, Any] = {}

    # -- step hooks -------------------------------------------------------------
    def new_step_starts(self) -> None:
        """Called once per scheduling step before any allocation (bookkeeping reset)."""
        return None

    @property
    def usage(self) -> float:
        return self.block_pool.usage

    # -- allocation -------------------------------------------------------------
    def allocate_slots(self, request, num_new_tokens: int, num_lookahead_tokens: int = 0,
                       **_unused) -> Optional[KVCacheBlocks]:
        """Make room for ``num_new_tokens`` (+ lookahead) more tokens of ``request``.

        Returns the newly allocated blocks (possibly empty) or ``None`` when the pool is
        exhausted — callers break out and fall back (omni_generation_scheduler.py:105-110).
        All-or-nothing: a failed call leaves the request's blocks unchanged.
        """
        if num_new_tokens < 0 or num_lookahead_tokens < 0:
            raise ValueError("token counts must be non-negative")
        rid = request.request_id
        have = self.req_to_blocks.get(rid)
        total_tokens = int(request.num_computed_tokens) + num_new_tokens + num_lookahead_tokens
        need = cdiv(total_tokens, self.block_size) - (len(have) if have else 0)
        if need <= 0:
            if have is None:
                self.req_to_blocks[rid] = []
            return KVCacheBlocks(([],))
        new = self.block_pool.get_new_blocks(need)
        if new is None:
            return None
        if have is None:
            self.req_to_blocks[rid] = new[:]
        else:
            have.extend(new)
        if self._new_block_ids is not None:
            self._new_block_ids.extend(new)
        if self._events_on:
            self._events.append(BlockStored(new[:], rid))
        return KVCacheBlocks((new,))

    def get_blocks(self, request_id: str) -> KVCacheBlocks:
        return KVCacheBlocks((list(self.req_to_blocks.get(request_id, [])),))

    def get_block_ids(self, request_id: str) -> tuple[list[int], ...]:
        """One list per KV-cache group (the reference takes element 0)."""
        return (list(self.req_to_blocks.get(request_id, [])),)

    def mark_request_for_kv_transfer(self, request_id: str, seq_len: int) -> dict:
        """A finishing request whose KV goes to the next stage (``_mark_request_for_kv_transfer``,
        omni_ar_scheduler.py:607-660): snapshot ``{"seq_len", "block_ids"}`` with the ids truncated to
        ceil(seq_len / block_size), and hold the blocks — ``free`` is deferred until ``on_kv_extracted`` names
        the request (the runner's ``kv_extracted_req_ids``)."""
        ids = list(self.req_to_blocks.get(request_id, []))
        n = cdiv(seq_len, self.block_size)
        if len(ids) > n:
            ids = ids[:n]
        self.active_kv_transfers.add(request_id)
        return {"seq_len": int(seq_len), "block_ids": ids}

    def on_kv_extracted(self, request_ids: Iterable[str]) -> list[str]:
        """The runner has copied these requests' KV out of the paged cache (``kv_extracted_req_ids``): their
        transfers are no longer active, and the ones whose ``free`` was deferred are freed now
        (omni_ar_scheduler.py:521-539).  Returns the ids actually freed."""
        freed = []
        for rid in request_ids or ():
            self.active_kv_transfers.discard(rid)
            req = self.waiting_for_transfer_free.pop(rid, None)
            if req is not None:
                self._free_now(req)
                freed.append(rid)
        return freed

    def free(self, request) -> None:
        """Return the request's blocks to the pool tail, LAST block first (so a later request
        re-uses the tail of a sequence before its head, as upstream's eviction order does).  While the request's
        KV transfer is active the blocks are kept (the extraction reads them) and freed by ``on_kv_extracted``."""
        if request.request_id in self.active_kv_transfers:
            self.waiting_for_transfer_free[request.request_id] = request
            return
        self._free_now(request)

    def _free_now(self, request) -> None:
        blocks = self.req_to_blocks.pop(request.request_id, None)
        if blocks:
            self.block_pool.free_blocks(list(reversed(blocks)))
            if self._events_on:
                self._events.append(BlockRemoved(blocks, request.request_id))

    def get_num_common_prefix_blocks(self, running_request_id: str) -> list[int]:
        """Blocks shared by every running request.  Without prefix sharing that is 0 unless a
        single request runs (then all of its blocks trivially qualify)."""
        if len(self.req_to_blocks) == 1 and running_request_id in self.req_to_blocks:
            return [len(self.req_to_blocks[running_request_id])]
        return [0]

    def take_new_block_ids(self) -> list[int]:
        if self._new_block_ids is None:
            return []
        out, self._new_block_ids = self._new_block_ids, []
        return out

    def take_events(self) -> list[Any]:
        out, self._events = self._events, []
        return out

    def reset_prefix_cache(self) -> bool:
        return True


# ------------------------------------------------------------------- cache tensors
@dataclass
class KVCacheSpec:
    """Geometry of one layer's paged cache (per TP rank)."""

    block_size: int
    num_kv_heads: int
    head_dim: int
    dtype: torch.dtype

    @property
    def page_bytes(self) -> int:
        """Bytes of K+V for one block of one layer."""
        return 2 * self.block_size * self.num_kv_heads * self.head_dim * torch.finfo(self.dtype).bits // 8


def num_blocks_for_budget(available_bytes: int, num_layers: int, spec: KVCacheSpec) -> int:
    """A15: how many blocks fit the KV budget (all layers share block ids)."""
    return max(0, int(available_bytes) // (num_layers * spec.page_bytes))


def determine_available_memory(total_bytes: int, gpu_memory_utilization: float, used_by_process: int,
                               kv_cache_memory_bytes: Optional[int] = None) -> int:
    """KV budget: requested = total * utilisation; available = requested - this process's use.

    Restates ``OmniGPUWorkerBase.determine_available_memory`` (vllm_omni/worker/base.py:77-156):
    an explicit ``kv_cache_memory_bytes`` wins; never negative.
    """
    if kv_cache_memory_bytes is not None:
        return int(kv_cache_memory_bytes)
    requested = int(total_bytes * gpu_memory_utilization)
    return max(0, requested - int(used_by_process))


KV_LAYOUTS = ("NB2", "2NB", "NB2_HEAD_MAJOR")


def allocate_kv_cache(num_layers: int, num_blocks: int, spec: KVCacheSpec, device, layout: str = "NB2"):
    """Per-layer cache tensors.  Logical views are always ``[num_blocks, block_size, kv_heads,
    head_dim]`` for K and V, in one of the layouts ``normalize_layer_kv`` accepts
    (vllm_omni/distributed/omni_connectors/utils/kv_utils.py:378-450):

      NB2             tensor [num_blocks, 2, block_size, kv_heads, head_dim]   (K/V at dim 1)
      2NB             tensor [2, num_blocks, block_size, kv_heads, head_dim]   (K/V at dim 0)
      NB2_HEAD_MAJOR  (K, V) tuple of views over [num_blocks, 2, kv_heads, block_size, head_dim]
                      — each (block, head) is one contiguous block_size*head_dim*2 B run.
    """
    b, h, d = spec.block_size, spec.num_kv_heads, spec.head_dim
    out = []
    for _ in range(num_layers):
        if layout == "NB2":
            out.append(torch.zeros(num_blocks, 2, b, h, d, dtype=spec.dtype, device=device))
        elif layout == "2NB":
            out.append(torch.zeros(2, num_blocks, b, h, d, dtype=spec.dtype, device=device))
        elif layout == "NB2_HEAD_MAJOR":
            t = torch.zeros(num_blocks, 2, h, b, d, dtype=spec.dtype, device=device)
            out.append((t[:, 0].permute(0, 2, 1, 3), t[:, 1].permute(0, 2, 1, 3)))
        else:
            raise ValueError(f"unknown KV layout {layout!r}; expected one of {KV_LAYOUTS}")
    return out


def normalize_layer_kv(layer_kv):
    """(key_blocks, value_blocks) views of any accepted layout, else ``None``.

    Same acceptance rules and precedence as the reference's ``normalize_layer_kv``
    (kv_utils.py:419-450): ``[2, nb, ...]`` first, then ``[nb, 2, ...]``, then a 2-tuple.
    """
    if isinstance(layer_kv, torch.Tensor):
        if layer_kv.ndim >= 3 and layer_kv.shape[0] == 2:
            kb, vb = layer_kv[0], layer_kv[1]
        elif layer_kv.ndim >= 3 and layer_kv.shape[1] == 2:
            kb, vb = layer_kv[:, 0], layer_kv[:, 1]
        else:
            return None
    elif isinstance(layer_kv, tuple):
        if len(layer_kv) != 2:
            return None
        kb, vb = layer_kv
        if not isinstance(kb, torch.Tensor) or not isinstance(vb, torch.Tensor):
            return None
    else:
        return None
    if kb.ndim < 2 or vb.ndim < 2:
        return None
    return kb, vb


def extract_request_kv(kv_caches: Sequence, block_ids: Sequence[int], seq_len: int):
    """Per layer ``[seq_len, kv_heads, head_dim]`` K and V of one request (None for bad layers).

    Restates ``_extract_kv_cache`` (vllm_omni/distributed/omni_connectors/
    kv_transfer_manager.py:816-848): out-of-range block ids are dropped, the overlapping block
    range is used when K and V disagree in length, truncation only when seq_len is shorter than
    the gathered rows.  Indexing only (torch gather on whatever device the cache lives on).
    """
    keys: list[Optional[torch.Tensor]] = []
    vals: list[Optional[torch.Tensor]] = []
    for layer_kv in kv_caches:
        pair = normalize_layer_kv(layer_kv)
        if pair is None:
            keys.append(None)
            vals.append(None)
            continue
        kb, vb = pair
        max_block = min(kb.shape[0], vb.shape[0]) - 1
        valid = [b for b in block_ids if 0 <= b <= max_block]
        if not valid:
            keys.append(None)
            vals.append(None)
            continue
        idx = torch.as_tensor(valid, dtype=torch.int64, device=kb.device)
        fk = kb.index_select(0, idx).flatten(0, 1)
        fv = vb.index_select(0, idx).flatten(0, 1)
        if seq_len < fk.shape[0]:
            fk, fv = fk[:seq_len], fv[:seq_len]
        keys.append(fk.detach().contiguous())
        vals.append(fv.detach().contiguous())
    return keys, vals

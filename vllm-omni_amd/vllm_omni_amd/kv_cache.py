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
  * ``get_computed_blocks`` / ``cache_blocks`` / ``reset_prefix_cache``: upstream's prefix caching, reached from
    ``super().schedule()`` (omni_ar_scheduler.py:188) when ``enable_prefix_caching`` is set

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

    # -- prefix-cache index (full blocks by 128-bit content hash) -------------------
    def cache_insert(self, block_id: int, digest: bytes) -> None:
        lo, hi = int.from_bytes(digest[:8], "little"), int.from_bytes(digest[8:16], "little")
        _C.check("mi_blockpool_cache_insert", lib.mi_blockpool_cache_insert(self._h, block_id, lo, hi))

    def cache_lookup(self, digests: Sequence[bytes]) -> list[int]:
        """Block ids of the longest indexed PREFIX of ``digests``."""
        n = len(digests)
        if n == 0:
            return []
        words = (ctypes.c_uint64 * (2 * n)).from_buffer_copy(b"".join(digests))
        out = (ctypes.c_int32 * n)()
        hits = int(lib.mi_blockpool_cache_lookup(self._h, words, n, out))
        if hits < 0:
            _C.check("mi_blockpool_cache_lookup", hits)
        return list(out[:hits])

    def num_cached(self) -> int:
        return int(lib.mi_blockpool_num_cached(self._h))

    def reset_cache(self) -> bool:
        return int(lib.mi_blockpool_reset_cache(self._h)) == _C.MI_OK

    def take_evicted(self) -> list[int]:
        out = []
        buf = (ctypes.c_int32 * 256)()
        while True:
            n = int(lib.mi_blockpool_take_evicted(self._h, buf, 256))
            out.extend(buf[:n])
            if n < 256:
                return out

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


@dataclass
class PrefixCacheStats:
    """upstream's ``vllm.v1.metrics.stats.PrefixCacheStats`` fields (what its stat loggers read from
    ``SchedulerStats.prefix_cache_stats``), with mapping access for this package's own callers."""

    requests: int = 0
    queries: int = 0
    hits: int = 0
    reset: bool = False

    def __getitem__(self, key: str):
        return getattr(self, key)

    def keys(self):
        return ("requests", "queries", "hits", "reset")


def _is_finished(request) -> bool:
    st = getattr(request, "status", None)
    if st is None:
        return True          # a bare request object (tests, the reference's generation scheduler): free means gone
    fin = getattr(request, "is_finished", None)
    if callable(fin):
        return bool(fin())
    return "FINISHED" in getattr(st, "name", str(st))


# ----------------------------------------------------------------------- hashing
def _request_tokens(request) -> Sequence[int]:
    toks = getattr(request, "all_token_ids", None)
    if toks is not None:
        return toks
    return list(request.prompt_token_ids) + list(getattr(request, "output_token_ids", []) or [])


def _prompt_embed_source(request):
    """The rows that ARE the prompt's content when its token ids are placeholders: ``request.prompt_embeds``, or — the
    reference's thinker -> talker hand-over, where the model's preprocess hook builds the rows from
    ``additional_information`` (qwen2_5_omni.py:681-764) — the thinker's prompt rows, one row standing for the speaker
    token (a constant per voice: the voice name is hashed with it) and the first reply row.  Returns (Tensor | None, tag)."""
    pe = getattr(request, "prompt_embeds", None)
    if isinstance(pe, torch.Tensor):
        return pe, b""
    info = getattr(request, "additional_information", None)
    if isinstance(info, dict) and isinstance(info.get("thinker_result"), torch.Tensor):
        tr = info["thinker_result"]
        tp = info.get("prompt_embeds")
        e = tr.shape[-1]
        tp = tp.reshape(-1, e).to(tr.dtype).cpu() if isinstance(tp, torch.Tensor) else tr.new_zeros(0, e).cpu()
        rows = torch.cat([tp, tr.new_zeros(1, e).cpu(), tr[:1].reshape(1, e).cpu()], dim=0)
        return rows, b"voice:" + str(info.get("voice_type", "Chelsie")).encode()
    return None, b""


# keys of ``additional_information`` whose effect on the KV contents the hasher accounts for: the thinker hand-over's
# prompt rows and first reply row are hashed as bytes (``_prompt_embed_source``), the reply queue salts generated blocks
# with the request id (``_has_reply_rows``), the id lists are the token ids themselves, the voice is hashed by name
_HASHED_INFO_KEYS = frozenset({"prompt_embeds", "thinker_result", "thinker_reply_part", "prompt_token_ids",
                               "thinker_output_token_ids", "voice_type"})


def _unhashed_conditioning(request) -> bool:
    """True when the request carries per-request conditioning the block hash does NOT cover: any other
    ``additional_information`` key (Qwen3-style ``thinker_prefill_embeddings`` / ``thinker_hidden_states``, whatever a
    custom ``model_cls`` preprocess hook consumes).  Talker prompts are placeholder ids, so two such requests would
    collide on ids alone: their blocks are then salted with the request id — never shared, only re-hit by the request
    itself after a preemption (fail closed; the reference ships enable_prefix_caching: false for every stage, so no
    reference behaviour defines this case)."""
    info = getattr(request, "additional_information", None)
    if not info:
        return False
    if not isinstance(info, dict):
        return True
    return any(k not in _HASHED_INFO_KEYS for k in info)


def _token_slice(request, lo: int, hi: int) -> Sequence[int]:
    """Token ids [lo, hi) of prompt + output without materialising the concatenation."""
    toks = getattr(request, "all_token_ids", None)
    if toks is not None:
        return toks[lo:hi]
    prompt = request.prompt_token_ids
    n = len(prompt)
    if hi <= n:
        return prompt[lo:hi]
    out = getattr(request, "output_token_ids", None) or []
    if lo >= n:
        return out[lo - n:hi - n]
    return list(prompt[lo:]) + list(out[:hi - n])


def _prompt_rows(request) -> int:
    pe, _ = _prompt_embed_source(request)
    return int(pe.shape[0]) if pe is not None else 0


def _has_reply_rows(request) -> bool:
    info = getattr(request, "additional_information", None)
    return isinstance(info, dict) and (info.get("thinker_reply_part") is not None or info.get("thinker_result") is not None)


def block_content_hash(parent: bytes, token_ids: Sequence[int], extra: bytes = b"") -> bytes:
    """Chained 128-bit content hash of one FULL block: blake2b-128(parent || int64 token ids || extra) — upstream
    chains ``hash((parent_hash, token_ids, extra_keys))`` the same way.  ``extra`` carries what the token ids do not
    determine: the rows of ``prompt_embeds`` the block covers (every talker prompt is embeddings behind placeholder ids,
    stage_input_processors/qwen2_5_omni.py:6-61) and, for blocks that hold generated positions of a request whose decode
    inputs also depend on per-request data (``thinker_reply_part``), the request id."""
    import hashlib

    import numpy as np

    h = hashlib.blake2b(parent, digest_size=16)
    h.update(np.asarray(token_ids, dtype=np.int64).tobytes())
    if extra:
        h.update(extra)
    return h.digest()


# ----------------------------------------------------------------------- manager
class KVCacheManager:
    """Request -> physical block ids.  Grows by ceil((computed + new + lookahead) / b); with
    ``enable_caching`` full blocks are indexed by a chained content hash and re-used by later requests with the
    same prefix (upstream's prefix caching, the precondition of ``OmniTensorPrefixCache``,
    vllm_omni/core/prefix_cache.py:255-264).

    Two ways to build it: ``KVCacheManager(num_blocks, block_size, ...)``, or upstream's own call
    ``KVCacheManager(kv_cache_config, max_model_len, hash_block_size, enable_caching=..., enable_kv_cache_events=...)``
    as ``vllm.v1.core.sched.scheduler.Scheduler.__init__`` makes it (``kv_cache_config.num_blocks`` and the block size
    of its single KV-cache group are read)."""

    def __init__(self, *args, **kwargs):
        first = args[0] if args else None
        if isinstance(first, int) or (not args and "num_blocks" in kwargs):
            self._init(*args, **kwargs)
        else:
            self._init_upstream(*args, **kwargs)

    def _init_upstream(self, kv_cache_config=None, max_model_len: Optional[int] = None,
                       hash_block_size: Optional[int] = None, enable_caching: bool = True, use_eagle: bool = False,
                       log_stats: bool = False, enable_kv_cache_events: bool = False, dcp_world_size: int = 1,
                       pcp_world_size: int = 1, metrics_collector=None, track_new_block_ids: bool = False):
        """upstream's own argument order, as ``Scheduler.__init__`` calls it"""
        if kv_cache_config is None:
            raise TypeError("KVCacheManager needs (num_blocks, block_size) or a kv_cache_config")
        groups = list(getattr(kv_cache_config, "kv_cache_groups", []) or [])
        if len(groups) > 1:
            raise ValueError("one KV-cache group only (the talker is a uniform full-attention stack)")
        block_size = int(groups[0].kv_cache_spec.block_size) if groups else hash_block_size
        self._init(int(kv_cache_config.num_blocks), block_size, enable_kv_cache_events, track_new_block_ids,
                   enable_caching, max_model_len=max_model_len, use_eagle=use_eagle, log_stats=log_stats,
                   dcp_world_size=dcp_world_size, pcp_world_size=pcp_world_size)

    def _init(self, num_blocks: int = None, block_size: int = None, enable_kv_cache_events: bool = False,
              track_new_block_ids: bool = False, enable_caching: bool = False, *, max_model_len: Optional[int] = None,
              use_eagle: bool = False, log_stats: bool = False, dcp_world_size: int = 1, pcp_world_size: int = 1):
        if num_blocks is None or block_size is None:
            raise TypeError("KVCacheManager needs (num_blocks, block_size) or a kv_cache_config")
        if dcp_world_size != 1 or pcp_world_size != 1 or use_eagle:
            raise ValueError("context-parallel / eagle block layouts are not built (not on the talker path)")
        self.block_size = block_size
        self.num_blocks = num_blocks
        self.max_model_len = max_model_len
        self.enable_caching = bool(enable_caching)
        self.log_stats = log_stats
        self.block_pool = BlockPool(num_blocks)
        self.req_to_blocks: dict[str, list[int]] = {}
        self._req_hashes: dict[str, list[bytes]] = {}       # chained hashes of the request's full blocks so far
        self._num_cached_blocks: dict[str, int] = {}        # leading blocks of the request already indexed
        self._req_src: dict[str, list] = {}                 # per request: [prompt rows on the host, voice tag, #rows, salted, private]
        # optional hook ``f(request, lo, hi) -> bytes | None``: extra content of the tokens [lo, hi) beyond ids and
        # prompt rows (a model class with its own preprocess hook); None = never share this block
        self.prefix_cache_extra_keys = None
        self.prefix_cache_queries = 0                        # tokens looked up / found (upstream's PrefixCacheStats)
        self.prefix_cache_hits = 0
        self.prefix_cache_requests = 0
        self._prefix_reset = False
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
        self._drain_evictions()

    def _drain_evictions(self) -> None:
        # the pool records every eviction of a hashed block for KV events; with events off nobody consumes them, and
        # the list would grow for the life of the server
        if self.enable_caching and not self._events_on:
            self.block_pool.take_evicted()

    @property
    def usage(self) -> float:
        return self.block_pool.usage

    # -- prefix cache -----------------------------------------------------------
    def _block_hashes(self, request, num_blocks: int) -> list[bytes]:
        """Chained hashes of the request's first ``num_blocks`` full blocks (extended incrementally: a block's
        content never changes once it is full)."""
        rid = request.request_id
        hs = self._req_hashes.setdefault(rid, [])
        if len(hs) >= num_blocks:
            return hs[:num_blocks]
        b = self.block_size
        # the prompt-row source is built ONCE per request (the thinker hand-over's is a concatenation + a device->host
        # copy of the whole prompt) and dropped when the last block that covers prompt rows has been hashed
        src = self._req_src.get(rid)
        if src is None:
            pe, voice = _prompt_embed_source(request)
            if pe is not None:
                pe = pe.detach().contiguous().cpu()
            src = self._req_src[rid] = [pe, voice, int(pe.shape[0]) if pe is not None else 0,
                                        _has_reply_rows(request), _unhashed_conditioning(request)]
        pe, voice, rows, salted, private = src
        plen = max(rows, len(request.prompt_token_ids))
        parent = hs[-1] if hs else b"\0" * 16
        for i in range(len(hs), num_blocks):
            lo, hi = i * b, (i + 1) * b
            extra = b""
            if lo < rows:   # the block covers prompt rows given as embeddings: their bytes ARE the content
                extra = pe[lo:min(hi, rows)].view(torch.uint8).numpy().tobytes() + voice
            if private or (hi > plen and salted):
                extra += b"|" + rid.encode()
            if self.prefix_cache_extra_keys is not None:   # a model class's own statement of what else the KV depends on
                more = self.prefix_cache_extra_keys(request, lo, hi)
                extra += b"|" + (rid.encode() if more is None else bytes(more))
            parent = block_content_hash(parent, _token_slice(request, lo, hi), extra)
            hs.append(parent)
        if pe is not None and len(hs) * b >= rows:
            src[0] = None   # every block with prompt rows is hashed: the host copy of the rows is no longer needed
        return hs

    def get_computed_blocks(self, request) -> tuple["KVCacheBlocks", int]:
        """Longest cached prefix of the request: ``(blocks, num_computed_tokens)`` (upstream's
        ``KVCacheManager.get_computed_blocks``, called for a request with no computed tokens yet).  At least the last
        token is always recomputed — the step needs its logits (upstream: ``max_cache_hit_length = num_tokens - 1``)."""
        if not self.enable_caching:
            return KVCacheBlocks(([],)), 0
        max_hit = int(request.num_tokens) - 1
        n = max(max_hit, 0) // self.block_size
        if n == 0:
            return KVCacheBlocks(([],)), 0
        hashes = self._block_hashes(request, n)
        ids = self.block_pool.cache_lookup(hashes)
        self.prefix_cache_requests += 1
        self.prefix_cache_queries += int(request.num_tokens)
        self.prefix_cache_hits += len(ids) * self.block_size
        return KVCacheBlocks((ids,)), len(ids) * self.block_size

    def cache_blocks(self, request, num_computed_tokens: int) -> None:
        """Index the request's full blocks up to ``num_computed_tokens`` (upstream's ``cache_blocks``)."""
        if not self.enable_caching:
            return
        rid = request.request_id
        blocks = self.req_to_blocks.get(rid) or []
        n_full = min(int(num_computed_tokens) // self.block_size, len(blocks))
        done = self._num_cached_blocks.get(rid, 0)
        if n_full <= done:
            return
        hashes = self._block_hashes(request, n_full)
        for i in range(done, n_full):
            self.block_pool.cache_insert(blocks[i], hashes[i])
        self._num_cached_blocks[rid] = n_full
        if self._events_on:
            self._events.append(BlockStored(list(blocks[done:n_full]), rid))

    # -- allocation -------------------------------------------------------------
    def allocate_slots(self, request, num_new_tokens: int, num_new_computed_tokens: int = 0,
                       new_computed_blocks: Optional["KVCacheBlocks"] = None, num_lookahead_tokens: int = 0,
                       num_external_computed_tokens: int = 0, delay_cache_blocks: bool = False,
                       num_encoder_tokens: int = 0) -> Optional[KVCacheBlocks]:
        """Make room for ``num_new_tokens`` (+ lookahead) more tokens of ``request`` — upstream's signature
        (``Scheduler.schedule`` calls it with ``num_new_computed_tokens`` / ``new_computed_blocks`` for a request
        that just hit the prefix cache; the reference's generation scheduler with ``num_lookahead_tokens=``,
        omni_generation_scheduler.py:100-110,167-177).

        Returns the newly allocated blocks (possibly empty; cache-hit blocks are not repeated — ``get_block_ids``
        has the whole row) or ``None`` when the pool is exhausted: callers break out and fall back
        (omni_generation_scheduler.py:105-110).  All-or-nothing: a failed call leaves the request's blocks unchanged."""
        if num_new_tokens < 0 or num_lookahead_tokens < 0 or num_new_computed_tokens < 0:
            raise ValueError("token counts must be non-negative")
        if num_external_computed_tokens or num_encoder_tokens:
            raise ValueError("external / encoder tokens are not on the talker path")
        rid = request.request_id
        have = self.req_to_blocks.get(rid)
        hit = list(new_computed_blocks.blocks[0]) if new_computed_blocks is not None else []
        if hit and have:
            raise ValueError("prefix-cache hits are only attached to a request without blocks")
        computed = int(request.num_computed_tokens) + num_new_computed_tokens
        total_tokens = computed + num_new_tokens + num_lookahead_tokens
        if self.max_model_len is not None:
            total_tokens = min(total_tokens, self.max_model_len)
        need = cdiv(total_tokens, self.block_size) - (len(have) if have else 0) - len(hit)
        if need > 0:
            # a hit on a block that currently sits on the free list takes it OUT of the free list: it cannot also
            # serve this allocation (upstream: `num_evictable_computed_blocks`)
            revived = sum(1 for b in hit if self.block_pool.ref_cnt(b) == 0)
            if need > self.block_pool.get_num_free_blocks() - revived:
                return None
        self.block_pool.touch(hit)
        new = self.block_pool.get_new_blocks(max(need, 0))
        if new is None:   # cannot happen after the check above; keep the all-or-nothing contract regardless
            self.block_pool.free_blocks(list(reversed(hit)))
            return None
        if have is None:
            have = self.req_to_blocks[rid] = []
        have.extend(hit)
        have.extend(new)
        if hit:
            self._num_cached_blocks[rid] = len(hit)
        if self._new_block_ids is not None:
            self._new_block_ids.extend(new)
        if self._events_on:
            if self.enable_caching:   # upstream's events: a block is "stored" when indexed, "removed" when evicted
                for b in self.block_pool.take_evicted():
                    self._events.append(BlockRemoved([b], rid))
            elif new:                 # without the index: allocation / release of the blocks themselves
                self._events.append(BlockStored(new[:], rid))
        elif need > 0:
            self._drain_evictions()   # (upstream's scheduler never calls new_step_starts: drain here too)
        if not delay_cache_blocks:
            self.cache_blocks(request, min(computed + num_new_tokens, int(request.num_tokens)))
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
        rid = request.request_id
        blocks = self.req_to_blocks.pop(rid, None)
        self._num_cached_blocks.pop(rid, None)
        # a FINISHED request never comes back: its hash chain goes with it (upstream keeps hashes on the Request and
        # never calls free_block_hashes — a reused request id must not meet a stale chain); a preempted request
        # (status not finished) keeps its chain and re-hits its own blocks
        if _is_finished(request):
            self._req_hashes.pop(rid, None)
            self._req_src.pop(rid, None)
        if blocks:
            self.block_pool.free_blocks(list(reversed(blocks)))
            if self._events_on and not self.enable_caching:
                self._events.append(BlockRemoved(blocks, rid))

    def free_block_hashes(self, request) -> None:
        """Forget the request's hash chain (upstream calls this when the request leaves the scheduler for good; a
        preempted request keeps it and re-hits its own blocks)."""
        self._req_hashes.pop(request.request_id, None)
        self._req_src.pop(request.request_id, None)

    def get_num_common_prefix_blocks(self, running_request_id: str) -> list[int]:
        """Leading blocks of ``running_request_id`` shared by EVERY request that holds blocks (upstream: blocks whose
        ``ref_cnt`` equals the number of requests; the reference forwards the result as
        ``num_common_prefix_blocks``, omni_generation_scheduler.py:207-210)."""
        blocks = self.req_to_blocks.get(running_request_id)
        if not blocks:
            return [0]
        n_req = len(self.req_to_blocks)
        n = 0
        for b in blocks:
            if self.block_pool.ref_cnt(b) != n_req:
                break
            n += 1
        return [n]

    def take_new_block_ids(self) -> list[int]:
        if self._new_block_ids is None:
            return []
        out, self._new_block_ids = self._new_block_ids, []
        return out

    def take_events(self) -> list[Any]:
        out, self._events = self._events, []
        return out

    def reset_prefix_cache(self) -> bool:
        """Drop every cached-block entry; False (nothing dropped) while any block is still in use — upstream's
        contract."""
        if not self.block_pool.reset_cache():
            return False
        self._req_hashes.clear()
        self._req_src.clear()
        self._num_cached_blocks.clear()
        self._prefix_reset = True
        return True

    def make_prefix_cache_stats(self) -> "PrefixCacheStats":
        """Counters since the last call, in the shape upstream's ``SchedulerStats.prefix_cache_stats`` consumers read
        (``.requests`` / ``.queries`` / ``.hits`` / ``.reset``; ``stats["queries"]`` keeps working)."""
        out = PrefixCacheStats(requests=self.prefix_cache_requests, queries=self.prefix_cache_queries,
                               hits=self.prefix_cache_hits, reset=self._prefix_reset)
        self.prefix_cache_queries = self.prefix_cache_hits = self.prefix_cache_requests = 0
        self._prefix_reset = False
        return out

    # -- upstream's small helpers the scheduler calls on its manager ----------------
    @property
    def empty_kv_cache_blocks(self) -> KVCacheBlocks:
        return KVCacheBlocks(([],))

    def create_kv_cache_blocks(self, blocks: tuple[list[int], ...]) -> KVCacheBlocks:
        return KVCacheBlocks(tuple(list(b) for b in blocks)) if any(blocks) else self.empty_kv_cache_blocks


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


def get_process_gpu_memory(local_rank: int) -> Optional[int]:
    """Device memory held by THIS process on its GPU, from the driver's per-process accounting — the role NVML's
    ``nvmlDeviceGetComputeRunningProcesses`` plays in the reference (vllm_omni/worker/gpu_memory_utils.py:69-124;
    ``usedGpuMemory`` of the entry whose pid is ours), here through ``amdsmi``.  With several workers sharing one card
    (``gpu_memory_utilization`` split between stages) the allocator's own ``memory_reserved`` cannot see the neighbours;
    the driver's figure per pid can.  ``local_rank`` indexes the VISIBLE devices (``HIP_VISIBLE_DEVICES`` /
    ``CUDA_VISIBLE_DEVICES``, then ``ROCR_VISIBLE_DEVICES`` integer lists are honoured; the worker additionally never
    takes a figure below the allocator's own ``memory_reserved``).  Returns None when amdsmi is missing, not permitted or
    reports nothing for this pid — the caller then takes the profiling branch (base.py:134-154)."""
    import os

    try:
        import amdsmi
    except Exception:  # noqa: BLE001 - not installed
        return None
    try:
        amdsmi.amdsmi_init()
    except Exception:  # noqa: BLE001 - no driver access in this container
        return None
    try:
        handles = amdsmi.amdsmi_get_processor_handles()
        index = local_rank
        # HIP_/CUDA_VISIBLE_DEVICES index the devices ROCR_VISIBLE_DEVICES leaves visible, which index the physical ones
        def through(idx: int, var: str) -> Optional[int]:
            vis = os.environ.get(var)
            if not vis:
                return idx
            ids = [v.strip() for v in vis.split(",") if v.strip()]
            if idx < len(ids) and ids[idx].isdigit():
                return int(ids[idx])
            return None   # UUID-style entries or an index past the list: no trustworthy mapping

        index = through(index, "HIP_VISIBLE_DEVICES" if os.environ.get("HIP_VISIBLE_DEVICES") else "CUDA_VISIBLE_DEVICES")
        if index is not None:
            index = through(index, "ROCR_VISIBLE_DEVICES")
        if index is None:
            return None   # profiling branch
        if not 0 <= index < len(handles):
            raise RuntimeError(f"Invalid GPU device {index}. Only {len(handles)} GPU(s) available. "
                               "Check HIP_VISIBLE_DEVICES or stage config 'devices' setting.")
        me = os.getpid()
        for proc in amdsmi.amdsmi_get_gpu_process_list(handles[index]):
            if int(proc.get("pid", -1)) == me:
                usage = proc.get("memory_usage") or {}
                vram = usage.get("vram_mem")
                return int(vram if vram else proc.get("mem", 0))
        return None     # the driver lists no entry for this pid (not permitted to see it): profiling branch
    except RuntimeError:
        raise
    except Exception:  # noqa: BLE001 - any amdsmi failure: profiling branch
        return None
    finally:
        try:
            amdsmi.amdsmi_shut_down()
        except Exception:  # noqa: BLE001
            pass


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

"""``scheduler_cls`` for an AR stage on MI355X: the scheduler that owns THIS package's KV block manager.

The reference reaches the KV block allocator only through its scheduler — ``OmniARScheduler(VLLMScheduler)``
(vllm_omni/core/sched/omni_ar_scheduler.py:41), ``schedule()`` -> ``super().schedule()`` ->
``self.kv_cache_manager.allocate_slots(...)`` (:183-193), ``get_block_ids`` (:628), ``free`` (:528),
``take_events`` (:480) — and picks the scheduler class by name per stage
(``scheduler_cls:`` in vllm_omni/platforms/rocm/stage_configs/qwen2_5_omni.yaml:14,42).  ``MI355XARScheduler`` is the
class to name there:

  * with vllm + vllm_omni importable it IS an ``OmniARScheduler`` whose ``kv_cache_manager`` has been replaced,
    right after upstream's constructor ran, by ``vllm_omni_amd.kv_cache.KVCacheManager`` built from the same
    ``kv_cache_config`` / ``max_model_len`` / ``enable_caching`` arguments upstream passes its own manager —
    everything else (queues, budgets, stop checks, KV-transfer bookkeeping) stays the reference's code;
  * without them (this build container, the GPU test boxes) it is ``StandaloneARScheduler`` below: a restatement
    of the scheduling loop the reference inherits (waiting / running queues, token budget, chunked prefill,
    preemption of the newest running request when the pool is exhausted, prefix-cache hits for new requests) plus
    the reference's own additions (KV-transfer triggers, deferred free while an extraction is in flight,
    ``finished_requests_needing_kv_transfer`` on the scheduler output), producing this package's
    ``SchedulerOutput`` — the type ``GPUARModelRunner.execute_model`` consumes.

Same pattern as ``platform.py``: real base classes when the reference stack is installed, a local stand-in with the
same surface otherwise, so the class is importable, nameable from a stage YAML and testable in both worlds.
"""
from __future__ import annotations

import time
from collections import deque
from typing import Any, Iterable, Optional, Union

from .compat import _get
from .kv_cache import KVCacheManager
from .sched_output import (CachedRequestData, EngineCoreOutput, EngineCoreOutputs, NewRequestData, Request,
                           RequestStatus, SchedulerOutput)

SCHEDULER_QUALNAME = "vllm_omni_amd.scheduler.MI355XARScheduler"


def check_stop(request: Request, max_model_len: int) -> bool:
    """Stop rule applied after each appended token (upstream's ``check_stop``, reached from
    ``_update_request_with_output``, omni_ar_scheduler.py:349-350): EOS unless ``ignore_eos``, then the request's
    ``stop_token_ids`` (the talker's codec end token, qwen2_5_omni.yaml:59), then the length caps."""
    sp = request.sampling_params
    last = request.output_token_ids[-1]
    if not getattr(sp, "ignore_eos", False) and request.eos_token_id is not None and last == request.eos_token_id:
        request.status = RequestStatus.FINISHED_STOPPED
        return True
    if last in (getattr(sp, "stop_token_ids", None) or ()):
        request.status = RequestStatus.FINISHED_STOPPED
        request.stop_reason = last
        return True
    if request.num_tokens >= max_model_len or request.num_output_tokens >= request.max_tokens:
        request.status = RequestStatus.FINISHED_LENGTH_CAPPED
        return True
    return False


class StandaloneARScheduler:
    """Continuous-batching scheduler over ``KVCacheManager`` with the reference's omni additions.

    Constructor arguments follow upstream's ``Scheduler(vllm_config, kv_cache_config, structured_output_manager,
    block_size, ...)``; only duck-typed fields are read: ``scheduler_config.{max_num_seqs, max_num_batched_tokens,
    enable_chunked_prefill, long_prefill_token_threshold}``, ``cache_config.{block_size, enable_prefix_caching}``,
    ``model_config.{max_model_len, omni_kv_config}``, ``kv_cache_config.num_blocks``."""

    def __init__(self, vllm_config: Any = None, kv_cache_config: Any = None, structured_output_manager: Any = None,
                 block_size: Optional[int] = None, mm_registry: Any = None, include_finished_set: bool = False,
                 log_stats: bool = False, *, num_blocks: Optional[int] = None, **_ignored):
        sc, cc, mc = _get(vllm_config, "scheduler_config"), _get(vllm_config, "cache_config"), _get(vllm_config, "model_config")
        self.vllm_config = vllm_config
        self.scheduler_config, self.cache_config = sc, cc
        self.max_num_running_reqs = int(_get(sc, "max_num_seqs", 256))
        self.max_num_scheduled_tokens = int(_get(sc, "max_num_batched_tokens", 8192))
        self.enable_chunked_prefill = bool(_get(sc, "enable_chunked_prefill", True))
        self.long_prefill_token_threshold = int(_get(sc, "long_prefill_token_threshold", 0) or 0)
        self.max_model_len = int(_get(mc, "max_model_len", None) or _get(sc, "max_model_len", 4096))
        self.block_size = int(block_size or _get(cc, "block_size", 16))
        nb = num_blocks if num_blocks is not None else _get(kv_cache_config, "num_blocks", None) or _get(cc, "num_gpu_blocks", None)
        if nb is None:
            raise ValueError("the scheduler needs kv_cache_config.num_blocks (or num_blocks=)")
        self.log_stats = log_stats
        self.kv_cache_manager = KVCacheManager(
            int(nb), self.block_size, enable_caching=bool(_get(cc, "enable_prefix_caching", False)),
            enable_kv_cache_events=bool(_get(_get(vllm_config, "kv_events_config"), "enable_kv_cache_events", False)),
            max_model_len=self.max_model_len, log_stats=log_stats)
        self.requests: dict[str, Request] = {}
        self.waiting: deque[Request] = deque()
        self.running: list[Request] = []
        self.finished_req_ids: set[str] = set()
        self.finished_req_ids_dict: Optional[dict[int, set[str]]] = {} if include_finished_set else None
        # --- the reference's additions (omni_ar_scheduler.py:49-77)
        self.requests_needing_kv_transfer: dict[str, dict[str, Any]] = {}   # rid -> {"seq_len", "block_ids"}
        self.waiting_for_transfer_free: set[str] = set()                    # finished, blocks held for the extraction
        self.active_kv_transfers: set[str] = set()                          # handed to the runner, not acked yet
        self.pending_stop_after_extraction: set[str] = set()
        self.transfer_triggered_requests: set[str] = set()
        self.kv_transfer_criteria = _get(_get(mc, "omni_kv_config"), "kv_transfer_criteria", None)

    # ------------------------------------------------------------------ queue interface
    def add_request(self, request: Request) -> None:
        request.status = RequestStatus.WAITING
        if not request.arrival_time:
            request.arrival_time = time.monotonic()
        self.requests[request.request_id] = request
        self.waiting.append(request)

    def finish_requests(self, request_ids: Union[str, Iterable[str]], finished_status: RequestStatus) -> None:
        """Abort / stop from outside the step loop (client disconnect, stop strings found by the front end)."""
        assert RequestStatus.is_finished(finished_status)
        ids = [request_ids] if isinstance(request_ids, str) else list(request_ids)
        for rid in ids:
            req = self.requests.get(rid)
            if req is None or req.is_finished():
                continue
            if req.status == RequestStatus.RUNNING:
                self.running.remove(req)
            else:
                try:
                    self.waiting.remove(req)
                except ValueError:
                    pass
            req.status = finished_status
            self._free_request(req)

    def get_num_unfinished_requests(self) -> int:
        return len(self.waiting) + len(self.running)

    def _has_pending_kv_transfers(self) -> bool:
        return bool(self.requests_needing_kv_transfer or self.active_kv_transfers or self.waiting_for_transfer_free)

    def has_unfinished_requests(self) -> bool:
        # the engine loop must keep stepping until every extraction has been acked (omni_ar_scheduler.py:692-700)
        return self._has_pending_kv_transfers() or self.get_num_unfinished_requests() > 0

    def has_finished_requests(self) -> bool:
        return self._has_pending_kv_transfers() or bool(self.finished_req_ids)

    def has_requests(self) -> bool:
        return self.has_unfinished_requests() or self.has_finished_requests()

    def get_request_counts(self) -> tuple[int, int]:
        return len(self.running), len(self.waiting)

    def reset_prefix_cache(self) -> bool:
        return self.kv_cache_manager.reset_prefix_cache()

    def make_stats(self) -> Optional[dict]:
        if not self.log_stats:
            return None
        return {"num_running_reqs": len(self.running), "num_waiting_reqs": len(self.waiting),
                "kv_cache_usage": self.kv_cache_manager.usage,
                "prefix_cache_stats": self.kv_cache_manager.make_prefix_cache_stats()}

    def get_grammar_bitmask(self, scheduler_output):   # structured output is not on the talker path
        return None

    def update_draft_token_ids(self, draft_token_ids) -> None:
        return None

    def get_kv_connector(self):
        return None

    def shutdown(self) -> None:
        return None

    # ------------------------------------------------------------------------- schedule
    def _preempt(self, request: Request) -> None:
        self.kv_cache_manager.free(request)
        request.status = RequestStatus.PREEMPTED
        request.num_computed_tokens = 0
        request.num_preemptions += 1
        self.waiting.appendleft(request)

    def schedule(self) -> SchedulerOutput:
        """One scheduling step.  No separate prefill / decode phases: every request has ``num_computed_tokens`` and
        ``num_tokens``; each step hands out tokens so the former catches up with the latter, running requests first
        (a pool that cannot serve one preempts the NEWEST running request and retries), then waiting requests while
        budget, batch rows and blocks last (a new request first takes its longest cached prefix)."""
        kv = self.kv_cache_manager
        kv.new_step_starts()
        budget = self.max_num_scheduled_tokens
        num_scheduled: dict[str, int] = {}
        new_blocks_of: dict[str, Optional[tuple[list[int], ...]]] = {}
        scheduled_new: list[Request] = []
        scheduled_resumed: list[Request] = []
        scheduled_running: list[Request] = []
        preempted: list[Request] = []

        i = 0
        while i < len(self.running) and budget > 0:
            req = self.running[i]
            n = req.num_tokens - req.num_computed_tokens
            if 0 < self.long_prefill_token_threshold < n:
                n = self.long_prefill_token_threshold
            n = min(n, budget, self.max_model_len - 1 - req.num_computed_tokens)
            if n <= 0:
                i += 1
                continue
            blocks = None
            while True:
                blocks = kv.allocate_slots(req, n)
                if blocks is not None:
                    break
                victim = self.running.pop()          # FCFS: the newest running request has the lowest priority
                self._preempt(victim)
                preempted.append(victim)
                if victim in scheduled_running:      # it had been given tokens earlier in this very step
                    scheduled_running.remove(victim)
                    budget += num_scheduled.pop(victim.request_id)
                    new_blocks_of.pop(victim.request_id, None)
                if victim is req:
                    break
            if blocks is None:
                break                                # `req` itself was preempted: nothing behind it can run either
            scheduled_running.append(req)
            num_scheduled[req.request_id] = n
            new_blocks_of[req.request_id] = blocks.get_block_ids() if blocks.blocks[0] else None
            budget -= n
            i += 1

        if not preempted:
            while self.waiting and budget > 0 and len(self.running) < self.max_num_running_reqs:
                req = self.waiting[0]
                hit, n_hit = None, 0
                if req.num_computed_tokens == 0:
                    hit, n_hit = kv.get_computed_blocks(req)
                computed = req.num_computed_tokens + n_hit
                n = req.num_tokens - computed
                if 0 < self.long_prefill_token_threshold < n:
                    n = self.long_prefill_token_threshold
                if n > budget:
                    if not self.enable_chunked_prefill:
                        break
                    n = budget
                if n <= 0:
                    break
                blocks = kv.allocate_slots(req, n, num_new_computed_tokens=n_hit, new_computed_blocks=hit)
                if blocks is None:
                    break                            # the pool is full: the request keeps waiting
                self.waiting.popleft()
                self.running.append(req)
                (scheduled_resumed if req.status == RequestStatus.PREEMPTED else scheduled_new).append(req)
                req.status = RequestStatus.RUNNING
                req.num_computed_tokens = computed
                if req.num_cached_tokens < 0:
                    req.num_cached_tokens = n_hit
                num_scheduled[req.request_id] = n
                new_blocks_of[req.request_id] = kv.get_block_ids(req.request_id)
                budget -= n

        total = sum(num_scheduled.values())
        assert total <= self.max_num_scheduled_tokens and len(self.running) <= self.max_num_running_reqs
        new_data = [NewRequestData(r.request_id, list(r.prompt_token_ids), r.sampling_params,
                                   kv.get_block_ids(r.request_id), r.num_computed_tokens, prompt_embeds=r.prompt_embeds,
                                   additional_information=r.additional_information) for r in scheduled_new]
        cached = CachedRequestData()
        for r in scheduled_running + scheduled_resumed:
            rid = r.request_id
            cached.req_ids.append(rid)
            cached.new_token_ids.append([])
            cached.new_block_ids.append(new_blocks_of.get(rid))
            cached.num_computed_tokens.append(r.num_computed_tokens)
            cached.num_output_tokens.append(r.num_output_tokens)
        cached.resumed_req_ids = {r.request_id for r in scheduled_resumed}
        common = [0]
        if self.running:
            common = kv.get_num_common_prefix_blocks(self.running[0].request_id)
        out = SchedulerOutput(scheduled_new_reqs=new_data, scheduled_cached_reqs=cached,
                              num_scheduled_tokens=num_scheduled, total_num_scheduled_tokens=total,
                              finished_req_ids=self.finished_req_ids, num_common_prefix_blocks=common,
                              finished_requests_needing_kv_transfer=self.get_finished_requests_needing_kv_transfer())
        self.finished_req_ids = set()
        # upstream's `_update_after_schedule`: the tokens count as computed as soon as they are scheduled, so the next
        # `schedule()` may run before this step's output is back
        for rid, n in num_scheduled.items():
            self.requests[rid].num_computed_tokens += n
        return out

    # --------------------------------------------------------------- update_from_output
    def _should_transfer_kv_for_request(self, req_id: str) -> bool:
        cfg = _get(_get(self.vllm_config, "model_config"), "omni_kv_config")
        if not _get(cfg, "need_send_cache", False):
            return False
        req = self.requests.get(req_id)
        info = getattr(req, "additional_information", None) if req is not None else None
        # a request that finalises at stage 0 (text only) never feeds the next stage (omni_ar_scheduler.py:91-112)
        return not (isinstance(info, dict) and info.get("omni_final_stage_id") == 0)

    def _mark_request_for_kv_transfer(self, req_id: str, seq_len: int) -> None:
        """Queue ``{"seq_len", "block_ids"}`` for the runner, ids truncated to ceil(seq_len / block_size)
        (omni_ar_scheduler.py:607-660)."""
        if req_id in self.requests_needing_kv_transfer or not self._should_transfer_kv_for_request(req_id):
            return
        ids = list(self.kv_cache_manager.get_block_ids(req_id)[0])
        n = -(-int(seq_len) // self.block_size)
        self.requests_needing_kv_transfer[req_id] = {"seq_len": int(seq_len), "block_ids": ids[:n]}

    def get_finished_requests_needing_kv_transfer(self) -> dict[str, dict]:
        """Hand the queued extractions to the runner (they ride on the scheduler output) and mark them active."""
        out = dict(self.requests_needing_kv_transfer)
        self.active_kv_transfers.update(out)
        self.requests_needing_kv_transfer.clear()
        return out

    def _process_kv_transfer_trigger(self, request: Request, new_token_ids: list[int]) -> bool:
        """``kv_transfer_criteria`` of the stage config (omni_ar_scheduler.py:114-181): queue an extraction when the
        prompt is done (``prefill_finished``) or a given token was sampled (``special_token``); with
        ``stop_after_transfer`` the request is stopped one step AFTER the runner has acked the extraction, so the
        ``kv_ready`` signal goes out while the request is still alive.  True = stop the request now."""
        crit = self.kv_transfer_criteria
        rid = request.request_id
        if not crit or rid in self.waiting_for_transfer_free or not self._should_transfer_kv_for_request(rid):
            return False
        if rid in self.transfer_triggered_requests:
            if rid in self.pending_stop_after_extraction and rid not in self.active_kv_transfers:
                self.pending_stop_after_extraction.discard(rid)
                request.status = RequestStatus.FINISHED_STOPPED
                return True
            return False
        kind = _get(crit, "type")
        snapshot = None
        if kind == "prefill_finished" and request.num_computed_tokens >= request.num_prompt_tokens:
            snapshot = request.num_computed_tokens
        elif kind == "special_token" and _get(crit, "token_id") is not None and _get(crit, "token_id") in new_token_ids:
            after = len(new_token_ids) - (new_token_ids.index(_get(crit, "token_id")) + 1)
            snapshot = request.num_computed_tokens - after
        if snapshot is None:
            return False
        self.transfer_triggered_requests.add(rid)
        self._mark_request_for_kv_transfer(rid, snapshot)
        if _get(crit, "stop_after_transfer", True) and rid in self.requests_needing_kv_transfer:
            self.pending_stop_after_extraction.add(rid)
        return False

    def _free_request(self, request: Request) -> Optional[dict]:
        """A finished request leaves the scheduler; its blocks go back to the pool unless its KV still has to be
        extracted for the next stage — then they are held until the runner names the request in
        ``kv_extracted_req_ids`` (omni_ar_scheduler.py:541-605).  Returns the ``kv_transfer_params`` to report."""
        assert request.is_finished()
        rid = request.request_id
        self.finished_req_ids.add(rid)
        if self.finished_req_ids_dict is not None:
            self.finished_req_ids_dict.setdefault(request.client_index, set()).add(rid)
        params = None
        if self._should_transfer_kv_for_request(rid):
            if rid in self.transfer_triggered_requests:
                if rid in self.active_kv_transfers:
                    self.waiting_for_transfer_free.add(rid)     # extraction still in flight
                    return None
                if rid in self.waiting_for_transfer_free:
                    return None
                # triggered earlier and already acked: nothing holds the blocks
            else:
                self.waiting_for_transfer_free.add(rid)
                self._mark_request_for_kv_transfer(rid, request.num_computed_tokens)
                data = self.requests_needing_kv_transfer.get(rid)
                if data is not None:
                    params = {"past_key_values": data["block_ids"],
                              "kv_metadata": {"seq_len": data["seq_len"], "block_ids": data["block_ids"]}}
                    if request.additional_information is None:
                        request.additional_information = {}
                    if isinstance(request.additional_information, dict):
                        request.additional_information.update(params)
                return params
        self._free_blocks(request)
        return params

    def _free_blocks(self, request: Request) -> None:
        self.kv_cache_manager.free(request)
        self.kv_cache_manager.free_block_hashes(request)
        self.requests.pop(request.request_id, None)
        self.transfer_triggered_requests.discard(request.request_id)
        self.active_kv_transfers.discard(request.request_id)
        self.pending_stop_after_extraction.discard(request.request_id)

    def update_from_output(self, scheduler_output: SchedulerOutput, model_runner_output) -> dict[int, EngineCoreOutputs]:
        """Fold one step's runner output back into the request states (omni_ar_scheduler.py:238-540): append the
        sampled tokens, apply the stop rules and the KV-transfer triggers, free what finished, release blocks whose
        extraction the runner has acked.  Returns ``{client_index: EngineCoreOutputs}``."""
        outputs: dict[int, list[EngineCoreOutput]] = {}
        sampled = model_runner_output.sampled_token_ids
        pooled = model_runner_output.pooler_output
        extracted = list(getattr(model_runner_output, "kv_extracted_req_ids", None) or [])
        # acks first: the per-request loop below must see up-to-date `active_kv_transfers`, and `kv_ready` has to be
        # emitted while the request is still alive
        for rid in extracted:
            self.active_kv_transfers.discard(rid)
            req = self.requests.get(rid)
            if req is not None and not req.is_finished():
                outputs.setdefault(req.client_index, []).append(
                    EngineCoreOutput(rid, [], kv_transfer_params={"kv_ready": True}))
        stopped_running: list[Request] = []
        for rid, n_sched in scheduler_output.num_scheduled_tokens.items():
            assert n_sched > 0
            req = self.requests.get(rid)
            if req is None or req.is_finished():
                continue      # aborted while the step was running
            idx = model_runner_output.req_id_to_index.get(rid)
            if idx is None:
                continue
            new_ids = list(sampled[idx]) if sampled else []
            pooler = pooled[idx] if pooled else None
            stopped = False
            for k, tok in enumerate(new_ids):
                req.output_token_ids.append(int(tok))
                if check_stop(req, self.max_model_len):
                    new_ids = new_ids[:k + 1]
                    stopped = True
                    break
            if not stopped and self._process_kv_transfer_trigger(req, new_ids):
                stopped = True
            finish_reason, params = None, None
            if stopped:
                finish_reason = req.get_finished_reason()
                params = self._free_request(req)
                stopped_running.append(req)
            if new_ids or pooler is not None or params or stopped:
                outputs.setdefault(req.client_index, []).append(EngineCoreOutput(
                    rid, new_ids, finish_reason=finish_reason, stop_reason=req.stop_reason, pooling_output=pooler,
                    kv_transfer_params=params, num_cached_tokens=max(req.num_cached_tokens, 0)))
        if stopped_running:
            gone = set(map(id, stopped_running))
            self.running = [r for r in self.running if id(r) not in gone]
        for req in stopped_running:
            if req.request_id not in self.waiting_for_transfer_free:
                self.transfer_triggered_requests.discard(req.request_id)
                self.active_kv_transfers.discard(req.request_id)
                self.pending_stop_after_extraction.discard(req.request_id)
        self.kv_events = self.kv_cache_manager.take_events()
        # blocks held for an extraction the runner has now acked
        for rid in extracted:
            if rid in self.waiting_for_transfer_free:
                req = self.requests.get(rid)
                if req is not None:
                    self._free_blocks(req)
                self.waiting_for_transfer_free.discard(rid)
        result = {ci: EngineCoreOutputs(outputs=outs) for ci, outs in outputs.items()}
        if self.finished_req_ids_dict:
            for ci, done in self.finished_req_ids_dict.items():
                result.setdefault(ci, EngineCoreOutputs()).finished_requests = set(done)
            self.finished_req_ids_dict.clear()
        stats = self.make_stats()
        if stats is not None:
            if not result:
                result[0] = EngineCoreOutputs()
            next(iter(result.values())).scheduler_stats = stats
        return result


try:  # the reference's scheduler, when the reference stack is installed
    from vllm_omni.core.sched.omni_ar_scheduler import OmniARScheduler as _ReferenceARScheduler  # type: ignore

    HAVE_REFERENCE_STACK = True
except Exception:  # noqa: BLE001 - any import problem means "not installed here"
    _ReferenceARScheduler = None
    HAVE_REFERENCE_STACK = False


if HAVE_REFERENCE_STACK:

    class MI355XARScheduler(_ReferenceARScheduler):  # type: ignore[misc,valid-type]
        """``OmniARScheduler`` with this package's KV block manager underneath."""

        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            up = self.kv_cache_manager
            kv_cache_config = kwargs.get("kv_cache_config", args[1] if len(args) > 1 else None)
            self.kv_cache_manager = KVCacheManager(
                kv_cache_config=kv_cache_config, max_model_len=self.max_model_len,
                hash_block_size=getattr(self, "block_size", None),
                enable_caching=bool(getattr(up, "enable_caching", self.cache_config.enable_prefix_caching)),
                use_eagle=getattr(self, "use_eagle", False), log_stats=getattr(self, "log_stats", False),
                enable_kv_cache_events=getattr(up, "enable_kv_cache_events", False),
                dcp_world_size=getattr(self, "dcp_world_size", 1), pcp_world_size=getattr(self, "pcp_world_size", 1))

else:

    class MI355XARScheduler(StandaloneARScheduler):
        """Stand-in used where vllm / vllm_omni are not importable: same ``scheduler_cls`` name, same method surface."""

"""CPU: the ``scheduler_cls`` of this package (vllm_omni_amd/scheduler.py) driving its own KV block manager —
schedule -> allocate -> update -> free, chunked prefill, preemption, prefix-cache reuse, and the reference's
KV-transfer bookkeeping (vllm_omni/core/sched/omni_ar_scheduler.py:183-236,238-540,541-660).

The reference holds no scheduler-level test vectors (tests/core/sched covers the chunk coordinator only), so the
pool geometry is the one its own allocator-facing tests use: NUM_BLOCKS 10 / BLOCK_SIZE 4
(tests/core/test_prefix_cache.py:8-13) and block 8 / seq_len 20 -> 3 blocks for the hand-off truncation
(tests/distributed/omni_connectors/test_kv_flow.py:63,98-141)."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from vllm_omni_amd.sched_output import OmniModelRunnerOutput, Request, RequestStatus, SamplingParams

NB, BS = 10, 4


def _cfg(nb=NB, bs=BS, max_seqs=4, budget=64, caching=False, max_len=64, kv=None, chunked=True):
    return NS(scheduler_config=NS(max_num_seqs=max_seqs, max_num_batched_tokens=budget, enable_chunked_prefill=chunked,
                                  long_prefill_token_threshold=0),
              cache_config=NS(block_size=bs, enable_prefix_caching=caching),
              model_config=NS(max_model_len=max_len, omni_kv_config=kv)), NS(num_blocks=nb)


def _sched(**kw):
    from vllm_omni_amd.scheduler import MI355XARScheduler

    vc, kvc = _cfg(**kw)
    return MI355XARScheduler(vc, kvc, None, vc.cache_config.block_size)


def _runner_output(so, token=7, extracted=None):
    """What the runner returns for a step: one token per request whose prompt is finished after it."""
    ids = list(so.num_scheduled_tokens)
    return OmniModelRunnerOutput(ids, {r: i for i, r in enumerate(ids)}, [[token] for _ in ids], [None] * len(ids),
                                 kv_extracted_req_ids=extracted)


def _step(s, token=7, partial=(), extracted=None):
    so = s.schedule()
    out = _runner_output(so, token, extracted)
    for r in partial:
        out.sampled_token_ids[out.req_id_to_index[r]] = []
    return so, s.update_from_output(so, out)


def test_class_is_nameable_from_a_stage_yaml():
    import importlib

    from vllm_omni_amd.scheduler import SCHEDULER_QUALNAME

    mod, cls = SCHEDULER_QUALNAME.rsplit(".", 1)
    c = getattr(importlib.import_module(mod), cls)
    for m in ("schedule", "update_from_output", "add_request", "finish_requests", "has_requests",
              "has_unfinished_requests", "has_finished_requests", "get_num_unfinished_requests", "reset_prefix_cache",
              "get_finished_requests_needing_kv_transfer", "get_request_counts", "make_stats", "shutdown"):
        assert callable(getattr(c, m)), m


def test_schedule_allocate_decode_free():
    s = _sched()
    kv = s.kv_cache_manager
    r = Request("a", list(range(9)), SamplingParams(max_tokens=4))
    s.add_request(r)
    so = s.schedule()
    assert so.total_num_scheduled_tokens == 9 and so.num_scheduled_tokens == {"a": 9}
    nr = so.scheduled_new_reqs[0]
    assert nr.req_id == "a" and nr.block_ids == ([1, 2, 3],) and nr.num_computed_tokens == 0   # ceil(9 / 4), never block 0
    assert r.status == RequestStatus.RUNNING and r.num_computed_tokens == 9                    # advanced at schedule time
    outs = s.update_from_output(so, _runner_output(so, 11))
    assert outs[0].outputs[0].new_token_ids == [11] and outs[0].outputs[0].finish_reason is None
    new_blocks = []
    for step in range(3):
        so = s.schedule()
        c = so.scheduled_cached_reqs
        assert c.req_ids == ["a"] and c.num_computed_tokens == [9 + step] and so.num_scheduled_tokens == {"a": 1}
        new_blocks.append(c.new_block_ids[0])
        outs = s.update_from_output(so, _runner_output(so, 12 + step))
    # tokens 9, 10, 11: slot 9..11 live in block 3 (index 2) -> nothing new until token 12 opens the 4th block
    assert new_blocks == [None, None, None]
    assert outs[0].outputs[0].finish_reason == "length" and r.status == RequestStatus.FINISHED_LENGTH_CAPPED
    assert r.output_token_ids == [11, 12, 13, 14]
    assert kv.block_pool.get_num_free_blocks() == NB - 1 and "a" not in kv.req_to_blocks and not s.running
    so = s.schedule()
    assert so.finished_req_ids == {"a"} and so.total_num_scheduled_tokens == 0
    assert not s.has_requests()


def test_block_growth_and_stop_token():
    s = _sched()
    r = Request("a", list(range(8)), SamplingParams(max_tokens=16, stop_token_ids=[8294]))
    s.add_request(r)
    _step(s, 5)
    so = s.schedule()                       # token 8 -> position 8 needs the third block
    assert so.scheduled_cached_reqs.new_block_ids == [([3],)]
    outs = s.update_from_output(so, _runner_output(so, 8294))
    o = outs[0].outputs[0]
    assert o.finish_reason == "stop" and o.stop_reason == 8294 and r.status == RequestStatus.FINISHED_STOPPED
    e = Request("e", [1, 2, 3], SamplingParams(max_tokens=16), eos_token_id=2)
    s.add_request(e)
    _, outs = _step(s, 2)
    assert outs[0].outputs[0].finish_reason == "stop" and outs[0].outputs[0].stop_reason is None


def test_chunked_prefill_respects_the_token_budget():
    s = _sched(nb=40, budget=8, max_len=128)
    a, b = Request("a", list(range(13)), SamplingParams(max_tokens=2)), Request("b", list(range(5)), SamplingParams(max_tokens=2))
    s.add_request(a)
    s.add_request(b)
    so = s.schedule()
    assert so.num_scheduled_tokens == {"a": 8}                       # the budget is spent on the head of the queue
    s.update_from_output(so, _runner_output(so))                      # (runner discards the partial chunk's sample)
    a.output_token_ids.clear()
    so, _ = _step(s, partial=("a",))
    assert so.num_scheduled_tokens == {"a": 5, "b": 3}               # a's tail, then b chunked into what is left
    assert so.scheduled_new_reqs[0].req_id == "b" and so.scheduled_cached_reqs.req_ids == ["a"]
    s2 = _sched(nb=40, budget=8, max_len=128, chunked=False)
    s2.add_request(Request("c", list(range(13)), SamplingParams(max_tokens=2)))
    assert s2.schedule().total_num_scheduled_tokens == 0             # without chunking a 13-token prompt never fits 8


def test_preemption_of_the_newest_request_and_resume():
    s = _sched(nb=7, budget=64)                                       # 6 usable blocks of 4
    a, b = Request("a", list(range(12)), SamplingParams(max_tokens=8)), Request("b", list(range(12)), SamplingParams(max_tokens=8))
    s.add_request(a)
    s.add_request(b)
    so, _ = _step(s)
    assert set(so.num_scheduled_tokens) == {"a", "b"} and s.kv_cache_manager.block_pool.get_num_free_blocks() == 0
    so = s.schedule()                                                 # both need a 4th block for position 12; none left
    assert so.num_scheduled_tokens == {"a": 1}
    assert b.status == RequestStatus.PREEMPTED and b.num_computed_tokens == 0 and list(s.waiting) == [b]
    assert "b" not in s.kv_cache_manager.req_to_blocks
    s.update_from_output(so, _runner_output(so))
    s.finish_requests("a", RequestStatus.FINISHED_ABORTED)
    so = s.schedule()
    assert so.finished_req_ids == {"a"}
    c = so.scheduled_cached_reqs
    assert c.req_ids == ["b"] and c.resumed_req_ids == {"b"} and c.num_computed_tokens == [0]
    assert so.num_scheduled_tokens == {"b": 13}                       # prompt + the token it had generated
    assert c.new_block_ids[0] == s.kv_cache_manager.get_block_ids("b") and len(c.new_block_ids[0][0]) == 4


def test_prefix_cache_reuse_lru_eviction_and_revival():
    s = _sched(nb=NB, caching=True, budget=64)
    kv, pool = s.kv_cache_manager, s.kv_cache_manager.block_pool
    shared = list(range(100, 108))                                    # two full blocks
    a = Request("a", shared + [1, 2, 3], SamplingParams(max_tokens=1))
    s.add_request(a)
    so, _ = _step(s)
    assert so.scheduled_new_reqs[0].block_ids == ([1, 2, 3],) and a.num_cached_tokens == 0
    assert pool.num_cached() == 2 and pool.get_num_free_blocks() == NB - 1   # a finished: blocks free but still indexed
    b = Request("b", shared + [9, 9], SamplingParams(max_tokens=2))
    s.add_request(b)
    so = s.schedule()
    nr = so.scheduled_new_reqs[0]
    # slots 0..7 <-> blocks 1, 2 re-used; only the tail is new work (core/prefix_cache.py:255-264 reads block_table[req, :8 // 4])
    assert nr.block_ids[0][:2] == [1, 2] and nr.num_computed_tokens == 8 and so.num_scheduled_tokens == {"b": 2}
    assert b.num_cached_tokens == 8 and pool.ref_cnt(1) == 1 and pool.ref_cnt(2) == 1
    c = Request("c", shared + [5], SamplingParams(max_tokens=2))
    s.add_request(c)
    so2 = s.schedule()
    assert so2.scheduled_new_reqs[0].block_ids[0][:2] == [1, 2] and pool.ref_cnt(1) == 2
    assert so2.num_common_prefix_blocks == [2]                       # blocks 1, 2 are shared by every request
    # a prompt that is exactly the cached blocks still recomputes its last token (the step needs its logits)
    d = Request("d", shared, SamplingParams(max_tokens=1))
    s.add_request(d)
    so3 = s.schedule()
    assert so3.scheduled_new_reqs[0].num_computed_tokens == 4 and so3.num_scheduled_tokens["d"] == 4
    for r in (b, c, d):
        s.finish_requests(r.request_id, RequestStatus.FINISHED_ABORTED)
    assert pool.get_num_free_blocks() == NB - 1
    # eviction: allocating the whole pool pops every free block, the cached ones last-freed last
    big = Request("big", list(range(200, 200 + 4 * (NB - 1) - 1)), SamplingParams(max_tokens=1))
    s.add_request(big)
    so = s.schedule()
    assert so.scheduled_new_reqs[0].num_computed_tokens == 0 and pool.get_num_free_blocks() == 0
    assert kv.get_computed_blocks(Request("z", shared + [1], SamplingParams()))[1] == 0   # evicted: no longer a hit
    s.finish_requests("big", RequestStatus.FINISHED_ABORTED)
    assert s.reset_prefix_cache() and pool.num_cached() == 0


def test_prefix_hash_covers_prompt_embeds_and_reply_rows():
    from vllm_omni_amd.kv_cache import KVCacheManager

    kv = KVCacheManager(32, BS, enable_caching=True)
    ids = [8293] + [8292] * 10 + [8294]                              # every talker prompt has the same placeholder ids
    e1, e2 = torch.randn(12, 8), torch.randn(12, 8)
    r1 = Request("r1", ids, SamplingParams(), prompt_embeds=e1)
    assert kv.allocate_slots(r1, 12) is not None
    same = Request("r2", ids, SamplingParams(), prompt_embeds=e1.clone())
    other = Request("r3", ids, SamplingParams(), prompt_embeds=e2)
    assert kv.get_computed_blocks(same)[1] == 8 and kv.get_computed_blocks(other)[1] == 0
    # generated positions of a request with thinker_reply_part depend on per-request rows: never shared across requests
    info = {"thinker_reply_part": torch.randn(4, 8)}
    g1 = Request("g1", ids, SamplingParams(), prompt_embeds=e1, additional_information=info, output_token_ids=[5] * 9)
    g1.num_computed_tokens = 0
    kvb = KVCacheManager(32, BS, enable_caching=True)
    assert kvb.allocate_slots(g1, 21) is not None
    g2 = Request("g2", ids, SamplingParams(), prompt_embeds=e1, additional_information=info, output_token_ids=[5] * 9)
    assert kvb.get_computed_blocks(g2)[1] == 12                       # the prompt blocks only
    assert kvb.get_computed_blocks(g1)[1] == 20                       # the request itself (recompute after preemption)


def test_upstream_constructor_and_call_signatures():
    from vllm_omni_amd.kv_cache import KVCacheBlocks, KVCacheManager

    kvc = NS(num_blocks=NB, kv_cache_groups=[NS(kv_cache_spec=NS(block_size=BS))])
    kv = KVCacheManager(kvc, 64, BS, enable_caching=True, use_eagle=False, log_stats=True, enable_kv_cache_events=True,
                        dcp_world_size=1, pcp_world_size=1, metrics_collector=None)
    assert kv.num_blocks == NB and kv.block_size == BS and kv.max_model_len == 64 and kv.enable_caching
    r = Request("x", list(range(9)), SamplingParams())
    hit, n = kv.get_computed_blocks(r)
    assert isinstance(hit, KVCacheBlocks) and n == 0
    got = kv.allocate_slots(r, 9, 0, hit, num_lookahead_tokens=0, num_external_computed_tokens=0, delay_cache_blocks=False)
    assert got.get_block_ids() == ([1, 2, 3],) and kv.get_block_ids("x") == ([1, 2, 3],)
    ev = kv.take_events()
    assert ev and ev[0].block_ids == [1, 2]                           # the two FULL blocks were indexed
    # exhaustion -> None, nothing changed (omni_generation_scheduler.py:105-110 falls back on None)
    big = Request("y", list(range(40)), SamplingParams())
    assert kv.allocate_slots(big, 40) is None and "y" not in kv.req_to_blocks
    assert kv.block_pool.get_num_free_blocks() == NB - 1 - 3
    with pytest.raises(ValueError):
        KVCacheManager(NS(num_blocks=4, kv_cache_groups=[1, 2]), 64)


def test_kv_transfer_holds_blocks_until_the_runner_acks():
    """need_send_cache: a finished request's blocks stay allocated until the runner lists it in kv_extracted_req_ids;
    block ids are truncated to ceil(seq_len / block_size) — block 8, seq_len 20 -> 3 blocks (test_kv_flow.py:98-141)."""
    s = _sched(nb=12, bs=8, kv=NS(need_send_cache=True, kv_transfer_criteria=None), max_len=64)
    pool = s.kv_cache_manager.block_pool
    r = Request("a", list(range(19)), SamplingParams(max_tokens=1))
    s.add_request(r)
    so, outs = _step(s, 3)
    o = outs[0].outputs[0]
    assert o.finish_reason == "length" and o.kv_transfer_params["kv_metadata"]["seq_len"] == 19
    assert o.kv_transfer_params["past_key_values"] == [1, 2, 3]
    assert pool.get_num_free_blocks() == 11 - 3 and s.waiting_for_transfer_free == {"a"} and s.has_unfinished_requests()
    so = s.schedule()                                                 # the extraction rides on the next scheduler output
    assert so.finished_requests_needing_kv_transfer == {"a": {"seq_len": 19, "block_ids": [1, 2, 3]}}
    assert s.active_kv_transfers == {"a"} and pool.get_num_free_blocks() == 8
    s.update_from_output(so, _runner_output(so, extracted=["a"]))
    assert pool.get_num_free_blocks() == 11 and not s.waiting_for_transfer_free and not s.active_kv_transfers
    assert "a" not in s.requests and not s.has_unfinished_requests()
    # truncation: 4 blocks allocated (lookahead of a 25-token request), snapshot at seq_len 20 -> ceil(20 / 8) = 3
    s.add_request(Request("b", list(range(25)), SamplingParams(max_tokens=8)))
    s.schedule()
    s._mark_request_for_kv_transfer("b", 20)
    assert len(s.kv_cache_manager.get_block_ids("b")[0]) == 4
    assert len(s.requests_needing_kv_transfer["b"]["block_ids"]) == 3


def test_prefill_finished_trigger_stops_after_the_ack():
    crit = {"type": "prefill_finished", "stop_after_transfer": True}
    s = _sched(nb=12, bs=8, kv={"need_send_cache": True, "kv_transfer_criteria": crit})
    r = Request("a", list(range(10)), SamplingParams(max_tokens=50))
    s.add_request(r)
    so, outs = _step(s)                                               # prompt done -> extraction queued, request lives on
    assert "a" in s.transfer_triggered_requests and "a" in s.pending_stop_after_extraction and not r.is_finished()
    so = s.schedule()
    assert "a" in so.finished_requests_needing_kv_transfer and so.num_scheduled_tokens == {"a": 1}
    outs = s.update_from_output(so, _runner_output(so, extracted=["a"]))
    kinds = [o.kv_transfer_params for o in outs[0].outputs]
    assert {"kv_ready": True} in kinds                                # emitted while the request is alive
    assert r.status == RequestStatus.FINISHED_STOPPED                 # ... and stopped in the same update, after the ack
    assert s.kv_cache_manager.block_pool.get_num_free_blocks() == 11


def test_random_workload_keeps_the_pool_consistent():
    rng = np.random.default_rng(0)
    s = _sched(nb=24, bs=4, max_seqs=6, budget=24, caching=True, max_len=80)
    kv, pool = s.kv_cache_manager, s.kv_cache_manager.block_pool
    prefixes = [list(rng.integers(0, 50, size=8)) for _ in range(3)]
    n_added = 0
    for step in range(400):
        if rng.random() < 0.3 and n_added < 60:
            p = prefixes[int(rng.integers(0, 3))] + list(rng.integers(0, 50, size=int(rng.integers(1, 12))))
            s.add_request(Request(f"r{n_added}", [int(x) for x in p], SamplingParams(max_tokens=int(rng.integers(1, 9)))))
            n_added += 1
        so = s.schedule()
        assert so.total_num_scheduled_tokens <= 24 and len(s.running) <= 6
        partial = [r for r, n in so.num_scheduled_tokens.items()
                   if s.requests[r].num_computed_tokens < s.requests[r].num_tokens]
        out = _runner_output(so, int(rng.integers(0, 50)))
        for r in partial:
            out.sampled_token_ids[out.req_id_to_index[r]] = []
        s.update_from_output(so, out)
        if rng.random() < 0.05 and s.running:
            s.finish_requests(s.running[int(rng.integers(0, len(s.running)))].request_id, RequestStatus.FINISHED_ABORTED)
        # invariants: a block's refcount == number of requests holding it; free + held == usable; rows never hold block 0
        held: dict[int, int] = {}
        for rid, blocks in kv.req_to_blocks.items():
            assert 0 not in blocks and len(set(blocks)) == len(blocks)
            assert len(blocks) >= -(-s.requests[rid].num_computed_tokens // 4)
            for b in blocks:
                held[b] = held.get(b, 0) + 1
        for b in range(1, 24):
            assert pool.ref_cnt(b) == held.get(b, 0), (step, b)
        assert pool.get_num_free_blocks() + len(held) == 23
    for r in list(s.requests.values()):
        s.finish_requests(r.request_id, RequestStatus.FINISHED_ABORTED)
    assert pool.get_num_free_blocks() == 23 and n_added == 60


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the prefix-cache hash fails closed on conditioning it does not understand
def test_prefix_hash_fails_closed_on_unknown_per_request_conditioning():
    """Equal placeholder ids + different ``additional_information`` under a key the hasher does not account for (a
    custom model_cls hook's rows, Qwen3-style thinker_prefill_embeddings): 0 hit tokens — the blocks are salted with the
    request id, never shared, yet the request re-hits its OWN blocks after a preemption.  Known keys keep sharing."""
    from vllm_omni_amd.kv_cache import KVCacheManager

    kv = KVCacheManager(32, BS, enable_caching=True)
    ids = [7] * 9
    a = Request("a", ids, SamplingParams(), additional_information={"my_rows": torch.randn(9, 8)})
    assert kv.allocate_slots(a, 9) is not None
    b = Request("b", ids, SamplingParams(), additional_information={"my_rows": torch.randn(9, 8)})
    assert kv.get_computed_blocks(b) == (kv.empty_kv_cache_blocks, 0) or kv.get_computed_blocks(b)[1] == 0
    c = Request("c", ids, SamplingParams(), additional_information={"thinker_prefill_embeddings": torch.randn(9, 8)})
    assert kv.get_computed_blocks(c)[1] == 0
    assert kv.get_computed_blocks(a)[1] == 8                      # the request itself (recompute after a preemption)
    plain = Request("p", ids, SamplingParams())                   # no conditioning at all: ids ARE the content
    assert kv.get_computed_blocks(plain)[1] == 0                  # ... and a's salted blocks are not served to it either
    assert kv.allocate_slots(plain, 9) is not None
    assert kv.get_computed_blocks(Request("q", ids, SamplingParams()))[1] == 8
    # known keys still share: prompt rows by content
    rows = torch.randn(9, 8)
    k1 = Request("k1", ids, SamplingParams(), additional_information={"thinker_result": rows[8:], "prompt_embeds": rows[:7],
                                                                       "prompt_token_ids": [1, 2], "voice_type": "Ethan"})
    kv2 = KVCacheManager(32, BS, enable_caching=True)
    assert kv2.allocate_slots(k1, 9) is not None
    k2 = Request("k2", ids, SamplingParams(), additional_information={"thinker_result": rows[8:].clone(),
                                                                       "prompt_embeds": rows[:7].clone(),
                                                                       "prompt_token_ids": [1, 2], "voice_type": "Ethan"})
    assert kv2.get_computed_blocks(k2)[1] == 8
    k3 = Request("k3", ids, SamplingParams(), additional_information={"thinker_result": rows[8:], "prompt_embeds": rows[:7],
                                                                       "voice_type": "Chelsie"})
    assert kv2.get_computed_blocks(k3)[1] == 0                    # another voice: another speaker row


def test_prefix_cache_extra_keys_hook_and_stats_shape():
    from vllm_omni_amd.kv_cache import KVCacheManager, PrefixCacheStats

    kv = KVCacheManager(32, BS, enable_caching=True)
    kv.prefix_cache_extra_keys = lambda req, lo, hi: getattr(req, "lora", b"")     # a model class's statement
    ids = list(range(9))
    x = Request("x", ids, SamplingParams())
    x.lora = b"A"
    assert kv.allocate_slots(x, 9) is not None
    y = Request("y", ids, SamplingParams())
    y.lora = b"B"
    z = Request("z", ids, SamplingParams())
    z.lora = b"A"
    w = Request("w", ids, SamplingParams())
    w.lora = None                                                                   # None: never share
    assert kv.get_computed_blocks(y)[1] == 0 and kv.get_computed_blocks(z)[1] == 8 and kv.get_computed_blocks(w)[1] == 0
    st = kv.make_prefix_cache_stats()
    assert isinstance(st, PrefixCacheStats) and (st.requests, st.queries, st.hits, st.reset) == (3, 27, 8, False)
    assert st["hits"] == 8 and dict(st)["queries"] == 27                            # mapping access keeps working
    assert kv.make_prefix_cache_stats().queries == 0
    assert kv.create_kv_cache_blocks(([1, 2],)).get_block_ids() == ([1, 2],)
    assert kv.create_kv_cache_blocks(([],)).get_block_ids() == ([],) == kv.empty_kv_cache_blocks.get_block_ids()


def test_finished_requests_leave_no_hash_chain_and_evictions_do_not_pile_up():
    """free() of a FINISHED request drops its hash chain and prompt-row copy (upstream never calls free_block_hashes; a
    reused request id must not meet a stale chain); a preempted request keeps its chain.  With caching on and KV events
    off the pool's eviction record is drained every step instead of growing for the life of the server."""
    from vllm_omni_amd.kv_cache import KVCacheManager

    kv = KVCacheManager(6, BS, enable_caching=True)          # 5 usable blocks
    r = Request("same-id", [1] * 9, SamplingParams(), prompt_embeds=torch.randn(9, 4))
    assert kv.allocate_slots(r, 9) is not None and "same-id" in kv._req_hashes
    r.status = RequestStatus.PREEMPTED
    kv.free(r)
    assert "same-id" in kv._req_hashes                        # preempted: re-hits its own blocks
    assert kv.get_computed_blocks(r)[1] == 8
    got = kv.allocate_slots(r, 1, 8, kv.get_computed_blocks(r)[0])
    assert got is not None
    r.status = RequestStatus.FINISHED_STOPPED
    kv.free(r)
    assert "same-id" not in kv._req_hashes and "same-id" not in kv._req_src
    again = Request("same-id", [2] * 9, SamplingParams(), prompt_embeds=torch.randn(9, 4))   # the id comes back
    assert kv.get_computed_blocks(again)[1] == 0
    # churn: every allocation evicts cached blocks; the record stays drained
    for i in range(40):
        q = Request(f"c{i}", [100 + i] * 17, SamplingParams())
        kv.new_step_starts()
        assert kv.allocate_slots(q, 17) is not None
        q.status = RequestStatus.FINISHED_STOPPED
        kv.free(q)
    assert len(kv.block_pool.take_evicted()) <= 5

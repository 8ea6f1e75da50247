"""CPU: host-side logic of the path — block manager façade, block table / slot mapping, persistent
batch, platform plugin surface, head partition rules.  No GPU compute."""
import numpy as np
import pytest
import torch

from oracle import ref_ops


# ------------------------------------------------------------------------- KV block manager (A1)
def _req(rid, n_prompt, computed=0):
    from vllm_omni_amd.sched_output import Request

    r = Request(rid, list(range(n_prompt)))
    r.num_computed_tokens = computed
    return r


def test_allocate_slots_growth_and_exhaustion():
    from vllm_omni_amd.kv_cache import KVCacheBlocks, KVCacheManager

    m = KVCacheManager(num_blocks=8, block_size=16)  # 7 usable blocks
    m.new_step_starts()
    a = _req("a", 40)
    got = m.allocate_slots(a, 40)
    assert isinstance(got, KVCacheBlocks) and got.get_block_ids() == ([1, 2, 3],)   # ceil(40/16) = 3, never block 0
    assert m.get_block_ids("a") == ([1, 2, 3],)
    a.num_computed_tokens = 40
    for step in range(8):                      # tokens 41..48 fit in block 3
        assert m.allocate_slots(a, 1).get_block_ids() == ([],)
        a.num_computed_tokens += 1
    assert m.allocate_slots(a, 1).get_block_ids() == ([4],)   # token 49 opens a new block
    a.num_computed_tokens += 1
    assert m.allocate_slots(a, 1, num_lookahead_tokens=16).get_block_ids() == ([5],)  # lookahead rounds up
    b = _req("b", 64)
    assert m.allocate_slots(b, 64) is None      # needs 4, only 2 free: None, and nothing taken
    assert m.block_pool.get_num_free_blocks() == 2 and m.get_block_ids("b") == ([],)
    assert m.get_num_common_prefix_blocks("a") == [5]
    m.free(a)
    assert m.block_pool.get_num_free_blocks() == 7 and m.get_block_ids("a") == ([],)
    got_b = m.allocate_slots(b, 64)
    assert got_b is not None and len(got_b.blocks[0]) == 4
    assert 0 not in got_b.blocks[0]
    assert 0.0 < m.usage <= 1.0


def test_events_and_new_block_ids_tracking():
    from vllm_omni_amd.kv_cache import BlockRemoved, BlockStored, KVCacheManager

    m = KVCacheManager(16, 4, enable_kv_cache_events=True, track_new_block_ids=True)
    r = _req("x", 9)
    m.allocate_slots(r, 9)
    assert m.take_new_block_ids() == [1, 2, 3] and m.take_new_block_ids() == []
    m.free(r)
    ev = m.take_events()
    assert isinstance(ev[0], BlockStored) and isinstance(ev[1], BlockRemoved) and m.take_events() == []


def test_truncation_rule_matches_scheduler():
    """ceil(seq_len / block_size) leading ids (omni_ar_scheduler.py:644-651)."""
    from vllm_omni_amd.kv_cache import KVCacheManager

    m = KVCacheManager(32, 8)
    r = _req("t", 30)
    m.allocate_slots(r, 30, num_lookahead_tokens=8)
    ids = m.get_block_ids("t")[0]
    assert len(ids) == 5
    assert ref_ops.truncate_block_ids(ids, 20, 8) == ids[:3]


def test_kv_budget_and_layouts():
    from vllm_omni_amd.kv_cache import (KVCacheSpec, allocate_kv_cache, determine_available_memory, extract_request_kv,
                                        normalize_layer_kv, num_blocks_for_budget)

    spec = KVCacheSpec(16, 4, 128, torch.bfloat16)
    assert spec.page_bytes == 32768                                   # 32 KiB / block / layer
    assert num_blocks_for_budget(int(270e9 * 0.9), 28, spec) == int(270e9 * 0.9) // 917504   # ~2.6e5 blocks (A1)
    assert determine_available_memory(288 * 2**30, 0.9, 20 * 2**30) == int(288 * 2**30 * 0.9) - 20 * 2**30
    assert determine_available_memory(100, 0.5, 80) == 0
    assert determine_available_memory(100, 0.5, 80, kv_cache_memory_bytes=7) == 7
    small = KVCacheSpec(8, 4, 16, torch.float32)
    for layout in ("NB2", "2NB", "NB2_HEAD_MAJOR"):
        caches = allocate_kv_cache(2, 10, small, "cpu", layout)
        kb, vb = normalize_layer_kv(caches[0])
        assert kb.shape == (10, 8, 4, 16) and vb.shape == kb.shape and kb.stride(3) == 1
        kb.copy_(torch.randn(10, 8, 4, 16))
        vb.copy_(torch.randn(10, 8, 4, 16))
        keys, vals = extract_request_kv(caches, [1, 3, 5], 20)         # the reference's gather constants
        exp = ref_ops.extract_kv(kb, vb, [1, 3, 5], 20)
        assert keys[0].shape == (20, 4, 16) and torch.equal(keys[0], exp[0]) and torch.equal(vals[0], exp[1])
        assert keys[1] is not None
    with pytest.raises(ValueError):
        allocate_kv_cache(1, 4, small, "cpu", "bogus")
    keys, vals = extract_request_kv([torch.randn(3, 8, 4, 16), (torch.randn(3, 8, 4, 16), torch.randn(2, 8, 4, 16))],
                                    [0, 1, 2], 32)
    assert keys[0] is None and keys[1].shape == (16, 4, 16)           # invalid layer skipped; K/V overlap range


# ------------------------------------------------------------------------- block table (A3)
def test_block_table_rows_and_slot_mapping():
    from vllm_omni_amd.block_table import PAD_SLOT_ID, MultiGroupBlockTable

    bt = MultiGroupBlockTable(max_num_reqs=4, max_model_len=64, max_num_batched_tokens=32, block_sizes=[4], device=None)
    bt.add_row(([2, 3],), 0)
    bt.append_row(([9],), 0)
    bt.add_row(([5],), 1)
    t0 = bt[0]
    assert t0.block_table.cpu[0, :3].tolist() == [2, 3, 9] and t0.num_blocks_per_row[0] == 3
    assert len(bt.block_tables) == 1
    # slots 8..15 <-> blocks 2,3 at block_size 4 (tests/core/test_prefix_cache.py:173-185)
    bt.compute_slot_mapping(np.zeros(8, dtype=np.int32), np.arange(8))
    assert t0.slot_mapping.cpu[:8].tolist() == list(range(8, 16))
    req = np.array([0, 0, 1], dtype=np.int32)
    pos = np.array([8, 11, 2])
    bt.compute_slot_mapping(req, pos)
    exp = ref_ops.slot_mapping(t0.block_table.cpu, torch.from_numpy(req), torch.from_numpy(pos), 4)
    assert t0.slot_mapping.cpu[:3].tolist() == exp.tolist() == [36, 39, 22]
    out = t0.commit_slot_mapping(3, 8)
    assert out[3:8].tolist() == [PAD_SLOT_ID] * 5                      # padded tail never writes the cache
    bt.swap_row(0, 1)
    assert t0.block_table.cpu[1, :3].tolist() == [2, 3, 9] and t0.block_table.cpu[0, 0] == 5
    bt.move_row(1, 2)
    assert t0.block_table.cpu[2, :3].tolist() == [2, 3, 9] and t0.num_blocks_per_row[2] == 3
    with pytest.raises(ValueError):
        bt.append_row((list(range(20)),), 3)
    with pytest.raises(Exception):
        bt.compute_slot_mapping(np.zeros(1, dtype=np.int32), np.array([64 * 4]))   # past the table row


def test_input_batch_condense_and_swap():
    from vllm_omni_amd.model_runner import CachedRequestState, InputBatch
    from vllm_omni_amd.sched_output import SamplingParams

    ib = InputBatch(4, 64, 16, 4, None)
    for i, name in enumerate("abc"):
        ib.add_request(CachedRequestState(name, [1, 2, 3 + i], SamplingParams(), ([10 + i],), num_computed_tokens=i))
    assert ib.req_ids == ["a", "b", "c"] and ib.last_token_ids_cpu[:3].tolist() == [3, 4, 5]
    ib.remove_request("a")                                            # last row moves into the hole
    assert ib.req_ids == ["c", "b"] and ib.req_id_to_index == {"c": 0, "b": 1}
    assert ib.num_computed_tokens_cpu[:2].tolist() == [2, 1] and ib.block_table[0].block_table.cpu[0, 0] == 12
    ib.swap(0, 1)
    assert ib.req_ids == ["b", "c"] and ib.block_table[0].block_table.cpu[0, 0] == 11
    ib.remove_request("zzz")                                          # unknown id: no-op
    assert ib.num_reqs == 2


# ------------------------------------------------------------------------- plugin boundary (L0)
def test_platform_plugin_surface():
    import vllm_omni_amd
    from vllm_omni_amd.platform import AR_WORKER_QUALNAME, MI355XOmniPlatform, mi355x_platform_plugin

    needed = ["get_omni_ar_worker_cls", "get_omni_generation_worker_cls", "get_default_stage_config_path",
              "get_diffusion_attn_backend_cls", "supports_torch_inductor", "get_torch_device", "get_device_count",
              "get_device_version", "synchronize", "get_free_memory", "set_device_control_env_var",
              "unset_device_control_env_var", "create_autocast_context", "supports_cpu_offload", "supports_float64",
              "get_profiler_cls", "check_if_supports_dtype"]
    for name in needed:                                              # vllm_omni/platforms/interface.py:26-165
        assert callable(getattr(MI355XOmniPlatform, name)), name
    p = MI355XOmniPlatform()
    assert p.is_rocm() and not p.is_cuda() and p.dist_backend == "nccl" and p.device_type == "cuda"
    assert MI355XOmniPlatform.get_omni_ar_worker_cls() == AR_WORKER_QUALNAME
    mod, cls = AR_WORKER_QUALNAME.rsplit(".", 1)
    assert hasattr(__import__(mod, fromlist=[cls]), cls)             # the named worker class resolves
    assert MI355XOmniPlatform.get_torch_device(3) == torch.device("cuda", 3)
    MI355XOmniPlatform.check_if_supports_dtype(torch.bfloat16)
    with pytest.raises(ValueError):
        MI355XOmniPlatform.check_if_supports_dtype(torch.float64)
    # no gfx950 GPU here -> the plugin declines (returns None) instead of half-activating
    assert vllm_omni_amd.register_platform() is None or torch.cuda.is_available()
    assert mi355x_platform_plugin() in (None, "vllm_omni_amd.platform.MI355XOmniPlatform")


def test_worker_refuses_non_gpu_device():
    from vllm_omni_amd.talker import TalkerConfig
    from vllm_omni_amd.worker import GPUARWorker, WorkerConfig

    w = GPUARWorker(WorkerConfig(TalkerConfig.from_json("mini_talker"), device_type="cpu"))
    with pytest.raises(RuntimeError, match="Not support device type"):    # gpu_ar_worker.py:89-90
        w.init_device()


# ------------------------------------------------------------------------- TP head rules (A12)
def test_partition_heads_rules():
    from vllm_omni_amd.distributed import kv_head_for_rank, partition_heads

    assert partition_heads(28, 4, 1) == (28, 4, 1)
    assert partition_heads(28, 4, 2) == (14, 2, 1)
    assert partition_heads(28, 4, 4) == (7, 1, 1)
    with pytest.raises(ValueError):
        partition_heads(28, 4, 8)                                     # 28 % 8 != 0 (qwen2_old.py:91)
    assert partition_heads(16, 2, 8) == (2, 1, 4)                     # KV replication (qwen2_old.py:98-102)
    assert [list(kv_head_for_rank(2, 8, r)) for r in range(8)] == [[0]] * 4 + [[1]] * 4
    assert [list(kv_head_for_rank(4, 2, r)) for r in range(2)] == [[0, 1], [2, 3]]
    with pytest.raises(ValueError):
        partition_heads(16, 3, 2)


def test_talker_config_shapes():
    from vllm_omni_amd.talker import TalkerConfig, bad_word_mask_spec

    c = TalkerConfig.from_json("qwen2_5_omni_talker")
    assert (c.num_hidden_layers, c.hidden_size, c.num_attention_heads, c.num_key_value_heads, c.head_dim,
            c.intermediate_size, c.vocab_size) == (28, 3584, 28, 4, 128, 18944, 8448)
    assert abs(c.num_params() - 6.57e9) < 0.05e9                      # 6.57 B params
    assert bad_word_mask_spec(8448, 8296, 8294, 8293) == ([(8296, 8448)], 8293)
    assert bad_word_mask_spec(8448, 8294, 8294, 8293) == ([(8295, 8448)], 8293)
    assert bad_word_mask_spec(8448, 8000, 8294, None) == ([(8000, 8294), (8295, 8448)], -1)
    assert bad_word_mask_spec(8448, None, 8294, 8293) == ([], 8293)
    with pytest.raises(ValueError):
        bad_word_mask_spec(8448, 8000, None, 1)


def test_linear_routing_policy_matches_the_measurements():
    """Which GEMM runs which projection of the headline talker (hidden 3584, intermediate 18944, 28/4 heads of 128)
    per (TP, rows per rank) — DESIGN.md 3.5 / profiles/r01_scaling_estimate.md: the streaming kernel takes the MLP at
    TP 1 and 2 for every decode batch, o_proj only when its split-K reduce can ride in the next RMSNorm, nothing at TP 4,
    and no batch above 256 rows."""
    from vllm_omni_amd import layers as L
    H, I = 3584, 18944
    for tp, down, gate_up in [(1, True, True), (2, True, True), (4, False, False), (8, False, False)]:
        for m in (64, 128, 256):
            assert L._streaming_kernel_wins(m, H, I // tp) is down, (tp, m)
        for m in (64, 128, 256):
            assert L._fused_gate_up_wins(m, I // tp, H) is gate_up, (tp, m)
        assert L._fused_gate_up_wins(32, I // tp, H)        # <= 32 rows: one launch instead of GEMM + SiluAndMul
    assert not L._streaming_kernel_wins(512, H, I) and not L._fused_gate_up_wins(512, I, H)
    # qkv_proj / o_proj are not "deep and narrow": the plain policy leaves them on the library ...
    assert not L._streaming_kernel_wins(256, 4608, H) and not L._streaming_kernel_wins(256, H, H)
    assert not L._streaming_kernel_wins(64, 4608, H) and not L._streaming_kernel_wins(64, H, H)
    # ... except at <= 32 rows per rank (the dp8 plan), where the kernel is ahead on every projection of a megaelement
    # or more: the talker's four and the code predictor's 1024 x 2048 / 2048 x 1024 / 6144 x 1024 alike
    for n, k in [(4608, H), (H, H), (H, I // 4), (H, I // 8), (2048, 1024), (1024, 2048), (1024, 3072)]:
        assert L._streaming_kernel_wins(32, n, k) and L._streaming_kernel_wins(1, n, k), (n, k)
    assert not L._streaming_kernel_wins(32, 512, 1024)
    # the code predictor's small MLP (6144 x 1024, 1024 x 3072): the fused pair up to 256 rows, not at 512
    for m in (32, 64, 128, 256):
        assert L._fused_gate_up_wins(m, 3072, 1024) and L._streaming_kernel_wins(m, 1024, 3072), m
    assert not L._fused_gate_up_wins(512, 3072, 1024) and not L._streaming_kernel_wins(512, 1024, 3072)
    assert not L._streaming_kernel_wins(64, 1024, 2048) and not L._streaming_kernel_wins(64, 4096, 1024)   # its o / qkv
    # ... o_proj joins only with the deferred reduce; the code predictor's 1024 x 2048 o_proj never does
    assert L._deferred_kernel_wins(256, H, H) and L._deferred_kernel_wins(32, H, H)
    assert not L._deferred_kernel_wins(256, 1024, 2048) and not L._deferred_kernel_wins(512, H, H)


# ------------------------------------------------------------------------- deferred free while a KV transfer is active
def test_free_is_deferred_while_kv_transfer_is_active():
    """omni_ar_scheduler.py:559-605: a finishing request whose KV goes to the next stage keeps its blocks until the
    runner reports the extraction; block ids in the transfer snapshot are truncated to ceil(seq_len / b)."""
    from vllm_omni_amd.kv_cache import KVCacheManager

    m = KVCacheManager(num_blocks=16, block_size=4)
    a, b = _req("a", 11), _req("b", 5)
    assert m.allocate_slots(a, 11, num_lookahead_tokens=4) is not None       # 4 blocks (lookahead), 3 hold tokens
    assert m.allocate_slots(b, 5) is not None
    free0 = m.block_pool.get_num_free_blocks()
    snap = m.mark_request_for_kv_transfer("a", seq_len=11)
    assert snap == {"seq_len": 11, "block_ids": m.get_block_ids("a")[0][:3]}
    m.free(a)                                                                # transfer active: nothing returns
    assert m.block_pool.get_num_free_blocks() == free0 and m.get_block_ids("a")[0]
    assert "a" in m.waiting_for_transfer_free
    m.free(b)                                                                # no transfer: immediate
    assert m.block_pool.get_num_free_blocks() == free0 + 2
    assert m.on_kv_extracted(["a", "unknown"]) == ["a"]
    assert m.block_pool.get_num_free_blocks() == free0 + 2 + 4 and m.get_block_ids("a") == ([],)
    assert not m.active_kv_transfers and not m.waiting_for_transfer_free
    # extraction reported BEFORE the request finishes: the later free is immediate
    c = _req("c", 8)
    m.allocate_slots(c, 8)
    m.mark_request_for_kv_transfer("c", 8)
    assert m.on_kv_extracted(["c"]) == []
    n = m.block_pool.get_num_free_blocks()
    m.free(c)
    assert m.block_pool.get_num_free_blocks() == n + 2


# ------------------------------------------------------------------------- thinker_reply_part pool (A11)
def test_reply_row_allocator_first_fit_and_coalescing():
    from vllm_omni_amd.model_runner import ReplyRowAllocator

    a = ReplyRowAllocator(10)
    s0, s1, s2 = a.alloc(3), a.alloc(4), a.alloc(3)
    assert (s0, s1, s2) == (0, 3, 7) and a.num_free == 0
    with pytest.raises(RuntimeError, match="pool is full"):
        a.alloc(1)
    a.release(s1, 4)
    assert a.alloc(2) == 3 and a.free == [(5, 2)]
    a.release(s0, 3)
    a.release(3, 2)
    assert a.free == [(0, 7)]                                                # neighbours merged back
    a.release(s2, 3)
    assert a.free == [(0, 10)]


def test_reply_cursor_survives_unscheduling_and_live_rows_are_never_overwritten():
    """a request that leaves the batch for a step continues its thinker_reply_part where it stopped
    (the reference persists the consumed queue, qwen2_5_omni.py:771-774); a full pool raises instead of wrapping
    over live rows; a finished request's rows are reusable."""
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, SamplingParams, SchedulerOutput
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("mini_talker")
    r = GPUARModelRunner(cfg, torch.device("cpu"), torch.bfloat16, max_num_reqs=4, max_model_len=64,
                         max_num_batched_tokens=16, block_size=16, enforce_eager=True, max_reply_rows=8)
    r._reply_pool = torch.zeros(8, cfg.embedding_size, dtype=torch.bfloat16)
    rp = {n: torch.full((k, cfg.embedding_size), float(i + 1)) for i, (n, k) in enumerate([("a", 3), ("b", 4)])}
    new = [NewRequestData(n, [1, 2, 3], SamplingParams(), ([1 + i],), 3, additional_information={"thinker_reply_part": rp[n]})
           for i, n in enumerate(("a", "b"))]
    r._update_states(SchedulerOutput(scheduled_new_reqs=new, num_scheduled_tokens={"a": 1, "b": 1},
                                     total_num_scheduled_tokens=2))
    ib = r.input_batch
    assert ib.reply_offset[:2].tolist() == [0, 3] and ib.reply_end[:2].tolist() == [3, 7]
    assert float(r._reply_pool[3, 0]) == 2.0
    ib.reply_offset[:2] += 2                                                 # two decode steps consumed two rows each
    # step 3: "a" is not scheduled (preempted / waiting) -> leaves the batch, keeps its cursor
    cached = CachedRequestData(req_ids=["b"], new_block_ids=[None], num_computed_tokens=[5], num_output_tokens=[2])
    r._update_states(SchedulerOutput(scheduled_cached_reqs=cached, num_scheduled_tokens={"b": 1}, total_num_scheduled_tokens=1))
    assert ib.req_ids == ["b"] and r.requests["a"].reply_cursor == 2
    # step 4: "a" comes back: same pool rows, cursor where it stopped (NOT row 0 again), nothing re-allocated
    cached = CachedRequestData(req_ids=["a", "b"], new_block_ids=[None, None], num_computed_tokens=[5, 6],
                               num_output_tokens=[2, 3])
    r._update_states(SchedulerOutput(scheduled_cached_reqs=cached, num_scheduled_tokens={"a": 1, "b": 1},
                                     total_num_scheduled_tokens=2))
    row_a = ib.req_id_to_index["a"]
    assert (int(ib.reply_offset[row_a]), int(ib.reply_end[row_a])) == (2, 3) and r._reply_alloc.num_free == 1
    # a third request that does not fit raises; nothing of a / b was touched
    big = NewRequestData("c", [1], SamplingParams(), ([5],), 1,
                         additional_information={"thinker_reply_part": torch.ones(2, cfg.embedding_size)})
    with pytest.raises(RuntimeError, match="pool is full"):
        r._update_states(SchedulerOutput(scheduled_new_reqs=[big], num_scheduled_tokens={"a": 1, "b": 1, "c": 1},
                                         total_num_scheduled_tokens=3))
    assert float(r._reply_pool[0, 0]) == 1.0 and float(r._reply_pool[6, 0]) == 2.0
    # "b" finishes: its 4 rows return to the pool and "c"'s part now fits into them
    r.requests.pop("c", None)
    r.input_batch.remove_request("c")
    cached = CachedRequestData(req_ids=["a"], new_block_ids=[None], num_computed_tokens=[6], num_output_tokens=[3])
    r._update_states(SchedulerOutput(scheduled_new_reqs=[big], scheduled_cached_reqs=cached, finished_req_ids={"b"},
                                     num_scheduled_tokens={"a": 1, "c": 1}, total_num_scheduled_tokens=2))
    assert r.requests["c"].reply_base == 3 and r._reply_alloc.num_free == 8 - 3 - 2


def test_cached_request_update_replaces_the_reply_queue():
    """`scheduled_cached_reqs.additional_information[rid]` (gpu_model_runner.py:1100-1116): a key's new value replaces
    the old one.  For the thinker's reply queue (async chunks arriving while the talker decodes) the request's segment of
    the device pool is swapped for the new rows: old rows freed, cursor back at the first new row."""
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, SamplingParams, SchedulerOutput
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("mini_talker")
    r = GPUARModelRunner(cfg, torch.device("cpu"), torch.bfloat16, max_num_reqs=4, max_model_len=64,
                         max_num_batched_tokens=16, block_size=16, enforce_eager=True, max_reply_rows=12)
    r._reply_pool = torch.zeros(12, cfg.embedding_size, dtype=torch.bfloat16)
    new = [NewRequestData(n, [1, 2, 3], SamplingParams(), ([1 + i],), 3,
                          additional_information={"thinker_reply_part": torch.full((3, cfg.embedding_size), float(i + 1))})
           for i, n in enumerate(("a", "b"))]
    r._update_states(SchedulerOutput(scheduled_new_reqs=new, num_scheduled_tokens={"a": 1, "b": 1}, total_num_scheduled_tokens=2))
    ib = r.input_batch
    ib.reply_offset[:2] += 2                                                   # two decode steps consumed two rows each
    cached = CachedRequestData(req_ids=["a", "b"], new_block_ids=[None, None], num_computed_tokens=[5, 5], num_output_tokens=[2, 2],
                               additional_information={"a": {"thinker_reply_part": torch.full((4, cfg.embedding_size), 9.0),
                                                             "other": 7},
                                                       "gone": {"thinker_reply_part": torch.ones(1, cfg.embedding_size)}})
    r._update_states(SchedulerOutput(scheduled_cached_reqs=cached, num_scheduled_tokens={"a": 1, "b": 1}, total_num_scheduled_tokens=2))
    a = r.requests["a"]
    row_a, row_b = ib.req_id_to_index["a"], ib.req_id_to_index["b"]
    assert a.reply_rows == 4 and a.reply_cursor == 0 and a.additional_information["other"] == 7
    assert int(ib.reply_offset[row_a]) == a.reply_base and int(ib.reply_end[row_a]) == a.reply_base + 4
    assert float(r._reply_pool[a.reply_base, 0]) == 9.0 and float(r._reply_pool[a.reply_base + 3, 0]) == 9.0
    b = r.requests["b"]
    assert (int(ib.reply_offset[row_b]), int(ib.reply_end[row_b])) == (b.reply_base + 2, b.reply_base + 3)   # untouched
    assert float(r._reply_pool[b.reply_base, 0]) == 2.0
    assert r._reply_alloc.num_free == 12 - 4 - 3                                # a's old three rows went back to the pool


def test_graph_batch_size_ladder():
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.talker import TalkerConfig

    r = GPUARModelRunner(TalkerConfig.from_json("mini_talker"), torch.device("cpu"), max_num_reqs=250, max_model_len=64)
    assert [r.graph_batch_size(n) for n in (1, 2, 3, 4, 5, 8, 9, 64, 249, 250)] == [1, 2, 4, 4, 8, 8, 16, 64, 250, 250]
    assert len({r.graph_batch_size(n) for n in range(1, 251)}) <= 250 // 8 + 4


@pytest.mark.parametrize("seed", range(6))
def test_block_manager_random_operations_keep_their_invariants(seed):
    """Random allocate / grow / finish / KV-hand-off sequences over the C++ pool against a plain-Python model:
    no block is ever owned twice, block 0 (the null block) is never handed out, a request owns exactly
    ceil(tokens / block_size) blocks in allocation order, a failed allocation changes nothing, free + owned == pool,
    blocks of a request whose KV transfer is active survive `free` until `on_kv_extracted`, and freed blocks come back
    tail first (the order the next allocations see)."""
    import numpy as np

    from vllm_omni_amd.kv_cache import KVCacheManager
    from vllm_omni_amd.sched_output import Request, SamplingParams

    rng = np.random.default_rng(seed)
    bs = int(rng.choice([4, 8, 16]))
    nb = int(rng.integers(8, 40))
    mgr = KVCacheManager(nb, bs, track_new_block_ids=True)
    usable = mgr.block_pool.get_num_free_blocks()
    assert usable == nb - 1                                   # block 0 is the null block
    live: dict[str, Request] = {}
    owned: dict[str, list[int]] = {}                          # the model
    held: dict[str, Request] = {}                             # freed while their transfer is active
    nxt = 0
    for _ in range(400):
        op = rng.random()
        if op < 0.45 and (live or nxt == 0) or not live:
            if rng.random() < 0.4 or not live:                # a new request
                r = Request(f"r{nxt}", [1], SamplingParams())
                nxt += 1
                live[r.request_id], owned[r.request_id] = r, []
            else:
                r = live[str(rng.choice(sorted(live)))]
            n = int(rng.integers(0, 3 * bs))
            look = int(rng.integers(0, 3)) if rng.random() < 0.2 else 0
            free_before = mgr.block_pool.get_num_free_blocks()
            before = list(mgr.get_block_ids(r.request_id)[0])
            got = mgr.allocate_slots(r, n, num_lookahead_tokens=look)
            need = max(0, -(-(r.num_computed_tokens + n + look) // bs) - len(before))
            if got is None:
                assert need > free_before                                      # only exhaustion fails ...
                assert list(mgr.get_block_ids(r.request_id)[0]) == before      # ... and leaves everything as it was
                assert mgr.block_pool.get_num_free_blocks() == free_before
            else:
                new = got.get_block_ids()[0]
                assert len(new) == need and 0 not in new
                owned[r.request_id] += new
                assert list(mgr.get_block_ids(r.request_id)[0]) == owned[r.request_id]
                assert mgr.take_new_block_ids() == new
                r.num_computed_tokens += n                                     # the step ran
        elif op < 0.75:
            rid = str(rng.choice(sorted(live)))
            r = live.pop(rid)
            if rng.random() < 0.4:                            # its KV goes to the next stage first
                info = mgr.mark_request_for_kv_transfer(rid, r.num_computed_tokens)
                assert info["block_ids"] == owned[rid][: -(-r.num_computed_tokens // bs)]
                mgr.free(r)
                assert list(mgr.get_block_ids(rid)[0]) == owned[rid]           # still held
                held[rid] = r
            else:
                free_before = mgr.block_pool.get_num_free_blocks()
                mgr.free(r)
                assert mgr.block_pool.get_num_free_blocks() == free_before + len(owned.pop(rid))
        elif held:
            rid = str(rng.choice(sorted(held)))
            held.pop(rid)
            assert mgr.on_kv_extracted([rid]) == [rid]
            owned.pop(rid)
            assert list(mgr.get_block_ids(rid)[0]) == []
        flat = [b for ids in owned.values() for b in ids]
        assert len(flat) == len(set(flat)) and 0 not in flat                   # nothing owned twice
        assert mgr.block_pool.get_num_free_blocks() + len(flat) == usable      # nothing lost
        assert all(mgr.block_pool.ref_cnt(b) == 1 for b in flat[:8])
    # tail-first reuse: free one request, the next allocation of as many blocks returns them LAST block first
    for rid in list(held):
        mgr.on_kv_extracted([rid])
        owned.pop(rid)
    for r in list(live.values()):
        mgr.free(r)
        owned.pop(r.request_id)
    assert mgr.block_pool.get_num_free_blocks() == usable
    a = Request("tail", [1], SamplingParams())
    ids = mgr.allocate_slots(a, 3 * bs).get_block_ids()[0]
    mgr.free(a)
    drain = mgr.block_pool.get_new_blocks(usable)
    assert drain[-3:] == list(reversed(ids))


def test_decode_rows_are_one_token_steps_behind_the_prompt():
    """Which rows of a step are DECODES (input = last token, decode kernel, graph replay) — `_prepare_inputs` on the
    host: a one-token step inside a prompt of EMBEDDINGS is a prefill row; for a token-id prompt the last prompt token
    alone counts as a decode (same token); a request resumed after a preemption is a prefill row until it is back at the
    frontier of the tokens it has; decodes are moved to the front of the batch."""
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, SamplingParams, SchedulerOutput
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("mini_talker")
    r = GPUARModelRunner(cfg, torch.device("cpu"), torch.bfloat16, max_num_reqs=8, max_model_len=64,
                         max_num_batched_tokens=32, block_size=16, enforce_eager=True)
    emb = torch.zeros(5, cfg.embedding_size)
    new = [
        NewRequestData("emb_mid", [1] * 5, SamplingParams(), ([1],), 2, prompt_embeds=emb),     # 1 token at 2 of 5: prefill
        NewRequestData("emb_last", [1] * 5, SamplingParams(), ([2],), 4, prompt_embeds=emb),    # the LAST prompt row: prefill
        NewRequestData("emb_done", [1] * 5, SamplingParams(), ([3],), 5, prompt_embeds=emb),    # behind the prompt: decode
        NewRequestData("tok_mid", [1] * 5, SamplingParams(), ([4],), 2),                        # inside a token prompt: prefill
        NewRequestData("tok_last", [1] * 5, SamplingParams(), ([5],), 4),                       # last prompt token alone: decode
        NewRequestData("chunk", [1] * 9, SamplingParams(), ([6],), 0),                          # a 4-token chunk
    ]
    sched = {"emb_mid": 1, "emb_last": 1, "emb_done": 1, "tok_mid": 1, "tok_last": 1, "chunk": 4}
    so = SchedulerOutput(scheduled_new_reqs=new, num_scheduled_tokens=sched, total_num_scheduled_tokens=9)
    r._update_states(so)
    n, total, num_sched, max_seq_len, num_decodes = r._prepare_inputs(so)
    ib = r.input_batch
    assert (n, total, num_decodes) == (6, 9, 2)
    assert sorted(ib.req_ids[:2]) == ["emb_done", "tok_last"]                  # decodes first
    assert sorted(ib.req_ids[2:]) == ["chunk", "emb_last", "emb_mid", "tok_mid"]
    assert not r._can_use_graph(5, 5, 2)                                       # one-token rows, but not all decodes

    # a resumed request recomputes: one-token steps behind its prompt are prefill rows until the frontier
    st = r.requests["emb_done"]
    st.output_token_ids += [7, 8, 9]                                           # it had generated three tokens (frontier 8)
    cached = CachedRequestData(req_ids=["emb_done"], resumed_req_ids={"emb_done"}, new_block_ids=[([7],)],
                               num_computed_tokens=[6], num_output_tokens=[3])
    so = SchedulerOutput(scheduled_cached_reqs=cached, num_scheduled_tokens={"emb_done": 1}, total_num_scheduled_tokens=1)
    r._update_states(so)
    assert r._prepare_inputs(so)[4] == 0 and st.recompute_until == 8           # position 6 of 8: still recomputing
    cached.resumed_req_ids, cached.new_block_ids, cached.num_computed_tokens = set(), [None], [7]
    r._update_states(so)
    assert r._prepare_inputs(so)[4] == 1                                       # the last known token: a decode again

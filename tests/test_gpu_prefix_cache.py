"""GPU: prefix caching end to end — scheduler class (``scheduler_cls``) -> KV block manager (hash-based reuse of
full blocks) -> runner (paged KV cache on the device + the Omni tensor prefix cache).

A second request that shares a prompt prefix with a finished one must (a) be scheduled with the first request's
block ids for the shared full blocks and only the tail as new work, (b) attend over KV rows it never computed, and
(c) still hand ALL its hidden rows to the next stage: ``cat([rows of the cached blocks, this step's rows])``
(vllm_omni/worker/gpu_ar_model_runner.py:205-260,846-856; vllm_omni/core/prefix_cache.py:212-264).  The expectation
is the CPU oracle prefilling the second request from scratch into its own fresh cache.

Tolerance: hidden states of the 2-layer mini talker, bf16 activations: rtol 2e-2 / atol 2e-2 as in
tests/test_gpu_talker_step.py (every op boundary rounds to 2^-8 relative)."""
from types import SimpleNamespace as NS

import pytest
import torch

from oracle import ref_ops, ref_talker

pytestmark = pytest.mark.gpu

BS, NB = 16, 48


def _oracle_prefill(cfg, w, embeds, csc):
    """Prefill ``embeds`` [T, E] from scratch on a fresh contiguous cache; returns hidden [T, H], logits of the last row."""
    T = embeds.shape[0]
    nblk = (T + BS - 1) // BS
    caches = [(torch.zeros(nblk + 1, BS, cfg.num_key_value_heads, cfg.head_dim),
               torch.zeros(nblk + 1, BS, cfg.num_key_value_heads, cfg.head_dim)) for _ in range(cfg.num_hidden_layers)]
    table = torch.arange(1, nblk + 1, dtype=torch.int32).view(1, -1)
    pos = torch.arange(T, dtype=torch.int64)
    slots = ref_ops.slot_mapping(table, torch.zeros(T, dtype=torch.int64), pos, BS)
    hidden = ref_talker.talker_step(cfg, w, embeds.float(), pos, caches, table, torch.tensor([T], dtype=torch.int32),
                                    torch.tensor([0, T], dtype=torch.int32), slots, csc, torch.bfloat16)
    logits, ids = ref_talker.logits_and_greedy(cfg, w, hidden[-1:], cfg.tts_codec_mask_token_id, torch.bfloat16)
    return hidden, logits, ids


def _drive(sched, runner, max_steps=32):
    """schedule -> execute_model -> sample_tokens -> update_from_output until nothing is left; returns the per-step
    (scheduler output, runner output) pairs."""
    trace = []
    for _ in range(max_steps):
        if not sched.has_unfinished_requests():
            break
        so = sched.schedule()
        out = runner.execute_model(so)
        if out is None:
            out = runner.sample_tokens()
        sched.update_from_output(so, out)
        trace.append((so, out))
    assert not sched.has_unfinished_requests()
    return trace


@pytest.mark.parametrize("budget", [64, 24], ids=["one-chunk", "chunked"])
def test_shared_prefix_reuses_blocks_and_matches_the_oracle(dev, budget):
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.scheduler import MI355XARScheduler
    from vllm_omni_amd.sched_output import Request, SamplingParams
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("mini_talker")
    cfg.mrope_section = None
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=4, max_model_len=128, max_num_batched_tokens=64,
                              block_size=BS, enforce_eager=True, max_reply_rows=0, seed=0, enable_prefix_caching=True)
    runner.load_model()
    w = ref_talker.random_weights(cfg, seed=3)
    assert runner.model.load_weights({k: v.to(dev) for k, v in w.items()}) == set(w.keys())
    runner.model.set_suppress_start_id(cfg.tts_codec_mask_token_id)
    runner.initialize_kv_cache(NB)
    assert runner.omni_prefix_cache is not None and runner.omni_prefix_cache.hidden_states_cache.shape[:2] == (NB, BS)
    vc = NS(scheduler_config=NS(max_num_seqs=4, max_num_batched_tokens=budget, enable_chunked_prefill=True),
            cache_config=NS(block_size=BS, enable_prefix_caching=True), model_config=NS(max_model_len=128, omni_kv_config=None))
    sched = MI355XARScheduler(vc, NS(num_blocks=NB), None, BS)
    csc = ref_ops.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)

    g = torch.Generator().manual_seed(11)
    shared = torch.randn(2 * BS + 5, cfg.embedding_size, generator=g).to(torch.bfloat16)     # 37 rows: 2 full blocks + 5
    tail_b = torch.randn(9, cfg.embedding_size, generator=g).to(torch.bfloat16)
    emb_a = shared
    emb_b = torch.cat([shared[:2 * BS + 2], tail_b])                                         # diverges inside block 3
    ids = lambda n: [8293] + [8292] * (n - 2) + [8294]   # noqa: E731 - the talker's placeholder prompt ids

    # ---- request A alone: fills the caches
    a = Request("a", ids(emb_a.shape[0]), SamplingParams(temperature=0.0, max_tokens=2), prompt_embeds=emb_a)
    sched.add_request(a)
    trace_a = _drive(sched, runner)
    blocks_a = list(trace_a[0][0].scheduled_new_reqs[0].block_ids[0])
    for so_, _ in trace_a[1:]:                                     # later chunks / decode steps append blocks
        for rid, new in zip(so_.scheduled_cached_reqs.req_ids, so_.scheduled_cached_reqs.new_block_ids):
            if rid == "a" and new is not None:
                blocks_a += new[0]
    rows_a = torch.cat([o.pooler_output[o.req_id_to_index["a"]]["hidden"] for _, o in trace_a if "a" in o.req_id_to_index])
    hid_a, _, ids_a = _oracle_prefill(cfg, w, emb_a, csc)
    torch.testing.assert_close(rows_a[:emb_a.shape[0]].float(), hid_a, rtol=2e-2, atol=2e-2)
    assert sched.kv_cache_manager.block_pool.num_cached() >= 2

    # ---- request B: same first 34 rows -> the two full blocks are a hit, 11 rows of new work
    b = Request("b", ids(emb_b.shape[0]), SamplingParams(temperature=0.0, max_tokens=1), prompt_embeds=emb_b)
    sched.add_request(b)
    so = sched.schedule()
    nr = so.scheduled_new_reqs[0]
    assert nr.req_id == "b" and nr.num_computed_tokens == 2 * BS and b.num_cached_tokens == 2 * BS
    assert nr.block_ids[0][:2] == blocks_a[:2] and nr.block_ids[0][2] != blocks_a[2]
    n_new = emb_b.shape[0] - 2 * BS
    assert so.num_scheduled_tokens == {"b": min(n_new, budget)} and n_new == 11
    assert runner.execute_model(so) is None
    out = runner.sample_tokens()
    sched.update_from_output(so, out)
    got = out.pooler_output[0]["hidden"]
    # all of B's rows: the 32 cached ones (computed by A's steps) + the 11 of this step
    assert got.shape == (emb_b.shape[0], cfg.hidden_size)
    hid_b, logits_b, ids_b = _oracle_prefill(cfg, w, emb_b, csc)
    torch.testing.assert_close(got.float(), hid_b, rtol=2e-2, atol=2e-2)
    assert torch.equal(got[:2 * BS], rows_a[:2 * BS])                 # bit-identical to what A's steps produced
    top2 = logits_b.topk(2, dim=-1).values[0]
    if float(top2[0] - top2[1]) > 0.05:
        assert out.sampled_token_ids[0] == [int(ids_b[0])]
    assert b.is_finished() and not sched.has_unfinished_requests()
    st = sched.kv_cache_manager.make_prefix_cache_stats()
    assert st["hits"] == 2 * BS and st["queries"] == emb_a.shape[0] + emb_b.shape[0]

    # ---- the KV rows B attended over in the shared blocks are A's, untouched; a third request with a DIFFERENT
    # prompt of the same placeholder ids gets no hit (the hash covers the embedding rows)
    c = Request("c", ids(40), SamplingParams(temperature=0.0, max_tokens=1),
                prompt_embeds=torch.randn(40, cfg.embedding_size, generator=g).to(torch.bfloat16))
    sched.add_request(c)
    so = sched.schedule()
    assert so.scheduled_new_reqs[0].num_computed_tokens == 0
    out = runner.execute_model(so) or runner.sample_tokens()
    sched.update_from_output(so, out)
    assert out.pooler_output[0]["hidden"].shape[0] == min(40, budget)


def test_reference_shaped_talker_request_through_scheduler_and_runner(dev):
    """A talker request as the reference's stage input processor hands it over — placeholder prompt ids
    ``[8293] + [8292] * P + [8294]`` and ``additional_information{prompt_embeds, thinker_result, prompt_token_ids,
    thinker_output_token_ids}``, no request-level prompt embeddings (stage_input_processors/qwen2_5_omni.py:6-61) —
    goes scheduler -> runner unchanged: the runner builds the P + 2 prompt rows the model's preprocess hook builds
    (qwen2_5_omni.py:681-764), prefills them, then decodes with one reply row per step.  Hidden states of the prefill
    and of every decode step against the oracle; a second request with the SAME thinker prompt and a different reply
    re-uses the prompt's full blocks (the hash covers the hand-over's rows)."""
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.scheduler import MI355XARScheduler
    from vllm_omni_amd.sched_output import Request, SamplingParams
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("mini_talker")
    cfg.mrope_section = None
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=4, max_model_len=128, max_num_batched_tokens=64,
                              block_size=BS, enforce_eager=True, max_reply_rows=64, seed=0, enable_prefix_caching=True)
    runner.load_model()
    w = ref_talker.random_weights(cfg, seed=3)
    g = torch.Generator().manual_seed(21)
    spk = torch.randn(cfg.embedding_size, generator=g).to(torch.bfloat16)
    table = torch.zeros(151873, cfg.embedding_size, dtype=torch.bfloat16)
    table[151872] = spk
    table[151860] = torch.randn(cfg.embedding_size, generator=g).to(torch.bfloat16)
    runner.model.load_weights([(k, v.to(dev)) for k, v in w.items()] + [("thinker_embedding.weight", table)])
    runner.model.set_suppress_start_id(cfg.tts_codec_mask_token_id)
    runner.initialize_kv_cache(NB)
    vc = NS(scheduler_config=NS(max_num_seqs=4, max_num_batched_tokens=64, enable_chunked_prefill=True),
            cache_config=NS(block_size=BS, enable_prefix_caching=True), model_config=NS(max_model_len=128, omni_kv_config=None))
    sched = MI355XARScheduler(vc, NS(num_blocks=NB), None, BS)
    csc = ref_ops.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)

    P, K, steps = 34, 4, 5
    tp = torch.randn(P, cfg.embedding_size, generator=g)                       # fp32 hidden states from the thinker stage
    ids = [8293] + [8292] * P + [8294]

    def request(rid, reply):
        info = {"prompt_embeds": tp, "thinker_result": reply, "prompt_token_ids": list(range(P)),
                "thinker_output_token_ids": list(range(K))}
        return Request(rid, ids, SamplingParams(temperature=0.0, max_tokens=steps), additional_information=info)

    def oracle(reply, tokens):
        """Prefill of the built prompt, then `len(tokens)` decode steps teacher-forced with the device's tokens."""
        _, rows, rp = ref_talker.thinker_to_talker_prefill(cfg, w, spk, tp.to(torch.bfloat16), reply.to(torch.bfloat16),
                                                          torch.bfloat16)
        nblk = (P + 2 + steps + BS) // BS
        caches = [(torch.zeros(nblk + 1, BS, cfg.num_key_value_heads, cfg.head_dim),
                   torch.zeros(nblk + 1, BS, cfg.num_key_value_heads, cfg.head_dim)) for _ in range(cfg.num_hidden_layers)]
        tbl = torch.arange(1, nblk + 1, dtype=torch.int32).view(1, -1)
        n = rows.shape[0]
        pos = torch.arange(n, dtype=torch.int64)
        hid = [ref_talker.talker_step(cfg, w, rows, pos, caches, tbl, torch.tensor([n], dtype=torch.int32),
                                      torch.tensor([0, n], dtype=torch.int32),
                                      ref_ops.slot_mapping(tbl, torch.zeros(n, dtype=torch.int64), pos, BS), csc, torch.bfloat16)]
        for j, tok in enumerate(tokens):
            vec = rp[j:j + 1].to(torch.bfloat16) if j < rp.shape[0] else None
            emb = ref_talker.decode_embeds(w, torch.tensor([tok]), vec, torch.bfloat16)
            p1 = torch.tensor([n + j], dtype=torch.int64)
            hid.append(ref_talker.talker_step(cfg, w, emb, p1, caches, tbl, torch.tensor([n + j + 1], dtype=torch.int32),
                                              torch.tensor([0, 1], dtype=torch.int32),
                                              ref_ops.slot_mapping(tbl, torch.zeros(1, dtype=torch.int64), p1, BS), csc,
                                              torch.bfloat16))
        return hid

    reply_a = torch.randn(K, cfg.embedding_size, generator=g)
    ra = request("a", reply_a)
    sched.add_request(ra)
    trace = _drive(sched, runner)
    assert ra.is_finished() and len(ra.output_token_ids) == steps
    got = [o.pooler_output[o.req_id_to_index["a"]]["hidden"].float() for _, o in trace if "a" in o.req_id_to_index]
    want = oracle(reply_a, ra.output_token_ids[:-1])
    assert got[0].shape == (P + 2, cfg.hidden_size) and len(got) == len(want) == steps
    for i, (x, y) in enumerate(zip(got, want)):
        torch.testing.assert_close(x, y, rtol=3e-2, atol=3e-2, msg=lambda m: f"step {i}: {m}")

    # same thinker prompt, different reply: blocks 0, 1 (rows 0..31 of the 34 thinker rows) are a hit, the rest is not
    reply_b = torch.randn(K, cfg.embedding_size, generator=g)
    rb = request("b", reply_b)
    sched.add_request(rb)
    so = sched.schedule()
    nr = so.scheduled_new_reqs[0]
    assert nr.num_computed_tokens == 2 * BS and so.num_scheduled_tokens == {"b": P + 2 - 2 * BS}
    out = runner.execute_model(so) or runner.sample_tokens()
    sched.update_from_output(so, out)
    hb = out.pooler_output[0]["hidden"].float()
    assert hb.shape == (P + 2, cfg.hidden_size)                                # cached rows + this step's
    torch.testing.assert_close(hb, oracle(reply_b, [])[0], rtol=3e-2, atol=3e-2)
    assert torch.equal(hb[:2 * BS], got[0][:2 * BS])

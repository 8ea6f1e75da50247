"""GPU parity of the whole decode step through the runner's two-phase API:
execute_model -> sample_tokens, eager and hipGraph replay, vs the CPU oracle step.

Tolerance: hidden states after 2 layers are compared with rtol 2e-2 / atol 2e-2 (bf16 activations:
every op boundary rounds to 2^-8 relative and GEMM accumulation order differs between hipBLASLt
and the oracle); greedy token ids must agree wherever the oracle's top-2 logit margin exceeds the
logit tolerance (multi-step greedy agreement is the telling test).
"""
import numpy as np
import pytest
import torch

from oracle import ref_ops, ref_talker

pytestmark = pytest.mark.gpu


def _setup(dev, cfg_name, ctxs, enforce_eager, mrope, quant=None):
    from vllm_omni_amd.kv_cache import KVCacheManager, normalize_layer_kv
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import Request, SamplingParams
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json(cfg_name)
    if not mrope:
        cfg.mrope_section = None
    B, bs = len(ctxs), 16
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=8, max_model_len=256, max_num_batched_tokens=64,
                              block_size=bs, enforce_eager=enforce_eager, max_reply_rows=64, seed=0, quantization=quant)
    runner.load_model()
    w = ref_talker.random_weights(cfg, seed=3)
    loaded = runner.model.load_weights({k: v.to(dev) for k, v in w.items()})
    assert loaded == set(w.keys())
    nb = 64
    runner.initialize_kv_cache(nb)
    g = torch.Generator().manual_seed(5)
    cpu_caches = []
    for kv in runner.kv_caches:
        init = torch.randn(nb, 2, bs, cfg.num_key_value_heads, cfg.head_dim, generator=g).to(torch.bfloat16)
        kb, vb = normalize_layer_kv(kv)   # whatever physical layout the runner chose
        kb.copy_(init[:, 0].to(dev))
        vb.copy_(init[:, 1].to(dev))
        cpu_caches.append((init[:, 0].clone(), init[:, 1].clone()))
    mgr = KVCacheManager(nb, bs)
    scramble = mgr.block_pool.get_new_blocks(nb - 1)
    mgr.block_pool.free_blocks(np.random.default_rng(0).permutation(scramble).tolist())
    reqs = []
    for i, c in enumerate(ctxs):
        r = Request(f"r{i}", list(range(c + 1)), SamplingParams(temperature=0.0))
        r.additional_information = {"thinker_reply_part": torch.randn(3, cfg.embedding_size, generator=g).to(torch.bfloat16)}
        assert mgr.allocate_slots(r, c + 1) is not None
        r.num_computed_tokens = c
        reqs.append(r)
    return cfg, runner, w, cpu_caches, mgr, reqs


@pytest.mark.parametrize("enforce_eager,quant,linear_mode",
                         [(True, None, "auto"), (False, None, "auto"), (True, "int8", "auto"), (False, "int8", "auto"),
                          (True, None, "kernel"), (False, None, "kernel")])
def test_decode_steps_match_oracle(dev, enforce_eager, quant, linear_mode, monkeypatch):
    """quant="int8": the decoder linears run W8A8 (online-quantised after load_weights) and the
    oracle quantises the same weights / activations with the reference's scheme.  linear_mode="kernel": every 16-bit
    projection runs the decode-batch streaming kernel (csrc/skinny_gemm.hip) instead of the library GEMM."""
    from vllm_omni_amd import layers, ops
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, SchedulerOutput

    monkeypatch.setattr(layers, "_LINEAR_MODE", linear_mode)
    kernel_calls = []
    real_linear = ops.linear
    monkeypatch.setattr(ops, "linear", lambda *a, **k: (kernel_calls.append(1), real_linear(*a, **k))[1])
    real_fused = ops.linear_silu_mul
    monkeypatch.setattr(ops, "linear_silu_mul", lambda *a, **k: (kernel_calls.append(1), real_fused(*a, **k))[1])
    real_part = ops.linear_partials   # o_proj / down_proj (/ qkv_proj) whose split-K reduce rides in the next kernel

    def counted_partials(*a, **k):
        r = real_part(*a, **k)
        if r is not None:
            kernel_calls.append(1)
        return r

    monkeypatch.setattr(ops, "linear_partials", counted_partials)

    ctxs = [5, 16, 40, 1]
    # decode positions are 1-D (all three M-RoPE rows equal at decode => plain RoPE)
    cfg, runner, w, cpu_caches, mgr, reqs = _setup(dev, "mini_talker", ctxs, enforce_eager, False, quant)
    if quant == "int8":
        lin = runner.model.language_model.model.layers[0].mlp.down_proj
        assert lin.weight is None and lin.qweight.dtype == torch.int8 and lin.weight_scale.dtype == torch.float32
    runner.model.set_suppress_start_id(cfg.tts_codec_mask_token_id)
    B, bs = len(reqs), 16
    csc = ref_ops.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
    last_ids = torch.tensor([r.prompt_token_ids[-1] for r in reqs], dtype=torch.int64)
    table = torch.zeros(B, 16, dtype=torch.int32)
    for i, r in enumerate(reqs):
        ids = mgr.get_block_ids(r.request_id)[0]
        table[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    agree = total = 0
    for step in range(3):
        sched = {r.request_id: 1 for r in reqs}
        if step == 0:
            so = SchedulerOutput(scheduled_new_reqs=[
                NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params, mgr.get_block_ids(r.request_id),
                               r.num_computed_tokens, additional_information=r.additional_information) for r in reqs],
                num_scheduled_tokens=sched, total_num_scheduled_tokens=B)
        else:
            new_blocks = []
            for r in reqs:
                nb_ = mgr.allocate_slots(r, 1)
                assert nb_ is not None
                new_blocks.append(nb_.get_block_ids() if nb_.blocks[0] else None)
            for i, r in enumerate(reqs):
                ids = mgr.get_block_ids(r.request_id)[0]
                table[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
            so = SchedulerOutput(scheduled_cached_reqs=CachedRequestData(
                req_ids=[r.request_id for r in reqs], new_block_ids=new_blocks,
                num_computed_tokens=[r.num_computed_tokens for r in reqs], num_output_tokens=[step] * B),
                num_scheduled_tokens=sched, total_num_scheduled_tokens=B)
        assert runner.execute_model(so) is None
        with pytest.raises(RuntimeError, match="State error"):
            runner.execute_model(so)
        out = runner.sample_tokens()
        with pytest.raises(RuntimeError, match="State error"):
            runner.sample_tokens()
        order = [out.req_id_to_index[r.request_id] for r in reqs]
        # ---- oracle for the same step ----
        pos = torch.tensor([r.num_computed_tokens for r in reqs], dtype=torch.int64)
        step_vec = torch.stack([r.additional_information["thinker_reply_part"][step] for r in reqs])
        emb = ref_talker.decode_embeds(w, last_ids, step_vec, torch.bfloat16)
        seq_lens = (pos + 1).to(torch.int32)
        slots = ref_ops.slot_mapping(table, torch.arange(B), pos, bs)
        hidden = ref_talker.talker_step(cfg, w, emb, pos, cpu_caches, table, seq_lens,
                                        torch.arange(B + 1, dtype=torch.int32), slots, csc, torch.bfloat16, quant=quant)
        logits, ids = ref_talker.logits_and_greedy(cfg, w, hidden, cfg.tts_codec_mask_token_id, torch.bfloat16)
        got_hidden = torch.stack([out.pooler_output[j]["hidden"][0] for j in order]).float()
        # int8: a one-ulp difference in a 16-bit activation can move a quantised value by one step (1/127 of the row max)
        tol = 2e-2 if quant is None else 5e-2
        torch.testing.assert_close(got_hidden, hidden, rtol=tol, atol=tol)
        got_ids = torch.tensor([out.sampled_token_ids[j][0] for j in order])
        top2 = logits.topk(2, dim=-1).values
        confident = (top2[:, 0] - top2[:, 1]) > 0.05
        total += int(confident.sum())
        agree += int((got_ids[confident] == ids[confident]).sum())
        # masked columns ([mask_id, V) and the codec BOS column) can never be sampled
        assert not ((got_ids >= cfg.tts_codec_mask_token_id) | (got_ids == cfg.tts_codec_start_token_id)).any()
        # the KV rows written this step match the oracle's cache bit-for-bit in placement
        from vllm_omni_amd.kv_cache import normalize_layer_kv
        for li, kv in enumerate(runner.kv_caches):
            kb, vb = normalize_layer_kv(kv)
            b_idx, off = (slots // bs), (slots % bs)
            torch.testing.assert_close(kb.cpu()[b_idx, off].float(), cpu_caches[li][0][b_idx, off].float(),
                                       rtol=3e-2, atol=3e-2)
        # feed the GPU's tokens to both sides so later steps stay comparable
        last_ids = got_ids.clone()
        for r, t in zip(reqs, got_ids.tolist()):
            r.num_computed_tokens += 1
            r.output_token_ids.append(t)
    assert total == 0 or agree == total, f"greedy ids disagree on {total - agree} of {total} confident rows"
    if linear_mode == "kernel":
        assert len(kernel_calls) >= 4 * cfg.num_hidden_layers   # qkv, o, gate_up (+ SiLU, fused), down of every layer
    elif quant is None:
        # 4 rows: "auto" hands the kernel exactly the projections of a megaelement or more (layers._streaming_kernel_wins)
        big = sum(layers._streaming_kernel_wins(B, *lin.weight.shape)
                  for layer in runner.model.language_model.model.layers
                  for lin in (layer.self_attn.qkv_proj, layer.self_attn.o_proj, layer.mlp.gate_up_proj, layer.mlp.down_proj))
        assert (len(kernel_calls) > 0) == (big > 0)
    else:
        assert not kernel_calls                                 # W8A8 layers never take the 16-bit kernel


def test_finished_requests_are_removed_and_kv_extracted(dev):
    from vllm_omni_amd.sched_output import NewRequestData, SchedulerOutput

    cfg, runner, w, cpu_caches, mgr, reqs = _setup(dev, "mini_talker", [20, 9], True, False)
    so = SchedulerOutput(scheduled_new_reqs=[
        NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params, mgr.get_block_ids(r.request_id),
                       r.num_computed_tokens) for r in reqs],
        num_scheduled_tokens={r.request_id: 1 for r in reqs}, total_num_scheduled_tokens=2)
    runner.execute_model(so)
    runner.sample_tokens()
    blocks = mgr.get_block_ids("r0")[0]
    so2 = SchedulerOutput(finished_req_ids={"r0"},
                          finished_requests_needing_kv_transfer={"r0": {"block_ids": blocks, "seq_len": 21}})
    out = runner.execute_model(so2)
    assert out.kv_extracted_req_ids == ["r0"] and "r0" not in runner.input_batch.req_id_to_index
    got = runner.kv_extracted["r0"]
    assert len(got["key_cache"]) == cfg.num_hidden_layers
    assert got["key_cache"][0].shape == (21, cfg.num_key_value_heads, cfg.head_dim)
    exp = ref_ops.extract_kv(cpu_caches[0][0], cpu_caches[0][1], blocks, 21)
    # rows 0..19 were pre-filled identically on both sides (row 20 was written by the step on the GPU only)
    assert torch.equal(got["key_cache"][0][:20].cpu(), exp[0][:20])
    assert got["metadata"]["seq_len"] == 21 and got["block_ids"] == blocks
    # the packed wire payload of the same request decodes to the same tensors (kv_transfer.pack_request_kv)
    from vllm_omni_amd.kv_transfer import KVCacheTransferData
    dec = KVCacheTransferData.from_bytes(runner.kv_payloads["r0"].cpu().numpy().tobytes())
    assert dec["request_id"] == "r0" and torch.equal(dec["layer_blocks"]["value_cache"][1], got["value_cache"][1].cpu())
    # r1 was not scheduled in that step: it leaves the persistent batch but keeps its cached state
    # (vllm_omni/worker/gpu_model_runner.py:282-301); r0 is gone for good
    assert runner.input_batch.req_ids == [] and "r1" in runner.requests and "r0" not in runner.requests


def test_prefill_then_mixed_batch_matches_oracle(dev):
    """Step 0: two prompts prefilled from embeddings.  Step 1: both decode while a third request
    prefills (mixed batch: the runner reorders decodes first).  Hidden states of every scheduled
    token and the greedy ids are compared with the oracle step."""
    from vllm_omni_amd.kv_cache import KVCacheManager
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, Request, SamplingParams, SchedulerOutput
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("mini_talker")
    cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.mrope_section = 128, 4, 2, None
    bs, nb = 16, 32
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=4, max_model_len=128, max_num_batched_tokens=64,
                              block_size=bs, enforce_eager=True, seed=0)
    runner.load_model()
    w = ref_talker.random_weights(cfg, seed=11)
    runner.model.load_weights({k: v.to(dev) for k, v in w.items()})
    runner.initialize_kv_cache(nb)
    cpu_caches = [(torch.zeros(nb, bs, cfg.num_key_value_heads, cfg.head_dim, dtype=torch.bfloat16),
                   torch.zeros(nb, bs, cfg.num_key_value_heads, cfg.head_dim, dtype=torch.bfloat16))
                  for _ in range(cfg.num_hidden_layers)]
    mgr = KVCacheManager(nb, bs)
    g = torch.Generator().manual_seed(21)
    csc = ref_ops.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)

    def new_req(rid, n):
        r = Request(rid, list(range(1, n + 1)), SamplingParams(temperature=0.0),
                    prompt_embeds=(torch.randn(n, cfg.embedding_size, generator=g) * 0.5).to(torch.bfloat16))
        assert mgr.allocate_slots(r, n) is not None
        return r

    def oracle(reqs_in_order, emb_rows, positions, num_sched):
        B = len(reqs_in_order)
        table = torch.zeros(B, 8, dtype=torch.int32)
        for i, r in enumerate(reqs_in_order):
            ids = mgr.get_block_ids(r.request_id)[0]
            table[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
        qsl = torch.tensor([0] + list(torch.tensor(num_sched).cumsum(0)), dtype=torch.int32)
        req_idx = torch.repeat_interleave(torch.arange(B), torch.tensor(num_sched))
        slots = ref_ops.slot_mapping(table, req_idx, positions, bs)
        seq_lens = torch.tensor([r.num_computed_tokens + n for r, n in zip(reqs_in_order, num_sched)], dtype=torch.int32)
        hidden = ref_talker.talker_step(cfg, w, emb_rows, positions, cpu_caches, table, seq_lens, qsl, slots, csc,
                                        torch.bfloat16)
        last = hidden[(qsl[1:] - 1).long()]
        logits, ids = ref_talker.logits_and_greedy(cfg, w, last, None, torch.bfloat16)
        return hidden, logits, ids, qsl

    # ---- step 0: prefill r0 (20 tokens) and r1 (5 tokens) ----
    r0, r1 = new_req("r0", 20), new_req("r1", 5)
    so = SchedulerOutput(scheduled_new_reqs=[
        NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params, mgr.get_block_ids(r.request_id), 0,
                       prompt_embeds=r.prompt_embeds) for r in (r0, r1)],
        num_scheduled_tokens={"r0": 20, "r1": 5}, total_num_scheduled_tokens=25)
    assert runner.execute_model(so) is None
    out = runner.sample_tokens()
    order = [runner.requests[rid] for rid in out.req_ids]
    reqs = {"r0": r0, "r1": r1}
    seq = [reqs[rid] for rid in out.req_ids]
    ns = [len(r.prompt_token_ids) for r in seq]
    emb = torch.cat([r.prompt_embeds for r in seq])
    pos = torch.cat([torch.arange(n) for n in ns])
    hidden, logits, ids, qsl = oracle(seq, emb, pos, ns)
    got = torch.cat([out.pooler_output[i]["hidden"] for i in range(len(seq))]).float()
    torch.testing.assert_close(got, hidden, rtol=3e-2, atol=3e-2)
    got_ids = torch.tensor([t[0] for t in out.sampled_token_ids])
    top2 = logits.topk(2, -1).values
    conf = (top2[:, 0] - top2[:, 1]) > 0.05
    assert torch.equal(got_ids[conf], ids[conf])
    for r, n, t in zip(seq, ns, got_ids.tolist()):
        r.num_computed_tokens = n
        r.output_token_ids.append(t)

    # ---- step 1: r0, r1 decode; r2 (17 tokens) prefills ----
    r2 = new_req("r2", 17)
    reqs["r2"] = r2
    for r in (r0, r1):
        assert mgr.allocate_slots(r, 1) is not None
    so = SchedulerOutput(
        scheduled_new_reqs=[NewRequestData("r2", r2.prompt_token_ids, r2.sampling_params, mgr.get_block_ids("r2"), 0,
                                           prompt_embeds=r2.prompt_embeds)],
        scheduled_cached_reqs=CachedRequestData(req_ids=["r0", "r1"], new_block_ids=[None, None],
                                                num_computed_tokens=[r0.num_computed_tokens, r1.num_computed_tokens],
                                                num_output_tokens=[1, 1]),
        num_scheduled_tokens={"r0": 1, "r1": 1, "r2": 17}, total_num_scheduled_tokens=19)
    # block tables of r0/r1 may have grown by a block
    for r in (r0, r1):
        ids_now = mgr.get_block_ids(r.request_id)[0]
        row = runner.input_batch.req_id_to_index[r.request_id]
        have = int(runner.input_batch.block_table[0].num_blocks_per_row[row])
        if len(ids_now) > have:
            so.scheduled_cached_reqs.new_block_ids[["r0", "r1"].index(r.request_id)] = (ids_now[have:],)
    assert runner.execute_model(so) is None
    out = runner.sample_tokens()
    seq = [reqs[rid] for rid in out.req_ids]
    assert [len(reqs[rid].prompt_token_ids) == 17 for rid in out.req_ids] == [False, False, True]  # decodes first
    ns = [17 if r is r2 else 1 for r in seq]
    emb_rows, pos_rows = [], []
    for r, n in zip(seq, ns):
        if r is r2:
            emb_rows.append(r.prompt_embeds)
            pos_rows.append(torch.arange(17))
        else:
            emb_rows.append(ref_talker.decode_embeds(w, torch.tensor([r.output_token_ids[-1]]), None, torch.bfloat16))
            pos_rows.append(torch.tensor([r.num_computed_tokens]))
    hidden, logits, ids, qsl = oracle(seq, torch.cat(emb_rows).float(), torch.cat(pos_rows), ns)
    got = torch.cat([out.pooler_output[i]["hidden"] for i in range(len(seq))]).float()
    assert got.shape == hidden.shape == (19, cfg.hidden_size)
    torch.testing.assert_close(got, hidden, rtol=3e-2, atol=3e-2)
    got_ids = torch.tensor([t[0] for t in out.sampled_token_ids])
    top2 = logits.topk(2, -1).values
    conf = (top2[:, 0] - top2[:, 1]) > 0.05
    assert torch.equal(got_ids[conf], ids[conf])


def test_runner_samples_on_device_like_the_oracle(dev):
    """N2 through the runner: per-request temperature / top-k / top-p / repetition penalty / seed; the drawn ids
    equal the sampler oracle applied to the step's masked logits, draw counter = tokens generated so far, the
    seen set = prompt + generated tokens."""
    import numpy as np

    from oracle import ref_sampler
    from vllm_omni_amd.kv_cache import KVCacheManager
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, Request, SamplingParams, SchedulerOutput
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("mini_talker")
    bs, nb = 16, 32
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=4, max_model_len=128, max_num_batched_tokens=64,
                              block_size=bs, enforce_eager=True, seed=0)
    runner.load_model()
    runner.initialize_kv_cache(nb)
    mgr = KVCacheManager(nb, bs)
    sps = [SamplingParams(temperature=0.9, top_k=40, top_p=0.8, repetition_penalty=1.05, seed=123),
           SamplingParams(temperature=0.0),
           SamplingParams(temperature=1.2, repetition_penalty=1.3)]          # seed derived from the request id
    reqs = [Request(f"s{i}", list(range(3 + i, 12 + 2 * i)), sp) for i, sp in enumerate(sps)]
    outs = {r.request_id: [] for r in reqs}
    for step in range(4):
        sched = {}
        for r in reqs:
            n = len(r.prompt_token_ids) if step == 0 else 1
            assert mgr.allocate_slots(r, n) is not None
            sched[r.request_id] = n
        if step == 0:
            so = SchedulerOutput(scheduled_new_reqs=[NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params,
                                                                    mgr.get_block_ids(r.request_id), 0) for r in reqs],
                                 num_scheduled_tokens=sched, total_num_scheduled_tokens=sum(sched.values()))
        else:
            cached = CachedRequestData(req_ids=[r.request_id for r in reqs], new_token_ids=[],
                                       new_block_ids=[mgr.get_block_ids(r.request_id) if False else None for r in reqs],
                                       num_computed_tokens=[r.num_computed_tokens for r in reqs],
                                       num_output_tokens=[len(outs[r.request_id]) for r in reqs])
            so = SchedulerOutput(scheduled_cached_reqs=cached, num_scheduled_tokens=sched,
                                 total_num_scheduled_tokens=len(reqs))
        runner.execute_model(so)
        n = len(reqs)
        logits = runner.execute_model_state.logits[:n].float().cpu().numpy()
        order = list(runner.input_batch.req_ids[:n])
        out = runner.sample_tokens()
        for row, rid in enumerate(order):
            r = next(q for q in reqs if q.request_id == rid)
            sp = r.sampling_params
            seen = np.zeros(cfg.vocab_size, dtype=bool)
            seen[np.asarray(r.prompt_token_ids + outs[rid])] = True
            seed = sp.seed if sp.seed is not None else runner.requests[rid].sample_seed
            want, m = ref_sampler.sample_row(logits[row], sp.temperature, sp.top_k, sp.top_p, sp.repetition_penalty,
                                             seen if sp.repetition_penalty != 1.0 else None, seed, len(outs[rid]))
            got = out.sampled_token_ids[out.req_id_to_index[rid]][0]
            assert got == want or m["topp_gap"] < 1e-5 or m["race_ratio"] < 1 + 1e-4, (step, rid, got, want, m)
            outs[rid].append(got)
        for r in reqs:
            r.num_computed_tokens += sched[r.request_id]
    # sampled rows differ from greedy ones at least once; the bitmap holds prompt + outputs of penalised requests
    st0 = runner.requests["s0"]
    words = runner.sampling_state.bitmap[st0.seen_slot].cpu().numpy().view(np.uint32)
    have = {i for i in range(cfg.vocab_size) if (words[i >> 5] >> (i & 31)) & 1}
    assert have == set(reqs[0].prompt_token_ids) | set(outs["s0"])
    assert runner.requests["s1"].seen_slot == -1


@pytest.mark.parametrize("enforce_eager", [True, False])
def test_async_output_pipeline_equals_sync(dev, enforce_eager):
    """N1: sample_tokens(async_output=True) — the next step is launched before the previous step's ids and hidden
    states are read back; drawn ids feed the next embedding lookup on the device.  Tokens and hidden states must be
    identical to the synchronous loop, through a steady decode phase, a request that finishes mid-way (the pending
    output is drained before the batch changes) and sampled requests (draw counter ahead of the host token list)."""
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, SamplingParams, SchedulerOutput

    def run(async_mode):
        cfg, runner, w, cpu_caches, mgr, reqs = _setup(dev, "mini_talker", [5, 16, 40, 1], enforce_eager, False)
        reqs[1].sampling_params = SamplingParams(temperature=0.8, top_k=20, top_p=0.9, repetition_penalty=1.1, seed=7)
        reqs[3].sampling_params = SamplingParams(temperature=1.0, seed=11)
        live = list(reqs)
        tokens = {r.request_id: [] for r in reqs}
        hiddens = {r.request_id: [] for r in reqs}
        pending = None

        def take(out):
            for rid, j in out.req_id_to_index.items():
                if rid in out.req_ids:
                    tokens[rid].append(out.sampled_token_ids[j][0])
                    hiddens[rid].append(out.pooler_output[j]["hidden"][0].clone())

        for step in range(7):
            finished = set()
            if step == 4:                      # r2 finishes: composition change -> the runner drains by itself
                finished = {"r2"}
                live = [r for r in live if r.request_id != "r2"]
            sched = {r.request_id: 1 for r in live}
            if step == 0:
                so = SchedulerOutput(scheduled_new_reqs=[
                    NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params, mgr.get_block_ids(r.request_id),
                                   r.num_computed_tokens, additional_information=r.additional_information) for r in live],
                    num_scheduled_tokens=sched, total_num_scheduled_tokens=len(live))
            else:
                new_blocks = []
                for r in live:
                    nb_ = mgr.allocate_slots(r, 1)
                    new_blocks.append(nb_.get_block_ids() if nb_.blocks[0] else None)
                so = SchedulerOutput(scheduled_cached_reqs=CachedRequestData(
                    req_ids=[r.request_id for r in live], new_block_ids=new_blocks,
                    num_computed_tokens=[r.num_computed_tokens for r in live], num_output_tokens=[step] * len(live)),
                    num_scheduled_tokens=sched, total_num_scheduled_tokens=len(live), finished_req_ids=finished)
            runner.execute_model(so)
            out = runner.sample_tokens(async_output=async_mode)
            if async_mode:
                if pending is not None:
                    take(pending.get_output())
                pending = out
            else:
                take(out)
            for r in live:
                r.num_computed_tokens += 1
        if pending is not None:
            take(pending.get_output())
        return tokens, hiddens, {rid: list(st.output_token_ids) for rid, st in runner.requests.items()}

    t_sync, h_sync, o_sync = run(False)
    t_async, h_async, o_async = run(True)
    assert t_sync == t_async and o_sync == o_async
    assert [len(v) for v in t_sync.values()] == [7, 7, 4, 7]
    for rid in h_sync:
        for a_, b_ in zip(h_sync[rid], h_async[rid]):
            assert torch.equal(a_, b_)


def test_moe_talker_decode_steps_match_oracle(dev):
    """Qwen3-Omni talker layer in small (q/k-norm attention + sparse-MoE MLP with a shared expert)
    through the runner — eager and as a hipGraph — against the oracle stack with ref_moe in the MLP position."""
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, SchedulerOutput

    for enforce_eager in (True, False):
        ctxs = [5, 16, 40, 1, 33]
        cfg, runner, w, cpu_caches, mgr, reqs = _setup(dev, "mini_moe_talker", ctxs, enforce_eager, False)
        B, bs = len(reqs), 16
        csc = ref_ops.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
        last_ids = torch.tensor([r.prompt_token_ids[-1] for r in reqs], dtype=torch.int64)
        table = torch.zeros(B, 16, dtype=torch.int32)
        for step in range(2):
            sched = {r.request_id: 1 for r in reqs}
            if step == 0:
                so = SchedulerOutput(scheduled_new_reqs=[
                    NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params, mgr.get_block_ids(r.request_id),
                                   r.num_computed_tokens, additional_information=r.additional_information) for r in reqs],
                    num_scheduled_tokens=sched, total_num_scheduled_tokens=B)
            else:
                new_blocks = []
                for r in reqs:
                    nb_ = mgr.allocate_slots(r, 1)
                    new_blocks.append(nb_.get_block_ids() if nb_.blocks[0] else None)
                so = SchedulerOutput(scheduled_cached_reqs=CachedRequestData(
                    req_ids=[r.request_id for r in reqs], new_block_ids=new_blocks,
                    num_computed_tokens=[r.num_computed_tokens for r in reqs], num_output_tokens=[step] * B),
                    num_scheduled_tokens=sched, total_num_scheduled_tokens=B)
            for i, r in enumerate(reqs):
                ids = mgr.get_block_ids(r.request_id)[0]
                table[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
            runner.execute_model(so)
            out = runner.sample_tokens()
            order = [out.req_id_to_index[r.request_id] for r in reqs]
            pos = torch.tensor([r.num_computed_tokens for r in reqs], dtype=torch.int64)
            step_vec = torch.stack([r.additional_information["thinker_reply_part"][step] for r in reqs])
            emb = ref_talker.decode_embeds(w, last_ids, step_vec, torch.bfloat16)
            slots = ref_ops.slot_mapping(table, torch.arange(B), pos, bs)
            hidden = ref_talker.talker_step(cfg, w, emb, pos, cpu_caches, table, (pos + 1).to(torch.int32),
                                            torch.arange(B + 1, dtype=torch.int32), slots, csc, torch.bfloat16)
            got = torch.stack([out.pooler_output[j]["hidden"][0] for j in order]).float()
            # an expert swapped on a near-tie of two router probabilities moves a whole row: allow one such row
            row_err = (got - hidden).abs().amax(-1)
            assert (row_err > 0.1).sum() <= 1, row_err
            ok = row_err <= 0.1
            torch.testing.assert_close(got[ok], hidden[ok], rtol=3e-2, atol=3e-2)
            got_ids = torch.tensor([out.sampled_token_ids[j][0] for j in order])
            last_ids = got_ids.clone()
            for r, t in zip(reqs, got_ids.tolist()):
                r.num_computed_tokens += 1
                r.output_token_ids.append(t)


@pytest.mark.parametrize("enforce_eager", [True, False])
def test_qwen3_talker_step_with_code_predictor(dev, enforce_eager):
    """`talker_mtp` through the runner (qwen3_omni.py:681-705): every decode step runs the code predictor on (last
    codec token, its embedding, last talker hidden state); the talker's input is the sum of the G codebook
    embeddings + the text step.  Oracle: ref_code_predictor (teacher-forced with the device's codes) -> sum -> talker
    stack with the MoE MLP.  Codes are handed back per request."""
    from oracle import ref_code_predictor as rcp
    from vllm_omni_amd.code_predictor import CodePredictorConfig
    from vllm_omni_amd.kv_cache import KVCacheManager, normalize_layer_kv
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, Request, SamplingParams, SchedulerOutput
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("mini_qwen3_talker")
    cpc = CodePredictorConfig(hidden_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                              head_dim=128, intermediate_size=256, vocab_size=256, num_code_groups=4)
    bs, nb, B = 16, 48, 3
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=4, max_model_len=128, max_num_batched_tokens=64,
                              block_size=bs, enforce_eager=enforce_eager, max_reply_rows=32, seed=0, code_predictor=cpc)
    runner.load_model()
    w = ref_talker.random_weights(cfg, seed=3)
    assert runner.model.load_weights({k: v.to(dev) for k, v in w.items()}) == set(w.keys())
    wcp = rcp.random_weights(cpc, seed=4)
    assert runner.code_predictor.load_weights({k: v.to(dev) for k, v in wcp.items()}) == set(wcp.keys())
    runner.initialize_kv_cache(nb)
    g = torch.Generator().manual_seed(5)
    cpu_caches = []
    for kv in runner.kv_caches:
        init = torch.randn(nb, 2, bs, cfg.num_key_value_heads, cfg.head_dim, generator=g).to(torch.bfloat16)
        kb, vb = normalize_layer_kv(kv)
        kb.copy_(init[:, 0].to(dev))
        vb.copy_(init[:, 1].to(dev))
        cpu_caches.append((init[:, 0].clone(), init[:, 1].clone()))
    mgr = KVCacheManager(nb, bs)
    reqs = []
    for i, c in enumerate([7, 20, 1]):
        r = Request(f"r{i}", list(range(c + 1)), SamplingParams(temperature=0.0, seed=100 + i))
        r.additional_information = {"thinker_reply_part": torch.randn(4, cfg.embedding_size, generator=g).to(torch.bfloat16)}
        assert mgr.allocate_slots(r, c + 1) is not None
        r.num_computed_tokens = c
        reqs.append(r)
    csc = ref_ops.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
    last_ids = torch.tensor([r.prompt_token_ids[-1] for r in reqs], dtype=torch.int64)
    last_hidden = torch.zeros(B, cfg.hidden_size)
    table = torch.zeros(B, 8, dtype=torch.int32)
    for step in range(3):
        sched = {r.request_id: 1 for r in reqs}
        if step == 0:
            so = SchedulerOutput(scheduled_new_reqs=[
                NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params, mgr.get_block_ids(r.request_id),
                               r.num_computed_tokens, additional_information=r.additional_information) for r in reqs],
                num_scheduled_tokens=sched, total_num_scheduled_tokens=B)
        else:
            new_blocks = []
            for r in reqs:
                nb_ = mgr.allocate_slots(r, 1)
                new_blocks.append(nb_.get_block_ids() if nb_.blocks[0] else None)
            so = SchedulerOutput(scheduled_cached_reqs=CachedRequestData(
                req_ids=[r.request_id for r in reqs], new_block_ids=new_blocks,
                num_computed_tokens=[r.num_computed_tokens for r in reqs], num_output_tokens=[step] * B),
                num_scheduled_tokens=sched, total_num_scheduled_tokens=B)
        for i, r in enumerate(reqs):
            ids = mgr.get_block_ids(r.request_id)[0]
            table[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
        runner.execute_model(so)
        out = runner.sample_tokens()
        order = [out.req_id_to_index[r.request_id] for r in reqs]
        codes = torch.stack([out.pooler_output[j]["code_predictor_codes"] for j in order])
        assert codes.shape == (B, cpc.num_code_groups) and torch.equal(codes[:, 0], last_ids)
        assert int(codes[:, 1:].min()) >= 0 and int(codes[:, 1:].max()) < cpc.vocab_size
        # ---- oracle
        emb0 = w["language_model.model.embed_tokens.weight"].float()[last_ids]
        _, buf, _ = rcp.forward(cpc, wcp, last_ids.view(B, 1), emb0.view(B, 1, -1), last_hidden, rt=torch.bfloat16,
                                forced_codes=codes)
        step_vec = torch.stack([r.additional_information["thinker_reply_part"][step] for r in reqs]).float()
        x = (buf[:, 1:].sum(1) + step_vec).to(torch.bfloat16).float()
        pos = torch.tensor([r.num_computed_tokens for r in reqs], dtype=torch.int64)
        slots = ref_ops.slot_mapping(table, torch.arange(B), pos, bs)
        hidden = ref_talker.talker_step(cfg, w, x, pos, cpu_caches, table, (pos + 1).to(torch.int32),
                                        torch.arange(B + 1, dtype=torch.int32), slots, csc, torch.bfloat16)
        got = torch.stack([out.pooler_output[j]["hidden"][0] for j in order]).float()
        row_err = (got - hidden).abs().amax(-1)
        assert (row_err > 0.15).sum() <= 1, row_err          # one row may move on an expert near-tie
        ok = row_err <= 0.15
        torch.testing.assert_close(got[ok], hidden[ok], rtol=4e-2, atol=4e-2)
        got_ids = torch.tensor([out.sampled_token_ids[j][0] for j in order])
        last_ids, last_hidden = got_ids.clone(), got.clone()     # feed the device's state to both sides
        for r, t in zip(reqs, got_ids.tolist()):
            r.num_computed_tokens += 1
            r.output_token_ids.append(t)


def _qwen3_runner(dev, enforce_eager, ctxs, reply_rows=8):
    """Mini Qwen3-Omni talker + code predictor through the runner, KV pre-filled, requests ready to decode."""
    from oracle import ref_code_predictor as rcp
    from vllm_omni_amd.code_predictor import CodePredictorConfig
    from vllm_omni_amd.kv_cache import KVCacheManager, normalize_layer_kv
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import Request, SamplingParams
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("mini_qwen3_talker")
    cpc = CodePredictorConfig(hidden_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                              head_dim=128, intermediate_size=256, vocab_size=256, num_code_groups=4)
    bs, nb = 16, 64
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=8, max_model_len=128, max_num_batched_tokens=64,
                              block_size=bs, enforce_eager=enforce_eager, max_reply_rows=64, seed=0, code_predictor=cpc)
    runner.load_model()
    w = ref_talker.random_weights(cfg, seed=3)
    runner.model.load_weights({k: v.to(dev) for k, v in w.items()})
    runner.code_predictor.load_weights({k: v.to(dev) for k, v in rcp.random_weights(cpc, seed=4).items()})
    runner.initialize_kv_cache(nb)
    g = torch.Generator().manual_seed(5)
    for kv in runner.kv_caches:
        init = torch.randn(nb, 2, bs, cfg.num_key_value_heads, cfg.head_dim, generator=g).to(torch.bfloat16)
        kb, vb = normalize_layer_kv(kv)
        kb.copy_(init[:, 0].to(dev))
        vb.copy_(init[:, 1].to(dev))
    mgr = KVCacheManager(nb, bs)
    reqs = []
    for i, c in enumerate(ctxs):
        r = Request(f"r{i}", list(range(c + 1)), SamplingParams(temperature=0.0, seed=100 + i))
        r.additional_information = {"thinker_reply_part": torch.randn(reply_rows, cfg.embedding_size, generator=g).to(torch.bfloat16)}
        assert mgr.allocate_slots(r, c + 1) is not None
        r.num_computed_tokens = c
        reqs.append(r)
    return cfg, cpc, runner, mgr, reqs


def test_graph_capture_in_the_middle_of_serving_keeps_predictor_state(dev):
    """the first time a batch size is seen the runner captures its hipGraph by running the
    live step several times.  With a code predictor a step ADVANCES device state (`last_talker_hidden`); capture
    must leave that state as the previous step left it.  Five requests decode; two finish after step 2, so the
    batch drops from the 8-row graph to the 4-row graph, which is captured mid-serving.  Tokens and codes must be
    identical to the eager run, step by step."""
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, SchedulerOutput

    def run(enforce_eager):
        cfg, cpc, runner, mgr, reqs = _qwen3_runner(dev, enforce_eager, [7, 20, 1, 33, 12])
        live, trace = list(reqs), []
        for step in range(6):
            finished = set()
            if step == 3:
                finished = {"r1", "r4"}
                live = [r for r in live if r.request_id not in finished]
            sched = {r.request_id: 1 for r in live}
            if step == 0:
                so = SchedulerOutput(scheduled_new_reqs=[
                    NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params, mgr.get_block_ids(r.request_id),
                                   r.num_computed_tokens, additional_information=r.additional_information) for r in live],
                    num_scheduled_tokens=sched, total_num_scheduled_tokens=len(live))
            else:
                new_blocks = []
                for r in live:
                    nb_ = mgr.allocate_slots(r, 1)
                    new_blocks.append(nb_.get_block_ids() if nb_.blocks[0] else None)
                so = SchedulerOutput(scheduled_cached_reqs=CachedRequestData(
                    req_ids=[r.request_id for r in live], new_block_ids=new_blocks,
                    num_computed_tokens=[r.num_computed_tokens for r in live], num_output_tokens=[step] * len(live)),
                    num_scheduled_tokens=sched, total_num_scheduled_tokens=len(live), finished_req_ids=finished)
            runner.execute_model(so)
            out = runner.sample_tokens()
            for r in live:
                j = out.req_id_to_index[r.request_id]
                trace.append((step, r.request_id, out.sampled_token_ids[j][0],
                              out.pooler_output[j]["code_predictor_codes"].tolist()))
                r.num_computed_tokens += 1
        return trace, sorted(runner._graphs)

    eager, _ = run(True)
    graph, sizes = run(False)
    assert sizes == [4, 8]                 # 5 rows ran in the 8-row graph, 3 rows in the 4-row graph (padded rows inert)
    assert eager == graph


def test_prefill_step_stores_last_hidden_in_the_requests_own_state_rows(dev):
    """a pure-prefill step — the first step of every request — must store each request's last
    hidden state in ITS state row (`last_talker_hidden`, qwen3_omni.py:634-640), not in the rows of the previous
    batch.  A decode-only batch runs first (so the device still holds ITS row mapping), then two new requests
    prefill alone; their state rows must hold their own last prompt token's hidden state and the other
    requests' rows must be untouched."""
    from vllm_omni_amd.sched_output import NewRequestData, Request, SamplingParams, SchedulerOutput

    cfg, cpc, runner, mgr, reqs = _qwen3_runner(dev, True, [7, 20, 5])
    so = SchedulerOutput(scheduled_new_reqs=[
        NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params, mgr.get_block_ids(r.request_id),
                       r.num_computed_tokens, additional_information=r.additional_information) for r in reqs],
        num_scheduled_tokens={r.request_id: 1 for r in reqs}, total_num_scheduled_tokens=3)
    runner.execute_model(so)
    runner.sample_tokens()
    before = runner._last_hidden.clone()
    old_slots = [runner.requests[r.request_id].state_slot for r in reqs]
    g = torch.Generator().manual_seed(9)
    news = []
    for name, n in (("p0", 9), ("p1", 21)):
        r = Request(name, list(range(1, n + 1)), SamplingParams(temperature=0.0),
                    prompt_embeds=(torch.randn(n, cfg.embedding_size, generator=g) * 0.5).to(torch.bfloat16))
        assert mgr.allocate_slots(r, n) is not None
        news.append(r)
    # the three decoding requests are NOT scheduled this step: a pure-prefill batch of the two new ones
    so = SchedulerOutput(scheduled_new_reqs=[
        NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params, mgr.get_block_ids(r.request_id), 0,
                       prompt_embeds=r.prompt_embeds) for r in news],
        num_scheduled_tokens={r.request_id: len(r.prompt_token_ids) for r in news}, total_num_scheduled_tokens=30)
    runner.execute_model(so)
    out = runner.sample_tokens()
    for r in news:
        slot = runner.requests[r.request_id].state_slot
        assert slot not in old_slots
        last = out.pooler_output[out.req_id_to_index[r.request_id]]["hidden"][-1]
        assert torch.equal(runner._last_hidden[slot].cpu(), last), r.request_id
    for s in old_slots:                                        # the waiting requests' state survived the prefill step
        assert torch.equal(runner._last_hidden[s], before[s])

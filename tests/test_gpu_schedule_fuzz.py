"""GPU parity, randomised: the runner driven by RANDOM schedules — requests arriving and finishing at random steps, prompts
prefilled in random chunks (one-token chunks included), decode rows and prefill chunks mixed in one step, graph replay on
the pure-decode steps — against the CPU oracle stepping the same schedule on its own copy of the paged cache.

What it pins beyond the scripted steps of test_gpu_talker_step.py: the persistent batch (add / remove / condense /
decodes-first reorder) over many steps, block tables growing a block at a time, a chunk of a prompt continuing where the
previous one stopped, a one-token step that is still inside its prompt taking its input from the PROMPT (not from the last
sampled token), the bookkeeping rule that an unfinished prompt generates nothing (``sampled_token_ids[i] == []``,
upstream's discarded samples), and preemption: a request that lost its blocks comes back as a RESUMED request with a new
block list and recomputes its prompt and the tokens it had generated (prefill rows, nothing generated) before it decodes
again — with the thinker's reply rows added to the recomputed generated tokens exactly as they were when those tokens were
decoded, and the request's reply cursor continuing behind them.

Tolerances as in test_gpu_talker_step.py (hidden states rtol / atol 3e-2 after 2 bf16 layers, a lone element to 5e-2), greedy ids wherever the
oracle's top-2 logit margin exceeds 0.05; the oracle is teacher-forced with the tokens the device drew."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_ops, ref_talker

pytestmark = pytest.mark.gpu

_EXTRA = int(os.environ.get("MI_FUZZ_EXTRA", "0"))


# seeds 50-52: schedules biased towards the talker's real flow under memory pressure — every prompt is embeddings with
# reply rows, preemption four times as likely — so that recomputed generated tokens WITH reply rows occur for sure
@pytest.mark.parametrize("seed", list(range(4)) + [50, 51, 52] + list(range(100, 100 + _EXTRA)))
def test_random_schedule_matches_oracle(dev, seed):
    from vllm_omni_amd.kv_cache import KVCacheManager
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, Request, SamplingParams, SchedulerOutput
    from vllm_omni_amd.talker import TalkerConfig

    rng = np.random.default_rng(9000 + seed)
    g = torch.Generator().manual_seed(seed)
    biased = 50 <= seed < 100
    cfg = TalkerConfig.from_json("mini_talker")
    d, nq, nkv = [(128, 4, 2), (64, 4, 2), (128, 2, 1), (64, 8, 4)][int(rng.integers(0, 4))]
    cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.mrope_section = d, nq, nkv, None
    bs, nb, max_reqs, budget = 16, 96, 5, 80
    eager = bool(rng.random() < 0.5)
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=max_reqs, max_model_len=192,
                              max_num_batched_tokens=budget, block_size=bs, enforce_eager=eager, seed=0,
                              max_reply_rows=64)
    runner.load_model()
    w = ref_talker.random_weights(cfg, seed=11 + seed)
    runner.model.load_weights({k: v.to(dev) for k, v in w.items()})
    runner.initialize_kv_cache(nb)
    cpu_caches = [(torch.zeros(nb, bs, nkv, d, dtype=torch.bfloat16), torch.zeros(nb, bs, nkv, d, dtype=torch.bfloat16))
                  for _ in range(cfg.num_hidden_layers)]
    mgr = KVCacheManager(nb, bs)
    scramble = mgr.block_pool.get_new_blocks(nb - 1)           # non-contiguous block tables
    mgr.block_pool.free_blocks(rng.permutation(scramble).tolist())
    csc = ref_ops.build_cos_sin_cache(d, cfg.max_position_embeddings, cfg.rope_theta)

    class Live:
        def __init__(self, rid):
            n = int(rng.choice([1, 2, 3, 15, 16, 17, 33, 60])) if rng.random() < 0.6 else int(rng.integers(1, 70))
            self.use_embeds = bool(rng.random() < 0.7) or biased   # the talker's prompts are embeddings; token ids too
            emb = (torch.randn(n, cfg.embedding_size, generator=g) * 0.5).to(torch.bfloat16) if self.use_embeds else None
            toks = torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist()
            self.r = Request(rid, toks, SamplingParams(temperature=0.0), prompt_embeds=emb)
            # the thinker's reply rows, one added to the input of each decode step until they run out (a request that
            # sits a step out keeps its cursor); with prompts of embeddings only — the reference's talker flow
            self.reply = None
            if self.use_embeds and (rng.random() < 0.6 or biased):
                self.reply = (torch.randn(int(rng.integers(1, 5)), cfg.embedding_size, generator=g) * 0.5).to(torch.bfloat16)
                self.r.additional_information = {"thinker_reply_part": self.reply}
            self.n_decoded = 0
            self.recompute_until = 0      # preempted and resumed: the tokens it recomputes before it decodes again
            self.resume = False           # the next time it is scheduled it comes back as a RESUMED request
            self.plen = n
            self.max_new = int(rng.integers(3, 9)) if biased else int(rng.integers(1, 7))
            self.out = []                 # tokens the device generated
            self.computed = 0
            self.known = False            # the runner has seen it

    live, finished_last, next_id, steps_with_graph = [], set(), 0, 0
    n_partial = n_one_token_prompt_rows = n_mixed = n_recomputed = n_recomputed_with_reply = 0
    for step in range(int(rng.integers(14, 22))):
        # ---- a random schedule for this step ----
        while len(live) < max_reqs and rng.random() < (0.6 if len(live) < 2 else 0.3):
            live.append(Live(f"q{next_id}"))
            next_id += 1
        left = budget
        plan = []
        # preemption (recompute): a running request loses its blocks and starts over at token 0 when it is scheduled
        # again, as a resumed request with a new block list; it recomputes prompt + generated tokens before it decodes
        for q in live:
            if q.known and q.computed > 0 and not q.resume and rng.random() < (0.25 if biased else 0.06):
                mgr.free(q.r)
                q.r.num_computed_tokens = q.computed = 0
                q.recompute_until = q.plen + len(q.out)
                q.resume = True
        for q in live:
            target = max(q.plen, q.recompute_until)
            if q.computed >= target:
                n = 1                                                # decode
            else:
                rem = target - q.computed
                n = rem if rng.random() < 0.4 else int(rng.integers(1, rem + 1))
                if rng.random() < 0.15:
                    n = 1                                            # a one-token chunk inside the prompt
            if n > left or (q.known is False and rng.random() < 0.2) or rng.random() < 0.1 or (q.resume and rng.random() < 0.5):
                continue                                             # not scheduled this step
            left -= n
            plan.append((q, n))
        if not plan:
            continue
        new_reqs, c_ids, c_blocks, c_computed, c_nout, resumed = [], [], [], [], [], set()
        for q, n in plan:
            blk = mgr.allocate_slots(q.r, n)
            assert blk is not None
            if not q.known:
                new_reqs.append(NewRequestData(q.r.request_id, q.r.prompt_token_ids, q.r.sampling_params,
                                               mgr.get_block_ids(q.r.request_id), 0, prompt_embeds=q.r.prompt_embeds,
                                               additional_information=q.r.additional_information))
                q.known = True
            else:
                c_ids.append(q.r.request_id)
                if q.resume:
                    resumed.add(q.r.request_id)
                    q.resume = False
                    c_blocks.append(mgr.get_block_ids(q.r.request_id))            # the whole new list
                else:
                    c_blocks.append(blk.get_block_ids() if blk.blocks[0] else None)
                c_computed.append(q.computed)
                c_nout.append(len(q.out))
        so = SchedulerOutput(
            scheduled_new_reqs=new_reqs,
            scheduled_cached_reqs=CachedRequestData(req_ids=c_ids, resumed_req_ids=resumed, new_block_ids=c_blocks,
                                                    num_computed_tokens=c_computed, num_output_tokens=c_nout),
            num_scheduled_tokens={q.r.request_id: n for q, n in plan}, total_num_scheduled_tokens=sum(n for _, n in plan),
            finished_req_ids=set(finished_last))
        finished_last = set()
        assert runner.execute_model(so) is None
        out = runner.sample_tokens()
        by_id = {q.r.request_id: (q, n) for q, n in plan}
        assert sorted(out.req_ids) == sorted(by_id)
        order = [by_id[rid] for rid in out.req_ids]
        # decode rows first (the runner's reorder), in the runner's order
        is_dec = [n == 1 and q.computed >= (q.plen if q.use_embeds else max(q.plen - 1, 0))
                  and q.computed + 1 >= q.recompute_until for q, n in order]
        assert is_dec == sorted(is_dec, reverse=True)
        if all(is_dec) and not eager:
            steps_with_graph += 1
        n_mixed += int(any(is_dec) and not all(is_dec))
        # ---- the oracle on the same step ----
        B = len(order)
        table = torch.zeros(B, 16, dtype=torch.int32)
        emb_rows, pos_rows, ns = [], [], []
        for i, (q, n) in enumerate(order):
            ids = mgr.get_block_ids(q.r.request_id)[0]
            table[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
            c = q.computed
            if c >= max(q.plen, q.recompute_until):                  # decode: embedding of the last generated token
                vec = q.reply[q.n_decoded:q.n_decoded + 1] if (q.reply is not None and q.n_decoded < q.reply.shape[0]) else None
                emb_rows.append(ref_talker.decode_embeds(w, torch.tensor([q.out[-1]]), vec, torch.bfloat16))
                q.n_decoded += 1
            else:                                                    # prompt rows, then (recomputation) generated tokens
                a = min(n, max(0, q.plen - c))
                if a:
                    if q.use_embeds:
                        emb_rows.append(q.r.prompt_embeds[c:c + a].float())
                        n_one_token_prompt_rows += int(n == 1)
                    else:
                        emb_rows.append(ref_talker.decode_embeds(w, torch.tensor(q.r.prompt_token_ids[c:c + a]), None, torch.bfloat16))
                if a < n:
                    # generated token j went into the model as embed(token j) + reply[j] when it was decoded
                    # (qwen2_5_omni.py:766-806): the recomputation must feed the same rows to rebuild the same KV
                    j0, j1 = c + a - q.plen, c + n - q.plen
                    toks = q.out[j0:j1]
                    assert len(toks) == n - a
                    for j, tok in zip(range(j0, j1), toks):
                        vec = q.reply[j:j + 1] if (q.reply is not None and j < q.reply.shape[0]) else None
                        emb_rows.append(ref_talker.decode_embeds(w, torch.tensor([tok]), vec, torch.bfloat16))
                        n_recomputed_with_reply += int(vec is not None)
                    q.n_decoded = max(q.n_decoded, j1)               # the last recomputed row IS the next decode step
                    n_recomputed += n - a
            pos_rows.append(torch.arange(c, c + n))
            ns.append(n)
        qsl = torch.tensor([0] + list(np.cumsum(ns)), dtype=torch.int32)
        positions = torch.cat(pos_rows)
        req_idx = torch.repeat_interleave(torch.arange(B), torch.tensor(ns))
        slots = ref_ops.slot_mapping(table, req_idx, positions, bs)
        seq_lens = torch.tensor([q.computed + n for q, n in order], dtype=torch.int32)
        hidden = ref_talker.talker_step(cfg, w, torch.cat(emb_rows).float(), positions, cpu_caches, table, seq_lens, qsl, slots,
                                        csc, torch.bfloat16)
        logits, ids = ref_talker.logits_and_greedy(cfg, w, hidden[(qsl[1:] - 1).long()], None, torch.bfloat16)
        got = torch.cat([out.pooler_output[i]["hidden"] for i in range(B)]).float()
        assert got.shape == hidden.shape
        # over hundreds of steps a lone element lands a little beyond the 3e-2 of the scripted tests: 5e-2 for any element,
        # 3e-2 for all but one in a thousand, and a mean error of about one bf16 ulp of the values (0.0063-0.0073 measured)
        where = f"seed {seed} step {step} plan {[(q.r.request_id, q.computed, n) for q, n in order]}"
        torch.testing.assert_close(got, hidden, rtol=3e-2, atol=5e-2, msg=lambda m: f"{where}: {m}")
        err = (got - hidden).abs()
        assert int((err > 3e-2 + 3e-2 * hidden.abs()).sum()) <= max(1, err.numel() // 1000), where
        assert err.mean() < 1.2e-2, (where, float(err.mean()))
        top2 = logits.topk(2, -1).values
        for i, (q, n) in enumerate(order):
            q.computed += n
            if q.computed < max(q.plen, q.recompute_until):          # prompt (or recomputation) not finished: nothing is generated
                assert out.sampled_token_ids[i] == [], (seed, step, q.r.request_id)
                n_partial += 1
                continue
            assert len(out.sampled_token_ids[i]) == 1
            tok = int(out.sampled_token_ids[i][0])
            if float(top2[i, 0] - top2[i, 1]) > 0.05:
                assert tok == int(ids[i]), (seed, step, q.r.request_id)
            q.out.append(tok)                                        # teacher-force the oracle with the device's token
            q.r.num_computed_tokens = q.computed
            assert runner.requests[q.r.request_id].output_token_ids == q.out
        for q, _ in order:
            q.r.num_computed_tokens = q.computed
        for q in list(live):
            if len(q.out) >= q.max_new:
                live.remove(q)
                finished_last.add(q.r.request_id)
                mgr.free(q.r)
    # the schedule exercised what the test is about (deterministic per seed; checked over the default seeds together below)
    test_random_schedule_matches_oracle.stats = getattr(test_random_schedule_matches_oracle, "stats", [])
    test_random_schedule_matches_oracle.stats.append((n_partial, n_one_token_prompt_rows, n_mixed, steps_with_graph, n_recomputed,
                                                      n_recomputed_with_reply))


def test_schedules_covered_the_cases():
    stats = getattr(test_random_schedule_matches_oracle, "stats", None)
    if not stats:
        pytest.skip("runs after test_random_schedule_matches_oracle")
    tot = np.asarray(stats).sum(0)
    assert tot[0] > 0 and tot[1] > 0 and tot[2] > 0, f"partial chunks / one-token prompt rows / mixed steps: {tot}"
    if len(stats) >= 7:   # the biased seeds ran
        assert tot[4] > 0, "no schedule recomputed generated tokens after a preemption"
        assert tot[5] > 0, "no recomputed generated token carried a reply row"

"""GPU: the model-hook half of the runner contract: ``has_preprocess`` / ``preprocess``,
``has_postprocess`` / ``postprocess`` (vllm_omni/worker/gpu_model_runner.py:1264-1309,1032-1070) and
``prefer_model_sampler`` / ``sample`` (gpu_ar_model_runner.py:626-657).

The talker class's own ``preprocess`` is what the runner's vectorised step implements for the whole batch (device
reply pool + one gather-add launch).  A subclass that clears ``batched_preprocess`` is driven through the reference's
per-request hook loop instead; both routes must hand the next stage the SAME hidden states and tokens, bit for bit —
they feed identical rows to identical kernels."""
from types import SimpleNamespace as NS

import pytest
import torch

from oracle import ref_talker

pytestmark = pytest.mark.gpu

BS, NB = 16, 48


def _runner(dev, model_cls=None):
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json("mini_talker")
    cfg.mrope_section = None
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=4, max_model_len=128, max_num_batched_tokens=96,
                              block_size=BS, enforce_eager=True, max_reply_rows=64, seed=0, model_cls=model_cls)
    runner.load_model()
    w = ref_talker.random_weights(cfg, seed=3)
    g = torch.Generator().manual_seed(21)
    table = torch.zeros(151873, cfg.embedding_size, dtype=torch.bfloat16)
    table[151872] = torch.randn(cfg.embedding_size, generator=g).to(torch.bfloat16)
    table[151860] = torch.randn(cfg.embedding_size, generator=g).to(torch.bfloat16)
    runner.model.load_weights([(k, v.to(dev)) for k, v in w.items()] + [("thinker_embedding.weight", table)])
    runner.model.set_suppress_start_id(cfg.tts_codec_mask_token_id)
    runner.initialize_kv_cache(NB)
    return cfg, runner


def _scheduler():
    from vllm_omni_amd.scheduler import MI355XARScheduler

    vc = NS(scheduler_config=NS(max_num_seqs=4, max_num_batched_tokens=96, enable_chunked_prefill=True),
            cache_config=NS(block_size=BS, enable_prefix_caching=False), model_config=NS(max_model_len=128, omni_kv_config=None))
    return MI355XARScheduler(vc, NS(num_blocks=NB), None, BS)


def _requests(cfg, steps):
    """Two talker requests as the reference's stage input processor hands them over (thinker hand-over in
    additional_information, placeholder prompt ids of P + 2 tokens), different prompt lengths and reply lengths — one
    reply queue runs dry before the request ends."""
    from vllm_omni_amd.sched_output import Request, SamplingParams

    g = torch.Generator().manual_seed(7)
    reqs = []
    for rid, P, K in (("a", 21, 3), ("b", 9, 8)):
        info = {"prompt_embeds": torch.randn(P, cfg.embedding_size, generator=g),
                "thinker_result": torch.randn(K, cfg.embedding_size, generator=g),
                "prompt_token_ids": list(range(P)), "thinker_output_token_ids": list(range(K))}
        reqs.append(Request(rid, [8293] + [8292] * P + [8294], SamplingParams(temperature=0.0, max_tokens=steps),
                            additional_information=info))
    return reqs


def _drive(sched, runner, max_steps=32):
    trace = []
    for _ in range(max_steps):
        if not sched.has_unfinished_requests():
            break
        so = sched.schedule()
        out = runner.execute_model(so)
        if out is None:
            out = runner.sample_tokens()
        sched.update_from_output(so, out)
        trace.append(out)
    assert not sched.has_unfinished_requests()
    return trace


def test_per_request_hook_loop_equals_the_vectorised_step(dev):
    from vllm_omni_amd.talker import Qwen2_5OmniTalkerForConditionalGeneration as Talker

    calls = {"prefill": 0, "decode": 0}

    class HookedTalker(Talker):
        batched_preprocess = False          # same preprocess, driven per request by the runner's hook loop

        def preprocess(self, input_ids, input_embeds, **info):
            calls["prefill" if input_ids.shape[0] > 1 else "decode"] += 1
            assert "request_id" in info
            return super().preprocess(input_ids, input_embeds, **info)

    steps = 6
    traces = []
    for cls in (None, HookedTalker):
        cfg, runner = _runner(dev, cls)
        assert runner._generic_hooks is (cls is not None)
        sched = _scheduler()
        reqs = _requests(cfg, steps)
        for r in reqs:
            sched.add_request(r)
        trace = _drive(sched, runner)
        assert all(r.is_finished() and len(r.output_token_ids) == steps for r in reqs)
        traces.append((trace, [list(r.output_token_ids) for r in reqs]))
    (fast, fast_tokens), (hooked, hooked_tokens) = traces
    assert calls["prefill"] == 2 and calls["decode"] == 2 * (steps - 1)
    assert fast_tokens == hooked_tokens
    assert len(fast) == len(hooked)
    for i, (x, y) in enumerate(zip(fast, hooked)):
        assert x.req_ids == y.req_ids and x.sampled_token_ids == y.sampled_token_ids, i
        for px, py in zip(x.pooler_output, y.pooler_output):
            assert torch.equal(px["hidden"], py["hidden"]), f"step {i}"


def test_postprocess_updates_reach_the_next_preprocess_and_model_sampler_is_preferred(dev):
    from vllm_omni_amd.talker import Qwen2_5OmniTalkerForConditionalGeneration as Talker

    seen = []

    class StatefulTalker(Talker):
        """A stage model in the reference's style (qwen3_omni.py:634-640 `talker_postprocess`: the last hidden row goes
        back into the request's buffer; the next `preprocess` reads it) with its own sampler."""
        batched_preprocess = False
        has_postprocess = True
        prefer_model_sampler = True
        gpu_resident_buffer_keys = {"last_talker_hidden"}

        def preprocess(self, input_ids, input_embeds, **info):
            if input_ids.shape[0] == 1:
                h = info.get("last_talker_hidden")
                assert isinstance(h, torch.Tensor) and h.is_cuda and h.shape == (self.config.hidden_size,)
                seen.append((info["request_id"], h.clone(), info.get("steps_done")))
            return super().preprocess(input_ids, input_embeds, **info)

        def postprocess(self, hidden_states, multimodal_outputs=None, **info):
            return {"last_talker_hidden": hidden_states[-1], "steps_done": int(info.get("steps_done", 0)) + 1}

        def sample(self, logits, sampling_metadata):
            assert sampling_metadata.all_greedy and sampling_metadata.temperature is None
            assert len(sampling_metadata.output_token_ids) == logits.shape[0]
            # second-best token of every row: distinguishable from the runner's own (greedy) sampler
            return NS(sampled_token_ids=logits.float().topk(2, dim=-1).indices[:, 1:2])

    steps = 4
    cfg, runner = _runner(dev, StatefulTalker)
    sched = _scheduler()
    reqs = _requests(cfg, steps)
    for r in reqs:
        sched.add_request(r)
    trace = _drive(sched, runner)
    # the sampler hook decided every token
    for out in trace:
        for row, rid in enumerate(out.req_ids):
            assert len(out.sampled_token_ids[row]) == 1
    assert all(len(r.output_token_ids) == steps for r in reqs)
    # every decode step's preprocess saw the PREVIOUS step's last hidden row of that request, kept on the device,
    # and the step counter the postprocess hook maintains
    per_req = {"a": [], "b": []}
    for out in trace:
        for row, rid in enumerate(out.req_ids):
            per_req[rid].append(out.pooler_output[row]["hidden"][-1])
    got = {"a": [], "b": []}
    for rid, h, k in seen:
        got[rid].append((h, k))
    for rid in ("a", "b"):
        assert len(got[rid]) == steps - 1
        for j, (h, k) in enumerate(got[rid]):
            assert k == j + 1
            assert torch.equal(h.cpu(), per_req[rid][j])


def test_model_sampler_returning_none_falls_back_to_the_runner_sampler(dev):
    from vllm_omni_amd.talker import Qwen2_5OmniTalkerForConditionalGeneration as Talker

    class Deferring(Talker):
        prefer_model_sampler = True
        calls = 0

        def sample(self, logits, sampling_metadata):
            Deferring.calls += 1
            return None

    outs = []
    for cls in (None, Deferring):
        cfg, runner = _runner(dev, cls)
        assert not runner._generic_hooks          # hooks of the vectorised step: the graph / fast path stays
        sched = _scheduler()
        reqs = _requests(cfg, 3)
        for r in reqs:
            sched.add_request(r)
        _drive(sched, runner)
        outs.append([list(r.output_token_ids) for r in reqs])
    assert Deferring.calls > 0 and outs[0] == outs[1]

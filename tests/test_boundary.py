"""CPU: the drop-in boundary as the REFERENCE would drive it.

entry point -> ``register_platform()`` -> ``get_omni_ar_worker_cls()`` -> import by string ->
``worker_cls(vllm_config=..., local_rank, rank, distributed_init_method, is_driver_worker)``
(vllm_omni/engine/stage_init_utils.py:86-101, upstream ``Worker.__init__``) ->
``GPUARModelRunner(vllm_config, device)`` (vllm_omni/worker/gpu_ar_worker.py:102) ->
``model.load_weights(Iterable[(checkpoint name, tensor)])`` (qwen2_5_omni_talker.py:183-211,
qwen2_old.py:328-377) -> ``execute_model(upstream SchedulerOutput)``.
``vllm`` is not importable here, so the config objects are ``SimpleNamespace`` stand-ins carrying the fields
the reference reads.  The GPU half (same path, then one decode step) is tests/test_gpu_entry.py.
"""
import importlib
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import ref_talker


def hf_talker_config(**over):
    """Shaped like transformers' ``Qwen2_5OmniTalkerConfig`` (the reference's ``model_config.hf_config``)."""
    d = dict(model_type="qwen2_5_omni_talker", hidden_size=512, embedding_size=384, num_hidden_layers=2,
             num_attention_heads=8, num_key_value_heads=2, head_dim=64, intermediate_size=1024, vocab_size=512,
             rms_norm_eps=1e-6, rope_theta=1e6, max_position_embeddings=2048,
             rope_scaling={"mrope_section": [8, 12, 12], "rope_type": "default"},
             tts_codec_start_token_id=501, tts_codec_end_token_id=502, tts_codec_pad_token_id=500,
             tts_codec_mask_token_id=504)
    d.update(over)
    return NS(**d)


def make_vllm_config(hf=None, tp=1, **model_over):
    mc = dict(hf_config=hf or hf_talker_config(), dtype=torch.bfloat16, max_model_len=256, seed=7, enforce_eager=True,
              quantization=None, engine_output_type="latent")
    mc.update(model_over)
    return NS(model_config=NS(**mc),
              cache_config=NS(block_size=16, gpu_memory_utilization=0.5, kv_cache_memory_bytes=None),
              parallel_config=NS(tensor_parallel_size=tp, pipeline_parallel_size=1, world_size=tp,
                                 data_parallel_rank_local=0, data_parallel_index=0),
              scheduler_config=NS(max_num_seqs=8, max_num_batched_tokens=64),
              device_config=NS(device_type="cuda", device=torch.device("cuda")))


def _entry_point_target() -> str:
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "vllm-omni_amd", "pyproject.toml")).read()
    m = re.search(r'\[project\.entry-points\."vllm_omni\.platform_plugins"\]\s*\n\s*\w+\s*=\s*"([^"]+)"', txt)
    assert m, "entry point group vllm_omni.platform_plugins is missing from pyproject.toml"
    return m.group(1)


def test_entry_point_to_worker_construction_from_a_vllm_config():
    # 1. the entry point the reference's plugin loader calls (vllm_omni/plugins/__init__.py:24-58)
    mod_name, fn_name = _entry_point_target().split(":")
    plugin = getattr(importlib.import_module(mod_name), fn_name)
    qual = plugin()
    assert qual in (None, "vllm_omni_amd.platform.MI355XOmniPlatform")       # None here: no gfx950 GPU
    # 2. the platform names the worker by string; the reference imports it (stage_init_utils.py:86-101)
    from vllm_omni_amd.platform import PLATFORM_QUALNAME

    pm, pc = PLATFORM_QUALNAME.rsplit(".", 1)
    platform = getattr(importlib.import_module(pm), pc)
    wm, wc = platform.get_omni_ar_worker_cls().rsplit(".", 1)
    worker_cls = getattr(importlib.import_module(wm), wc)
    # 3. constructed exactly as upstream constructs workers
    vc = make_vllm_config()
    w = worker_cls(vllm_config=vc, local_rank=0, rank=0, distributed_init_method="tcp://127.0.0.1:29555",
                   is_driver_worker=True)
    c = w.config
    assert (c.model_config.hidden_size, c.model_config.num_attention_heads, c.model_config.num_key_value_heads,
            c.model_config.head_dim, c.model_config.embedding_size) == (512, 8, 2, 64, 384)
    assert c.model_config.mrope_section == [8, 12, 12] and c.model_config.rope_theta == 1e6
    assert c.model_config.input_projection and not c.model_config.qk_norm and c.model_config.attention_bias
    assert (c.block_size, c.max_num_seqs, c.max_model_len, c.max_num_batched_tokens) == (16, 8, 256, 64)
    assert c.dtype == torch.bfloat16 and c.enforce_eager and c.seed == 7 and c.gpu_memory_utilization == 0.5
    assert w.vllm_config is vc and w.distributed_init_method == "tcp://127.0.0.1:29555"
    # 4. no GPU here: init_device refuses the way the reference does (gpu_ar_worker.py:89-90)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="Not support device type"):
            w.init_device()
    with pytest.raises(TypeError):
        worker_cls(vllm_config=object())
    # data-parallel local rank shifts the device index, in init_device and only under the reference's condition:
    # DP_LOCAL_RANK * TP + TP_LOCAL_RANK unless ray / an external launcher (torchrun) already hands every worker its
    # device index, the DP backend is ray, or DP spans nodes (gpu_ar_worker.py:34-49)
    vc2 = make_vllm_config(tp=2)
    vc2.parallel_config.data_parallel_rank_local = 1
    w2 = worker_cls(vllm_config=vc2, local_rank=1, rank=3)
    assert w2.local_rank == 1 and w2.dp_adjusted_local_rank() == 3
    for field, value in (("distributed_executor_backend", "external_launcher"), ("distributed_executor_backend", "ray"),
                         ("data_parallel_backend", "ray"), ("nnodes_within_dp", 2)):
        vc3 = make_vllm_config(tp=2)
        vc3.parallel_config.data_parallel_rank_local = 1
        setattr(vc3.parallel_config, field, value)
        assert worker_cls(vllm_config=vc3, local_rank=1, rank=3).dp_adjusted_local_rank() == 1, field


def test_runner_is_constructed_from_vllm_config_and_device():
    """``GPUARModelRunner(self.vllm_config, self.device)`` (gpu_ar_worker.py:102)."""
    from vllm_omni_amd.model_runner import GPUARModelRunner

    r = GPUARModelRunner(make_vllm_config(), torch.device("cpu"))
    assert (r.max_num_reqs, r.max_model_len, r.block_size, r.max_num_batched_tokens) == (8, 256, 16, 64)
    assert r.config.vocab_size == 512 and r.dtype == torch.bfloat16 and r.enforce_eager
    assert r.engine_output_type == "latent" and r.max_reply_rows == 8 * 64
    with pytest.raises(TypeError):
        GPUARModelRunner(object(), torch.device("cpu"))
    with pytest.raises(ValueError, match="quantization"):
        GPUARModelRunner(make_vllm_config(quantization="fp8"), torch.device("cpu"))


def test_qwen3_omni_nested_hf_config_is_mapped():
    """Qwen3-Omni hands ``talker_config`` with a ``text_config`` inside (qwen3_omni_moe_talker.py:100-108)."""
    from vllm_omni_amd.compat import talker_config_from_hf

    text = NS(hidden_size=1024, num_hidden_layers=20, num_attention_heads=16, num_key_value_heads=2, head_dim=128,
              intermediate_size=2048, moe_intermediate_size=384, num_experts=128, num_experts_per_tok=8,
              norm_topk_prob=True, shared_expert_intermediate_size=768, vocab_size=3072, rms_norm_eps=1e-6,
              rope_theta=1e6, max_position_embeddings=32768, rope_scaling=None,
              rope_parameters={"rope_theta": 1e6, "mrope_section": [24, 20, 20]})
    hf = NS(model_type="qwen3_omni_moe_talker", text_config=text, tts_codec_start_token_id=None)
    c = talker_config_from_hf(NS(talker_config=hf, model_type="qwen3_omni_moe"))
    assert (c.hidden_size, c.head_dim, c.num_experts, c.num_experts_per_tok, c.moe_intermediate_size) == (1024, 128, 128, 8, 384)
    assert c.qk_norm and not c.input_projection and not c.attention_bias and c.norm_topk_prob
    assert c.mrope_section == [24, 20, 20] and c.embedding_size == 1024 and c.shared_expert_intermediate_size == 768


def checkpoint_names(cfg, fused: dict):
    """The fused test weights re-expressed as a released checkpoint names them: ``talker.`` prefix, codec_head,
    separate q/k/v and gate/up tensors — plus tensors a talker loader must skip."""
    d, nq, nkv = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads
    yield "thinker.model.layers.0.self_attn.q_proj.weight", torch.zeros(4, 4)        # skipped prefix
    yield "token2wav.code2wav.weight", torch.zeros(2)
    for name, w in fused.items():
        if name.startswith("language_model.model."):
            base = "talker.model." + name[len("language_model.model."):]
        elif name.startswith("language_model.lm_head."):
            base = "talker.codec_head." + name[len("language_model.lm_head."):]
        else:
            base = "talker." + name
        if "qkv_proj" in base:
            q, k, v = w.split([nq * d, nkv * d, nkv * d], dim=0)
            for sub, t in (("q_proj", q), ("k_proj", k), ("v_proj", v)):
                yield base.replace("qkv_proj", sub), t
        elif "gate_up_proj" in base:
            g, u = w.chunk(2, dim=0)
            yield base.replace("gate_up_proj", "gate_proj"), g
            yield base.replace("gate_up_proj", "up_proj"), u
        else:
            yield base, w
    yield "talker.model.layers.0.self_attn.rotary_emb.inv_freq", torch.zeros(d // 2)


@pytest.mark.parametrize("tp,rank", [(1, 0), (2, 1)])
def test_load_weights_takes_checkpoint_names_through_the_stacked_mapping(tp, rank):
    """Iterable[(checkpoint name, tensor)] must fill the same shards as the fused-name dict does, for every TP
    rank (q heads partitioned, KV heads partitioned / replicated, gate and up halves, row-parallel inputs)."""
    from vllm_omni_amd import distributed as D
    from vllm_omni_amd.talker import Qwen2_5OmniTalkerForConditionalGeneration as Talker, TalkerConfig

    cfg = TalkerConfig.from_json("mini_talker")
    fused = ref_talker.random_weights(cfg, seed=3)
    try:
        if tp > 1:
            D.simulate_tensor_parallel_rank(tp, rank)
        a = Talker(cfg, torch.bfloat16, torch.device("cpu"), seed=0)
        b = Talker(cfg, torch.bfloat16, torch.device("cpu"), seed=1)
        assert a.load_weights(fused) == set(fused.keys())
        consumed = []

        def stream():                                   # a generator: tensors are consumed one at a time
            for n, t in checkpoint_names(cfg, fused):
                consumed.append(n)
                yield n, t

        loaded = b.load_weights(stream())
        assert loaded == set(fused.keys())              # parameter (fused) names, as the reference returns
        assert any(n.startswith("thinker.") for n in consumed)
        for (na, wa), (nb_, wb) in zip(a.named_weights(), b.named_weights()):
            assert na == nb_ and torch.equal(wa, wb), na
        if tp > 1:                                      # rank 1 of 2: second half of the q heads, second KV head
            d = cfg.head_dim
            qkv = dict(b.named_weights())["language_model.model.layers.0.self_attn.qkv_proj.weight"]
            full = fused["language_model.model.layers.0.self_attn.qkv_proj.weight"]
            assert qkv.shape[0] == (4 + 2 * 1) * d and torch.equal(qkv[:4 * d], full[4 * d:8 * d])
            assert torch.equal(qkv[4 * d:5 * d], full[8 * d + d:8 * d + 2 * d])
        with pytest.raises(KeyError):
            b.load_weights([("talker.model.layers.0.self_attn.nonsense.weight", torch.zeros(1))])
    finally:
        D.destroy_distributed_environment()


def test_moe_checkpoint_expert_tensors_fill_the_stacked_parameters():
    from vllm_omni_amd.talker import Qwen2_5OmniTalkerForConditionalGeneration as Talker, TalkerConfig

    cfg = TalkerConfig.from_json("mini_moe_talker")
    fused = ref_talker.random_weights(cfg, seed=5)
    a = Talker(cfg, torch.bfloat16, torch.device("cpu"), seed=0)
    b = Talker(cfg, torch.bfloat16, torch.device("cpu"), seed=1)
    a.load_weights(fused)
    mi = cfg.moe_intermediate_size

    def stream():
        for n, t in checkpoint_names(cfg, {k: v for k, v in fused.items() if ".mlp.experts." not in k}):
            yield n, t
        for n, t in fused.items():
            if n.endswith(".mlp.experts.gate_up_proj"):
                base = "talker.model." + n[len("language_model.model."):-len(".gate_up_proj")]
                for e in range(cfg.num_experts):
                    yield f"{base}.{e}.gate_proj.weight", t[e, :mi]
                    yield f"{base}.{e}.up_proj.weight", t[e, mi:]
            elif n.endswith(".mlp.experts.down_proj"):
                base = "talker.model." + n[len("language_model.model."):-len(".down_proj")]
                for e in range(cfg.num_experts):
                    yield f"{base}.{e}.down_proj.weight", t[e]

    assert b.load_weights(stream()) == set(fused.keys())
    for (na, wa), (_, wb) in zip(a.named_weights(), b.named_weights()):
        assert torch.equal(wa, wb), na

    # a released Qwen3-Omni talker checkpoint also carries the thinker->talker resize MLPs and the code predictor
    # (hf_to_vllm_mapper, qwen3_omni_moe_talker.py:83-98): the former are the stage-input processor's (skipped by name
    # and recorded), the latter has its own loader — neither may raise or land in a decoder parameter
    c = Talker(cfg, torch.bfloat16, torch.device("cpu"), seed=2)

    def full_checkpoint():
        yield "talker.text_projection.linear_fc1.weight", torch.zeros(4, 4)
        yield "talker.text_projection.linear_fc1.bias", torch.zeros(4)
        yield "talker.hidden_projection.linear_fc2.weight", torch.zeros(4, 4)
        yield "talker.code_predictor.model.codec_embedding.0.weight", torch.zeros(4, 4)
        yield "thinker.model.layers.0.self_attn.q_proj.weight", torch.zeros(4, 4)
        yield "code2wav.decoder.0.weight", torch.zeros(4)
        yield from stream()

    assert c.load_weights(full_checkpoint()) == set(fused.keys())
    assert c.skipped_weights == ["talker.text_projection.linear_fc1.weight", "talker.text_projection.linear_fc1.bias",
                                 "talker.hidden_projection.linear_fc2.weight"]
    for (na, wa), (_, wc) in zip(a.named_weights(), c.named_weights()):
        assert torch.equal(wa, wc), na


def test_upstream_scheduler_output_is_normalised():
    """Upstream field names, msgspec-style payloads (engine/__init__.py:16-57), `resumed_from_preemption` flags,
    top_k = -1, missing omni extras."""
    from vllm_omni_amd.compat import normalize_scheduler_output
    from vllm_omni_amd.sched_output import SchedulerOutput

    emb = torch.arange(12, dtype=torch.float32).view(3, 4)
    bf = torch.randn(2, 4).to(torch.bfloat16)
    payload = NS(data=emb.numpy().tobytes(), shape=[3, 4], dtype="float32")
    info = NS(entries={"thinker_reply_part": NS(tensor_data=bf.view(torch.int16).numpy().tobytes(), tensor_shape=[2, 4],
                                                tensor_dtype="bfloat16", list_data=None, scalar_data=None),
                       "ids": NS(tensor_data=None, list_data=[1, 2], scalar_data=None),
                       "voice": NS(tensor_data=None, list_data=None, scalar_data="Chelsie")})
    sp = NS(temperature=0.9, top_k=-1, top_p=0.8, repetition_penalty=1.05, max_tokens=64, seed=None, stop_token_ids=None)
    up = NS(scheduled_new_reqs=[NS(req_id="n0", prompt_token_ids=[5, 6, 7], mm_features=[], sampling_params=sp,
                                   pooling_params=None, block_ids=([3, 9],), num_computed_tokens=0, lora_request=None,
                                   prompt_embeds=payload, additional_information=info)],
            scheduled_cached_reqs=NS(req_ids=["c0", "c1"], resumed_from_preemption=[False, True], new_token_ids=[],
                                     new_block_ids=[None, ([4],)], num_computed_tokens=[10, 0], num_output_tokens=[2, 0]),
            num_scheduled_tokens={"n0": 3, "c0": 1, "c1": 5}, total_num_scheduled_tokens=9,
            scheduled_spec_decode_tokens={}, scheduled_encoder_inputs={}, num_common_prefix_blocks=[0],
            finished_req_ids={"old"}, free_encoder_mm_hashes=[], kv_connector_metadata=None)
    so = normalize_scheduler_output(up)
    assert isinstance(so, SchedulerOutput) and normalize_scheduler_output(so) is so
    n0 = so.scheduled_new_reqs[0]
    assert torch.equal(n0.prompt_embeds, emb) and n0.block_ids == ([3, 9],) and n0.sampling_params.top_k == 0
    assert n0.sampling_params.temperature == 0.9 and n0.sampling_params.repetition_penalty == 1.05
    assert torch.equal(n0.additional_information["thinker_reply_part"], bf)
    assert n0.additional_information["ids"] == [1, 2] and n0.additional_information["voice"] == "Chelsie"
    c = so.scheduled_cached_reqs
    assert c.resumed_req_ids == {"c1"} and c.new_block_ids == [None, ([4],)] and c.num_computed_tokens == [10, 0]
    assert so.finished_req_ids == {"old"} and so.finished_requests_needing_kv_transfer == {} and so.new_block_ids_to_zero is None
    # and the runner's state update takes it (host logic only)
    from vllm_omni_amd.model_runner import GPUARModelRunner

    r = GPUARModelRunner(make_vllm_config(), torch.device("cpu"), max_reply_rows=8)
    r._reply_pool = torch.zeros(8, 4, dtype=torch.bfloat16)
    r.requests["c0"] = r.requests["c1"] = None
    so.scheduled_cached_reqs.req_ids, so.num_scheduled_tokens = [], {"n0": 3}
    r._update_states(so)
    assert r.input_batch.req_ids == ["n0"] and r.requests["n0"].reply_rows == 2


def test_talker_prompt_is_built_from_the_thinker_hand_over():
    """The reference hands a talker request over as ``additional_information{prompt_embeds, thinker_result, ...}`` and the
    model's preprocess hook builds the prompt on the prefill span (qwen2_5_omni.py:653-764): P thinker rows, one
    speaker + codec-pad row, one first-reply + codec-bos row; the rest of the reply is the per-step queue.  The product's
    ``thinker_to_talker_process`` against the oracle's restatement, bit for bit (the sums are single bf16 roundings),
    the speaker rows coming out of a checkpoint's ``thinker_embedding.weight``."""
    from vllm_omni_amd.talker import Qwen2_5OmniTalkerForConditionalGeneration as Talker, TalkerConfig

    cfg = TalkerConfig.from_json("mini_talker")
    w = ref_talker.random_weights(cfg, seed=4)
    m = Talker(cfg, torch.bfloat16, torch.device("cpu"), seed=0)
    g = torch.Generator().manual_seed(1)
    table = torch.zeros(151873, cfg.embedding_size, dtype=torch.bfloat16)     # a thinker embedding table (ids to 151872)
    for tok in (151860, 151870, 151872):
        table[tok] = torch.randn(cfg.embedding_size, generator=g).to(torch.bfloat16)
    loaded = m.load_weights(list(w.items()) + [("thinker_embedding.weight", table)])
    assert loaded == set(w.keys())                                            # the table is not a talker parameter
    assert set(m.thinker_special_embeddings) == {151860, 151870, 151872}
    P, K = 11, 5
    info = {"prompt_embeds": torch.randn(P, cfg.embedding_size, generator=g),                 # fp32, as the thinker stage sends them
            "thinker_result": torch.randn(K, cfg.embedding_size, generator=g),
            "prompt_token_ids": list(range(P)), "thinker_output_token_ids": [7, 8, 9, 10, 11]}
    for voice, tok in (("Chelsie", 151872), ("Ethan", 151870), ("nobody", 151860)):
        ids, rows, upd = m.thinker_to_talker_process(info, voice)
        r_ids, r_rows, r_reply = ref_talker.thinker_to_talker_prefill(
            cfg, w, table[tok], info["prompt_embeds"].to(torch.bfloat16), info["thinker_result"].to(torch.bfloat16),
            torch.bfloat16)
        assert ids == r_ids == [cfg.tts_codec_mask_token_id] * P + [cfg.tts_codec_pad_token_id, cfg.tts_codec_start_token_id]
        assert rows.shape == (P + 2, cfg.embedding_size) and torch.equal(rows.float(), r_rows)
        assert torch.equal(upd["thinker_reply_part"].float(), r_reply.float()) and upd["thinker_reply_part"].shape[0] == K - 1
    with pytest.raises(ValueError):
        m.thinker_to_talker_process({"prompt_embeds": info["prompt_embeds"], "thinker_result": torch.zeros(0, cfg.embedding_size)})
    m2 = Talker(cfg, torch.bfloat16, torch.device("cpu"), seed=0)
    with pytest.raises(RuntimeError, match="thinker's embedding table"):
        m2.thinker_to_talker_process(info)


# ------------------------------------------------------------------ per-request hook buffer (runner contract)
def _cpu_runner_with_request(rid="r1"):
    from vllm_omni_amd.model_runner import CachedRequestState, GPUARModelRunner
    from vllm_omni_amd.sched_output import SamplingParams
    from vllm_omni_amd.talker import TalkerConfig

    runner = GPUARModelRunner(TalkerConfig.from_json("mini_talker"), torch.device("cpu"), torch.bfloat16, max_num_reqs=4,
                              max_model_len=64, max_num_batched_tokens=32, block_size=16)
    runner.requests[rid] = CachedRequestState(rid, [1, 2], SamplingParams(), ([],), 0)
    return runner


def test_update_intermediate_buffer_rules_of_the_reference():
    """The four rules the reference's own tests pin for ``_update_intermediate_buffer``
    (tests/worker/test_omni_gpu_model_runner.py:171-243): tensors and lists are stored, successive updates accumulate
    keys, an empty update and an unknown request id are no-ops."""
    runner = _cpu_runner_with_request()
    runner._update_intermediate_buffer("r1", {"my_tensor": torch.tensor([1.0, 2.0]), "my_list": [3, 4]})
    buf = runner.model_intermediate_buffer["r1"]
    assert torch.allclose(buf["my_tensor"], torch.tensor([1.0, 2.0])) and buf["my_list"] == [3, 4]
    assert runner.requests["r1"].additional_information is buf

    runner = _cpu_runner_with_request()
    runner._update_intermediate_buffer("r1", {"a": torch.tensor([1.0])})
    runner._update_intermediate_buffer("r1", {"b": torch.tensor([2.0])})
    buf = runner.model_intermediate_buffer["r1"]
    assert torch.allclose(buf["a"], torch.tensor([1.0])) and torch.allclose(buf["b"], torch.tensor([2.0]))
    runner._update_intermediate_buffer("r1", {"a": torch.tensor([5.0])})          # a key's new value replaces the old one
    assert torch.allclose(runner.model_intermediate_buffer["r1"]["a"], torch.tensor([5.0]))

    runner = _cpu_runner_with_request()
    runner._update_intermediate_buffer("r1", {})
    assert "r1" not in runner.model_intermediate_buffer
    runner._update_intermediate_buffer("unknown_req", {"key": torch.tensor([1.0])})
    assert "unknown_req" not in runner.model_intermediate_buffer and "unknown_req" not in runner.requests

    # tensors are detached (`.detach().to("cpu").contiguous()`, as the reference stores them), list items too
    runner = _cpu_runner_with_request()
    src = torch.tensor([1.0, 2.0], requires_grad=True)
    runner._update_intermediate_buffer("r1", {"t": src, "l": [src, 7]})
    buf = runner.model_intermediate_buffer["r1"]
    assert not buf["t"].requires_grad and buf["t"].device.type == "cpu"
    assert not buf["l"][0].requires_grad and buf["l"][1] == 7

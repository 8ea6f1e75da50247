"""BASELINE config #1 — "OPT-125m greedy decode, batch=1, CPU backend (plumbing, no GPU)".

The reference has no CPU execution path and this product has none either (every op raises on a CPU
tensor), so config #1 is what SURVEY §8d says it is: a plumbing smoke test of the platform -> worker -> scheduler ->
runner wiring with a STAND-IN DEVICE.  Everything on the host side is product code — plugin entry point, platform
class, worker class resolved by name and built from a VllmConfig-shaped object, the KV block manager, the persistent
batch, block table / slot mapping, the two-phase execute_model / sample_tokens API, output packing.  The arithmetic of
the device (embedding fix-up + decoder stack + logits + argmax) is replaced, in this test only, by the CPU oracle
reading the SAME metadata and the SAME paged KV cache the kernels would.

Expectation: an independent oracle pass that re-prefills the whole sequence into a fresh, contiguous cache every
step (no persistent batch, no block reuse, no incremental KV) must pick the same greedy token wherever the top-2
logit margin is not a numerical tie.  Shape: OPT-125m's (12 layers, hidden 768, 12 heads of 64, FFN 3072, vocab 50272) on this repo's decoder
wiring (the reference's talker stack; OPT's own LayerNorm/ReLU block is not on the path being built).
"""
import importlib
from types import SimpleNamespace as NS

import torch

from oracle import ref_ops, ref_talker


def _opt125m_vllm_config():
    hf = NS(model_type="opt-125m-shaped", hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
            num_key_value_heads=12, head_dim=64, intermediate_size=3072, vocab_size=50272, rms_norm_eps=1e-5,
            rope_theta=10000.0, max_position_embeddings=2048, rope_scaling=None, embedding_size=768,
            tts_codec_start_token_id=None, tts_codec_end_token_id=None, tts_codec_pad_token_id=None,
            tts_codec_mask_token_id=None, attention_bias=True)
    return NS(model_config=NS(hf_config=hf, dtype="bfloat16", max_model_len=64, seed=0, enforce_eager=True, quantization=None,
                              engine_output_type="text"),
              cache_config=NS(block_size=16, gpu_memory_utilization=0.9, kv_cache_memory_bytes=None),
              parallel_config=NS(tensor_parallel_size=1, pipeline_parallel_size=1, world_size=1, data_parallel_rank_local=0),
              scheduler_config=NS(max_num_seqs=1, max_num_batched_tokens=64),
              device_config=NS(device_type="cuda"))


def test_opt125m_shaped_greedy_decode_through_the_plumbing():
    import vllm_omni_amd
    from vllm_omni_amd.kv_cache import KVCacheManager, normalize_layer_kv
    from vllm_omni_amd.platform import PLATFORM_QUALNAME
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, Request, SamplingParams, SchedulerOutput

    # --- platform -> worker class (by name) -> worker built from the config object
    assert vllm_omni_amd.register_platform() in (None, PLATFORM_QUALNAME)
    pm, pc = PLATFORM_QUALNAME.rsplit(".", 1)
    platform = getattr(importlib.import_module(pm), pc)
    wm, wc = platform.get_omni_ar_worker_cls().rsplit(".", 1)
    worker = getattr(importlib.import_module(wm), wc)(vllm_config=_opt125m_vllm_config(), local_rank=0, rank=0,
                                                      distributed_init_method=None, is_driver_worker=True)
    cfg = worker.config.model_config
    assert (cfg.num_hidden_layers, cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.vocab_size) == (12, 768, 12, 64, 50272)
    # --- the runner as the worker builds it, on the stand-in device
    from vllm_omni_amd.model_runner import GPUARModelRunner

    dev = torch.device("cpu")
    runner = GPUARModelRunner(worker.vllm_config, dev)
    runner.load_model()
    w = ref_talker.random_weights(cfg, seed=1)
    assert runner.model.load_weights(w) == set(w.keys())
    bs, nb = worker.config.block_size, 8
    runner.initialize_kv_cache(nb)
    csc = ref_ops.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
    fw = {k: v.float() for k, v in w.items()}
    cpu_caches = [normalize_layer_kv(kv) for kv in runner.kv_caches]     # the runner's own paged cache (host tensors here)

    # the stand-in device: the oracle computes what the HIP kernels would, from the runner's own buffers and metadata
    def decode_embeds(nd):
        ids = runner.last_token_ids.gpu[:nd]
        runner.inputs_embeds[:nd] = ref_talker.decode_embeds(fw, ids, None, torch.bfloat16).to(runner.dtype)

    def forward(n, total, md):
        hidden = ref_talker.talker_step(cfg, fw, runner.inputs_embeds[:total].float(), runner.positions.gpu[:total],
                                        cpu_caches, md.block_table, md.seq_lens, md.query_start_loc, md.slot_mapping, csc,
                                        torch.bfloat16)
        sample_h = hidden if total == n else hidden.index_select(0, runner.logits_indices.gpu[:n])
        logits, ids = ref_talker.logits_and_greedy(cfg, fw, sample_h, None, torch.bfloat16)
        runner.sampled_ids_gpu[:n] = ids
        return hidden.to(runner.dtype), logits, runner.sampled_ids_gpu[:n]

    runner._decode_embeds, runner._forward = decode_embeds, forward

    # --- scheduler side: one request, prompt prefilled in ONE step, then greedy decode one token per step
    prompt = [2, 100, 2000, 31, 7, 45000, 9, 12, 5, 77, 1234, 50271, 8, 3, 19, 4, 21]      # 17 tokens: crosses a block
    steps = 10
    mgr = KVCacheManager(nb, bs)
    req = Request("r0", prompt, SamplingParams(temperature=0.0, max_tokens=steps))
    got = []
    for step in range(steps):
        n_new = len(prompt) if step == 0 else 1
        new_blocks = mgr.allocate_slots(req, n_new)
        assert new_blocks is not None
        if step == 0:
            so = SchedulerOutput(scheduled_new_reqs=[NewRequestData("r0", prompt, req.sampling_params,
                                                                    mgr.get_block_ids("r0"), 0)],
                                 num_scheduled_tokens={"r0": n_new}, total_num_scheduled_tokens=n_new)
        else:
            so = SchedulerOutput(scheduled_cached_reqs=CachedRequestData(
                req_ids=["r0"], new_block_ids=[new_blocks.get_block_ids() if new_blocks.blocks[0] else None],
                num_computed_tokens=[req.num_computed_tokens], num_output_tokens=[len(got)]),
                num_scheduled_tokens={"r0": 1}, total_num_scheduled_tokens=1)
        assert runner.execute_model(so) is None
        out = runner.sample_tokens()
        assert out.req_ids == ["r0"] and out.pooler_output == [None]        # engine_output_type "text": ids only
        got.append(out.sampled_token_ids[0][0])
        req.num_computed_tokens += n_new
        req.output_token_ids.append(got[-1])
    # the second block was opened when the 17-token prompt crossed 16 and the table never re-ordered
    assert len(mgr.get_block_ids("r0")[0]) == 2 and runner.input_batch.block_table[0].num_blocks_per_row[0] == 2

    # --- independent expectation: re-prefill the whole sequence (teacher-forced with the tokens the plumbing produced)
    # into a fresh contiguous cache every step; the greedy token must agree wherever the top-2 logit margin is not a
    # numerical tie (random weights give near-ties; prefill and incremental decode reassociate fp32 sums differently)
    confident = 0
    for k in range(steps):
        seq = prompt + got[:k]
        t = len(seq)
        fresh = [(torch.zeros(4, bs, cfg.num_key_value_heads, cfg.head_dim), torch.zeros(4, bs, cfg.num_key_value_heads, cfg.head_dim))
                 for _ in range(cfg.num_hidden_layers)]
        table = torch.arange(1, 4, dtype=torch.int32).view(1, 3)
        pos = torch.arange(t)
        emb = ref_talker.decode_embeds(fw, torch.tensor(seq), None, torch.bfloat16)
        hidden = ref_talker.talker_step(cfg, fw, emb, pos, fresh, table, torch.tensor([t], dtype=torch.int32),
                                        torch.tensor([0, t], dtype=torch.int32),
                                        ref_ops.slot_mapping(table, torch.zeros(t, dtype=torch.long), pos, bs), csc, torch.bfloat16)
        logits, ids = ref_talker.logits_and_greedy(cfg, fw, hidden[-1:], None, torch.bfloat16)
        top2 = logits[0].topk(2).values
        if float(top2[0] - top2[1]) > 0.02:
            confident += 1
            assert got[k] == int(ids[0]), (k, got, int(ids[0]))
    assert confident >= steps // 2, confident
    mgr.free(req)
    assert mgr.block_pool.get_num_free_blocks() == nb - 1

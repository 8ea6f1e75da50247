"""The driver's hooks in ONE process, in the order build() -> smoke(): the package is imported (and the C-ABI library
loaded) before anything has touched torch.cuda — the order that once left two HIP runtimes in the process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_then_smoke_in_one_process():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "smoke ok" in r.stdout


def test_reference_style_bringup_then_one_decode_step(dev):
    """The drop-in path end to end on the GPU: plugin entry point -> platform -> worker class by
    string -> ``worker_cls(vllm_config=...)`` -> init_device / load_model(checkpoint-named weights) / KV sizing /
    initialize_cache -> ``execute_model(upstream-shaped SchedulerOutput)`` -> ``sample_tokens`` — and the step's
    hidden states against the oracle."""
    import importlib
    from types import SimpleNamespace as NS

    import torch

    from oracle import ref_ops, ref_talker
    from test_boundary import checkpoint_names, make_vllm_config

    import vllm_omni_amd

    qual = vllm_omni_amd.register_platform()
    assert qual == "vllm_omni_amd.platform.MI355XOmniPlatform"           # gfx950 + library present: activates
    pm, pc = qual.rsplit(".", 1)
    platform = getattr(importlib.import_module(pm), pc)
    wm, wc = platform.get_omni_ar_worker_cls().rsplit(".", 1)
    worker_cls = getattr(importlib.import_module(wm), wc)
    vc = make_vllm_config()
    vc.cache_config.kv_cache_memory_bytes = 64 << 20
    w = worker_cls(vllm_config=vc, local_rank=0, rank=0, distributed_init_method=None, is_driver_worker=True)
    w.init_device()
    cfg = w.config.model_config
    fused = ref_talker.random_weights(cfg, seed=3)
    w.load_model(weights=((n, t.to(dev)) for n, t in checkpoint_names(cfg, fused)))
    assert w.determine_available_memory() == 64 << 20
    # KV budget without the explicit override (base.py:77-156): utilisation x total minus what THIS process holds —
    # from the driver's per-process accounting (amdsmi, the counterpart of the reference's NVML per-PID query) when the
    # box lets us see it, else from the allocator
    w.config.kv_cache_memory_bytes = None
    free, total = torch.cuda.mem_get_info(dev)
    avail = w.determine_available_memory()
    acct = w.memory_accounting
    used = acct["amdsmi_process_bytes"] if acct["amdsmi_process_bytes"] else acct["profiled_bytes"]
    assert avail == max(0, int(total * w.config.gpu_memory_utilization) - used) and 0 < avail < total
    if acct["amdsmi_process_bytes"]:   # the two views of this (only) process agree to within the runtime's own pools
        assert abs(acct["amdsmi_process_bytes"] - acct["profiled_bytes"]) < (4 << 30), acct
    print("memory accounting:", acct)
    w.config.kv_cache_memory_bytes = 64 << 20
    nb = min(w.determine_num_available_blocks(), 64)
    assert nb >= 16
    w.initialize_cache(nb)
    bs = w.config.block_size
    # upstream-shaped scheduler output: one request, 12 tokens cached, decoding its 13th
    from vllm_omni_amd.kv_cache import normalize_layer_kv

    g = torch.Generator().manual_seed(5)
    cpu_caches = []
    for kv in w.model_runner.kv_caches:
        init = torch.randn(nb, 2, bs, cfg.num_key_value_heads, cfg.head_dim, generator=g).to(torch.bfloat16)
        kb, vb = normalize_layer_kv(kv)
        kb.copy_(init[:, 0].to(dev))
        vb.copy_(init[:, 1].to(dev))
        cpu_caches.append((init[:, 0].clone(), init[:, 1].clone()))
    sp = NS(temperature=0.0, top_k=-1, top_p=1.0, repetition_penalty=1.0, max_tokens=8, seed=None, stop_token_ids=None)
    up = NS(scheduled_new_reqs=[NS(req_id="r0", prompt_token_ids=list(range(13)), mm_features=[], sampling_params=sp,
                                   pooling_params=None, block_ids=([7],), num_computed_tokens=12, lora_request=None,
                                   prompt_embeds=None, additional_information=None)],
            scheduled_cached_reqs=NS(req_ids=[], resumed_req_ids=set(), new_token_ids=[], new_block_ids=[],
                                     num_computed_tokens=[], num_output_tokens=[]),
            num_scheduled_tokens={"r0": 1}, total_num_scheduled_tokens=1, finished_req_ids=set(),
            num_common_prefix_blocks=[0], free_encoder_mm_hashes=[])
    assert w.execute_model(up) is None
    out = w.sample_tokens(None)
    assert out.req_ids == ["r0"] and len(out.sampled_token_ids[0]) == 1
    got = out.pooler_output[0]["hidden"].float()
    pos = torch.tensor([12])
    table = torch.tensor([[7]], dtype=torch.int32)
    emb = ref_talker.decode_embeds(fused, torch.tensor([12]), None, torch.bfloat16)
    csc = ref_ops.build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
    hidden = ref_talker.talker_step(cfg, fused, emb, pos, cpu_caches, table, torch.tensor([13], dtype=torch.int32),
                                    torch.tensor([0, 1], dtype=torch.int32), ref_ops.slot_mapping(table, torch.tensor([0]), pos, bs),
                                    csc, torch.bfloat16)
    torch.testing.assert_close(got, hidden, rtol=2e-2, atol=2e-2)

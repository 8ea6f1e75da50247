"""Whole-model chunked prefill through the runner (BASELINE config #4 shape: one long prompt, chunked).

    python tools/bench_prefill_model.py [--prompt 8192 --chunk 2048 --config qwen2_5_omni_talker]

Schedules ONE request of --prompt tokens in chunks of --chunk tokens (what OmniARScheduler does with
max_num_batched_tokens = chunk), random-init weights, and reports prompt tokens/s plus the per-kernel-family
split from HIP events around the attention launches.  Every chunk runs the full decoder stack: embedding,
RMSNorm, QKV GEMM, RoPE + cache write, paged prefill attention against the growing cache, o_proj, MLP.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", type=int, default=8192)
    ap.add_argument("--chunk", type=int, default=2048)
    ap.add_argument("--config", default="qwen2_5_omni_talker")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--host-profile", action="store_true", help="cProfile of one prompt: where the host time of a step goes")
    a = ap.parse_args()
    from vllm_omni_amd import ops
    from vllm_omni_amd.kv_cache import KVCacheManager
    from vllm_omni_amd.model_runner import GPUARModelRunner
    from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, Request, SamplingParams, SchedulerOutput
    from vllm_omni_amd.talker import TalkerConfig

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if os.environ.get("MI_BLAS"):   # experiment: which library serves GEMM shapes TunableOp has no entry for
        torch.backends.cuda.preferred_blas_library(os.environ["MI_BLAS"])
    cfg = TalkerConfig.from_json(a.config)
    bs = 16
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=4, max_model_len=a.prompt + bs,
                              max_num_batched_tokens=a.chunk, block_size=bs, enforce_eager=True, seed=0)
    runner.load_model()
    nblocks = (a.prompt + bs) // bs * (a.iters + 2) + 8
    runner.initialize_kv_cache(nblocks)
    mgr = KVCacheManager(nblocks, bs)
    rng = np.random.default_rng(0)

    attn_events = []
    orig = ops.paged_attn_prefill

    def timed(*x, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*x, **k)
        e1.record()
        attn_events.append((e0, e1))
        return r

    # MI_PREFILL_STEP_MS=1: wall time of every step (a step at 80 ms among steps at 8 was the container's CPU quota being
    # exhausted by torch's intra-op thread pool: profiles/r02_host_threads.md; MI_PREFILL_HOST_TRACE=1 times the host side)
    step_ms = [] if os.environ.get("MI_PREFILL_STEP_MS") else None

    def one_prompt(i, instrument=False):
        r = Request(f"p{i}", rng.integers(0, cfg.vocab_size, size=a.prompt).tolist(), SamplingParams(temperature=0.0))
        done = 0
        first = True
        if instrument:
            ops.paged_attn_prefill = timed
        try:
            while done < a.prompt:
                n = min(a.chunk, a.prompt - done)
                new_blocks = mgr.allocate_slots(r, n)
                assert new_blocks is not None
                if first:
                    so = SchedulerOutput(scheduled_new_reqs=[NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params,
                                                                             mgr.get_block_ids(r.request_id), 0)],
                                         num_scheduled_tokens={r.request_id: n}, total_num_scheduled_tokens=n)
                    first = False
                else:
                    cached = CachedRequestData(req_ids=[r.request_id], new_token_ids=[], new_block_ids=[new_blocks.get_block_ids()],
                                               num_computed_tokens=[done], num_output_tokens=[0])
                    so = SchedulerOutput(scheduled_cached_reqs=cached, num_scheduled_tokens={r.request_id: n},
                                         total_num_scheduled_tokens=n)
                t_s = time.perf_counter()
                runner.execute_model(so)
                runner.sample_tokens()
                if step_ms is not None:
                    torch.cuda.synchronize()
                    step_ms.append(round((time.perf_counter() - t_s) * 1e3, 2))
                done += n
                r.num_computed_tokens = done
        finally:
            ops.paged_attn_prefill = orig
        so = SchedulerOutput(finished_req_ids={r.request_id}, num_scheduled_tokens={}, total_num_scheduled_tokens=0)
        runner.execute_model(so)
        mgr.free(r)

    one_prompt(0)                      # warm-up: TunableOp picks, allocator
    if os.environ.get("MI_PREFILL_HOST_TRACE"):
        # where does the HOST stall?  perf_counter around every decoder layer and every C-ABI / F.linear call inside it,
        # garbage-collector runs, and the caching allocator's device-malloc counter; prints the calls over 3 ms
        import gc
        import torch.nn.functional as F

        slow = []
        gc_t = {}

        def gc_cb(phase, info):
            if phase == "start":
                gc_t["t"] = time.perf_counter()
            else:
                slow.append(("gc gen%d" % info["generation"], (time.perf_counter() - gc_t["t"]) * 1e3))
        gc.callbacks.append(gc_cb)

        def wrap(owner, name, label):
            f = getattr(owner, name)

            def g(*x, **k):
                m0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
                t = time.perf_counter()
                r = f(*x, **k)
                d = (time.perf_counter() - t) * 1e3
                if d > 3.0:
                    slow.append((label, round(d, 2), "device mallocs +%d" % (torch.cuda.memory_stats().get("num_device_alloc", 0) - m0)))
                return r
            setattr(owner, name, g)
        for n in ("fused_moe", "moe_route", "silu_and_mul", "fused_add_rms_norm", "rms_norm", "paged_attn_prefill",
                  "rope_neox_and_cache"):
            if hasattr(ops, n):
                wrap(ops, n, "ops." + n)
        wrap(F, "linear", "F.linear")
        wrap(torch, "empty", "torch.empty")
        one_prompt(70)
        one_prompt(71)
        torch.cuda.synchronize()
        for x in slow:
            print("host stall:", x)
        return
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.iters):
        one_prompt(1 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    if a.host_profile:
        import cProfile
        import pstats

        pr = cProfile.Profile()
        pr.enable()
        one_prompt(50)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    one_prompt(99, instrument=True)
    torch.cuda.synchronize()
    attn_ms = sum(e0.elapsed_time(e1) for e0, e1 in attn_events)
    nl = cfg.num_hidden_layers
    d = cfg.head_dim
    attn_flops = nl * sum(2 * 2 * d * cfg.num_attention_heads * c * (i * a.chunk + (c + 1) / 2)
                          for i, c in enumerate([min(a.chunk, a.prompt - s) for s in range(0, a.prompt, a.chunk)]))
    if step_ms:
        print("wall ms per step (every prompt, in order):", step_ms)
    line = f"{a.config}: prompt {a.prompt} in chunks of {a.chunk}: {dt * 1e3:.1f} ms = {a.prompt / dt:,.0f} prompt tokens/s; "
    if attn_ms > 0:
        line += (f"attention {attn_ms:.1f} ms ({attn_ms / (dt * 1e3):.0%} of the prefill, {attn_flops / attn_ms / 1e9:.0f} TFLOP/s over "
                 f"{len(attn_events)} eager launches)")
    else:
        line += "no eager attention launch was seen"
    print(line)


if __name__ == "__main__":
    main()

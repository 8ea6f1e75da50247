#!/usr/bin/env python3
"""Headline benchmark: talker decode tokens/s (+ achieved HBM GB/s of the paged-attention decode
kernel) at BASELINE.json's shape — Qwen2.5-Omni-7B talker, global batch 256, ctx 4096, bf16.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one engine step of the decode hot path over the whole batch: scheduler output ->
``GPUARModelRunner.execute_model`` (persistent-batch update, input prep, 28 decoder layers with the
HIP kernels, logits, mask, argmax) -> ``sample_tokens`` (ids + hidden states back on the host).
Synthetic data: random-init weights of the named architecture, KV cache pre-filled N(0,1),
non-contiguous block tables, every sequence at exactly ``ctx`` tokens and NOT growing during the
timed region, so bytes/step are constant.  Inputs are resident in HBM when the
timed region starts.

Parallelism for N > 1: the global batch is fixed (strong scaling).  The HEADLINE value is the widest tensor-parallel
plan the model allows — tp2, tp4, and dp2 x tp4 at 8 GPUs (28 query heads do not divide by 8,
vllm_omni/model_executor/models/qwen2_5_omni/qwen2_old.py:91): the "TP scaling" line north_star asks for, RCCL
all-reduces inside the step.  Decode sequences are independent, so pure data parallelism (dpN, no data-path collective)
is also a valid way to use N GPUs: it is measured in the same run, right after, and reported beside the headline as
`alt_plans` — nothing is estimated from tables.  MI_BENCH_PLAN=tpT|dp runs only that plan; MI_BENCH_ALT=0 skips the
alternative.  `python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run on
127.0.0.1) before anything touches the GPU.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from typing import Optional

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="GLOBAL batch (sequences decoded per step)")
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--config", default="qwen2_5_omni_talker")
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--kv-layout", default="NB2_HEAD_MAJOR")
    ap.add_argument("--eager", action="store_true", help="no hipGraph replay")
    ap.add_argument("--sync-output", action="store_true",
                    help="wait for every step's ids + hidden states on the host before launching the next step (the "
                         "default pipelines them one step behind: sample_tokens(async_output=True))")
    ap.add_argument("--code-predictor", action="store_true",
                    help="Qwen3-Omni: run the code predictor inside every decode step (talker_mtp); needs a config with "
                         "input_projection=false, e.g. --config qwen3_omni_talker")
    ap.add_argument("--simulate-shard", default=None, metavar="TP,DP",
                    help="ONE GPU runs rank 0 of a (tp, dp) plan with the collectives stubbed: per-rank compute time of "
                         "that plan, i.e. its step time without communication.  A sizing estimate, flagged in the JSON")
    ap.add_argument("--quantization", default=None, choices=[None, "int8"],
                    help="W8A8 decoder linears (BASELINE config #5 family; NOT the headline: the default line is bf16)")
    ap.add_argument("--roofline-steps", type=int, default=3, help="instrumented eager steps for per-launch timing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic", default=os.environ.get("MI_BENCH_TRAFFIC", "pmc"), choices=["pmc", "recorded", "none"],
                    help="roofline.traffic: 'pmc' = measure in THIS run (two child passes of this script under "
                         "`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, before the parent touches the GPU; N=1, "
                         "headline workload only), 'recorded' = the figure kept in profiles/, 'none' = null")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the profiled pass itself
    ap.add_argument("--cpu-baseline-batch", type=int, default=4)
    a = ap.parse_args()
    if a.pmc_child:   # a few steps of the same workload, nothing else: the parent reads the counters of the decode launches
        a.steps, a.warmup, a.no_cpu_baseline, a.roofline_steps, a.traffic = 2, 1, True, 0, "none"
    return a


def measure_decode_traffic_pmc(argv_tail: list, timeout_s: float = 170.0):
    """HBM bytes per launch of the decode kernel from PMC counters, collected as MI355X_MICROARCH.md's HBM section
    prescribes: two SEPARATE `rocprofv3 --pmc` passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), no trace domains;
    FETCH_SIZE / WRITE_SIZE are KiB, and on gfx950 FETCH_SIZE reports exactly half the bytes of wide (16 B per lane)
    coalesced streaming reads, so it is doubled; WRITE_SIZE is exact.  Each pass is a CHILD process running this script
    (`--pmc-child`: 1 warm-up + 2 steps of the same workload); the parent has not touched the GPU yet.
    Returns (bytes per launch, description) or (None, reason)."""
    import csv
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    got = {}
    t0 = time.monotonic()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"mi_pmc_{counter.lower()}_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "-d", d, "-o", "run", "--output-format", "csv", "--", sys.executable,
               os.path.abspath(__file__), "--pmc-child"] + argv_tail
        try:
            r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                               timeout=timeout_s, cwd="/tmp")
        except subprocess.TimeoutExpired:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"rocprofv3 --pmc {counter} pass exceeded {timeout_s:.0f} s"
        if r.returncode != 0:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"rocprofv3 --pmc {counter} pass exited {r.returncode}: {r.stderr.decode(errors='replace')[-200:]}"
        per_dispatch: dict = {}
        for root, _, files in os.walk(d):
            for f in files:
                if f.endswith("counter_collection.csv"):
                    with open(os.path.join(root, f)) as fh:
                        for row in csv.DictReader(fh):
                            if "decode_mfma_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                                k = row["Dispatch_Id"]
                                per_dispatch[k] = per_dispatch.get(k, 0.0) + float(row["Counter_Value"])
        shutil.rmtree(d, ignore_errors=True)
        if not per_dispatch:
            return None, f"no {counter} rows for the decode kernel"
        got[counter] = (sum(per_dispatch.values()) / len(per_dispatch), len(per_dispatch))
    hbm = int(2 * got["FETCH_SIZE"][0] * 1024 + got["WRITE_SIZE"][0] * 1024)
    return hbm, (f"measured in this run: two child passes under rocprofv3 --pmc (FETCH_SIZE mean {got['FETCH_SIZE'][0]:.0f} KiB "
                 f"over {got['FETCH_SIZE'][1]} decode launches, doubled per the guide's gfx950 correction; WRITE_SIZE mean "
                 f"{got['WRITE_SIZE'][0]:.1f} KiB over {got['WRITE_SIZE'][1]}), {time.monotonic() - t0:.0f} s")


def parallel_plan(n_gpus: int, cfg) -> tuple[int, int, str]:
    """(tp, dp, label) of the widest tensor-parallel plan: TP must divide the query heads; what is left
    becomes data parallel."""
    tp = n_gpus
    while tp > 1 and (cfg.num_attention_heads % tp or (cfg.num_key_value_heads < tp and tp % cfg.num_key_value_heads)
                      or (cfg.num_key_value_heads >= tp and cfg.num_key_value_heads % tp)):
        tp //= 2
    dp = n_gpus // tp
    return tp, dp, plan_label(tp, dp)


def _code_predictor_config():
    from vllm_omni_amd.code_predictor import CodePredictorConfig

    return CodePredictorConfig()


def plan_label(tp: int, dp: int) -> str:
    return f"tp{tp}" if dp == 1 else (f"dp{dp}" if tp == 1 else f"dp{dp}xtp{tp}")


def plans_to_run(world: int, cfg, batch: int) -> list[tuple[int, int, str, str]]:
    """[(tp, dp, label, role)]: the widest tensor-parallel plan first (role "headline": the TP scaling line), then
    pure data parallelism (role "alt") when it is a different plan.  MI_BENCH_PLAN=tpT|dp|tpmax runs one plan only."""
    tp_max, dp_max, label = parallel_plan(world, cfg)
    forced = os.environ.get("MI_BENCH_PLAN", "")
    if forced == "dp":
        return [(1, world, plan_label(1, world), "headline (forced data parallel)")]
    if forced.startswith("tp") and forced != "tpmax":
        t = int(forced[2:])
        if world % t or t > tp_max:
            raise SystemExit(f"MI_BENCH_PLAN={forced}: not a valid tensor-parallel degree for {world} GPUs")
        return [(t, world // t, plan_label(t, world // t), f"headline (forced {forced})")]
    out = [(tp_max, dp_max, label, "headline: widest tensor-parallel plan")]
    if forced != "tpmax" and tp_max > 1 and batch % world == 0 and os.environ.get("MI_BENCH_ALT", "1") == "1":
        out.append((1, world, plan_label(1, world), "alt: pure data parallel (no data-path collective)"))
    return out


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) with torch.distributed.run on
    127.0.0.1 — BEFORE this process has touched the GPU — and pass their exit code on.  Counting devices does not
    initialise the runtime."""
    import socket
    import subprocess

    rehearsal = os.environ.get("MI_BENCH_REHEARSAL") == "1"
    have = torch.cuda.device_count()
    if not rehearsal and have < n:
        print(f"[bench] --gpus {n} but only {have} GPU(s) are visible: refusing to report a {have}-GPU number as "
              f"n_gpus={n}", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    # torch.distributed.run answers 1 for ANY failing rank: the ranks leave their own code (4 = a stage hung or raised
    # after the line was printed) in a file, and this launcher passes that on
    import tempfile

    code_file = os.path.join(tempfile.gettempdir(), f"mi_bench_exit_{os.getpid()}_{port}")
    rc = subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MI_BENCH_EXIT_FILE=code_file)).returncode
    try:
        if rc != 0 and os.path.exists(code_file):
            rc = int(open(code_file).read().strip() or rc)
    except (OSError, ValueError):
        pass
    finally:
        try:
            os.remove(code_file)
        except OSError:
            pass
    return rc


def leave(code: int) -> None:
    """os._exit with the code also left where a self-spawned launcher (``spawn_ranks``) can read it."""
    f = os.environ.get("MI_BENCH_EXIT_FILE")
    if f:
        try:
            with open(f, "w") as fh:
                fh.write(str(code))
        except OSError:
            pass
    os._exit(code)


class SyntheticScheduler:
    """Feeds the runner what ``OmniARScheduler.schedule()`` would for a steady decode batch, using the
    real block manager for the block ids.  Sequences sit at ``ctx`` tokens and do not grow."""

    def __init__(self, batch: int, ctx: int, block_size: int, vocab: int, rank_seed: int):
        from vllm_omni_amd.kv_cache import KVCacheManager
        from vllm_omni_amd.sched_output import Request, SamplingParams

        self.batch, self.ctx = batch, ctx
        blocks_per_seq = -(-ctx // block_size)
        self.num_blocks = batch * blocks_per_seq + 1
        self.kv = KVCacheManager(self.num_blocks, block_size)
        # fragment the pool so block tables are NON-contiguous: take every block, free in random order
        all_ids = self.kv.block_pool.get_new_blocks(self.num_blocks - 1)
        rng = np.random.default_rng(2 + rank_seed)
        self.kv.block_pool.free_blocks(rng.permutation(all_ids).tolist())
        self.requests = []
        for i in range(batch):
            r = Request(f"req-{i}", rng.integers(0, vocab, size=ctx).tolist(), SamplingParams(temperature=0.0))
            blocks = self.kv.allocate_slots(r, ctx)
            assert blocks is not None
            r.num_computed_tokens = ctx - 1  # everything but the token decoded this step is cached
            self.requests.append(r)
        self._first = True

    def schedule(self):
        from vllm_omni_amd.sched_output import CachedRequestData, NewRequestData, SchedulerOutput

        self.kv.new_step_starts()
        n = self.batch
        sched = {r.request_id: 1 for r in self.requests}
        if self._first:
            self._first = False
            new = [NewRequestData(r.request_id, r.prompt_token_ids, r.sampling_params,
                                  self.kv.get_block_ids(r.request_id), r.num_computed_tokens) for r in self.requests]
            return SchedulerOutput(scheduled_new_reqs=new, num_scheduled_tokens=sched, total_num_scheduled_tokens=n)
        cached = CachedRequestData(req_ids=[r.request_id for r in self.requests], new_token_ids=[],
                                   new_block_ids=[None] * n,
                                   num_computed_tokens=[self.ctx - 1] * n, num_output_tokens=[1] * n)
        return SchedulerOutput(scheduled_cached_reqs=cached, num_scheduled_tokens=sched, total_num_scheduled_tokens=n)


def fill_kv_cache(runner, seed: int) -> None:
    from vllm_omni_amd.kv_cache import normalize_layer_kv

    g = torch.Generator(device=runner.device).manual_seed(seed)
    for kv in runner.kv_caches:
        kb, vb = normalize_layer_kv(kv)
        base = kv if isinstance(kv, torch.Tensor) else kb._base
        base.normal_(0.0, 1.0, generator=g)


def time_decode_launches(runner, steps: int):
    """Instrumented pass: HIP events around every paged-attention decode CALL of an eager step (on the stream the kernel
    is launched on), same inputs as the timed region.  A call = the attention launch (with RoPE + KV write inside on
    pure-decode steps) + the partition-merge launch where the context is partitioned, and the interval includes the
    host's launch latency — on a 45 us kernel that is 10-20 %, so this figure is reported beside the kernel-only one of
    ``time_decode_kernel``, not as the roofline figure.  Returns (avg ms, calls)."""
    from vllm_omni_amd import ops

    events: list[tuple[torch.cuda.Event, torch.cuda.Event]] = []
    origs = {n: getattr(ops, n) for n in ("paged_attn_decode", "rope_cache_paged_attn_decode")}

    def wrap(orig):
        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            events.append((e0, e1))
            return r
        return timed

    for n, f in origs.items():
        setattr(ops, n, wrap(f))
    was = runner.enforce_eager
    runner.enforce_eager = True
    try:
        sched = runner._bench_sched
        for _ in range(steps):
            runner.execute_model(sched.schedule())
            runner.sample_tokens()
        torch.cuda.synchronize()
    finally:
        for n, f in origs.items():
            setattr(ops, n, f)
        runner.enforce_eager = was
    ms = [a.elapsed_time(b) for a, b in events]
    return (float(np.mean(ms)) if ms else float("nan")), len(ms)


def time_decode_kernel(runner, reps: int = 4):
    """The decode attention KERNEL alone, on the step's own inputs: one launch per decoder layer — that layer's KV cache,
    the step's block tables, sequence lengths, positions and slots as they sit in the runner's device buffers after the
    timed region, the launch bound and partition choice of the captured step, RoPE + KV write inside exactly as the
    step runs it — issued back to back, HIP events around each pass of L launches on the launching stream.  Back to
    back: the host's launch latency hides behind the previous launch (as it does inside the captured graph) instead
    of sitting inside the interval; every layer has its own KV cache (GBs apart), so nothing is re-read from a cache.
    The partition-merge launch is left out (MI_DECODE_SKIP_MERGE), so the figure is the kernel rocprofv3's kernel trace
    reports under the same name.  Returns (avg ms per launch, launches timed, variant)."""
    from vllm_omni_amd import ops

    layers = getattr(runner.model, "language_model", runner.model).model.layers
    attn0 = layers[0].self_attn
    n = runner.input_batch.num_reqs
    nq, d = attn0.attn.num_heads, attn0.attn.head_size
    bt = runner.input_batch.block_table[0]
    table, lens = bt.block_table.gpu[:n], runner.seq_lens.gpu[:n]
    pos, slots = runner.positions.gpu[:n], bt.slot_mapping.gpu[:n]
    bound = runner.max_model_len
    g = torch.Generator(device=runner.device).manual_seed(7)
    qkv = torch.randn(n, (nq + 2 * attn0.attn.num_kv_heads) * d, device=runner.device, generator=g).to(torch.bfloat16)
    q, k, v = qkv.split([nq * d, attn0.attn.num_kv_heads * d, attn0.attn.num_kv_heads * d], dim=-1)
    out = torch.empty(n, nq * d, dtype=torch.bfloat16, device=runner.device)
    rope = attn0.rotary_pos_emb
    fused = ops.rope_decode_fusable(q, attn0.attn._kb, pos, rope.cos_sin_cache.shape[1], 0, n, bound)

    def one_pass():
        for ly in layers:
            at = ly.self_attn.attn
            if fused:
                sa = ly.self_attn
                ops.rope_cache_paged_attn_decode(out, q, k, v, pos, rope.cos_sin_cache, at._kb, at._vb, slots, table, lens,
                                                 bound, at.scale, workspace=runner._decode_ws, _skip_merge=True,
                                                 q_norm_weight=sa.q_norm.weight if sa.q_norm else None,
                                                 k_norm_weight=sa.k_norm.weight if sa.k_norm else None, norm_eps=sa.norm_eps)
            else:
                ops.paged_attn_decode(out, q, at._kb, at._vb, table, lens, bound, at.scale, workspace=runner._decode_ws,
                                      _skip_merge=True)

    with torch.inference_mode():
        one_pass()
        torch.cuda.synchronize()
        ev = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            one_pass()
            e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
    total = sum(a.elapsed_time(b) for a, b in ev)
    return total / (reps * len(layers)), reps * len(layers), ("RoPE + KV write inside" if fused else "attention only")


def cpu_baseline(cfg, batch: int, ctx: int, block_size: int, budget_s: float = 20.0):
    """The oracle (CPU restatement of the reference's decode-step semantics — the reference has no CPU execution
    path) timed on this box's host cores on a BOUNDED SAMPLE OF THE SAME WORKLOAD: the whole decode step
    — every one of the model's layers, lm_head, mask, argmax — at the full context length, for ``batch`` sequences
    instead of 256.  Nothing is extrapolated: value = batch / measured step time.  (All layers run the same
    random-init weight tensors: 28 distinct fp32 layers would be 26 GB of host memory and a minute of RNG for an
    identical amount of arithmetic.)"""
    from oracle import ref_ops, ref_talker

    threads = max(1, min(os.cpu_count() or 1, 16))
    torch.set_num_threads(threads)
    nl = cfg.num_hidden_layers
    w1 = {k: v.float() for k, v in ref_talker.random_weights(cfg, seed=0, dtype=torch.bfloat16, num_layers=1).items()}
    w = dict(w1)
    p0 = "language_model.model.layers.0."
    for li in range(1, nl):
        for k, v in w1.items():
            if k.startswith(p0):
                w[f"language_model.model.layers.{li}." + k[len(p0):]] = v
    bps = -(-ctx // block_size)
    nb = batch * bps + 1
    d, nkv = cfg.head_dim, cfg.num_key_value_heads
    # every layer has its own KV cache (bf16 values held in fp32, what the oracle's attention reads)
    kvs = [(torch.randn(nb, block_size, nkv, d), torch.randn(nb, block_size, nkv, d)) for _ in range(nl)]
    table = (torch.randperm(nb - 1)[: batch * bps] + 1).view(batch, bps).to(torch.int32)
    seq_lens = torch.full((batch,), ctx, dtype=torch.int32)
    qsl = torch.arange(batch + 1, dtype=torch.int32)
    pos = torch.full((batch,), ctx - 1, dtype=torch.int64)
    slots = ref_ops.slot_mapping(table, torch.arange(batch), pos, block_size)
    csc = ref_ops.build_cos_sin_cache(d, ctx + 1, cfg.rope_theta)
    emb = torch.randn(batch, cfg.embedding_size)

    def run():
        h = ref_talker.talker_step(cfg, w, emb, pos, kvs, table, seq_lens, qsl, slots, csc, None)
        return ref_talker.logits_and_greedy(cfg, w, h, None)

    run()
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or time.perf_counter() - t0 < budget_s:
        run()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": batch / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"torch-fp32 oracle of the WHOLE decode step ({nl} layers + lm_head + mask + argmax) for {batch} "
                      f"sequences at ctx {ctx} (the headline batch is 256): {reps} steps timed, {dt * 1e3:.0f} ms/step, "
                      "no extrapolation; CPU restatement of reference semantics, not a reference CPU backend"}


class StageSkipped(RuntimeError):
    """A staged attempt declined to run (agreed by all ranks, nothing left in a bad state): the next stage may go on."""


def validate_in_tree_allreduce(dev, rows: int, cols: int) -> None:
    """Before anything is timed on the in-tree all-reduce (csrc/allreduce.hip) it must reproduce, on THESE ranks, what the
    transport every other plan used gives: the decode step's message, a ragged one, eager and replayed from a hipGraph
    with fresh inputs, against an fp32 all-reduce over the TP group (its sum is rank-ordered fp32 rounded once, the
    library's order is its own: one 16-bit ulp of slack).  The verdict is agreed over the group, so either every rank
    raises or none does."""
    from vllm_omni_amd.distributed import get_parallel_state

    st = get_parallel_state()
    ar = st.custom_ar
    ok = ar is not None
    if ok:
        def agree(got: torch.Tensor, x: torch.Tensor) -> bool:
            want = x.float()
            dist.all_reduce(want, group=st.tp_group)
            return bool(((got.float() - want).abs() <= want.abs() * 2.0 ** -7 + 1e-3).all())

        g = torch.Generator(device=dev).manual_seed(1234 + st.tp_rank)
        for shape in ((rows, cols), (13, 8), (1, 8)):
            x = torch.randn(*shape, device=dev, generator=g).to(torch.bfloat16)
            ok = ok and agree(ar.all_reduce(x.clone()), x)
        buf = torch.zeros(rows, cols, dtype=torch.bfloat16, device=dev)
        out = torch.zeros_like(buf)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ar.all_reduce(buf, out)
        torch.cuda.current_stream().wait_stream(s)
        from vllm_omni_amd.model_runner import no_gc_during_capture

        gr = torch.cuda.CUDAGraph()
        with no_gc_during_capture(), torch.cuda.graph(gr):
            ar.all_reduce(buf, out)
        for _ in range(3):
            x = torch.randn(rows, cols, device=dev, generator=g).to(torch.bfloat16)
            buf.copy_(x)
            gr.replay()
            torch.cuda.synchronize()
            ok = ok and agree(out, x)
        try:
            ar.check()
        except Exception:  # noqa: BLE001
            ok = False
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=st.tp_group)
    if flag.item() < 1.0:
        raise StageSkipped("the in-tree all-reduce did not reproduce the library all-reduce on these ranks")


def run_plan(a, cfg, dev, world: int, rank: int, tp: int, dp: int, rehearsal: bool, eager: bool = False,
             allreduce: str = "rccl"):
    """Build the runner of one (tp, dp) plan on this rank, warm up, time EXACTLY ``a.steps`` steps between barriers,
    return (max-over-ranks seconds, runner, local batch)."""
    from vllm_omni_amd.distributed import init_distributed_environment
    from vllm_omni_amd.model_runner import GPUARModelRunner

    if world > 1:
        os.environ["VLLM_OMNI_AMD_ALLREDUCE"] = allreduce
        init_distributed_environment(world, rank, tp, backend="gloo" if rehearsal else "nccl",
                                     device=None if rehearsal else dev)
        if allreduce == "custom" and tp > 1:
            # inference mode like the runner's own captures: the capture bookkeeping of the default generator was created
            # under it by the earlier plans and may only be updated under it
            with torch.inference_mode():
                validate_in_tree_allreduce(dev, a.batch // dp, cfg.hidden_size)
    local_batch = a.batch // dp
    max_model_len = a.ctx + a.block_size
    runner = GPUARModelRunner(cfg, dev, torch.bfloat16, max_num_reqs=local_batch, max_model_len=max_model_len,
                              max_num_batched_tokens=max(local_batch, 256), block_size=a.block_size,
                              kv_layout=a.kv_layout, enforce_eager=a.eager or eager, seed=0, quantization=a.quantization,
                              code_predictor=_code_predictor_config() if a.code_predictor else None,
                              copy_outputs=False)   # views of the pinned output ring
    runner.load_model()
    # every rank of a TP group must see the same schedule: seed by data-parallel replica only
    sched = SyntheticScheduler(local_batch, a.ctx, a.block_size, cfg.vocab_size, rank_seed=rank // tp)
    runner.initialize_kv_cache(sched.num_blocks)
    fill_kv_cache(runner, seed=1)
    runner._bench_sched = sched
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(k: int, stamps: Optional[list] = None) -> None:
        """k engine steps; every step's output (ids + hidden states) is on the host when this returns.  Pipelined
        mode retrieves step i's output after step i+1 has been launched (upstream's async scheduling).  ``stamps``
        receives the wall clock at which each step's output reached the host (per-step distribution)."""
        pending = None
        for _ in range(k):
            runner.execute_model(sched.schedule())
            out = runner.sample_tokens(async_output=not a.sync_output)
            if pending is not None:
                pending.get_output()
                if stamps is not None:
                    stamps.append(time.perf_counter())
            elif a.sync_output and stamps is not None:
                stamps.append(time.perf_counter())
            pending = None if a.sync_output else out
        if pending is not None:
            pending.get_output()
            if stamps is not None:
                stamps.append(time.perf_counter())

    run_steps(a.warmup)
    barrier()
    stamps: list = []
    t0 = time.perf_counter()
    run_steps(a.steps, stamps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # per-step wall times of THIS rank inside the timed region: the interval between consecutive outputs reaching the
    # host (the first one, which also holds the pipeline fill, is measured from the start of the region)
    runner._bench_step_ms = [(b - a_) * 1e3 for a_, b in zip([t0] + stamps[:-1], stamps)]
    return elapsed, runner, local_batch


def step_distribution(step_ms: list) -> dict:
    """p10 / p50 / p90 of the per-step wall times recorded inside the timed region (median
    with p10 / p90, so that box-to-box spread can be told from a regression); `ms_per_step` stays the mean of the
    region, which is what `value` is computed from."""
    if len(step_ms) < 2:
        return {}
    v = sorted(step_ms[1:]) if len(step_ms) > 4 else sorted(step_ms)   # the first interval also holds the pipeline fill
    q = lambda f: v[min(len(v) - 1, max(0, int(round(f * (len(v) - 1)))))]   # noqa: E731
    return {"ms_per_step_p10": round(q(0.10), 3), "ms_per_step_p50": round(q(0.50), 3), "ms_per_step_p90": round(q(0.90), 3),
            "ms_per_step_how": f"rank 0, {len(v)} intervals between consecutive step outputs reaching the host inside "
                               "the timed region (first interval excluded: pipeline fill)"}


def time_allreduce(dev, rows: int, cols: int, calls: int = 50) -> dict:
    """The decode step's tensor-parallel exchange on its own: ``calls`` back-to-back all-reduces of the step's message
    ([rows, hidden] bf16, what o_proj / down_proj of every layer reduce) through the product's own call
    (``tensor_model_parallel_all_reduce``: RCCL or the in-tree kernel, whichever the plan runs), wall clock around the
    burst + synchronize, eager (inside the captured step the same launches carry no host cost).  Every rank of the
    group calls it the same number of times."""
    from vllm_omni_amd.distributed import tensor_model_parallel_all_reduce

    x = torch.randn(rows, cols, device=dev, dtype=torch.bfloat16)
    with torch.inference_mode():
        for _ in range(5):
            tensor_model_parallel_all_reduce(x.clone())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            tensor_model_parallel_all_reduce(x)     # in place on the same buffer: values do not matter, timing does
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"message_bytes": rows * cols * 2, "us_per_message": round(dt / calls * 1e6, 1), "calls": calls,
            "how": "back-to-back eager calls of the product's all-reduce on the step's [rows, hidden] bf16 message"}


def main():
    a = parse_args()
    launched = "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not launched:
        # no launcher around us: become one.  Nothing has touched the GPU yet (device_count() does not).
        raise SystemExit(spawn_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # roofline.traffic, measured in this run: the profiled child passes go FIRST, while this process has not touched
    # the GPU (they are ordinary child processes; nothing is exec'ed from a process that has initialised HIP)
    pmc_traffic = (None, None)
    under_profiler = ("rocprof" in os.environ.get("LD_PRELOAD", "").lower()
                      or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ))
    if world == 1 and a.traffic == "pmc" and not a.pmc_child and not a.simulate_shard and a.roofline_steps > 0:
        if under_profiler:   # this process is itself being profiled (kernel trace): no nested profiler passes
            pmc_traffic = (None, "bench.py is running under rocprofv3 itself")
        else:
            pmc_traffic = measure_decode_traffic_pmc([x for x in sys.argv[1:] if x != "--pmc-child"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # rehearsal mode for a 1-GPU box: every rank on cuda:0, collectives over gloo (functional check of
    # the N>1 code path only; never used by the driver)
    rehearsal = os.environ.get("MI_BENCH_REHEARSAL") == "1"
    if rehearsal:
        os.environ["VLLM_OMNI_AMD_TP_GRAPH"] = "0"  # gloo collectives are not capturable
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)

    from vllm_omni_amd.talker import TalkerConfig

    cfg = TalkerConfig.from_json(a.config)
    plans = [(1, 1, "tp1", "headline")]
    if a.simulate_shard:
        if world > 1:
            raise SystemExit("--simulate-shard runs on one GPU")
        from vllm_omni_amd.distributed import simulate_tensor_parallel_rank

        tp, dp = (int(x) for x in a.simulate_shard.split(","))
        plans = [(tp, dp, f"SIMULATED rank 0 of dp{dp}xtp{tp}, collectives stubbed", "simulated")]
        if tp > 1:
            simulate_tensor_parallel_rank(tp)
        a.no_cpu_baseline = True
    n_ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not rehearsal:
            dist.init_process_group(backend="nccl", init_method="env://", world_size=world, rank=rank, device_id=dev)
        elif not dist.is_initialized():
            dist.init_process_group(backend="gloo", init_method="env://", world_size=world, rank=rank)
        # n_gpus is what the communicator saw, not what the command line asked for
        one = torch.ones(1, device=dev if not rehearsal else "cpu")
        dist.all_reduce(one)
        n_ranks_seen = int(one.item())
        plans = plans_to_run(world, cfg, a.batch)
    for tp, dp, label, _ in plans:
        if a.batch % dp:
            raise SystemExit(f"global batch {a.batch} must divide by the data-parallel degree of {label}")

    def compose(results: list[dict], failures: list[dict]) -> dict:
        """The ONE JSON line: the widest tensor-parallel plan that produced a number is the headline (the TP scaling
        line the metric names); every other measured plan goes to `alt_plans`, failed attempts to `failed_plans`."""
        tp_top = max(r["tp"] for r in results)
        head = max((r for r in results if r["tp"] == tp_top), key=lambda r: r["value"])
        ms_per_step, tokens_per_s = head["ms_per_step"], head["value"]
        tp = head["tp"]
        roofline = head["roofline"]
        if a.simulate_shard:   # the whole plan would decode the global batch in this rank's step time
            tokens_per_s = a.batch / (ms_per_step / 1e3)
        out = {
            "simulated": bool(a.simulate_shard),
            "metric": ("decode tokens/s, Qwen-Omni-7B talker, bs=256 ctx=4k"
                       if (a.config, a.batch, a.ctx, a.quantization, a.code_predictor) == ("qwen2_5_omni_talker", 256, 4096, None, False)
                       else f"decode tokens/s, {cfg.name}{' int8' if a.quantization else ''}"
                            f"{' + code predictor' if a.code_predictor else ''}, bs={a.batch} ctx={a.ctx} (not the headline shape)"),
            "value": round(tokens_per_s, 1),
            "unit": "tokens/s", "n_gpus": n_ranks_seen, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3),
            **{k: head[k] for k in ("ms_per_step_p10", "ms_per_step_p50", "ms_per_step_p90", "ms_per_step_how") if k in head},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if a.quantization is None else "bf16 (attention, norms, activations) + int8 W8A8 decoder linears",
            "data": "synthetic (random-init weights, N(0,1) KV cache, random non-contiguous block tables, fixed ctx)",
            "config": {"workload": f"{cfg.name} decode step, global batch {a.batch}, ctx {a.ctx}, block {a.block_size}",
                       "global_batch": a.batch, "seq_len": a.ctx, "parallelism": head["parallelism"],
                       "plan_role": head["role"],
                       "hipgraph": head["hipgraph"], "kv_layout": a.kv_layout,
                       "output": "sync" if a.sync_output else "pipelined one step behind (all outputs on the host inside the timed region)",
                       "decode_partition": head["decode_partition"],
                       "tuned_gemm_csv_loaded": head["tuned_gemm_csv_loaded"],
                       "allreduce": head["allreduce"],
                       "linear": os.environ.get("VLLM_OMNI_AMD_LINEAR", "auto")
                       + " (auto: hand-written streaming GEMM where measured faster than the TunableOp-selected library "
                         "kernel, library GEMM elsewhere; same bf16 inputs, fp32 accumulation)"},
            "roofline": roofline,
        }
        if "allreduce_probe" in head:   # measured time of the headline plan's all-reduce per message (tp > 1)
            out["allreduce_probe"] = head["allreduce_probe"]
        others = [{k: v for k, v in r.items() if k != "roofline"} for r in results if r is not head]
        if others:
            out["alt_plans"] = others
        if failures:
            out["failed_plans"] = failures
        return out

    def roofline_of(runner, local_batch: int, tp: int) -> dict:
        """Roofline of the dominant kernel (paged-attention decode) on this rank, for the plan `runner` was built for."""
        attn = runner.model.language_model.model.layers[0].self_attn
        # unique KV bytes one launch must read: K+V, local kv heads, head_dim, 2 B, ctx tokens, local batch
        algo_bytes = 2 * attn.num_kv_heads * attn.head_dim * 2 * a.ctx * local_batch
        if a.roofline_steps <= 0:
            return {"bound": "hbm", "kernel": "mi::decode_mfma_kernel (paged-attention decode)", "achieved": None,
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None,
                    "algorithmic_bytes_per_launch": algo_bytes}
        call_ms, calls = time_decode_launches(runner, a.roofline_steps)
        kern_ms, launches, variant = time_decode_kernel(runner)
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        prof = os.path.join(ROOT, "profiles", "pmc_decode_bytes_per_launch.json")
        if pmc_traffic[0] is not None and tp == 1 and local_batch == a.batch:
            traffic, traffic_src = pmc_traffic
        elif a.traffic != "none" and os.path.exists(prof):
            try:
                with open(prof) as f:
                    pj = json.load(f)
                if pj.get("batch") == local_batch and pj.get("ctx") == a.ctx and pj.get("tp") == tp:
                    traffic = pj.get("hbm_bytes_per_launch")
                    traffic_src = ("NOT measured in this run: rocprofv3 --pmc pass of the same command, "
                                   "profiles/pmc_decode_bytes_per_launch.json (" + str(pj.get("source", "")) + ")"
                                   + ("" if pmc_traffic[1] is None else "; in-run measurement failed: " + str(pmc_traffic[1])))
            except Exception:
                traffic = None
        return {"bound": "hbm", "kernel": "mi::decode_mfma_kernel (paged-attention decode)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(kern_ms, 4),
                "launches_timed": launches,
                "timing": "HIP events around passes of one decode-kernel launch per layer, issued back to back on the "
                          "step's own inputs right after the timed region (" + variant + "; the partition-merge launch "
                          "left out): the kernel rocprofv3's kernel trace lists as decode_mfma_kernel",
                "eager_call_ms": round(call_ms, 4), "eager_calls_timed": calls,
                "eager_call_how": "HIP events around each decode call of an instrumented EAGER step: attention launch + "
                                  "merge launch (if partitioned) + the host's launch latency — the round-1..3 figure"}

    def measure(tp: int, dp: int, label: str, role: str, eager: bool = False, allreduce: str = "rccl") -> dict:
        # fault injection for rehearsing the safety net (never set by the driver): "skip:<text>", "raise:<text>",
        # "hang:<text>" act on the stage whose label contains <text>
        kind, _, what = os.environ.get("MI_BENCH_INJECT_FAULT", "").partition(":")
        if kind and what in label:
            if kind == "skip":
                raise StageSkipped("injected")
            if kind == "raise":
                raise RuntimeError("injected")
            if kind == "hang":
                time.sleep(10 ** 6)
        elapsed, runner, local_batch = run_plan(a, cfg, dev, world, rank, tp, dp, rehearsal, eager, allreduce)
        ms = elapsed / a.steps * 1e3
        res = {"parallelism": label, "role": role, "tp": tp, "allreduce": allreduce if tp > 1 else "none",
               "ms_per_step": round(ms, 3), **step_distribution(getattr(runner, "_bench_step_ms", [])),
               "value": round(a.batch * a.steps / elapsed, 1), "hipgraph": bool(runner._graphs),
               "decode_partition": runner._decode_part,
               "tuned_gemm_csv_loaded": bool(getattr(runner, "tuned_gemms", False)),
               "roofline": roofline_of(runner, local_batch, tp)}
        if tp > 1:   # measured all-reduce time per message of this plan's transport (2 x layers of them per step)
            try:
                res["allreduce_probe"] = dict(time_allreduce(dev, local_batch, cfg.hidden_size),
                                              messages_per_step=2 * cfg.num_hidden_layers)
            except Exception as e:  # noqa: BLE001 - the probe must never cost the plan its number
                res["allreduce_probe"] = {"error": repr(e)[:200]}
        del runner
        import gc

        gc.collect()               # the runner's object graph has cycles; the next plan needs its HBM back
        torch.cuda.empty_cache()
        return res

    # With more than one GPU the measurements go from the safest to the fastest, each kept once it exists:
    #   1. the plan WITHOUT a data-path collective (pure data parallel, captured step);
    #   2. the tensor-parallel plan run EAGERLY (RCCL all-reduces as plain stream operations);
    #   3. the tensor-parallel plan captured in a hipGraph with the in-tree all-reduce, after an on-rank validation;
    #   4. the tensor-parallel plan captured with RCCL inside the capture (never exercised on the 1-GPU boxes this repo
    #      is developed on).
    # 2 to 4 run under a wall-clock watchdog.  Whatever raises or hangs, rank 0 still prints ONE line from what was
    # measured before it — the widest tensor-parallel plan that produced a number is the headline, its faster variant
    # if both exist — with what happened to the failed attempt next to it, instead of nothing.
    results: list[dict] = []
    failures: list[dict] = []
    first_ar = os.environ.get("VLLM_OMNI_AMD_ALLREDUCE", "rccl")
    ordered = [(tp, dp, label, role, False, first_ar) for tp, dp, label, role in sorted(plans, key=lambda pl: pl[0])]   # tp == 1 first
    safe_first = world > 1 and len(ordered) > 1 and ordered[0][0] == 1
    if safe_first and not a.eager and os.environ.get("MI_BENCH_TP_EAGER_FIRST", "1") == "1":
        staged = [ordered[0]]
        for tp, dp, label, role, _, ar in ordered[1:]:
            staged.append((tp, dp, label + " (eager)", role + ", eager launches", True, ar))
            # 3. the same plan, captured, on the in-tree all-reduce (csrc/allreduce.hip) once it has reproduced the
            #    library's sums on these very ranks (validate_in_tree_allreduce): the multi-GPU run DESIGN.md 5.1 waits
            #    for.  Before 4 because a failed capture of library collectives leaves the process unusable.
            if ar != "custom" and os.environ.get("MI_BENCH_TRY_IN_TREE_ALLREDUCE", "1") == "1":
                staged.append((tp, dp, label + " (in-tree all-reduce)", role + ", in-tree two-shot all-reduce", False, "custom"))
            staged.append((tp, dp, label, role, False, ar))   # 4. captured, library collectives inside the graph
        ordered = staged

    # A stage that hung or raised has touched the GPU and may have left kernels / collectives behind: the process
    # reports what it measured BEFORE that stage (one JSON line, `failed_plans` + `exit_code` inside) and then ends
    # with a distinct NON-ZERO code, so torchrun and the driver do not record a hang as a clean run.
    EXIT_STAGE_FAILED = 4

    def emit_and_exit(reason: str, what: str) -> None:      # watchdog thread
        failures.append({"parallelism": what, "error": reason})
        if rank == 0 and results:
            line = compose(results, failures)
            line["exit_code"] = EXIT_STAGE_FAILED
            print(json.dumps(line), flush=True)
        leave(EXIT_STAGE_FAILED if results else 3)

    t_bench0 = time.monotonic()
    budget = float(os.environ.get("MI_BENCH_BUDGET_S", "420"))   # no further stage is STARTED past this many seconds
    for i, (tp, dp, label, role, eager, ar_kind) in enumerate(ordered):
        guard = None
        if safe_first and i > 0:
            import threading

            # every rank must take the same decision: rank 0's clock decides
            late = torch.tensor([1.0 if time.monotonic() - t_bench0 > budget else 0.0], device=dev if not rehearsal else "cpu")
            dist.broadcast(late, src=0)
            if late.item() > 0:
                failures.append({"parallelism": label, "error": f"not started: {budget:.0f} s budget of the run used up"})
                continue

            limit = float(os.environ.get("MI_BENCH_TP_WATCHDOG_S", "200"))
            guard = threading.Timer(limit, emit_and_exit, args=(f"no result within {limit:.0f} s (hung collective?)", label))
            guard.daemon = True
            guard.start()
        try:
            results.append(measure(tp, dp, label, role, eager, ar_kind))
        except StageSkipped as e:
            failures.append({"parallelism": label, "error": str(e)})
            continue
        except Exception as e:  # noqa: BLE001
            if not (safe_first and i > 0):
                raise
            import traceback

            failures.append({"parallelism": label, "error": f"{type(e).__name__}: {str(e)[:300]}",
                             "where": traceback.format_exc()[-700:]})
            if guard is not None:
                guard.cancel()
            # a failed capture / collective leaves the process group in an unknown state: report and leave
            if rank == 0:
                line = compose(results, failures)
                line["exit_code"] = EXIT_STAGE_FAILED
                print(json.dumps(line), flush=True)
            leave(EXIT_STAGE_FAILED)
        finally:
            if guard is not None:
                guard.cancel()
    out = compose(results, failures)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_baseline_batch, a.ctx, a.block_size)
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

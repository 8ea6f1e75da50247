"""Decode-step driver: persistent batch, vectorised input prep, hipGraph replay.

Mirrors the two-phase contract of ``GPUARModelRunner``
(vllm_omni/worker/gpu_ar_model_runner.py:262-624 ``execute_model`` -> ``None`` with stashed state,
:676-953 ``sample_tokens`` -> ``OmniModelRunnerOutput``; out-of-order calls raise
``RuntimeError("State error: ...")`` :268-269) and the persistent-batch update of
``OmniGPUModelRunner._update_states`` (vllm_omni/worker/gpu_model_runner.py:241-556).

What is restructured for MI355X (SURVEY §3 hot loop 3, §7 "host overhead at bs=256"):
  * no per-request Python in the steady-state decode step: positions / slots / seq_lens are numpy
    vector ops + one C call, one pinned H2D copy each;
  * the talker's per-request ``preprocess`` hook (``thinker_reply_part[0] + embed(last token)``,
    qwen2_5_omni.py:766-806) is one batched gather-add over a device-resident reply pool;
  * the 28-layer step (+ logits, mask, argmax) of a pure-decode batch is captured in a hipGraph
    per batch size and replayed; the reference ships ``enforce_eager: true``
    (vllm_omni/platforms/rocm/stage_configs/qwen2_5_omni.yaml:44);
  * hidden states / sampled ids leave through pinned buffers on a side stream
    (the reference's synchronous ``hidden_states.to("cpu")``, gpu_ar_model_runner.py:835).
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .block_table import MultiGroupBlockTable
from .kv_cache import KVCacheSpec, allocate_kv_cache, extract_request_kv
from .layers import AttentionMetadata, set_forward_context
from .sched_output import (EMPTY_MODEL_RUNNER_OUTPUT, CachedRequestData, NewRequestData, OmniModelRunnerOutput,
                           SamplingMetadata, SamplingParams, SchedulerOutput)
from .talker import Qwen2_5OmniTalkerForConditionalGeneration, TalkerConfig


@dataclass
class CachedRequestState:
    req_id: str
    prompt_token_ids: list[int]
    sampling_params: SamplingParams
    block_ids: tuple[list[int], ...]
    num_computed_tokens: int
    output_token_ids: list[int] = field(default_factory=list)
    prompt_embeds: Optional[torch.Tensor] = None          # [prompt_len, E] talker prefill embeddings
    additional_information: Optional[dict] = None
    sample_seed: int = 0                                  # Philox key of the request's draws
    seen_slot: int = -1                                   # row of the device seen-token bitmap, -1 = none
    state_slot: int = -1                                  # row of the device per-request state (last talker hidden)
    reply_base: int = -1                                  # first row of the request's thinker_reply_part in the device pool
    reply_rows: int = 0                                   # rows it owns there
    reply_cursor: int = 0                                 # rows consumed so far (the reference's `q[1:]` queue, qwen2_5_omni.py:771-774)
    recompute_until: int = 0                              # resumed after a preemption: tokens to recompute before it decodes again

    @property
    def num_tokens(self) -> int:
        return len(self.prompt_token_ids) + len(self.output_token_ids)


class SamplingState:
    """Per-request state of the device sampler (ops.sample_tokens): the Philox key, the draw
    counter (= number of tokens drawn so far) and, for requests with a repetition penalty, a row of the
    device bitmap of prompt + output tokens (the sampler sets the drawn token's bit itself).  Mirrors what
    upstream keeps in `SamplingMetadata` (`prompt_token_ids`, `output_token_ids`, per-request generators) for
    the sampler call at vllm_omni/worker/gpu_ar_model_runner.py:626-657."""

    def __init__(self, max_num_reqs: int, vocab: int, device, base_seed: int, arena=None):
        from .block_table import CpuGpuBuffer

        self.arena = arena
        self.vocab, self.words = vocab, (vocab + 31) // 32
        self.base_seed = int(base_seed or 0)
        self.free_slots = list(range(max_num_reqs - 1, -1, -1))
        self.scratch_slot = max_num_reqs                      # rows without a penalty record their draws here
        self.bitmap = torch.zeros(max_num_reqs + 1, self.words, dtype=torch.int32, device=device)
        mk = lambda dt: (arena.buffer(max_num_reqs, dtype=dt) if arena is not None   # noqa: E731
                         else CpuGpuBuffer(max_num_reqs, dtype=dt, device=device))
        self.temperature, self.top_p, self.rep = mk(torch.float32), mk(torch.float32), mk(torch.float32)
        self.top_k, self.seen_row = mk(torch.int32), mk(torch.int32)
        self.seed, self.offset = mk(torch.int64), mk(torch.int64)

    def add(self, st: CachedRequestState) -> None:
        sp = st.sampling_params
        if sp.seed is not None:
            st.sample_seed = int(sp.seed) & (2**63 - 1)
        else:  # stable across runs and processes (str hash is salted): FNV-1a of the request id
            h = 0xCBF29CE484222325 ^ self.base_seed
            for ch in st.req_id.encode():
                h = ((h ^ ch) * 0x100000001B3) & (2**64 - 1)
            st.sample_seed = h & (2**63 - 1)
        if sp.repetition_penalty != 1.0 and st.seen_slot < 0 and self.free_slots:
            st.seen_slot = self.free_slots.pop()
            row = np.zeros(self.words, dtype=np.uint32)
            toks = np.asarray(list(st.prompt_token_ids) + list(st.output_token_ids), dtype=np.int64)
            toks = toks[(toks >= 0) & (toks < self.vocab)]
            np.bitwise_or.at(row, toks >> 5, (np.uint32(1) << (toks & 31).astype(np.uint32)))
            self.bitmap[st.seen_slot].copy_(torch.from_numpy(row.view(np.int32)))

    def remove(self, st: Optional[CachedRequestState]) -> None:
        if st is not None and st.seen_slot >= 0:
            self.free_slots.append(st.seen_slot)
            st.seen_slot = -1

    @staticmethod
    def needed(params: list[SamplingParams]) -> bool:
        return any(p.temperature > 0 or p.repetition_penalty != 1.0 for p in params)

    def sample(self, logits: torch.Tensor, states: list[CachedRequestState], ids: torch.Tensor,
               drawn_ahead: int = 0, discard: Optional[np.ndarray] = None) -> torch.Tensor:
        """``drawn_ahead``: draws already made for these requests whose tokens are not yet in ``output_token_ids``
        (an async output still pending on the host).  ``discard``: rows whose draw will be thrown away (unfinished
        prompt): it must not enter the request's seen set."""
        n = len(states)
        for i, st in enumerate(states):
            sp = st.sampling_params
            self.temperature.np[i] = max(sp.temperature, 0.0)
            self.top_p.np[i] = sp.top_p
            self.rep.np[i] = sp.repetition_penalty if st.seen_slot >= 0 else 1.0
            self.top_k.np[i] = sp.top_k
            keep = st.seen_slot >= 0 and not (discard is not None and discard[i])
            self.seen_row.np[i] = st.seen_slot if keep else self.scratch_slot
            if not keep:
                self.rep.np[i] = 1.0
            self.seed.np[i] = st.sample_seed
            self.offset.np[i] = len(st.output_token_ids) + drawn_ahead
        dev_args = [b.copy_to_gpu(n) for b in (self.temperature, self.seed, self.offset, self.top_k, self.top_p, self.rep,
                                                self.seen_row)]
        if self.arena is not None:
            self.arena.flush()      # ONE host->device copy for the seven per-row arrays, before the launch reads them
        temperature, seed, offset, top_k, top_p, rep, seen_row = dev_args
        return ops.sample_tokens(logits[:n], temperature, seed, offset, top_k=top_k, top_p=top_p, rep_penalty=rep,
                                 seen_bitmap=self.bitmap, seen_row=seen_row, update_seen=True, ids=ids[:n])


class ReplyRowAllocator:
    """First-fit free list over the rows of the device-resident ``thinker_reply_part`` pool.  A request owns one
    contiguous segment from its first scheduling until it finishes; ``alloc`` raises when no segment fits — live
    rows are never overwritten."""

    def __init__(self, rows: int):
        self.rows = rows
        self.free: list[tuple[int, int]] = [(0, rows)] if rows else []   # (start, length), sorted by start

    def alloc(self, n: int) -> int:
        for i, (start, length) in enumerate(self.free):
            if length >= n:
                if length == n:
                    self.free.pop(i)
                else:
                    self.free[i] = (start + n, length - n)
                return start
        raise RuntimeError(f"thinker_reply_part pool is full: {n} rows requested, largest free segment "
                           f"{max((ln for _, ln in self.free), default=0)} of {self.rows} (raise max_reply_rows)")

    def release(self, start: int, n: int) -> None:
        if n <= 0:
            return
        import bisect

        i = bisect.bisect_left(self.free, (start, 0))
        self.free.insert(i, (start, n))
        # coalesce with the neighbours
        if i + 1 < len(self.free) and self.free[i][0] + self.free[i][1] == self.free[i + 1][0]:
            self.free[i] = (self.free[i][0], self.free[i][1] + self.free[i + 1][1])
            self.free.pop(i + 1)
        if i > 0 and self.free[i - 1][0] + self.free[i - 1][1] == self.free[i][0]:
            self.free[i - 1] = (self.free[i - 1][0], self.free[i - 1][1] + self.free[i][1])
            self.free.pop(i)

    @property
    def num_free(self) -> int:
        return sum(ln for _, ln in self.free)


class InputBatch:
    """Persistent batch: dense rows [0, num_reqs), condensed on removal."""

    def __init__(self, max_num_reqs: int, max_model_len: int, max_num_batched_tokens: int, block_size: int, device,
                 arena=None):
        self.max_num_reqs = max_num_reqs
        self.req_ids: list[str] = []
        self.req_id_to_index: dict[str, int] = {}
        self.num_computed_tokens_cpu = np.zeros(max_num_reqs, dtype=np.int64)
        self.num_prompt_tokens = np.zeros(max_num_reqs, dtype=np.int64)
        self.decode_from = np.zeros(max_num_reqs, dtype=np.int64)   # a one-token step at >= this many computed tokens is a decode
        # a request resumed after a preemption recomputes the tokens it already has: prefill rows, nothing generated,
        # until it has this many computed tokens again (0 = not recomputing)
        self.recompute_until = np.zeros(max_num_reqs, dtype=np.int64)
        self.last_token_ids_cpu = np.zeros(max_num_reqs, dtype=np.int64)
        self.reply_offset = np.full(max_num_reqs, -1, dtype=np.int64)   # row of the next reply vector, -1 = none
        self.reply_end = np.zeros(max_num_reqs, dtype=np.int64)
        self.block_table = MultiGroupBlockTable(max_num_reqs, max_model_len, max_num_batched_tokens, [block_size],
                                                device, arena=arena)

    @property
    def num_reqs(self) -> int:
        return len(self.req_ids)

    def add_request(self, st: CachedRequestState) -> int:
        i = self.num_reqs
        if i >= self.max_num_reqs:
            raise RuntimeError("persistent batch is full (max_num_reqs)")
        self.req_ids.append(st.req_id)
        self.req_id_to_index[st.req_id] = i
        self.num_computed_tokens_cpu[i] = st.num_computed_tokens
        # upstream's length_from_prompt_token_ids_or_embeds: a prompt may come as embeddings only
        self.num_prompt_tokens[i] = len(st.prompt_token_ids) if st.prompt_token_ids else (
            int(st.prompt_embeds.shape[0]) if st.prompt_embeds is not None else 0)
        # A one-token step is a DECODE (input = embedding of the last token, + the thinker's reply row) once the prompt is
        # behind it.  A prompt given as EMBEDDINGS (every talker prompt the reference builds) must be consumed from the
        # embeddings to its last row, so the rule is strict; for a token-id prompt the last prompt token alone in a
        # step goes through the decode path too — same token (`last_token_ids_cpu` starts at prompt[-1]), and what the
        # synthetic steady-state batches of bench.py / the tests schedule ("everything but this token is cached").
        self.decode_from[i] = self.num_prompt_tokens[i] if st.prompt_embeds is not None else max(
            int(self.num_prompt_tokens[i]) - 1, 0)
        self.recompute_until[i] = st.recompute_until
        last = st.output_token_ids[-1] if st.output_token_ids else (st.prompt_token_ids[-1] if st.prompt_token_ids else 0)
        self.last_token_ids_cpu[i] = last
        self.reply_offset[i] = -1
        self.reply_end[i] = 0
        self.block_table.add_row(st.block_ids, i)
        return i

    def remove_request(self, req_id: str) -> None:
        """Remove and condense: the last row moves into the hole (one row move, no full shift)."""
        i = self.req_id_to_index.pop(req_id, None)
        if i is None:
            return
        last = self.num_reqs - 1
        if i != last:
            moved = self.req_ids[last]
            self.req_ids[i] = moved
            self.req_id_to_index[moved] = i
            for arr in (self.num_computed_tokens_cpu, self.num_prompt_tokens, self.decode_from, self.recompute_until,
                        self.last_token_ids_cpu, self.reply_offset, self.reply_end):
                arr[i] = arr[last]
            self.block_table.move_row(last, i)
        self.block_table.clear_row(last)
        self.req_ids.pop()

    def swap(self, a: int, b: int) -> None:
        if a == b:
            return
        ra, rb = self.req_ids[a], self.req_ids[b]
        self.req_ids[a], self.req_ids[b] = rb, ra
        self.req_id_to_index[ra], self.req_id_to_index[rb] = b, a
        for arr in (self.num_computed_tokens_cpu, self.num_prompt_tokens, self.decode_from, self.recompute_until,
                    self.last_token_ids_cpu, self.reply_offset, self.reply_end):
            arr[[a, b]] = arr[[b, a]]
        self.block_table.swap_row(a, b)


@dataclass
class ExecuteModelState:
    """Ephemeral state handed from execute_model() to sample_tokens()."""

    scheduler_output: SchedulerOutput
    logits: torch.Tensor
    sampled_ids: Optional[torch.Tensor]       # greedy ids already computed by the fused mask+argmax
    hidden_states: torch.Tensor
    num_reqs: int
    num_tokens: int
    query_start_loc_np: np.ndarray
    kv_extracted_req_ids: Optional[list[str]]
    num_decodes: int = 0
    # Omni tensor prefix cache (A14): the step's slot mapping (host copy; the pinned buffer is rewritten next step) and,
    # per request that entered on a KV prefix-cache hit, the block ids of its cached full blocks
    slot_mapping_np: Optional[np.ndarray] = None
    prefix_hit_blocks: Optional[dict] = None
    # rows whose prompt is not finished after this step (a chunk of a chunked prefill): their sampled token is
    # discarded, as upstream's bookkeeping does (`discard_sampled_tokens_req_indices`)
    partial: Optional[np.ndarray] = None


class AsyncOmniModelRunnerOutput:
    """What ``sample_tokens(async_output=True)`` returns (upstream's ``AsyncGPUModelRunnerOutput`` idea applied to
    the reference's hidden-state hand-back, gpu_ar_model_runner.py:835,868-912): the device->host copy of the step's
    ids and hidden states is in flight on a side stream; ``get_output()`` waits for it, then does the host
    bookkeeping (token lists, pooler payloads) and returns the ordinary ``OmniModelRunnerOutput``.  Outputs finish
    in step order."""

    def __init__(self, runner: "GPUARModelRunner", st: "ExecuteModelState", slot: int, req_ids: list, req_index: dict,
                 prev: Optional["AsyncOmniModelRunnerOutput"]):
        self.runner, self.st, self.slot = runner, st, slot
        self.req_ids, self.req_index, self.prev = req_ids, req_index, prev
        self._out: Optional[OmniModelRunnerOutput] = None

    @property
    def done(self) -> bool:
        return self._out is not None

    def get_output(self) -> OmniModelRunnerOutput:
        if self._out is None:
            if self.prev is not None:
                self.prev.get_output()
                self.prev = None
            self._out = self.runner._finish_output(self)
        return self._out


class no_gc_during_capture:
    """The cyclic garbage collector must not run while a stream is capturing: collecting a dead tensor / graph / event
    there issues HIP calls that are illegal during capture and the process aborts (seen: "Fatal Python error: Aborted
    ... Garbage-collecting" inside a capture, after many runners had lived in one process).  Collect first, then keep the
    collector off until the capture has ended — what upstream does around its own captures."""

    def __enter__(self):
        import gc

        self._was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc

        if self._was_enabled:
            gc.enable()
        return False


def _limit_host_threads() -> None:
    """One intra-op CPU thread per worker unless the user asked for more — what vLLM does for its GPU workers
    (``OMP_NUM_THREADS`` defaults to 1 there; the reference's workers inherit it).  The step's host-side tensor work is a
    few index arrays; with torch's default (one OpenMP thread per visible core: 256 on an MI355X host) every array
    over the parallel grain wakes the whole pool, the idle threads spin, and a container with a CPU quota is throttled
    for the rest of the scheduler period.  Measured on a 16-core-quota box: chunked prefill of the MoE talker
    (8192 tokens in chunks of 2048) 165 ms -> 29 ms, `nr_throttled` 14 -> 0 (profiles/r02_host_threads.md).
    ``VLLM_OMNI_AMD_HOST_THREADS`` or ``OMP_NUM_THREADS`` override."""
    want = os.environ.get("VLLM_OMNI_AMD_HOST_THREADS")
    if want is None and "OMP_NUM_THREADS" in os.environ:
        return
    n = max(1, int(want)) if want else 1
    if torch.get_num_threads() != n:
        torch.set_num_threads(n)


class GPUARModelRunner:
    OUT_SLOTS = 4   # ring of pinned output buffers: a step's hidden-state views live for OUT_SLOTS - 1 more steps

    def __init__(self, config, device: torch.device, dtype: Optional[torch.dtype] = None,
                 max_num_reqs: Optional[int] = None, max_model_len: Optional[int] = None,
                 max_num_batched_tokens: Optional[int] = None, block_size: Optional[int] = None,
                 kv_layout: str = "NB2_HEAD_MAJOR", enforce_eager: Optional[bool] = None,
                 engine_output_type: Optional[str] = None, max_reply_rows: Optional[int] = None,
                 seed: Optional[int] = 0, quantization: Optional[str] = None, code_predictor=None,
                 copy_outputs: bool = True, enable_prefix_caching: Optional[bool] = None, model_cls=None):
        """``model_cls``: the model class ``load_model`` builds (same constructor as the talker class, which is the
        default) — how a stage brings its own ``preprocess`` / ``postprocess`` / ``sample`` hooks, as the reference's
        model classes do (gpu_model_runner.py:1264-1309,1045-1070; gpu_ar_model_runner.py:633-650).

        ``copy_outputs``: hand every step's ``pooler_output`` tensors back as caller-owned host tensors, as the
        reference does (``hidden_states.to("cpu")``, gpu_ar_model_runner.py:835).  False returns VIEWS of the pinned
        output ring, valid until ``OUT_SLOTS - 1`` further steps have been sampled (what bench.py uses: fresh 1.8 MB
        host allocations every step were measured to stall the device queues)."""
        from .compat import is_vllm_config, worker_config_from_vllm_config
        from .distributed import get_parallel_state

        _limit_host_threads()
        wc = None
        if is_vllm_config(config):
            # ``GPUARModelRunner(self.vllm_config, self.device)`` (gpu_ar_worker.py:102): everything else comes
            # from the config object; explicit keyword arguments still win
            self.vllm_config = config
            wc = worker_config_from_vllm_config(config)
            config = wc.model_config
            quantization = quantization or wc.quantization
            seed = wc.seed if seed == 0 else seed

        def pick(explicit, from_cfg, default):
            return explicit if explicit is not None else (from_cfg if wc is not None else default)

        dtype = pick(dtype, wc and wc.dtype, torch.bfloat16)
        max_num_reqs = pick(max_num_reqs, wc and wc.max_num_seqs, 256)
        max_model_len = pick(max_model_len, wc and wc.max_model_len, 4096)
        max_num_batched_tokens = pick(max_num_batched_tokens, wc and wc.max_num_batched_tokens, 8192)
        block_size = pick(block_size, wc and wc.block_size, 16)
        enforce_eager = pick(enforce_eager, wc and wc.enforce_eager, False)
        engine_output_type = pick(engine_output_type, wc and wc.engine_output_type, "latent")
        max_reply_rows = pick(max_reply_rows, wc and (wc.max_reply_rows or wc.max_num_seqs * 64), 0)
        # `cache_config.enable_prefix_caching`: the hidden states handed to the next stage are cached per KV slot too,
        # so a request that hits the KV prefix cache still gets ALL its rows back (gpu_model_runner.py:79-85)
        self.enable_prefix_caching = bool(pick(enable_prefix_caching, wc and wc.enable_prefix_caching, False))
        self.omni_prefix_cache = None
        if not isinstance(config, TalkerConfig):
            raise TypeError(f"GPUARModelRunner needs a VllmConfig-shaped object or a TalkerConfig, got {type(config).__name__}")
        self.config = config
        self.device = device
        self.dtype = dtype
        self.max_num_reqs = max_num_reqs
        self.max_model_len = max_model_len
        self.max_num_batched_tokens = max(max_num_batched_tokens, max_num_reqs)
        self.block_size = block_size
        self.kv_layout = kv_layout
        self.enforce_eager = enforce_eager
        self.engine_output_type = engine_output_type
        self.seed = seed
        if quantization not in (None, "int8"):
            raise ValueError(f"unsupported quantization {quantization!r} (None or 'int8')")
        self.quantization = quantization
        # Qwen3-Omni: a CodePredictorConfig turns every decode step into `talker_mtp`
        # (vllm_omni/model_executor/models/qwen3_omni/qwen3_omni.py:681-705): the code predictor expands the last
        # codec token into all residual codebooks, the talker's input is the SUM of their embeddings + the text step
        self.code_predictor_config = code_predictor
        self.code_predictor = None
        self.model_cls = model_cls or Qwen2_5OmniTalkerForConditionalGeneration
        self._generic_hooks = False      # set by load_model: drive the model's hooks per request (eager)
        self.parallel = get_parallel_state()
        self.model: Optional[Qwen2_5OmniTalkerForConditionalGeneration] = None
        self.kv_caches: list = []
        self.requests: dict[str, CachedRequestState] = {}
        from .block_table import StepArena

        # the step's small index arrays live in ONE pinned block / device block: one host->device copy per step
        # instead of one per array (block_table.StepArena); generous: 16 int64 arrays of the larger of tokens / requests
        self._arena = StepArena(16 * 8 * (max(self.max_num_batched_tokens, max_num_reqs) + 64) + 64 * StepArena.ALIGN, device)
        self.input_batch = InputBatch(max_num_reqs, max_model_len, self.max_num_batched_tokens, block_size, device,
                                      arena=self._arena)
        self.execute_model_state: Optional[ExecuteModelState] = None
        self.kv_extracted: dict[str, Any] = {}
        self.kv_payloads: dict[str, torch.Tensor] = {}     # packed wire payloads (1-D uint8, device)
        self.max_kept_kv = 4 * max_num_reqs                # hand-offs nobody collected: oldest dropped beyond this
        self._graphs: dict[int, tuple[torch.cuda.CUDAGraph, dict]] = {}
        self._graph_pool = None                            # one memory pool shared by every captured batch size
        self.max_reply_rows = max_reply_rows
        self._reply_pool: Optional[torch.Tensor] = None
        self._reply_alloc = ReplyRowAllocator(max_reply_rows)
        self.copy_outputs = copy_outputs
        self._alloc_buffers()

    # ------------------------------------------------------------------ buffers
    def _alloc_buffers(self) -> None:
        from .block_table import CpuGpuBuffer

        dev, n, t = self.device, self.max_num_reqs, self.max_num_batched_tokens
        ar = self._arena
        self.positions = ar.buffer(t, dtype=torch.int64)
        self.seq_lens = ar.buffer(n, dtype=torch.int32)
        self.query_start_loc = ar.buffer(n + 1, dtype=torch.int32)
        self.last_token_ids = CpuGpuBuffer(n, dtype=torch.int64, device=dev)   # also written on the DEVICE: not in the arena
        self.reply_idx = ar.buffer(n, dtype=torch.int64)
        self.logits_indices = ar.buffer(n, dtype=torch.int64)
        on_gpu = torch.device(dev).type == "cuda"
        e = self.config.embedding_size
        self.inputs_embeds = torch.zeros(t, e, dtype=self.dtype, device=dev)
        self.sampled_ids_gpu = torch.zeros(n, dtype=torch.int64, device=dev)
        # step outputs go device -> staging (same stream, ~2 us) -> pinned host on a side stream, through a ring of
        # OUT_SLOTS buffers so that the copy of step N can still be in flight while step N+1 runs
        # (sample_tokens(async_output=True)).  The hidden-state tensors handed out are VIEWS of the pinned ring: they
        # stay valid until OUT_SLOTS - 1 further steps have been sampled.  (Cloning them into fresh host tensors
        # every step was measured to stall the GPU ~48 ms every third step: the 1.8 MB malloc/free churn unmaps
        # pages, and the MMU notifier that follows blocks the device queues.)
        hs = self.config.hidden_size
        k = self.OUT_SLOTS
        self._out_ids_cpu = [torch.zeros(n, dtype=torch.int64, pin_memory=on_gpu) for _ in range(k)]
        self._out_hidden_cpu = [torch.zeros(t, hs, dtype=self.dtype, pin_memory=on_gpu) for _ in range(k)]
        self._out_ids_stage = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(k)] if on_gpu else None
        self._out_hidden_stage = [torch.zeros(t, hs, dtype=self.dtype, device=dev) for _ in range(k)] if on_gpu else None
        self.copy_stream = torch.cuda.Stream(device=dev) if on_gpu else None
        self._out_done = [torch.cuda.Event() for _ in range(k)] if on_gpu else None
        self._out_inflight: list[Optional["AsyncOmniModelRunnerOutput"]] = [None] * k
        self._out_slot = 0
        self._last_output: Optional["AsyncOmniModelRunnerOutput"] = None   # newest output whose host part is pending
        self._ids_on_device_for: Optional[tuple] = None   # batch rows whose last token id already sits in last_token_ids.gpu
        self._h2d_done = torch.cuda.Event() if on_gpu else None
        self._h2d_pending = False
        self._decode_ws: Optional[torch.Tensor] = None
        self._decode_part = 0
        self.sampling_state: Optional[SamplingState] = None   # built with the model (needs the vocabulary size)

    # -------------------------------------------------------------------- model
    def load_model(self) -> None:
        if torch.device(self.device).type == "cuda":
            from .tuned import enable_tuned_gemms

            self.tuned_gemms = enable_tuned_gemms()
        self.model = self.model_cls(self.config, self.dtype, self.device, self.seed)
        m = self.model
        # a model whose hooks are not the ones the vectorised step implements goes through the per-request hook loop
        self._generic_hooks = bool((getattr(m, "has_preprocess", False) and not getattr(m, "batched_preprocess", False))
                                   or getattr(m, "has_postprocess", False))
        if self._generic_hooks and self.code_predictor_config is not None:
            raise ValueError("the per-request hook loop and the fused code-predictor step are alternatives: a model class "
                             "with its own preprocess / postprocess runs its predictor inside those hooks")
        if self.quantization == "int8":   # online W8A8 of the decoder linears (int8_config.py:391-418)
            self.model.quantize_int8()
        if self.code_predictor_config is not None:
            from .block_table import CpuGpuBuffer
            from .code_predictor import MI355XCodePredictor

            cpc = self.code_predictor_config
            if self.config.input_projection or self.config.embedding_size != cpc.hidden_size:
                raise ValueError("the code predictor feeds the talker's hidden space: needs input_projection=False and "
                                 "embedding_size == code predictor hidden_size")
            self.code_predictor = MI355XCodePredictor(cpc, self.dtype, self.device, max_batch=self.max_num_reqs,
                                                      seed=(self.seed or 0) + 17)
            n = self.max_num_reqs
            self._state_free = list(range(n - 1, -1, -1))
            # + one scratch row: where the padded rows of a bucketed graph replay park their state
            self._last_hidden = torch.zeros(n + 1, self.config.hidden_size, dtype=self.dtype, device=self.device)
            self._state_rows = self._arena.buffer(n, dtype=torch.int64)
            self._cp_seed = self._arena.buffer(n, dtype=torch.int64)
            self._cp_offset = self._arena.buffer(n, dtype=torch.int64)
            self._cp_codes = torch.zeros(n, cpc.num_code_groups, dtype=torch.int64, device=self.device)
            k = self.OUT_SLOTS
            on_gpu = torch.device(self.device).type == "cuda"
            self._out_codes_cpu = [torch.zeros(n, cpc.num_code_groups, dtype=torch.int64, pin_memory=on_gpu) for _ in range(k)]
            self._out_codes_stage = [torch.zeros(n, cpc.num_code_groups, dtype=torch.int64, device=self.device) for _ in range(k)]
        if torch.device(self.device).type == "cuda" and self.config.vocab_size <= 15744:
            self.sampling_state = SamplingState(self.max_num_reqs, self.config.vocab_size, self.device, self.seed or 0,
                                                arena=self._arena)
        if self.max_reply_rows:
            self._reply_pool = torch.zeros(self.max_reply_rows, self.config.embedding_size, dtype=self.dtype,
                                           device=self.device)

    def get_model(self):
        return self.model

    def get_kv_cache_spec(self) -> KVCacheSpec:
        a = self.model.language_model.model.layers[0].self_attn
        return KVCacheSpec(self.block_size, a.num_kv_heads, a.head_dim, self.dtype)

    def initialize_kv_cache(self, num_blocks: int) -> None:
        spec = self.get_kv_cache_spec()
        self.num_gpu_blocks = num_blocks
        self.kv_caches = allocate_kv_cache(self.config.num_hidden_layers, num_blocks, spec, self.device, self.kv_layout)
        for layer, kv in zip(self.model.language_model.model.layers, self.kv_caches):
            layer.self_attn.attn.bind_kv_cache(kv)
        a = self.model.language_model.model.layers[0].self_attn
        # The partition of a decode launch is chosen per call from the step's batch size and context bound (the fewest
        # partitions that leave >= 512 waves: ops.decode_auto_partition); the workspace is sized once for the largest
        # need any batch size of this runner can have.  `_decode_part` (reporting only) = the full batch's choice.
        self._decode_part = ops.decode_auto_partition(self.max_num_reqs, a.num_kv_heads, self.max_model_len,
                                                       self.block_size, head_dim=a.head_dim)
        nbytes = max(ops.decode_workspace_bytes(b, a.num_heads, a.head_dim, self.max_model_len,
                                                ops.decode_auto_partition(b, a.num_kv_heads, self.max_model_len,
                                                                          self.block_size, head_dim=a.head_dim))
                     for b in range(1, self.max_num_reqs + 1))
        if torch.device(self.device).type == "cuda":
            self._decode_ws = ops.new_decode_workspace(nbytes, self.device)   # zeroed counters: fused partition merge
        else:
            self._decode_ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
        if self.enable_prefix_caching and self.engine_output_type != "text":
            from .prefix_cache import OmniTensorPrefixCache

            # host tensor on the KV cache's own (block, offset) grid (gpu_model_runner.py:79-85)
            self.omni_prefix_cache = OmniTensorPrefixCache(num_blocks=num_blocks, block_size=self.block_size,
                                                           hidden_size=self.config.hidden_size, hs_dtype=self.dtype)

    def _zero_block_ids(self, block_ids: list[int]) -> None:
        idx = torch.as_tensor(block_ids, dtype=torch.int64, device=self.device)
        for kv in self.kv_caches:
            from .kv_cache import normalize_layer_kv
            kb, vb = normalize_layer_kv(kv)
            kb.index_fill_(0, idx, 0)
            vb.index_fill_(0, idx, 0)

    # ------------------------------------------------------------ state update
    def _update_states(self, so: SchedulerOutput) -> None:
        ib = self.input_batch
        if self.omni_prefix_cache is not None:     # (gpu_model_runner.py:252-253)
            self.omni_prefix_cache.reset_prefix_cached_new_req_ids()
        for rid in so.finished_req_ids:
            st = self.requests.pop(rid, None)
            if self.sampling_state is not None:
                self.sampling_state.remove(st)
            if self.code_predictor is not None and st is not None and st.state_slot >= 0:
                self._state_free.append(st.state_slot)
                st.state_slot = -1
            if st is not None and st.reply_base >= 0:
                self._reply_alloc.release(st.reply_base, st.reply_rows)
                st.reply_base, st.reply_rows = -1, 0
            ib.remove_request(rid)
        if so.new_block_ids_to_zero:
            self._zero_block_ids(so.new_block_ids_to_zero)
        cached = so.scheduled_cached_reqs
        scheduled = so.num_scheduled_tokens.keys()
        unscheduled = [r for r in ib.req_ids if r not in scheduled or r in cached.resumed_req_ids]
        for rid in unscheduled:
            # the request keeps its cached state, including how far it has consumed its reply rows
            # (the reference persists the consumed queue: update_dict["thinker_reply_part"] = q[1:])
            st, row = self.requests.get(rid), ib.req_id_to_index[rid]
            if st is not None and st.reply_base >= 0:
                st.reply_cursor = int(ib.reply_offset[row]) - st.reply_base
            ib.remove_request(rid)

        to_add: list[CachedRequestState] = []
        for nr in so.scheduled_new_reqs:
            prompt_embeds, info = nr.prompt_embeds, nr.additional_information
            if (prompt_embeds is None and isinstance(info, dict) and isinstance(info.get("thinker_result"), torch.Tensor)
                    and not self._generic_hooks):
                # The reference hands a talker request over as additional_information{prompt_embeds, thinker_result,
                # prompt_token_ids, thinker_output_token_ids} and lets the MODEL's preprocess hook build the prompt rows on
                # the prefill span (gpu_model_runner.py:1264-1309 -> talker_preprocess -> thinker_to_talker_process,
                # qwen2_5_omni.py:653-764).  Same construction, once, when the request enters: P + 2 prompt rows, the
                # remaining thinker rows become the request's reply queue.
                _, prompt_embeds, upd = self.model.thinker_to_talker_process(info, info.get("voice_type", "Chelsie"))
                if nr.prompt_token_ids and len(nr.prompt_token_ids) != prompt_embeds.shape[0]:
                    raise ValueError(f"request {nr.req_id}: {len(nr.prompt_token_ids)} prompt token ids but the thinker "
                                     f"hand-over builds {prompt_embeds.shape[0]} prompt rows (thinker prompt + 2)")
                info = dict(info, **upd)
            st = CachedRequestState(nr.req_id, list(nr.prompt_token_ids), nr.sampling_params,
                                    tuple(list(b) for b in nr.block_ids), nr.num_computed_tokens,
                                    prompt_embeds=prompt_embeds, additional_information=info)
            self.requests[nr.req_id] = st
            # first scheduling with computed tokens == a hit in the KV prefix cache: the rows of the cached blocks
            # come out of the tensor prefix cache when the step's output is packed (gpu_model_runner.py:315-320)
            if self.omni_prefix_cache is not None and nr.num_computed_tokens > 0:
                self.omni_prefix_cache.add_prefix_cached_new_req_id(nr.req_id)
            if self.sampling_state is not None:
                self.sampling_state.add(st)
            if self.code_predictor is not None:
                st.state_slot = self._state_free.pop()
                self._last_hidden[st.state_slot].zero_()
            to_add.append(st)

        for i, rid in enumerate(cached.req_ids):
            st = self.requests[rid]
            st.num_computed_tokens = cached.num_computed_tokens[i]
            new_ids = cached.new_block_ids[i] if cached.new_block_ids else None
            resumed = rid in cached.resumed_req_ids
            if not resumed:
                if new_ids is not None:
                    for have, add in zip(st.block_ids, new_ids):
                        have.extend(add)
            else:
                assert new_ids is not None
                st.block_ids = tuple(list(b) for b in new_ids)
                plen = len(st.prompt_token_ids) or (int(st.prompt_embeds.shape[0]) if st.prompt_embeds is not None else 0)
                st.recompute_until = plen + len(st.output_token_ids)
            idx = ib.req_id_to_index.get(rid)
            if idx is None:
                to_add.append(st)
                continue
            ib.num_computed_tokens_cpu[idx] = st.num_computed_tokens
            if new_ids is not None:
                ib.block_table.append_row(new_ids, idx)
        for st in to_add:
            i = ib.add_request(st)
            self._register_reply_part(st, i)
        # per-request updates riding on the cached requests (`scheduled_cached_reqs.additional_information`,
        # gpu_model_runner.py:1100-1116 -> `_update_intermediate_buffer`: a key's new value REPLACES the old one).  The
        # one key the talker's decode step consumes is the thinker's reply queue (async chunks: the thinker is still
        # generating while the talker decodes): the request's device segment is swapped for the new rows.
        for rid, upd in (cached.additional_information or {}).items():
            st = self.requests.get(rid)
            if st is None or not isinstance(upd, dict) or not upd:
                continue
            st.additional_information = dict(st.additional_information or {}, **upd)
            if "thinker_reply_part" in upd:
                if st.reply_base >= 0:
                    self._reply_alloc.release(st.reply_base, st.reply_rows)
                st.reply_base, st.reply_rows, st.reply_cursor = -1, 0, 0
                row = ib.req_id_to_index.get(rid)
                if row is not None:
                    ib.reply_offset[row], ib.reply_end[row] = -1, 0
                    self._register_reply_part(st, row)

    def _register_reply_part(self, st: CachedRequestState, row: int) -> None:
        """Move a request's ``thinker_reply_part`` rows (one consumed per decode step) into the device-resident
        pool ONCE; the batch rows only keep [offset, end) cursors into it.  A request that left the batch for a
        step (unscheduled / preempted) and comes back continues where it stopped (``reply_cursor``); its segment is
        released when it finishes.  A full pool raises — live rows are never overwritten."""
        if self._generic_hooks:     # the model's preprocess hook consumes the queue from additional_information itself
            return
        if st.reply_base < 0:
            info = st.additional_information or {}
            rp = info.get("thinker_reply_part")
            if not isinstance(rp, torch.Tensor) or rp.numel() == 0:
                return
            if self._reply_pool is None:
                raise RuntimeError("runner built with max_reply_rows=0 but a request carries thinker_reply_part")
            n = rp.shape[0]
            st.reply_base, st.reply_rows, st.reply_cursor = self._reply_alloc.alloc(n), n, 0
            self._reply_pool[st.reply_base:st.reply_base + n].copy_(rp.to(self.dtype), non_blocking=True)
        self.input_batch.reply_offset[row] = st.reply_base + st.reply_cursor
        self.input_batch.reply_end[row] = st.reply_base + st.reply_rows

    # ----------------------------------------------------------- input preparation
    def _is_decode(self, num_sched: np.ndarray, n: int) -> np.ndarray:
        """A decode row: ONE scheduled token and the prompt behind it (``InputBatch.decode_from``).  A one-token row that
        is still inside its prompt (the tail of a chunked prefill of prompt embeddings, a one-row prompt) is a prefill
        row: its input is the prompt's embedding at that position, not the last sampled token.
        A request that recomputes after a preemption is a prefill row until it is back at the frontier of the tokens it
        already has (``recompute_until``): its inputs are those tokens, not the last one."""
        ib = self.input_batch
        computed = ib.num_computed_tokens_cpu[:n]
        return (num_sched[:n] == 1) & (computed >= ib.decode_from[:n]) & (computed + 1 >= ib.recompute_until[:n])

    def _reorder_decodes_first(self, num_sched: np.ndarray) -> np.ndarray:
        """Partition: decode rows first (the attention metadata assumes decodes-first, as upstream backends request
        through ``reorder_batch``)."""
        ib = self.input_batch
        n = ib.num_reqs
        is_dec = self._is_decode(num_sched, n)
        if is_dec.all() or not is_dec.any():
            return num_sched
        i, j = 0, n - 1
        while True:
            while i < n and is_dec[i]:
                i += 1
            while j >= 0 and not is_dec[j]:
                j -= 1
            if i >= j:
                break
            ib.swap(i, j)
            num_sched[[i, j]] = num_sched[[j, i]]
            is_dec[[i, j]] = is_dec[[j, i]]
        return num_sched

    def _prepare_inputs(self, so: SchedulerOutput):
        ib = self.input_batch
        n = ib.num_reqs
        num_sched = np.fromiter((so.num_scheduled_tokens[r] for r in ib.req_ids), dtype=np.int64, count=n)
        num_sched = self._reorder_decodes_first(num_sched)
        total = int(num_sched.sum())
        if total > self.max_num_batched_tokens:
            raise RuntimeError("scheduled more tokens than max_num_batched_tokens")
        qsl = self.query_start_loc.np
        qsl[0] = 0
        np.cumsum(num_sched, out=qsl[1:n + 1])
        computed = ib.num_computed_tokens_cpu[:n]
        if total == n:  # pure decode: one token per request
            req_idx = np.arange(n, dtype=np.int32)
            pos = computed.copy()
        else:
            req_idx = np.repeat(np.arange(n, dtype=np.int32), num_sched)
            pos = np.arange(total, dtype=np.int64) - np.repeat(qsl[:n].astype(np.int64), num_sched) + computed[req_idx]
        self.positions.np[:total] = pos
        seq = (computed + num_sched).astype(np.int32)
        self.seq_lens.np[:n] = seq
        max_seq_len = int(seq.max())
        if max_seq_len > self.max_model_len:
            raise RuntimeError(f"sequence length {max_seq_len} exceeds max_model_len {self.max_model_len}")
        ib.block_table.compute_slot_mapping(req_idx, pos)
        self.logits_indices.np[:n] = qsl[1:n + 1] - 1
        num_decodes = int(self._is_decode(num_sched, n).sum())
        return n, total, num_sched, max_seq_len, num_decodes

    def _commit_inputs(self, n: int, total: int, padded_reqs: int = 0):
        """Host -> device copies of the step's index arrays.  ``padded_reqs`` > n (graph replay at a bucketed batch
        size; pure decode, so tokens == requests): the tail rows [n, padded_reqs) are inert — seq_len 0 (the decode
        kernel writes zeros and dereferences no block-table entry), slot -1 (no KV write), position 0."""
        ib = self.input_batch
        m = max(n, padded_reqs)
        if m > n:
            assert total == n, "only pure-decode batches are padded"
            self.positions.np[n:m] = 0
            self.seq_lens.np[n:m] = 0
            self.query_start_loc.np[n + 1:m + 1] = n + np.arange(1, m - n + 1)
        self.positions.copy_to_gpu(total + (m - n))
        self.seq_lens.copy_to_gpu(m)
        self.query_start_loc.copy_to_gpu(m + 1)
        self.logits_indices.copy_to_gpu(n)
        ib.block_table.commit_block_table(n)
        ib.block_table.commit_slot_mapping(total, total + (m - n))

    def _build_metadata(self, n, total, max_query_len, max_seq_len_bound, num_decodes) -> AttentionMetadata:
        bt = self.input_batch.block_table[0]
        return AttentionMetadata(
            num_actual_tokens=total, max_query_len=max_query_len, query_start_loc=self.query_start_loc.gpu[:n + 1],
            seq_lens=self.seq_lens.gpu[:n], max_seq_len=max_seq_len_bound, block_table=bt.block_table.gpu[:n],
            slot_mapping=bt.slot_mapping.gpu[:total], num_reqs=n, num_decodes=num_decodes,
            decode_partition_size=0, decode_workspace=self._decode_ws)   # 0 = auto, per call

    def _preprocess_embeds(self, n: int, total: int, num_sched: np.ndarray, num_decodes: int) -> None:
        """Fill ``self.inputs_embeds[:total]`` (the model's ``preprocess`` hook, batched)."""
        ib = self.input_batch
        if num_decodes:
            nd = num_decodes
            off = ib.reply_offset[:nd]
            has = (off >= 0) & (off < ib.reply_end[:nd])
            self.reply_idx.np[:nd] = np.where(has, off, -1)
            ib.reply_offset[:nd] = np.where(has, off + 1, off)   # consume one reply row per step
            if not (nd == n and self._ids_on_device_for == tuple(ib.req_ids[:n])):
                self.last_token_ids.np[:nd] = ib.last_token_ids_cpu[:nd]
                self.last_token_ids.copy_to_gpu(nd)
            self.reply_idx.copy_to_gpu(nd)
        # every step — a pure-prefill step too — ends by storing hidden[-1] of each request in its state row
        # (`last_talker_hidden`, qwen3_omni.py:634-640): the rows must belong to THIS step's batch
        self._commit_predictor_inputs(n)
        self._arena.flush()   # positions, lengths, offsets, slots, reply and predictor rows: one copy, before any reader
        if num_decodes:
            self._decode_embeds(num_decodes)
        # prefill rows: per-request prompt embeddings (thinker hidden states) or token embeddings
        t0 = num_decodes
        for i in range(num_decodes, n):
            st = self.requests[ib.req_ids[i]]
            k = int(num_sched[i])
            c = int(ib.num_computed_tokens_cpu[i])
            pl = int(ib.num_prompt_tokens[i])
            a = min(k, max(0, pl - c))          # rows of this chunk that are still inside the prompt
            if a:
                if st.prompt_embeds is not None:
                    self.inputs_embeds[t0:t0 + a].copy_(st.prompt_embeds[c:c + a].to(self.dtype), non_blocking=True)
                else:
                    ids = torch.as_tensor(st.prompt_token_ids[c:c + a], dtype=torch.int64)
                    self.inputs_embeds[t0:t0 + a] = self.model.embed_input_ids(ids.to(self.device))
            if a < k:
                # Recomputation after a preemption runs over tokens the request GENERATED.  Generated token j entered
                # the model as embed(token j) + thinker_reply_part[j] when it was decoded (one reply row per decode
                # step while rows last: qwen2_5_omni.py:766-806), so that is what rebuilds the same KV rows; the rows
                # are still in the request's segment of the device pool.  The chunk's last generated row is the input
                # of the NEXT decode step, so the request's reply cursor ends behind it.
                j0, j1 = c + a - pl, c + k - pl
                ids = torch.as_tensor(st.output_token_ids[j0:j1], dtype=torch.int64).to(self.device)
                emb = self.model.embed_input_ids(ids)
                if st.reply_base >= 0 and j0 < st.reply_rows:
                    r1 = min(j1, st.reply_rows)
                    rows = self._reply_pool[st.reply_base + j0:st.reply_base + r1]
                    emb = emb.clone()
                    emb[:r1 - j0] = (emb[:r1 - j0].float() + rows.float()).to(emb.dtype)
                self.inputs_embeds[t0 + a:t0 + k] = emb
                if st.reply_base >= 0:
                    ib.reply_offset[i] = st.reply_base + min(j1, st.reply_rows)
            t0 += k

    @property
    def model_intermediate_buffer(self) -> dict:
        """``{req_id: per-request dict}`` — the reference's name for what a request's hooks accumulate
        (gpu_model_runner.py:1388-1412); here it IS the request's ``additional_information``."""
        return {rid: st.additional_information for rid, st in self.requests.items() if st.additional_information}

    def _update_intermediate_buffer(self, req_id: str, upd) -> None:
        """Merge a hook's per-request update into the request's ``additional_information`` (the reference's
        ``model_intermediate_buffer``, gpu_model_runner.py:1388-1412): a key's new value replaces the old one; tensors go
        to the host unless the model lists the key in ``gpu_resident_buffer_keys``."""
        st = self.requests.get(req_id)
        if st is None or not isinstance(upd, dict) or not upd:
            return
        gpu_keys = getattr(self.model, "gpu_resident_buffer_keys", ())
        info = dict(st.additional_information or {})
        for k, v in upd.items():
            if isinstance(v, torch.Tensor):
                info[k] = v.detach().clone() if k in gpu_keys else v.detach().to("cpu").contiguous()
            elif isinstance(v, list):
                info[k] = [x.detach().to("cpu").contiguous() if isinstance(x, torch.Tensor) else x for x in v]
            else:
                info[k] = v
        st.additional_information = info

    def _preprocess_embeds_hooks(self, n: int, total: int, num_sched: np.ndarray, num_decodes: int) -> None:
        """``self.inputs_embeds[:total]`` through the MODEL's ``preprocess`` hook, one call per request on its span of
        the step — the reference's loop (gpu_model_runner.py:1264-1309): base rows are the token embeddings (or the
        request's prompt embeddings), the hook returns the rows to overlay and a per-request update."""
        ib = self.input_batch
        self._arena.flush()
        qsl = self.query_start_loc.np
        ids_np = np.zeros(total, dtype=np.int64)
        for i in range(n):
            st = self.requests[ib.req_ids[i]]
            s, k, c = int(qsl[i]), int(num_sched[i]), int(ib.num_computed_tokens_cpu[i])
            if i < num_decodes:
                ids_np[s] = ib.last_token_ids_cpu[i]
            elif st.prompt_token_ids:
                toks = (st.prompt_token_ids + st.output_token_ids)[c:c + k]
                ids_np[s:s + len(toks)] = toks
        input_ids = torch.from_numpy(ids_np).to(self.device)
        self.inputs_embeds[:total] = self.model.embed_input_ids(input_ids.clamp(0, self.config.vocab_size - 1))
        for i in range(n):
            rid = ib.req_ids[i]
            st = self.requests[rid]
            s, k, c = int(qsl[i]), int(num_sched[i]), int(ib.num_computed_tokens_cpu[i])
            if st.prompt_embeds is not None and c < st.prompt_embeds.shape[0]:
                a = min(k, st.prompt_embeds.shape[0] - c)
                self.inputs_embeds[s:s + a].copy_(st.prompt_embeds[c:c + a].to(self.dtype), non_blocking=True)
            if not getattr(self.model, "has_preprocess", False):
                continue
            infos = dict(st.additional_information or {})
            infos["request_id"] = rid
            req_ids, req_embeds, upd = self.model.preprocess(input_ids=input_ids[s:s + k],
                                                             input_embeds=self.inputs_embeds[s:s + k], **infos)
            self._update_intermediate_buffer(rid, upd)
            seg = min(k, int(req_embeds.shape[0]))
            self.inputs_embeds[s:s + seg] = req_embeds[:seg].to(self.dtype)

    def _postprocess_hooks(self, hidden: torch.Tensor, n: int, num_sched: np.ndarray) -> None:
        """The model's ``postprocess`` hook on every request's rows of the step's hidden states
        (``_process_additional_information_updates``, gpu_model_runner.py:1032-1070)."""
        if not getattr(self.model, "has_postprocess", False):
            return
        ib = self.input_batch
        qsl = self.query_start_loc.np
        for i in range(n):
            rid = ib.req_ids[i]
            st = self.requests[rid]
            s = int(qsl[i])
            upd = self.model.postprocess(hidden[s:s + int(num_sched[i])], multimodal_outputs=None,
                                         **(st.additional_information or {}))
            self._update_intermediate_buffer(rid, upd)

    def _commit_predictor_inputs(self, n: int, padded: int = 0) -> None:
        """Per-request state rows, Philox keys and draw bases of the code predictor for this step's batch rows
        (padded graph rows point at the scratch state row ``max_num_reqs``)."""
        if self.code_predictor is None:
            return
        pending = 1 if (self._last_output is not None and not self._last_output.done) else 0
        for i, rid in enumerate(self.input_batch.req_ids[:n]):
            st = self.requests[rid]
            self._state_rows.np[i] = st.state_slot
            self._cp_seed.np[i] = st.sample_seed
            self._cp_offset.np[i] = len(st.output_token_ids) + pending
        m = max(n, padded)
        if m > n:
            self._state_rows.np[n:m] = self.max_num_reqs
            self._cp_seed.np[n:m] = 0
            self._cp_offset.np[n:m] = 0
        self._state_rows.copy_to_gpu(m)
        self._cp_seed.copy_to_gpu(m)
        self._cp_offset.copy_to_gpu(m)

    def _decode_embeds(self, nd: int) -> None:
        if self.code_predictor is not None:
            # talker_mtp (qwen3_omni.py:681-705): codes = code_predictor(last codec token, its embedding, last talker
            # hidden state); input = sum over the G codebook embeddings + the text step of the thinker's reply
            ids = self.last_token_ids.gpu[:nd]
            emb0 = F.embedding(ids.clamp(0, self.config.vocab_size - 1), self.model.language_model.model.embed_tokens)
            last_h = self._last_hidden.index_select(0, self._state_rows.gpu[:nd])
            codes, buf = self.code_predictor._run(nd, ids, emb0, last_h, self._cp_seed.gpu[:nd], self._cp_offset.gpu[:nd],
                                                  True, False)
            summed = buf[:, 1:].float().sum(dim=1)
            if self._reply_pool is not None:
                ridx = self.reply_idx.gpu[:nd]
                step = self._reply_pool.index_select(0, ridx.clamp(min=0)).float() * (ridx >= 0).unsqueeze(-1)
                summed = summed + step
            self.inputs_embeds[:nd] = summed.to(self.dtype)
            self._cp_codes[:nd] = codes[:, :, 0]
            return
        ops.talker_decode_embeds(self.inputs_embeds[:nd], self.model.language_model.model.embed_tokens,
                                 self.last_token_ids.gpu[:nd], self._reply_pool,
                                 self.reply_idx.gpu[:nd] if self._reply_pool is not None else None)

    # ------------------------------------------------------------------- forward
    def _forward(self, n: int, total: int, md: AttentionMetadata):
        """Model forward + logits + fused mask/argmax on the static buffers."""
        with set_forward_context(md):
            hidden = self.model(input_ids=None, positions=self.positions.gpu[:total],
                                inputs_embeds=self.inputs_embeds[:total])
        sample_h = hidden if total == n else hidden.index_select(0, self.logits_indices.gpu[:n])
        if self.code_predictor is not None:   # `last_talker_hidden` of the next step (talker_postprocess, qwen3_omni.py:634-640)
            self._last_hidden.index_copy_(0, self._state_rows.gpu[:n], sample_h)
        logits, ids = self.model.compute_logits_and_greedy(sample_h, self.sampled_ids_gpu[:n])
        return hidden, logits, ids

    def _can_use_graph(self, n: int, total: int, num_decodes: Optional[int] = None) -> bool:
        """Pure-decode batches replay a hipGraph.  With TP > 1 the graph contains the all-reduces (RCCL collectives
        are capturable, probed on this stack with tools/probe_rccl_graph.py; the in-tree two-shot all-reduce is a
        plain kernel launch); VLLM_OMNI_AMD_TP_GRAPH=0 keeps TP > 1 eager."""
        if self.enforce_eager or total != n or torch.device(self.device).type != "cuda" or self._generic_hooks:
            return False
        if num_decodes is not None and num_decodes != n:   # a one-token row still inside its prompt
            return False
        return self.parallel.tp_size == 1 or os.environ.get("VLLM_OMNI_AMD_TP_GRAPH", "1") == "1"

    def graph_batch_size(self, n: int) -> int:
        """Batch size of the graph that runs ``n`` decode rows: 1, 2, 4, then multiples of 8, and ``max_num_reqs``
        itself (upstream's capture-size ladder) — at most max_num_reqs / 8 + 4 graphs, all in one memory pool,
        instead of one private pool per distinct batch size."""
        if n <= 4:
            b = 1 if n <= 1 else (2 if n <= 2 else 4)
        else:
            b = -(-n // 8) * 8
        return min(b, self.max_num_reqs)

    def capture_model(self, batch_sizes: Optional[list[int]] = None) -> None:
        """Capture before serving (the worker's ``compile_or_warm_up_model``); sizes are rounded up the ladder."""
        for b in sorted({self.graph_batch_size(b) for b in (batch_sizes or [])}, reverse=True):
            if b not in self._graphs:
                self._capture(b)

    def _capture(self, n: int):
        """Capture the pure-decode step for batch size ``n`` (grid sized for max_model_len).  The warm-up runs, the
        capture pass and (TP > 1) the watchdogged first replay all execute the step on whatever the static buffers
        hold; device state that a step ADVANCES — the code predictor's `last_talker_hidden` rows and codes — is
        snapshotted before and restored after, so a capture in the middle of serving (first time a batch size is
        seen) leaves the requests' state exactly as the previous step left it.  Everything else a step writes
        (KV rows at the step's slots, outputs) is rewritten with the same values by the replay that follows."""
        md = self._build_metadata(n, n, 1, self.max_model_len, n)
        saved = None
        if self.code_predictor is not None:
            saved = (self._last_hidden.clone(), self._cp_codes.clone())
        # warm-up on a side stream (library handles, hipBLASLt heuristics), then capture
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                if saved is not None:
                    self._last_hidden.copy_(saved[0])
                self._decode_embeds(n)
                self._forward(n, n, md)
        torch.cuda.current_stream().wait_stream(s)
        if saved is not None:
            self._last_hidden.copy_(saved[0])
        if self._graph_pool is None:
            self._graph_pool = torch.cuda.graph_pool_handle()
        g = torch.cuda.CUDAGraph()
        # thread_local: another thread's HIP calls (the RCCL watchdog polling events) must not invalidate
        # the capture.  A capture that does fail leaves a sticky HIP error behind, so it is not caught:
        # failing loudly beats limping on (VLLM_OMNI_AMD_TP_GRAPH=0 / enforce_eager avoid capture).
        with no_gc_during_capture(), torch.cuda.graph(
                g, pool=self._graph_pool, capture_error_mode="thread_local" if self.parallel.tp_size > 1 else "global"):
            self._decode_embeds(n)
            hidden, logits, ids = self._forward(n, n, md)
        if self.parallel.tp_size > 1:
            # first replay of a graph with collectives inside: complete it under a watchdog so that a
            # rank mismatch fails fast instead of hanging the node
            g.replay()
            done = torch.cuda.Event()
            done.record()
            deadline = time.monotonic() + float(os.environ.get("VLLM_OMNI_AMD_GRAPH_WATCHDOG_S", "60"))
            while not done.query():
                if time.monotonic() > deadline:
                    raise RuntimeError("first replay of the TP hipGraph did not finish within the watchdog "
                                       "window (set VLLM_OMNI_AMD_TP_GRAPH=0 to run TP eagerly)")
                time.sleep(0.001)
        if saved is not None:
            self._last_hidden.copy_(saved[0])
            self._cp_codes.copy_(saved[1])
        self._graphs[n] = (g, {"hidden": hidden, "logits": logits, "ids": ids})
        return self._graphs[n]

    # ---------------------------------------------------------------- the step
    @torch.inference_mode()
    def execute_model(self, scheduler_output: SchedulerOutput, intermediate_tensors=None):
        if self.execute_model_state is not None:
            raise RuntimeError("State error: sample_tokens() must be called after execute_model() returns None.")
        if not isinstance(scheduler_output, SchedulerOutput):   # upstream's / the reference's own type
            from .compat import normalize_scheduler_output

            scheduler_output = normalize_scheduler_output(scheduler_output)
        # an output whose host part is still pending may stay pending only across a pure steady-state decode step
        # (same requests, same rows, one token each): everything else reads the host-side token state
        if self._last_output is not None and not self._last_output.done and (
                self._generic_hooks or not self._steady_decode(scheduler_output)):
            self._last_output.get_output()
        kv_done = None
        if scheduler_output.finished_requests_needing_kv_transfer:
            kv_done = self._extract_finished_kv(scheduler_output.finished_requests_needing_kv_transfer)
        self._update_states(scheduler_output)
        if not scheduler_output.total_num_scheduled_tokens:
            if kv_done:
                return OmniModelRunnerOutput([], {}, [], [], kv_extracted_req_ids=kv_done)
            return EMPTY_MODEL_RUNNER_OUTPUT
        if self._h2d_pending:   # the previous step's host->device index copies read the pinned buffers written below
            self._h2d_done.synchronize()
            self._h2d_pending = False
        n, total, num_sched, max_seq_len, num_decodes = self._prepare_inputs(scheduler_output)
        use_graph = self._can_use_graph(n, total, num_decodes)
        if use_graph:
            b = self.graph_batch_size(n)
            self._commit_inputs(n, total, b)
            self._preprocess_host_only(n, b)
            entry = self._graphs.get(b)
            if entry is None:  # first time at this batch size: capture (static inputs are filled)
                entry = self._capture(b)
            self._h2d_done.record()
            self._h2d_pending = True
            g, outs = entry
            g.replay()
            hidden, logits, ids = outs["hidden"], outs["logits"], outs["ids"]
        else:
            self._commit_inputs(n, total)
            if self._generic_hooks:
                self._preprocess_embeds_hooks(n, total, num_sched, num_decodes)
            else:
                self._preprocess_embeds(n, total, num_sched, num_decodes)
            if self._h2d_done is not None:
                self._h2d_done.record()
                self._h2d_pending = True
            md = self._build_metadata(n, total, int(num_sched.max()), max_seq_len, num_decodes)
            hidden, logits, ids = self._forward(n, total, md)
            if self._generic_hooks:
                self._postprocess_hooks(hidden, n, num_sched)
        ib = self.input_batch
        # (a request recomputing after a preemption is partial until it is back at its frontier, prompt + the tokens it
        #  had generated — upstream's `seq_len < req_state.num_tokens`)
        partial = (ib.num_computed_tokens_cpu[:n] + num_sched[:n]) < np.maximum(ib.num_prompt_tokens[:n],
                                                                                ib.recompute_until[:n])
        self.execute_model_state = ExecuteModelState(scheduler_output, logits, ids, hidden, n, total,
                                                     self.query_start_loc.np[:n + 1].copy(), kv_done,
                                                     num_decodes=num_decodes, partial=partial if partial.any() else None)
        pc = self.omni_prefix_cache
        if pc is not None:
            st = self.execute_model_state
            st.slot_mapping_np = ib.block_table[0].slot_mapping.np[:total].copy()
            hits = {}
            for rid in pc._new_req_cache_hit_ids:
                row = ib.req_id_to_index.get(rid)
                if row is not None:   # block_table[req, :num_computed // block_size] (core/prefix_cache.py:255-264)
                    hits[rid] = pc._get_cached_block_ids(row, ib).clone()
            st.prefix_hit_blocks = hits
        return None

    def _steady_decode(self, so: SchedulerOutput) -> bool:
        ib = self.input_batch
        if so.scheduled_new_reqs or so.finished_req_ids or so.finished_requests_needing_kv_transfer:
            return False
        if so.scheduled_cached_reqs.resumed_req_ids or so.total_num_scheduled_tokens != ib.num_reqs:
            return False
        if self._ids_on_device_for != tuple(ib.req_ids) or len(so.num_scheduled_tokens) != ib.num_reqs:
            return False
        return all(so.num_scheduled_tokens.get(r) == 1 for r in ib.req_ids)

    def _preprocess_host_only(self, n: int, padded: int) -> None:
        """Graph path: only the host->device index copies; the gather-add itself is in the graph.  Rows [n, padded)
        of a bucketed replay get no reply row; their token id is whatever the buffer holds (any id is a valid
        embedding lookup, ids outside the table read as a zero row)."""
        ib = self.input_batch
        off = ib.reply_offset[:n]
        has = (off >= 0) & (off < ib.reply_end[:n])
        self.reply_idx.np[:n] = np.where(has, off, -1)
        self.reply_idx.np[n:padded] = -1
        ib.reply_offset[:n] = np.where(has, off + 1, off)
        if self._ids_on_device_for != tuple(ib.req_ids[:n]):   # else the last sample_tokens left them on the device
            self.last_token_ids.np[:n] = ib.last_token_ids_cpu[:n]
            self.last_token_ids.copy_to_gpu(n)
        self.reply_idx.copy_to_gpu(max(n, padded))
        self._commit_predictor_inputs(n, padded)
        self._arena.flush()   # one host->device copy for everything the replay reads

    @torch.inference_mode()
    def sample_tokens(self, grammar_output=None, async_output: bool = False):
        """Second phase of the step (gpu_ar_model_runner.py:626-953): sample, hand ids + hidden states back.

        ``async_output=True`` returns an ``AsyncOmniModelRunnerOutput`` right after the device->host copy is
        enqueued: the caller may run ``execute_model`` of the next step before ``get_output()`` — the drawn ids
        stay on the device for the next step's embedding lookup, the host bookkeeping happens in ``get_output()``."""
        st = self.execute_model_state
        if st is None:
            raise RuntimeError("State error: sample_tokens() called without a preceding execute_model().")
        self.execute_model_state = None
        ib = self.input_batch
        n = st.num_reqs
        ids = st.sampled_ids
        # non-greedy requests (or a repetition penalty): one launch of the device sampler on the masked
        # logits — penalty, temperature, top-k, top-p and the draw; greedy rows of the batch go through it too
        states = [self.requests[r] for r in ib.req_ids[:n]]
        model_ids = self._model_sampler(st.logits, states, n)
        if model_ids is not None:
            ids = model_ids
        elif SamplingState.needed([s_.sampling_params for s_ in states]):
            if self.sampling_state is None:
                raise RuntimeError("non-greedy sampling needs the device sampler (GPU, vocabulary <= 15744): "
                                   "there is no host fallback")
            pending = 1 if (self._last_output is not None and not self._last_output.done) else 0
            ids = self.sampling_state.sample(st.logits, states, self.sampled_ids_gpu, drawn_ahead=pending,
                                             discard=st.partial)
        want_hidden = self.engine_output_type != "text"
        slot = self._out_slot
        self._out_slot = (slot + 1) % self.OUT_SLOTS
        if self._out_inflight[slot] is not None:      # its pinned buffers are about to be reused
            self._out_inflight[slot].get_output()
        if self.copy_stream is not None:
            cur = torch.cuda.current_stream()
            # feed the drawn ids to the next step's embedding lookup on the device (pure-decode batches)
            if st.num_decodes == n:
                self.last_token_ids.gpu[:n].copy_(ids[:n])
                self._ids_on_device_for = tuple(ib.req_ids[:n])
            else:
                self._ids_on_device_for = None
            cur.wait_event(self._out_done[slot])      # the staging slot's previous copy (two steps ago) has drained
            self._out_ids_stage[slot][:n].copy_(ids[:n])
            if self.code_predictor is not None:
                self._out_codes_stage[slot][:n].copy_(self._cp_codes[:n])
            if want_hidden:
                self._out_hidden_stage[slot][:st.num_tokens].copy_(st.hidden_states[:st.num_tokens])
            self.copy_stream.wait_stream(cur)
            with torch.cuda.stream(self.copy_stream):
                self._out_ids_cpu[slot][:n].copy_(self._out_ids_stage[slot][:n], non_blocking=True)
                if self.code_predictor is not None:
                    self._out_codes_cpu[slot][:n].copy_(self._out_codes_stage[slot][:n], non_blocking=True)
                if want_hidden:
                    self._out_hidden_cpu[slot][:st.num_tokens].copy_(self._out_hidden_stage[slot][:st.num_tokens],
                                                                     non_blocking=True)
                self._out_done[slot].record(self.copy_stream)
        else:
            self._out_ids_cpu[slot][:n].copy_(ids[:n])
            if want_hidden:
                self._out_hidden_cpu[slot][:st.num_tokens].copy_(st.hidden_states[:st.num_tokens])
        prev = self._last_output if (self._last_output is not None and not self._last_output.done) else None
        out = AsyncOmniModelRunnerOutput(self, st, slot, list(ib.req_ids[:n]), dict(ib.req_id_to_index), prev)
        self._out_inflight[slot] = out
        self._last_output = out
        return out if async_output else out.get_output()

    def _model_sampler(self, logits: torch.Tensor, states: list, n: int) -> Optional[torch.Tensor]:
        """``prefer_model_sampler`` (gpu_ar_model_runner.py:633-650): a model with its own ``sample(logits,
        sampling_metadata)`` draws the step's tokens; None from it (or no such model) leaves the runner's sampler in
        charge.  Its ids land in the same device buffer the runner's sampler writes."""
        fn = getattr(self.model, "sample", None)
        if logits is None or not callable(fn) or not getattr(self.model, "prefer_model_sampler", False):
            return None
        sp = [s_.sampling_params for s_ in states]
        dev = logits.device
        greedy = [p.temperature <= 0 for p in sp]
        md = SamplingMetadata(
            temperature=None if all(greedy) else torch.tensor([max(p.temperature, 0.0) for p in sp], dtype=torch.float32, device=dev),
            all_greedy=all(greedy), all_random=not any(greedy),
            top_p=None if all(p.top_p >= 1.0 for p in sp) else torch.tensor([p.top_p for p in sp], dtype=torch.float32, device=dev),
            top_k=None if all(p.top_k <= 0 for p in sp) else torch.tensor([p.top_k for p in sp], dtype=torch.int32, device=dev),
            output_token_ids=[list(s_.output_token_ids) for s_ in states])
        out = fn(logits[:n], md)
        if out is None:
            return None
        toks = getattr(out, "sampled_token_ids", out)
        ids = self.sampled_ids_gpu[:n]
        ids.copy_(torch.as_tensor(toks, device=dev).reshape(-1)[:n].to(torch.int64))
        return ids

    def _finish_output(self, out: AsyncOmniModelRunnerOutput) -> OmniModelRunnerOutput:
        """Host half of a step's output: wait for its device->host copy, update the token state."""
        st, slot = out.st, out.slot
        n = st.num_reqs
        if self._out_done is not None:
            self._out_done[slot].synchronize()
        # the step's kernels have completed: an in-tree all-reduce whose peer never arrived left its output unwritten
        # and said so in a host-mapped status word — raise before these tokens / hidden states reach the scheduler
        ar = getattr(self.parallel, "custom_ar", None)
        if ar is not None:
            ar.poll()
        ids_np = self._out_ids_cpu[slot][:n].numpy().copy()
        ib = self.input_batch
        # bookkeeping: last token per row feeds the next step's embedding lookup.  Rows are addressed by request
        # id: the batch may have been reordered since the step ran.
        for i, (rid, tok) in enumerate(zip(out.req_ids, ids_np.tolist())):
            if st.partial is not None and st.partial[i]:
                continue   # the prompt is not finished: nothing was generated
            req = self.requests.get(rid)
            if req is not None:
                req.output_token_ids.append(tok)
            row = ib.req_id_to_index.get(rid)
            if row is not None:
                ib.last_token_ids_cpu[row] = tok
        sampled = ids_np.reshape(n, 1).tolist()
        if st.partial is not None:
            sampled = [[] if st.partial[i] else t for i, t in enumerate(sampled)]
        pooler: list[Optional[dict[str, Any]]]
        if self.engine_output_type != "text":
            q = st.query_start_loc_np
            hidden = self._out_hidden_cpu[slot][:st.num_tokens]   # views of the pinned ring (see __init__)
            if self.copy_outputs:                                 # caller-owned, as the reference's .to("cpu") is
                hidden = hidden.clone()
            chunks = list(torch.split(hidden, np.diff(q).tolist()))
            pc = self.omni_prefix_cache
            if pc is not None:
                # `_maybe_update_prefix_cache` + `_maybe_get_combined_prefix_cache_tensors`
                # (gpu_ar_model_runner.py:205-260,539,846-856), on the host copy this step already made: write the
                # step's rows at their KV slots, then give a request that entered on a prefix hit
                # cat([rows of its cached full blocks, this step's rows])
                pc.update_omni_tensor_prefix_cache(hidden, None, st.num_tokens, torch.from_numpy(st.slot_mapping_np))
                cache = pc.hidden_states_cache
                for i, rid in enumerate(out.req_ids):
                    blocks = (st.prefix_hit_blocks or {}).get(rid)
                    if blocks is not None and blocks.numel():
                        cached = cache[blocks.to(torch.int64)].reshape(-1, cache.shape[-1])
                        chunks[i] = torch.cat([cached, chunks[i]], dim=0)
            pooler = [{"hidden": c} for c in chunks]
        else:
            pooler = [None] * n
        if self.code_predictor is not None and st.num_decodes == n:
            # the codes that produced THIS step's talker input (multimodal output `code_predictor_codes`,
            # qwen3_omni.py:480-494), one row of G codes per decode request
            codes = self._out_codes_cpu[slot][:n]
            if self.copy_outputs:
                codes = codes.clone()
            pooler = [dict(p_ or {}, code_predictor_codes=codes[i]) for i, p_ in enumerate(pooler)]
        if self._out_inflight[slot] is out:
            self._out_inflight[slot] = None
        return OmniModelRunnerOutput(out.req_ids, out.req_index, sampled, pooler,
                                     kv_extracted_req_ids=st.kv_extracted_req_ids)

    # ---------------------------------------------------------------- KV hand-off
    def _extract_finished_kv(self, finished: dict[str, dict]) -> list[str]:
        """Gather finished requests' K/V before their blocks are reused (A2 consumer;
        kv_transfer_manager.py:721-862).  On the GPU each request becomes ONE packed payload — extraction,
        (optional) sender-side head slice and serialisation in one kernel launch (kv_transfer.pack_request_kv,
        the `to_gpu_tensor` wire format) — kept in ``self.kv_payloads[rid]``; ``self.kv_extracted[rid]`` is the
        decoded view of it (key_cache / value_cache lists, block_ids, metadata)."""
        from .kv_transfer import KVCacheTransferData, pack_request_kv

        done = []
        on_gpu = torch.device(self.device).type == "cuda"
        for rid, info in finished.items():
            block_ids, seq_len = list(info.get("block_ids", [])), int(info.get("seq_len", 0))
            if on_gpu:
                payload = pack_request_kv(self.kv_caches, rid, block_ids, seq_len, self.block_size, str(self.dtype))
                if payload is not None:
                    self.kv_payloads[rid] = payload
                    dec = KVCacheTransferData.from_bytes_gpu(payload)
                    self.kv_extracted[rid] = {"key_cache": dec["layer_blocks"]["key_cache"],
                                              "value_cache": dec["layer_blocks"]["value_cache"],
                                              "block_ids": dec["block_ids"], "metadata": dec["metadata"]}
            else:  # host-logic unit tests: the cache is a host tensor, indexing only
                keys, vals = extract_request_kv(self.kv_caches, block_ids, seq_len)
                self.kv_extracted[rid] = {"key_cache": keys, "value_cache": vals, "block_ids": block_ids,
                                          "metadata": {"block_size": self.block_size, "num_layers": len(self.kv_caches),
                                                       "dtype": str(self.dtype), "seq_len": seq_len}}
            done.append(rid)
        # hand-offs nobody collected (take_kv_payload / take_kv_extracted) do not pile up for ever
        for store in (self.kv_payloads, self.kv_extracted):
            while len(store) > self.max_kept_kv:
                store.pop(next(iter(store)))
        return done

    def take_kv_payload(self, req_id: str) -> Optional[torch.Tensor]:
        """Hand the packed wire payload of a finished request to the connector and forget it (the reference's
        transfer manager sends and drops: kv_transfer_manager.py:721-784)."""
        self.kv_extracted.pop(req_id, None)
        return self.kv_payloads.pop(req_id, None)

    def take_kv_extracted(self, req_id: str) -> Optional[dict]:
        self.kv_payloads.pop(req_id, None)
        return self.kv_extracted.pop(req_id, None)

"""Data types that cross the scheduler -> runner -> scheduler boundary (SURVEY §8b runner contract).

Field names follow upstream's ``SchedulerOutput`` / ``NewRequestData`` / ``CachedRequestData`` as the
reference consumes them (vllm_omni/worker/gpu_model_runner.py:241-556) and the reference's own
extensions (``additional_information`` on new/cached requests: vllm_omni/core/sched/output.py;
``OmniModelRunnerOutput``: vllm_omni/outputs.py:39-54).
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Any, Optional

import torch


@dataclass
class SamplingParams:
    temperature: float = 0.0
    top_k: int = 0
    top_p: float = 1.0
    repetition_penalty: float = 1.0
    max_tokens: int = 16
    seed: Optional[int] = None
    stop_token_ids: list[int] = field(default_factory=list)
    ignore_eos: bool = False


@dataclass
class SamplingMetadata:
    """What a model's own ``sample(logits, sampling_metadata)`` receives when it sets ``prefer_model_sampler``
    (gpu_ar_model_runner.py:626-650) — the fields of upstream's ``vllm.v1.sample.metadata.SamplingMetadata`` a model
    sampler reads: per-row temperature / top-k / top-p on the logits' device, the two batch flags, and the tokens
    generated so far per row."""

    temperature: Any = None          # Tensor[n] float32, or None when all rows are greedy
    all_greedy: bool = True
    all_random: bool = False
    top_p: Any = None                # Tensor[n] float32 or None (no row restricts)
    top_k: Any = None                # Tensor[n] int32 or None
    output_token_ids: list = field(default_factory=list)
    generators: dict = field(default_factory=dict)


class RequestStatus(enum.IntEnum):
    """Upstream's ``vllm.v1.request.RequestStatus`` as the reference uses it (omni_ar_scheduler.py:143,362-385):
    everything after ``PREEMPTED`` is a finished state."""

    WAITING = 0
    WAITING_FOR_CHUNK = 1
    RUNNING = 2
    PREEMPTED = 3
    FINISHED_STOPPED = 4
    FINISHED_LENGTH_CAPPED = 5
    FINISHED_ABORTED = 6
    FINISHED_ERROR = 7

    @staticmethod
    def is_finished(status: "RequestStatus") -> bool:
        return status > RequestStatus.PREEMPTED


_FINISH_REASON = {RequestStatus.FINISHED_STOPPED: "stop", RequestStatus.FINISHED_LENGTH_CAPPED: "length",
                  RequestStatus.FINISHED_ABORTED: "abort", RequestStatus.FINISHED_ERROR: "error"}


@dataclass(eq=False)
class Request:
    """Scheduler-side request state (the subset of upstream's ``vllm.v1.request.Request`` the scheduler, the
    allocator and the runner read)."""

    request_id: str
    prompt_token_ids: list[int]
    sampling_params: SamplingParams = field(default_factory=SamplingParams)
    prompt_embeds: Optional[torch.Tensor] = None
    additional_information: Optional[dict] = None
    num_computed_tokens: int = 0
    output_token_ids: list[int] = field(default_factory=list)
    status: RequestStatus = RequestStatus.WAITING
    eos_token_id: Optional[int] = None
    client_index: int = 0
    stop_reason: Optional[int] = None
    num_cached_tokens: int = -1          # tokens served by the prefix cache at first scheduling (-1 = not scheduled yet)
    num_preemptions: int = 0
    arrival_time: float = 0.0
    external_req_id: Optional[str] = None

    @property
    def num_prompt_tokens(self) -> int:
        # upstream's length_from_prompt_token_ids_or_embeds: a prompt may come as embeddings only
        if self.prompt_token_ids:
            return len(self.prompt_token_ids)
        return int(self.prompt_embeds.shape[0]) if self.prompt_embeds is not None else 0

    @property
    def num_output_tokens(self) -> int:
        return len(self.output_token_ids)

    @property
    def num_tokens(self) -> int:
        return self.num_prompt_tokens + len(self.output_token_ids)

    @property
    def all_token_ids(self) -> list[int]:
        prompt = self.prompt_token_ids if self.prompt_token_ids else [0] * self.num_prompt_tokens
        return list(prompt) + list(self.output_token_ids)

    @property
    def max_tokens(self) -> int:
        return int(self.sampling_params.max_tokens)

    def is_finished(self) -> bool:
        return RequestStatus.is_finished(self.status)

    def get_finished_reason(self) -> Optional[str]:
        return _FINISH_REASON.get(self.status)


@dataclass
class NewRequestData:
    req_id: str
    prompt_token_ids: list[int]
    sampling_params: SamplingParams
    block_ids: tuple[list[int], ...]
    num_computed_tokens: int
    prompt_embeds: Optional[torch.Tensor] = None
    additional_information: Optional[dict] = None


@dataclass
class CachedRequestData:
    req_ids: list[str] = field(default_factory=list)
    resumed_req_ids: set[str] = field(default_factory=set)
    new_token_ids: list[list[int]] = field(default_factory=list)
    new_block_ids: list[Optional[tuple[list[int], ...]]] = field(default_factory=list)
    num_computed_tokens: list[int] = field(default_factory=list)
    num_output_tokens: list[int] = field(default_factory=list)
    additional_information: dict[str, Optional[dict]] = field(default_factory=dict)

    @property
    def num_reqs(self) -> int:
        return len(self.req_ids)


@dataclass
class SchedulerOutput:
    scheduled_new_reqs: list[NewRequestData] = field(default_factory=list)
    scheduled_cached_reqs: CachedRequestData = field(default_factory=CachedRequestData)
    num_scheduled_tokens: dict[str, int] = field(default_factory=dict)
    total_num_scheduled_tokens: int = 0
    finished_req_ids: set[str] = field(default_factory=set)
    num_common_prefix_blocks: list[int] = field(default_factory=lambda: [0])
    new_block_ids_to_zero: Optional[list[int]] = None
    # finished requests whose KV must be extracted before their blocks are reused
    # (OmniSchedulerOutput.finished_requests_needing_kv_transfer, core/sched/output.py:73-77)
    finished_requests_needing_kv_transfer: dict[str, dict] = field(default_factory=dict)


@dataclass
class OmniModelRunnerOutput:
    """What ``sample_tokens`` returns.  ``pooler_output[i] = {"hidden": Tensor[n_i, H], **mm}``
    unless the stage's engine_output_type is "text" (gpu_ar_model_runner.py:883-926)."""

    req_ids: list[str]
    req_id_to_index: dict[str, int]
    sampled_token_ids: list[list[int]]
    pooler_output: list[Optional[dict[str, Any]]]
    multimodal_outputs: Optional[dict] = None
    kv_extracted_req_ids: Optional[list[str]] = None
    omni_connector_output: Any = None


EMPTY_MODEL_RUNNER_OUTPUT = OmniModelRunnerOutput([], {}, [], [])


@dataclass
class EngineCoreOutput:
    """One request's share of a step, as the scheduler reports it to the engine (the fields of upstream's
    ``EngineCoreOutput`` the reference fills, omni_ar_scheduler.py:286-292,416-432)."""

    request_id: str
    new_token_ids: list[int]
    finish_reason: Optional[str] = None
    stop_reason: Optional[int] = None
    pooling_output: Optional[dict] = None
    kv_transfer_params: Optional[dict] = None
    num_cached_tokens: int = 0

    @property
    def finished(self) -> bool:
        return self.finish_reason is not None


@dataclass
class EngineCoreOutputs:
    outputs: list[EngineCoreOutput] = field(default_factory=list)
    finished_requests: Optional[set[str]] = None
    scheduler_stats: Optional[dict] = None

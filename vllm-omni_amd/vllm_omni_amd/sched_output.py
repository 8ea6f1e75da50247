"""Data types that cross the scheduler -> runner -> scheduler boundary (SURVEY §8b runner contract).

Field names follow upstream's ``SchedulerOutput`` / ``NewRequestData`` / ``CachedRequestData`` as the
reference consumes them (vllm_omni/worker/gpu_model_runner.py:241-556) and the reference's own
extensions (``additional_information`` on new/cached requests: vllm_omni/core/sched/output.py;
``OmniModelRunnerOutput``: vllm_omni/outputs.py:39-54).
"""
from __future__ import annotations

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


@dataclass
class Request:
    """Scheduler-side request state (the subset the allocator and runner read)."""

    request_id: str
    prompt_token_ids: list[int]
    sampling_params: SamplingParams = field(default_factory=SamplingParams)
    prompt_embeds: Optional[torch.Tensor] = None
    additional_information: Optional[dict] = None
    num_computed_tokens: int = 0
    output_token_ids: list[int] = field(default_factory=list)

    @property
    def num_tokens(self) -> int:
        return len(self.prompt_token_ids) + len(self.output_token_ids)


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

"""Adapters from the objects the reference hands a worker to this package's own types.

The reference constructs ``worker_cls(vllm_config=..., local_rank, rank, distributed_init_method,
is_driver_worker)`` (upstream ``Worker.__init__``; class named by string,
vllm_omni/engine/stage_init_utils.py:86-101) and then ``GPUARModelRunner(self.vllm_config, self.device)``
(vllm_omni/worker/gpu_ar_worker.py:102); it feeds ``execute_model`` upstream's ``SchedulerOutput`` whose request
data carry msgspec payloads (``PromptEmbedsPayload`` / ``AdditionalInformationPayload``,
vllm_omni/engine/__init__.py:16-57, decoded at vllm_omni/worker/gpu_model_runner.py:353-386).  Everything here is
duck-typed — ``vllm`` is not importable in the build container, and a maintainer's ``VllmConfig`` is just an object
with ``model_config`` / ``cache_config`` / ``parallel_config`` / ``scheduler_config`` / ``device_config`` on it.
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch

from .sched_output import CachedRequestData, NewRequestData, SamplingParams, SchedulerOutput

_DTYPES = {"bfloat16": torch.bfloat16, "float16": torch.float16, "half": torch.float16, "float32": torch.float32,
           "float": torch.float32, "auto": torch.bfloat16}


def _get(obj: Any, key: str, default: Any = None) -> Any:
    """Attribute or item — stage configs arrive as objects, dicts or OmegaConf nodes
    (``_get_attr_or_item``, stage_init_utils.py:103-107)."""
    if obj is None:
        return default
    if isinstance(obj, dict):
        return obj.get(key, default)
    v = getattr(obj, key, default)
    return default if v is None else v


def is_vllm_config(obj: Any) -> bool:
    return obj is not None and hasattr(obj, "model_config") and hasattr(obj, "cache_config")


def torch_dtype(d: Any) -> torch.dtype:
    if isinstance(d, torch.dtype):
        return d
    try:
        return _DTYPES[str(d).replace("torch.", "")]
    except KeyError:
        raise ValueError(f"unsupported model dtype {d!r}") from None


# ------------------------------------------------------------------------------------ hf_config -> TalkerConfig
def talker_config_from_hf(hf: Any):
    """``vllm_config.model_config.hf_config`` -> ``TalkerConfig``.

    Reads the fields the reference's model code reads: the talker sub-config when the top-level Omni config is
    handed over (``config.talker_config``, qwen2_5_omni.py:644-647; Qwen3: ``talker_config.text_config``), head
    geometry (qwen2_old.py:84-106), ``rope_theta`` / ``rope_scaling.mrope_section`` (:126-134), the codec token ids
    of ``bad_word_processor`` (qwen2_5_omni_talker.py:150-169) and the sparse-MoE fields of the Qwen3-Omni talker
    (qwen3_omni_moe_talker.py:476-513)."""
    from .talker import TalkerConfig

    if isinstance(hf, TalkerConfig):
        return hf
    top = hf
    if _get(hf, "talker_config") is not None:
        hf = _get(hf, "talker_config")
    outer = hf
    if _get(hf, "text_config") is not None:       # Qwen3-Omni: talker_config.text_config holds the decoder shape
        hf = _get(hf, "text_config")
    hidden = int(_get(hf, "hidden_size"))
    heads = int(_get(hf, "num_attention_heads"))
    rope_params = _get(hf, "rope_parameters") or {}
    rope_scaling = _get(hf, "rope_scaling") or {}
    theta = _get(hf, "rope_theta") or _get(rope_params, "rope_theta") or _get(rope_params, "base") or 10000.0
    mrope = _get(rope_scaling, "mrope_section") or _get(rope_params, "mrope_section")
    model_type = str(_get(outer, "model_type", "") or _get(top, "model_type", "") or "")
    experts = int(_get(hf, "num_experts", 0) or 0)
    qwen3 = "qwen3" in model_type or experts > 0 or bool(_get(hf, "qk_norm", False))

    def tok(name):   # codec ids live on the talker config, not on its text_config
        v = _get(outer, name)
        return v if v is not None else _get(hf, name)

    return TalkerConfig(
        name=model_type or "talker", hidden_size=hidden,
        embedding_size=int(_get(hf, "embedding_size", hidden)), num_hidden_layers=int(_get(hf, "num_hidden_layers")),
        num_attention_heads=heads, num_key_value_heads=int(_get(hf, "num_key_value_heads", heads)),
        head_dim=int(_get(hf, "head_dim", hidden // heads)),
        intermediate_size=int(_get(hf, "intermediate_size", 0) or _get(hf, "moe_intermediate_size", 0)),
        vocab_size=int(_get(hf, "vocab_size")), rms_norm_eps=float(_get(hf, "rms_norm_eps", 1e-6)),
        rope_theta=float(theta), max_position_embeddings=int(_get(hf, "max_position_embeddings", 32768)),
        mrope_section=list(mrope) if mrope else None, qk_norm=qwen3,
        attention_bias=bool(_get(hf, "attention_bias", not qwen3)),
        tts_codec_start_token_id=tok("tts_codec_start_token_id"), tts_codec_end_token_id=tok("tts_codec_end_token_id"),
        tts_codec_pad_token_id=tok("tts_codec_pad_token_id"), tts_codec_mask_token_id=tok("tts_codec_mask_token_id"),
        num_experts=experts, num_experts_per_tok=int(_get(hf, "num_experts_per_tok", 0) or 0),
        moe_intermediate_size=int(_get(hf, "moe_intermediate_size", 0) or 0),
        norm_topk_prob=bool(_get(hf, "norm_topk_prob", False)),
        shared_expert_intermediate_size=int(_get(hf, "shared_expert_intermediate_size", 0) or 0),
        input_projection=not qwen3)


# ------------------------------------------------------------------------------------ VllmConfig -> WorkerConfig
def worker_config_from_vllm_config(vc: Any):
    """Map the sub-configs the reference's worker reads (gpu_ar_worker.py:29-106, base.py:77-156) onto
    ``WorkerConfig``: ``model_config`` (hf_config, dtype, max_model_len, seed, enforce_eager, quantization,
    engine_output_type), ``cache_config`` (block_size, gpu_memory_utilization, kv_cache_memory_bytes),
    ``parallel_config`` (tensor_parallel_size, world_size, data-parallel local rank), ``scheduler_config``
    (max_num_seqs, max_num_batched_tokens), ``device_config.device_type``."""
    from .worker import WorkerConfig

    mc, cc = vc.model_config, vc.cache_config
    pc, sc = _get(vc, "parallel_config"), _get(vc, "scheduler_config")
    dc = _get(vc, "device_config")
    tp = int(_get(pc, "tensor_parallel_size", 1))
    pp = int(_get(pc, "pipeline_parallel_size", 1))
    if pp != 1:
        raise ValueError("pipeline parallelism is outside this backend's path (tensor/data parallel only)")
    quant = _get(mc, "quantization")
    if quant is not None and str(quant).lower() not in ("int8",):
        raise ValueError(f"unsupported quantization {quant!r} (None or 'int8')")
    dev_type = str(_get(dc, "device_type", None) or _get(dc, "device", "cuda"))
    if dev_type.startswith("cuda"):
        dev_type = "cuda"
    block_size = int(_get(cc, "block_size", 16) or 16)
    max_len = int(_get(mc, "max_model_len", 4096))
    max_seqs = int(_get(sc, "max_num_seqs", 256))
    return WorkerConfig(
        model_config=talker_config_from_hf(mc.hf_config), dtype=torch_dtype(_get(mc, "dtype", "bfloat16")),
        tensor_parallel_size=tp, world_size=int(_get(pc, "world_size", tp)),
        gpu_memory_utilization=float(_get(cc, "gpu_memory_utilization", 0.9)),
        kv_cache_memory_bytes=_get(cc, "kv_cache_memory_bytes"), block_size=block_size,
        max_num_seqs=max_seqs, max_model_len=max_len,
        max_num_batched_tokens=int(_get(sc, "max_num_batched_tokens", max(max_len, max_seqs))),
        enforce_eager=bool(_get(mc, "enforce_eager", False)),
        engine_output_type=str(_get(mc, "engine_output_type", "latent") or "latent"),
        device_type=dev_type, distributed_backend=str(_get(vc, "dist_backend", "nccl")),
        seed=int(_get(mc, "seed", 0) or 0), quantization=None if quant is None else "int8",
        dp_local_rank=int(_get(pc, "data_parallel_rank_local", None) or _get(pc, "data_parallel_index", 0) or 0),
        distributed_executor_backend=_get(pc, "distributed_executor_backend"),
        data_parallel_backend=_get(pc, "data_parallel_backend"),
        nnodes_within_dp=int(_get(pc, "nnodes_within_dp", 1) or 1),
        enable_prefix_caching=bool(_get(cc, "enable_prefix_caching", False)))


# ------------------------------------------------------------------------------------ request payloads
def decode_prompt_embeds(payload: Any) -> Optional[torch.Tensor]:
    """``PromptEmbedsPayload`` (data bytes, shape, dtype name) or a tensor -> CPU tensor
    (gpu_model_runner.py:353-366)."""
    if payload is None or isinstance(payload, torch.Tensor):
        return payload
    data, shape, dt = _get(payload, "data"), _get(payload, "shape"), _get(payload, "dtype")
    if data is None or shape is None:
        raise TypeError(f"cannot decode prompt_embeds of type {type(payload).__name__}")
    if str(dt) == "bfloat16":      # numpy has no bf16: reinterpret the 16-bit words
        arr = np.frombuffer(data, dtype=np.int16).reshape(tuple(shape)).copy()
        return torch.from_numpy(arr).view(torch.bfloat16)
    return torch.from_numpy(np.frombuffer(data, dtype=np.dtype(str(dt))).reshape(tuple(shape)).copy())


def deserialize_additional_information(payload: Any) -> dict:
    """dict as-is; ``AdditionalInformationPayload`` (``.entries`` of tensor / list / scalar entries) decoded entry
    by entry; None -> {} (vllm_omni/engine/serialization.py:73-113)."""
    if payload is None:
        return {}
    if isinstance(payload, dict):
        return payload
    entries = getattr(payload, "entries", None)
    if not isinstance(entries, dict):
        return {}
    info: dict[str, Any] = {}
    for k, e in entries.items():
        if getattr(e, "tensor_data", None) is not None:
            dt = str(getattr(e, "tensor_dtype", None) or "float32")
            shape = tuple(getattr(e, "tensor_shape", None) or ())
            if dt == "bfloat16":
                info[k] = torch.from_numpy(np.frombuffer(e.tensor_data, dtype=np.int16).reshape(shape).copy()).view(torch.bfloat16)
            else:
                info[k] = torch.from_numpy(np.frombuffer(e.tensor_data, dtype=np.dtype(dt)).reshape(shape).copy())
        elif getattr(e, "list_data", None) is not None:
            info[k] = e.list_data
        elif getattr(e, "scalar_data", None) is not None:
            info[k] = e.scalar_data
        else:
            info[k] = None
    return info


def normalize_sampling_params(sp: Any) -> SamplingParams:
    """Upstream ``SamplingParams`` -> ours (same field names; upstream writes "top-k off" as 0 or -1)."""
    if isinstance(sp, SamplingParams):
        return sp
    if sp is None:
        return SamplingParams()
    top_k = int(_get(sp, "top_k", 0) or 0)
    return SamplingParams(
        temperature=float(_get(sp, "temperature", 0.0) or 0.0), top_k=max(top_k, 0),
        top_p=float(_get(sp, "top_p", 1.0) or 1.0), repetition_penalty=float(_get(sp, "repetition_penalty", 1.0) or 1.0),
        max_tokens=int(_get(sp, "max_tokens", 16) or 16), seed=_get(sp, "seed"),
        stop_token_ids=list(_get(sp, "stop_token_ids", []) or []))


def _block_ids(b: Any) -> tuple[list[int], ...]:
    return tuple(list(g) for g in b)


def normalize_scheduler_output(so: Any) -> SchedulerOutput:
    """Upstream ``SchedulerOutput`` / the reference's ``OmniSchedulerOutput`` (core/sched/output.py) -> ours.
    Our own type passes through untouched.  Fields the path does not use (encoder inputs, spec-decode tokens,
    grammar bitmasks, connector metadata) are ignored; the ones it does use keep their upstream names."""
    if isinstance(so, SchedulerOutput):
        return so
    new = []
    for nr in _get(so, "scheduled_new_reqs", []) or []:
        new.append(NewRequestData(
            req_id=nr.req_id, prompt_token_ids=list(_get(nr, "prompt_token_ids", []) or []),
            sampling_params=normalize_sampling_params(_get(nr, "sampling_params")),
            block_ids=_block_ids(nr.block_ids), num_computed_tokens=int(nr.num_computed_tokens),
            prompt_embeds=decode_prompt_embeds(_get(nr, "prompt_embeds")),
            additional_information=deserialize_additional_information(_get(nr, "additional_information")) or None))
    c = _get(so, "scheduled_cached_reqs")
    req_ids = list(_get(c, "req_ids", []) or [])
    resumed = _get(c, "resumed_req_ids")
    if resumed is None:   # older upstream: a list of flags parallel to req_ids
        flags = _get(c, "resumed_from_preemption", []) or []
        resumed = {r for r, f in zip(req_ids, flags) if f}
    nbi = [None if b is None else _block_ids(b) for b in (_get(c, "new_block_ids", []) or [])]
    cached = CachedRequestData(
        req_ids=req_ids, resumed_req_ids=set(resumed), new_token_ids=list(_get(c, "new_token_ids", []) or []),
        new_block_ids=nbi or [None] * len(req_ids), num_computed_tokens=list(_get(c, "num_computed_tokens", []) or []),
        num_output_tokens=list(_get(c, "num_output_tokens", []) or []),
        additional_information={rid: deserialize_additional_information(v)
                                for rid, v in dict(_get(c, "additional_information", {}) or {}).items()})
    nst = dict(_get(so, "num_scheduled_tokens", {}) or {})
    return SchedulerOutput(
        scheduled_new_reqs=new, scheduled_cached_reqs=cached, num_scheduled_tokens=nst,
        total_num_scheduled_tokens=int(_get(so, "total_num_scheduled_tokens", sum(nst.values()))),
        finished_req_ids=set(_get(so, "finished_req_ids", set()) or set()),
        num_common_prefix_blocks=list(_get(so, "num_common_prefix_blocks", [0]) or [0]),
        new_block_ids_to_zero=_get(so, "new_block_ids_to_zero"),
        finished_requests_needing_kv_transfer=dict(_get(so, "finished_requests_needing_kv_transfer", {}) or {}))
